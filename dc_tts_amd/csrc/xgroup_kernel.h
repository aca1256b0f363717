// xgroup_kernel.h -- a run of newest-row highway layers of the decode chain as ONE launch whose workgroups meet inside one XCD.
//
// The chain of a decode frame (synthesize.py:47-54 for the newest row) is 16 dependent highway layers (AudioDec HC_2..HC_7, then AudioEnc
// HC_4..HC_13, modules.py:143-197), and every layer is an all-to-all: a workgroup that owns 16 of the 256 channels needs all 256 channels of
// the previous layer's row.  As dependent launches a layer costs ~5.3 us (1.45 us of launch boundary + a cold start on every load + the work);
// round 2's persistent variant with hand-offs through device memory (sc1 stores polled with sc1 loads) cost about the same, because the
// eight XCDs' L2s are not coherent with each other and every hand-off is a trip over the fabric.
//
// MI355X-specific observation this kernel is built on (tools/micro/xcd_handoff.hip, profiles/r03_xcd_handoff.txt): the command processor
// deals the workgroups of a launch to the XCDs round-robin -- block b runs on XCD b % 8 -- and INSIDE an XCD the L2 is the coherence point:
// a plain store stays in it, a load that bypasses the CU's L1 (sc1) is served from it, an atomic without sc1 executes in it.  A team of 16
// workgroups on one XCD gets through "publish 512 B, barrier, read everybody's slice" in 0.87 us with one flag word per workgroup (1.02 us with
// an L2 atomic counter), against 1.8 us with agent-scope operations over the fabric and ~3 us for a launch boundary + cold loads.
//
// So: grid = 128 workgroups, always; block b belongs to team b % 8 (= its XCD) and owns column group (b / 8) % 16; a team owns FOUR utterances at a
// time (the newest row of each; utterance groups team, team + 8, ... in turn when B > 32) and all 16 column groups, i.e. everything a layer's
// layer-norm needs.  Per layer a workgroup contracts
// K = 256 for its (gate, info) pair of 16-column tiles over 8 waves (fixed-order LDS reduction, partial layer-norm statistics per column group; since
// round 5 on 4 x 4 x 1 fp32 MFMA blocks -- a team's layer has four rows -- instead of chain3_kernel's 16 x 16 x 4 tiles: see the body), publishes 4 x 32 pre-norm values + statistics with plain stores, arrives at the team's
// barrier (its own word of the team's 64-byte line: no read-modify-write), and reads the other 15 slices past its L1.  Everything that does not depend on the predecessor -- the next layer's
// weights, presum, layer-norm parameters, a dilation-1 layer's history row -- is requested before the barrier.
//
// Placement is used for SPEED; CORRECTNESS does not rest on it: the only thing a workgroup ever trusts is what it reads in ITS OWN L2 -- sixteen
// barrier words that carry the layer's sequence number AND the XCD their writer runs on (HW_REG_XCC_ID).  Words written on this XCD are in this L2
// together with their writers' plain stores (each writer drains its stores before its word); a word from another XCD, should its line ever get
// here through memory, is an error, not a pass.  (In a process with other launches in flight a launch does not start at XCD 0: block b runs
// on XCD (b + k) % 8 for some k, which keeps blocks b, b + 8, b + 16 ... together; the team rule needs no more than that.)  If a team is ever
// split, its barrier cannot complete: every spin is bounded, the error word is raised, dctts_decode_status reports the decode as invalid, and
// the host stops using this kernel (dctts_api.hip: xgroup_ok) in favour of one launch per layer (chain3_kernel), which assumes nothing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "decode3_kernels.h"

namespace dctts {

struct XGroupLayer {
  const float* wp;                       // [tile][16 or 32 k-groups][lane][4]
  const float* presum; int presum_bs;    // bias + older taps of row b at presum + b * presum_bs
  const float* g1; const float* b1; const float* g2; const float* b2;   // THIS layer's H1 / H2 layer-norm parameters (used by the next layer's rebuild)
  float* xm; int xm_bs;                  // this layer's INPUT row is kept at xm + b * xm_bs (history / residual for later launches); nullptr = not kept
  const float* xt; int xt_bs;            // tap2: the input history row one time step back
  int tap2; int pad_;
};
struct XGroupParams {
  int B, L;
  int U, padu_;                          // utterances per team and round (round 6: 4, or 2 / 1 for batches of at most 16 / 8, so that a small batch spreads over all eight teams); 0 = 4
  const float* P0; int p0_bs; const float* stats0;     // the pre-group producer's pre-norm rows (256 channels, a C layer without activation) + statistics
  const float* pg1; const float* pb1;                   // its layer-norm parameters
  XGroupLayer lay[10];
  float* pout; float* stats_out;                        // the LAST layer's pre-norm rows [b][512] and statistics [b][16][4]
  float* xch; float* sch;                               // exchange buffers: [2][B_pad][512] pre-norm rows, [2][B_pad][16][4] statistics
  int xch_set, sch_set;                                 // floats between the two parity copies
  unsigned* bar; unsigned bar_base;                     // team barriers: bar[team * 32] counts arrivals since the decode started; value before this launch
  int* err;                                             // error word: bit 0 = a bounded wait gave up (a split team, or the side stream never arrived)
  unsigned* sig; unsigned sig_val;                      // first launch of a chain piece: *sig = sig_val ("every earlier piece of this stream is complete")
  const unsigned* wait2; unsigned wait_val;             // first launch of a chain piece: the presums come from the side stream: poll *wait2 >= wait_val first
  // passengers: independent small GEMMs (hbulk_body items, decode_kernels.h) that ride in this launch on compute units nobody is using while the
  // teams run -- in the AudioDec run's launch (the merged form of round 4 carries them in xtail_kernel's launch, the first one of a chain piece), the AudioEnc presums of the next frame (consumed by the AudioEnc run that follows on this stream)
  // and the newest row of the C1Q . W2 cache, which the NEXT side-stream piece needs: those workgroups count themselves and the last one
  // publishes psig_val (rowc1_kernel's tail polls it on the side stream)
  const SplitParams* ptab; int p_blocks, p_ipl, p_step; // descriptors; workgroups behind the teams' (p_blocks = descriptors * p_ipl); items per descriptor; frame index
  int p_count_from;                                     // descriptors >= this one are counted (and are dispatched first)
  unsigned* pdone; unsigned pdone_target; unsigned* psig; unsigned psig_val;
  // round 4: the attention row of the newest frame + AudioDec C_1 BEHIND the run's last layer (the AudioEnc run: its output row is Q[j]) instead of two more
  // launches (attnq_kernel, chain3_kernel<RAW>): networks.py:140-151 on the <= 3 keys of the window, then C_1 = (Q . W_bot) + bias + sum_k a_k VW[p + k]
  int attn, N, win, k_stride;
  const int* pm; int* pm_next;                          // prev_max_attentions of this frame [b] / of the next one (= this row's arg-max key, synthesize.py:54)
  const float* K; const float* VW; long kv_bs; int vw_stride; int q_bs;      // rows (b * kv_bs + n) of K (stride k_stride) and of V . W_top (stride vw_stride)
  float* qhist;                                         // Q[j] of utterance b -> qhist + b * q_bs (AudioEnc's output history)
  const float* c1_wp; const float* c1_bias;             // C_1's Q half in 16-column tiles [tile][16 k-groups][lane][4]; its bias
  float* c1_raw; int raw_bs; int pad3_;                 // the bare contraction Q[j] . W_bot -> c1_raw + b * raw_bs (the side stream's C_1 row operation reads it)
  float* c1_pout; float* c1_stats;                      // C_1's pre-norm rows [b][256] + partial statistics [b][16][4] (the AudioDec run's P0 / stats0)
  long long* ts;                                        // measurement (DCTTS_TRACE): workgroup 0 / thread 0 records 100 MHz wall-clock stamps at its phase boundaries
};

// a pointer that came out of the LDS copy of the descriptors is generic to the compiler: say "global" (flat loads count in both wait counters)
__device__ __forceinline__ f32x4 ldg4(const float* base, unsigned off) {
  return *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(reinterpret_cast<uintptr_t>(base + off));
}
__device__ __forceinline__ float ldg1(const float* base, unsigned off) {
  return *reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<uintptr_t>(base + off));
}
__device__ __forceinline__ unsigned xg_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xfu;
}

// The kernel's body as a function: xgroup_kernel below is a launch of its own (the AudioEnc run of the first chain piece, the fallback forms); since round 5 the
// AudioEnc run of every other piece runs BEHIND xtail_kernel's layers in the same launch (xtail_kernel.h: xchain_kernel), where the team stays on its CUs and its XCD.
// `meet` is called once, by every thread of a team workgroup, when the first layer's requests that do not depend on the front of the launch are out (weights,
// layer-norm parameters, the history row) and before the first request that does (the producer's pre-norm rows): xchain_kernel
// passes the team's barrier, xgroup_kernel nothing.
template <bool TS = false, typename Meet>   // TS: the stamped instantiation (DCTTS_TRACE); the production kernel carries no trace of the stamps
__device__ __forceinline__ void xgroup_body(const XGroupParams* __restrict__ pp, Meet meet) {
  __shared__ __attribute__((aligned(16))) float red[8 * 2 * 4 * 64];
  __shared__ int s_go;
  // the layers' descriptors, copied once: read through the scalar cache a layer's fields arrive as several dependent scalar loads inside the layer
  __shared__ XGroupLayer s_lay[10];
  // per wave: the wave's 32 channels (k-groups w and w + 8) of the team's four rows, [row][e * 16 + c]: written in the COMPACT layout the rows are
  // rebuilt in (lane = (row, c): two values per lane), read in the MFMA A-operand layout (lane = (row, 4-channel group): two float4 per lane)
  __shared__ __attribute__((aligned(16))) float s_xs[8][4 * 32];
  __shared__ float s_att[8 * 4 * 4];      // attention tail: per wave and row, the partial dot products with the window's <= 3 keys
  typedef const __attribute__((address_space(4))) XGroupParams CP;
  CP& p = *(CP*)pp;
  // Block order of a launch with passengers: [the teams] [the counted passengers] [the others].  The teams come FIRST: when the launch starts, the side
  // stream's xcone_kernel holds 16 of every XCD's 32 CUs and the teams need exactly the other 16 -- with the counted passengers in front of them (they would
  // finish ~4 us earlier, worth 0.9 us per frame) some team workgroups have to wait for a CU of THEIR XCD, and one decode in ~100 then failed a team hand-off.
  const int tbid = (int)blockIdx.x;        // block index among the teams' blocks
  if (p.p_blocks) {
    const int first = (int)gridDim.x - p.p_blocks;
    if ((int)blockIdx.x >= first) {
      extern __shared__ __attribute__((aligned(16))) float pass_smem[];
      __shared__ long s_prow[2][32];
      const int nd = p.p_blocks / p.p_ipl, q = (int)blockIdx.x - first, qd = q / p.p_ipl, item = q - qd * p.p_ipl;
      const int ncount = nd - p.p_count_from, layer = qd < ncount ? p.p_count_from + qd : qd - ncount;
      ConstSplitParams& sp = *((ConstSplitParams*)p.ptab + layer);
      hbulk_body<8, ConstSplitParams>(sp, p.p_step + sp.step_val, item, p.p_ipl, p.p_ipl, pass_smem, s_prow);
      if (p.pdone && layer >= p.p_count_from) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                   // every thread's stores are out
        if (threadIdx.x == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");               // ... and written back past this XCD's L2
          const unsigned old = __hip_atomic_fetch_add(p.pdone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (old + 1u == p.pdone_target) __hip_atomic_store(p.psig, p.psig_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      return;
    }
  }
  // everything the first layer needs, in ONE batch of scalar loads (left alone the fields arrive lazily, a dependent scalar load at a time: ~3 us
  // until the first row was built)
  asm volatile("; xgroup: the first layer's parameters, one batch"
               :: "s"(p.B), "s"(p.L), "s"(p.P0), "s"(p.p0_bs), "s"(p.stats0), "s"(p.pg1), "s"(p.pb1),
                  "s"(p.lay[0].wp), "s"(p.lay[0].tap2), "s"(p.lay[0].xt), "s"(p.lay[0].xt_bs), "s"(p.lay[0].presum), "s"(p.lay[0].presum_bs),
                  "s"(p.xch), "s"(p.sch), "s"(p.xch_set), "s"(p.sch_set), "s"(p.bar), "s"(p.bar_base), "s"(p.err),
                  "s"(p.sig), "s"(p.sig_val), "s"(p.wait2), "s"(p.wait_val), "s"(p.pout), "s"(p.stats_out), "s"(p.U));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bx = tbid & 7, bq = tbid >> 3;
  // The grid is ALWAYS 128 team workgroups (8 teams, one per XCD), whatever the batch: a team takes the utterance groups team, team + 8, ... in turn.
  // More team workgroups than that can starve the other stream of CUs while they poll for it (both team kernels are one workgroup per CU by registers):
  // at B = 96 a third round of polling xgroup workgroups held the CUs xcone_kernel needed to finish -- a resource deadlock until the bounded waits gave up.
  const int grp = bq & 15, team = bx;
  const int U = p.U ? p.U : 4;                                             // utterances this team serves per round (the four-row machinery below runs whatever U is: slots past U repeat the group's first utterance and store nothing)
  if (team * U >= p.B) return;                                             // a team without utterances: uniform per workgroup
  const int arow = lane & 15, aq = lane >> 4, c4 = aq * 4;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  unsigned* const bar = p.bar + team * 32;
  const unsigned xcc = xg_xcc_id();
  const int erow = aq * 4 + (wave & 3), etile = wave >> 2, ecol = lane & 15;
  const int pcol = etile * 256 + grp * 16 + ecol;
  int nts = 0;
  auto stamp = [&]() { if constexpr (TS) { if (p.ts && tbid == 0 && tid == 0 && nts < 120) p.ts[nts++] = wall_clock64(); } };
  stamp();
  for (int round = 0, m0 = team * U; m0 < p.B; ++round, m0 += 8 * U) {
  const int mend = (m0 + U < p.B) ? m0 + U : p.B;                          // this round's utterances: m0 .. mend - 1
  if (round > 0) __syncthreads();                                          // the previous round's last reads of the LDS buffers
  const unsigned rbase = p.bar_base + (unsigned)round * (unsigned)(p.L - 1 + (p.attn ? 1 : 0)) * 16u;      // the team's barrier sequence numbers of this round (the attention tail adds one hand-off)
  const int b = m0 + arow;
  // Only 4 of the 16 rows of an MFMA tile are utterances.  The other lanes run the same instructions on row 0's addresses: what they compute lands in
  // output rows nobody reads (MFMA rows are independent), so nothing is masked or zeroed for them -- per layer that was ~100 v_mov / select / exec-mask
  // instructions per wave, on a path that is bound by instruction issue as much as by latency.
  const bool valid = arow < 4 && b < mend;
  const unsigned bb = valid ? (unsigned)b : 0u;
  const int eb = m0 + erow;
  const bool wr = erow < 4 && eb < mend;

  // ---- round 5: the contraction runs on v_mfma_f32_4x4x1_16b_f32 -- 16 independent 4 x 4 blocks per instruction, block = lanes 4 i .. 4 i + 3, D[row][lane j]
  //      += A[row of lane] . B[column of lane j].  A team's layer has FOUR rows: on 16 x 16 x 4 tiles 12 of the 16 rows were padding, and the 16 (K = 512: 32)
  //      MFMAs of a wave took 512 (1024) cycles of a pipe it shares with a second wave: 0.43 (0.85) us per layer of MFMA issue alone.  Here a wave's block
  //      (cb, kh) = (lane >> 2) & 7, lane >> 5 owns columns 4 cb .. 4 cb + 3 of the workgroup's 32 (cb < 4: the gate tile, else the info tile) and the 16
  //      channels of k-group wave + 8 kh; 16 instructions of 8 cycles walk over those channels.  The weights are read from the SAME packing
  //      ([tile][k-group][lane = (col, k / 4)][4]): lane (cb, kh, j) takes the four float4 of column 4 (cb & 3) + j, sixteen lanes 256 contiguous bytes.
  const int j4 = lane & 3, cb = (lane >> 2) & 7, kh = lane >> 5;
  const unsigned kgw = (unsigned)(wave + 8 * kh);
  const unsigned wlane = (unsigned)(((cb & 3) * 4 + j4) * 4);             // float offset of this lane's column inside a (tile, k-group) block; + 64 per k / 4
  const unsigned wtile = (unsigned)(grp * 2 + (cb >> 2));
  const unsigned bj = (m0 + j4 < mend) ? (unsigned)(m0 + j4) : (unsigned)m0;      // the utterance of this lane's A row
  f32x4 wq[4], wtq[4], atq[4] = {z4, z4, z4, z4};
  float p0c[2], g1c[2], b1c[2];
  f32x4 st0;
  const int cr = lane >> 4, cc = lane & 15;
  const unsigned crow = (m0 + cr < mend) ? (unsigned)(m0 + cr) : (unsigned)m0;
  {
    const bool t2 = p.lay[0].tap2 != 0;
    const unsigned nkg = t2 ? 32u : 16u, kc = t2 ? 16u : 0u;
    const float* wb = p.lay[0].wp + wlane;
    const unsigned w0 = (wtile * nkg + kc + kgw) * 256u, wt0 = (wtile * nkg + kgw) * 256u;
#pragma unroll
    for (int q = 0; q < 4; ++q) wq[q] = ldv(wb, w0 + 64u * q);
#pragma unroll
    for (int q = 0; q < 4; ++q) wtq[q] = t2 ? ldv(wb, wt0 + 64u * q) : z4;
    if (t2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) atq[q] = ldv(p.lay[0].xt, bj * (unsigned)p.lay[0].xt_bs + kgw * 16u + 4u * q);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) { const unsigned ch = (unsigned)((8 * e + wave) * 16 + cc); g1c[e] = p.pg1[ch]; b1c[e] = p.pb1[ch]; }
  }
  if (round == 0) {
    constexpr int NW32 = (int)(sizeof(XGroupLayer) * 10 / 4);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(pp->lay);
    if (tid < NW32) reinterpret_cast<uint32_t*>(s_lay)[tid] = src[tid];
    meet();
  }
  // ---- what the launch's front (or the launch before this one) produced: the pre-norm rows and their partial statistics
  {                                         // (compact: lane (cr, cc) owns row cr, channels 16 w + cc and 128 + 16 w + cc, and column group cc's statistics)
#pragma unroll
    for (int e = 0; e < 2; ++e) p0c[e] = p.P0[crow * (unsigned)p.p0_bs + (unsigned)((8 * e + wave) * 16 + cc)];
    st0 = ldv(p.stats0, crow * 64u + (unsigned)(cc * 4));
  }
  // ---- placement check, the stream signal, and the wait for the side stream (first launch of a chain piece), while those loads are in flight
  if (tid == 0 && round == 0) {
    int go = __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;     // an earlier launch of this decode already failed: no more waiting, the decode is reported invalid
    if (p.sig && tbid == 0) __hip_atomic_store(p.sig, p.sig_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (p.wait2 && go) {
      bool ok = false;
      for (int i = 0; i < (1 << 20) && !ok; ++i) {                         // bounded: about a second
        ok = __hip_atomic_load(p.wait2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= p.wait_val;
        if (!ok) __builtin_amdgcn_s_sleep(16);                              // 128 pollers over the fabric: ~0.4 us apart is plenty, and spares the side stream's bandwidth
      }
      if (!ok) atomicOr(p.err, 16);                                        // (bits of the error word: 1 / 2 xgroup barrier time-out / split team, 4 / 8 the same in xcone_kernel, 16 this wait)
    }
    s_go = go;
  }
  __syncthreads();
  const bool team_ok = s_go != 0;                                          // a misplaced workgroup still arrives at every barrier (its team-mates time out), but skips the waits
  float addv = 0.f;
  if (wr) {                                                                // presum of layer 0: written by the side stream -> read past the L1 / a possibly stale line
    const float* ap = p.lay[0].presum + (unsigned)(eb * p.lay[0].presum_bs) + (unsigned)pcol;
    addv = __hip_atomic_load(ap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (sc0 sc1; waited for where it is used, behind the first contraction)
  }
  // The rebuild of a layer's input (layer-norm of both halves, sigmoid gate, highway mix) is done in a COMPACT layout: lane (cr, cc) owns row cr and the
  // two channels 16 w + cc and 128 + 16 w + cc: 2 values per lane and an LDS hop into the operand layout (lane (cb, kh, j): row j, the 16 channels of k-group w + 8 kh).
  float* const xs = s_xs[wave];
  float xc[2];                                                             // this layer's input at (cr, channel e) = the next rebuild's highway residual
  f32x4 ax[4];                                                             // ... and as the contraction's A operand
  {
    const float m1 = row16_sum(st0[0]) * (1.0f / 16.0f);
    const float d1 = st0[0] - m1;
    const float r1 = rsqrt_fast(row16_sum(st0[1] + 16.0f * d1 * d1) * (1.0f / 256.0f) + 1e-12f);
#pragma unroll
    for (int e = 0; e < 2; ++e) { xc[e] = (p0c[e] - m1) * r1 * g1c[e] + b1c[e]; xs[cr * 32 + e * 16 + cc] = xc[e]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) ax[q] = *reinterpret_cast<const f32x4*>(&xs[j4 * 32 + kh * 16 + 4 * q]);
  }
  stamp();                                                                 // first input row built (the wait for the side stream is in here)
  for (int g = 0; g < p.L; ++g) {
    const bool last = (g + 1 == p.L);
    const bool tail = last && p.attn != 0;                                  // the run's output row goes on into the attention + C_1 block below
    const bool t2 = __builtin_amdgcn_readfirstlane(s_lay[g].tap2) != 0;
    // ---- contraction of layer g: 16 (tap2: 32) instructions, two accumulators in turn
    f32x4 acc0 = z4, acc1 = z4;
    if (t2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 a = atq[q], b = wtq[q];
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], b[0], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], b[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], b[2], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], b[3], acc1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 a = ax[q], b = wq[q];
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], b[0], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], b[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], b[2], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], b[3], acc1, 0, 0, 0);
    }
    // the wave's partial sums: [lane (cb, kh, j)][row] -- 1 KB per wave (the 16 x 16 tiles wrote 8 KB, three quarters of it padding rows)
    *reinterpret_cast<f32x4*>(&red[wave * 256 + lane * 4]) = acc0 + acc1;
    // ---- requests that do not depend on the other workgroups: the next layer's weights / presum / history row, this layer's LN parameters
    float ng1[2], nb1[2], ng2[2], nb2[2];                                   // compact (a last layer without the attention tail leaves them unloaded, and leaves the loop before they are used)
    float naddv = 0.f;
    if (!last || tail) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const unsigned chc = (unsigned)((8 * e + wave) * 16 + cc);
        ng1[e] = ldg1(s_lay[g].g1, chc); nb1[e] = ldg1(s_lay[g].b1, chc); ng2[e] = ldg1(s_lay[g].g2, chc); nb2[e] = ldg1(s_lay[g].b2, chc);
      }
    }
    if (!last) {
      const XGroupLayer& Ln = s_lay[g + 1];
      const bool nt2 = __builtin_amdgcn_readfirstlane(Ln.tap2) != 0;
      const unsigned nkg = nt2 ? 32u : 16u, kc = nt2 ? 16u : 0u;
      const float* wb = Ln.wp + wlane;
      const unsigned w0 = (wtile * nkg + kc + kgw) * 256u, wt0 = (wtile * nkg + kgw) * 256u;
#pragma unroll
      for (int q = 0; q < 4; ++q) wq[q] = ldg4(wb, w0 + 64u * q);
      if (nt2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) wtq[q] = ldg4(wb, wt0 + 64u * q);
#pragma unroll
        for (int q = 0; q < 4; ++q) atq[q] = ldg4(Ln.xt, bj * (unsigned)Ln.xt_bs + kgw * 16u + 4u * q);
      }
      if (wr) naddv = ldg1(Ln.presum, (unsigned)(eb * Ln.presum_bs) + (unsigned)pcol);      // behind the wait for the side stream; never read before in this launch
    }
    // this layer's input row is kept for later launches (history / residual): column group 0 stores it
    if (grp == 0 && m0 + cr < mend && s_lay[g].xm) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
        *reinterpret_cast<__attribute__((address_space(1))) float*>(reinterpret_cast<uintptr_t>(s_lay[g].xm + (long)(m0 + cr) * s_lay[g].xm_bs + (8 * e + wave) * 16 + cc)) = xc[e];
    }
    stamp();                                                               // contraction issued, partial sums written, prefetches issued
    __syncthreads();
    float v_ = 0.f;                                                          // (row wave & 3, column ecol of tile etile): 8 waves x 2 k-halves, fixed order
    {
      const float* rp = &red[((etile * 4 + (ecol >> 2)) * 4 + (ecol & 3)) * 4 + (wave & 3)];
#pragma unroll
      for (int w = 0; w < 8; ++w) { v_ += rp[w * 256]; v_ += rp[w * 256 + 128]; }
    }
    v_ += addv;
    const float mg = row16_sum(v_) * (1.0f / 16.0f);
    const float dv = v_ - mg;
    const float m2g = row16_sum(dv * dv);
    stamp();                                                               // slice reduced, statistics
    if (last && !tail) {
      if (wr) p.pout[(long)eb * 512 + pcol] = v_;
      if (wr && ecol == 0) { float* so = p.stats_out + ((long)eb * 16 + grp) * 4 + etile * 2; so[0] = mg; so[1] = m2g; }
      break;
    }
    // ---- publish this workgroup's slice of layer g with PLAIN stores (they stay in this XCD's L2), then arrive at the team's barrier
    //      (round 5: xtail_kernel's tagged hand-off -- a sequence tag beside every value, the consumers poll the data -- was tried here too: 4 x 8-byte + 2 x 16-byte
    //      requests per poll, and a poll that comes too early costs a whole L2 round trip; 78.8 against 78.7 us per frame on one box, so the barrier stays)
    const int par = g & 1;
    if (wr) {
      p.xch[(long)par * p.xch_set + (long)eb * 512 + pcol] = v_;
      if (ecol == 0) { float* so = p.sch + (long)par * p.sch_set + ((long)eb * 16 + grp) * 4 + etile * 2; so[0] = mg; so[1] = m2g; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the stores are in the L2 (and the prefetches have landed)
    stamp();                                                               // published
    __syncthreads();
    // The team's barrier: every workgroup writes the layer's sequence number into ITS word of the team's line (a plain store: it stays in the XCD's
    // L2), and polls all sixteen words with one 64-byte request.  No read-modify-write: sixteen atomics on one word take ~25 ns each in the L2, and
    // the pollers' reads of that word queue in between (measured: 128 arrivals per layer cost 3-5 us).
    // The word also carries the XCD the writer runs on: a word from another XCD (should its line ever get here through memory) is an error, never a pass.
    if (wave == 0) {
      const unsigned target = rbase + (unsigned)(g + 1) * 16u, mine = (target << 4) | xcc;
      if (lane == 0) __hip_atomic_store(bar + grp, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (team_ok) {
        int spins = 0;
        for (;;) {
          const unsigned v = lane < 16 ? __hip_atomic_load(bar + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : mine;   // sc1 loads: past the L1, served by the L2
          const bool there = (int)((v >> 4) - target) >= 0;
          if (__builtin_amdgcn_ballot_w64(there && (v & 15u) != xcc) != 0ull) { if (lane == 0) atomicOr(p.err, 2); break; }      // a split team
          if (__builtin_amdgcn_ballot_w64(!there) == 0ull) break;
          if (++spins > (1 << 16) || ((spins & 255) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { if (lane == 0) atomicOr(p.err, 1); break; }   // bounded (~20 ms), and nobody keeps waiting once anybody gave up
        }
      }
    }
    __syncthreads();
    stamp();                                                               // team barrier passed
    // ---- the team's rows of layer g, past the L1
    float hg[2], hi[2]; f32x4 stc;
    {
      const float* xr = p.xch + (long)par * p.xch_set + (long)crow * 512 + wave * 16 + cc;   // gate channel 16 w + cc; +128 floats = the second k-group; +256 = info
      const float* sr = p.sch + (long)par * p.sch_set + (long)crow * 64 + cc * 4;             // column group cc's partial statistics of the row
      asm volatile(
          "global_load_dword %0, %5, off sc1\n\t"
          "global_load_dword %1, %5, off offset:512 sc1\n\t"
          "global_load_dword %2, %5, off offset:1024 sc1\n\t"
          "global_load_dword %3, %5, off offset:1536 sc1\n\t"
          "global_load_dwordx4 %4, %6, off sc1\n\t"
          "s_waitcnt vmcnt(0)"
          : "=&v"(hg[0]), "=&v"(hg[1]), "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(stc)
          : "v"(xr), "v"(sr)
          : "memory");
    }
    stamp();                                                               // exchanged rows landed
    // ---- rebuild the next layer's input: gate(LN(exchanged rows)) mixed with this layer's input (the highway residual), compact, then through
    //      the LDS into the A-operand layout.  Chan-combine of the 16 per-group partials (mean_g, M2_g over 16 channels each): the row's 16 lanes hold one each.
    {
      const float m1 = row16_sum(stc[0]) * (1.0f / 16.0f), m2 = row16_sum(stc[2]) * (1.0f / 16.0f);
      const float d1 = stc[0] - m1, d2 = stc[2] - m2;
      const float r1 = rsqrt_fast(row16_sum(stc[1] + 16.0f * d1 * d1) * (1.0f / 256.0f) + 1e-12f);
      const float r2 = rsqrt_fast(row16_sum(stc[3] + 16.0f * d2 * d2) * (1.0f / 256.0f) + 1e-12f);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float s_ = sigmoid_fast((hg[e] - m1) * r1 * ng1[e] + nb1[e]);
        xc[e] = s_ * ((hi[e] - m2) * r2 * ng2[e] + nb2[e]) + (1.0f - s_) * xc[e];
        xs[cr * 32 + e * 16 + cc] = xc[e];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) ax[q] = *reinterpret_cast<const f32x4*>(&xs[j4 * 32 + kh * 16 + 4 * q]);
    }
    addv = naddv;
    if (tail) break;
  }
  if (p.attn) {
    stamp();                                                               // (tail: last layer's output row rebuilt)
    // ---- the attention tail, OUTSIDE the layer loop: with its operands requested in the loop's prefetch slot the loop carried 47 more registers and
    //      every layer got ~0.3 us slower (stamps); requested in front of the first layer and held through the run (19 registers, round 5) the frame did not
    //      get shorter (A/B on one box: 79.6 against 79.4 us).  None of the operands depends on this run -- C_1's 16-column tile, the window's K rows for this
    //      lane's row and channels, V . W_top of the window for the (row, column) this lane finishes, the window position itself -- so they are one batch here.
    f32x4 vc1[2];
    float kk[3][2], vwv[3], c1b;
    int pm_e;
    {
    const float* wb = p.c1_wp + lane * 4;
#pragma unroll
    for (int e = 0; e < 2; ++e) vc1[e] = ldv(wb, (unsigned)(grp * 16 + wave + 8 * e) * 256u);
    const int pm_c = p.pm[crow];
    pm_e = p.pm[wr ? eb : (int)crow];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int nc = pm_c + k; if (nc > p.N - 1) nc = p.N - 1;                 // beyond the window: clamped into the utterance, weight exactly 0
      const float* kr = p.K + ((long)crow * p.kv_bs + nc) * p.k_stride;
      kk[k][0] = kr[wave * 16 + cc]; kk[k][1] = kr[128 + wave * 16 + cc];
      int ne = pm_e + k; if (ne > p.N - 1) ne = p.N - 1;
      vwv[k] = p.VW[((wr ? (long)eb : (long)crow) * p.kv_bs + ne) * p.vw_stride + grp * 16 + ecol];
    }
    c1b = p.c1_bias[grp * 16 + ecol];
    }
    float4 x[2];                                                           // Q[j] as the 16 x 16 x 4 A operand (C_1's tile below keeps that shape: one 16-column tile, 8 MFMAs)
    x[0] = *reinterpret_cast<const float4*>(&xs[(arow & 3) * 32 + c4]);
    x[1] = *reinterpret_cast<const float4*>(&xs[(arow & 3) * 32 + 16 + c4]);
    // ---- xs / xc hold Q[j] of the team's four utterances.  Keep the row (column group 0), attend, and run C_1 on it.
    if (grp == 0 && m0 + cr < mend) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
        *reinterpret_cast<__attribute__((address_space(1))) float*>(reinterpret_cast<uintptr_t>(p.qhist + (long)(m0 + cr) * p.q_bs + (8 * e + wave) * 16 + cc)) = xc[e];
    }
    // logits of the window's keys (networks.py:140): this wave's 32 channels of row cr, summed over the 16 lanes of the row, then over the waves through LDS
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float pr = row16_sum(fmaf(xc[1], kk[k][1], xc[0] * kk[k][0]));
      if (cc == 0) s_att[(wave * 4 + cr) * 4 + k] = pr;
    }
    // C_1's contraction on the Q row does not depend on the attention result: Q[j] . W_bot, one 16-column tile per workgroup
    f32x4 accc = z4;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float4 a = x[e]; const f32x4 b0 = vc1[e];
      accc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0[0], accc, 0, 0, 0); accc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0[1], accc, 0, 0, 0);
      accc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0[2], accc, 0, 0, 0); accc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0[3], accc, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[((wave * 2 + 0) * 4 + j) * 64 + lane] = accc[j];
    stamp();                                                               // (tail: operands landed, logits' partial sums and C_1's contraction written)
    __syncthreads();
    if (wave < 4) {
      // wave w finishes row w of the team (lanes 0 .. 15: the tile's 16 columns): masked softmax over <= 3 keys, arg-max of the post-softmax row with the
      // first index on ties (networks.py:142-149), exactly attnq_kernel's arithmetic behind the dot products
      const int rw = wave;
      int nk = p.N - pm_e; if (nk > p.win) nk = p.win;
      float lg[3], a[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float t_ = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t_ += s_att[(w * 4 + rw) * 4 + k];
        lg[k] = (k < nk) ? t_ * 0.0625f : -INFINITY;                    // tf.rsqrt(256) = 1 / 16
      }
      const float mx = fmaxf(lg[0], fmaxf(lg[1], lg[2]));
      float se = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) { a[k] = (k < nk) ? expf(lg[k] - mx) : 0.f; se += a[k]; }
      const float inv = 1.0f / se;
      int am = 0;
      float best = a[0] * inv; a[0] = best;
#pragma unroll
      for (int k = 1; k < 3; ++k) { a[k] *= inv; if (a[k] > best) { best = a[k]; am = k; } }
      float v_ = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v_ += red[((w * 2 + 0) * 4 + rw) * 64 + lane];
      float ps = c1b;
#pragma unroll
      for (int k = 0; k < 3; ++k) ps = fmaf(a[k], vwv[k], ps);          // bias + sum_k a_k (V . W_top)[p + k]   (a_k == 0 beyond the window)
      const float vt = v_ + ps;
      const float mg = row16_sum(vt) * (1.0f / 16.0f);
      const float dv = vt - mg;
      const float m2g = row16_sum(dv * dv);
      const bool ok = aq == 0 && m0 + rw < mend;                         // (lanes 16 .. 63 hold the tile's padding rows)
      if (ok) {
        const long b_ = m0 + rw; const int col = grp * 16 + ecol;
        p.c1_raw[b_ * p.raw_bs + col] = v_;
        p.c1_pout[b_ * 256 + col] = vt;
        if (ecol == 0) { float* so = p.c1_stats + (b_ * 16 + grp) * 4; so[0] = mg; so[1] = m2g; }
        if (grp == 0 && ecol == 0) p.pm_next[b_] = pm_e + am;           // max_attentions[:, j] (synthesize.py:54)
      }
    }
    stamp();                                                               // (tail: softmax, C_1 row finished, stores issued)
  }
  }                                        // next utterance group of this team
}

// grid: 128 blocks of 512 threads, whatever the batch (+ p_blocks passengers, with hsplit_smem(32) bytes of dynamic LDS)
template <bool TS = false>
__global__ void __launch_bounds__(512) xgroup_kernel(const XGroupParams* __restrict__ pp) { xgroup_body<TS>(pp, []() {}); }

}  // namespace dctts
