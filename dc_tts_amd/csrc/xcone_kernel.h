// xcone_kernel.h -- the tail of AudioDec's dependency cone (HC_3 .. HC_7, networks.py:175-191) for one decode frame as ONE launch whose
// workgroups meet inside one XCD.
//
// Exact parity with synthesize.py:47-54 makes every frame re-evaluate AudioDec over the shrinking cone of rows its newest row depends on
// (46 / 16 / 6 / 4 / 1 rows per utterance for HC_3 .. HC_7, the last one of each being the presum row the chain finishes).  As launches that
// is, per frame and on the side stream, five small GEMMs and four layer-norm row passes = nine dependent launches, 99 us of which ~35 are
// arithmetic: hbulk_kernel<12> re-reads its weights for every 32-row item, the 16-row launches and the row passes are launch latency.
//
// Same idea as xgroup_kernel.h: a team of 16 workgroups that the command processor put on ONE XCD (blocks b, b + 8, b + 16 ...) owns four
// utterances; a workgroup owns one column group (16 gate + 16 info columns) and keeps that slice of a layer's weights (96 KB) in registers
// while it walks over the team's row tiles; pre-norm rows go to the XCD's L2 with plain stores; the team meets at an L2 atomic; then the
// team's waves normalise / gate the rows (one wave per row, two-pass layer-norm, highway mix: modules.py:189-193) and store the layer's
// output rows, meet again, and go on to the next layer.  The only thing a workgroup ever trusts is its team's barrier word reaching the
// count in ITS OWN L2 (which proves the team is on this XCD); every spin is bounded and raises the error word (dctts_decode_status).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "decode3_kernels.h"
#include "xgroup_kernel.h"

namespace dctts {

struct XConeLayer {
  const float* wp;                       // 16-column tiles, all three taps: [tile = 2 grp + h][48 k-groups][lane][4]
  const float* bias;                     // [512]
  const float* g1; const float* b1; const float* g2; const float* b2;
  const float* xin; long xin_bstride; long xin_row0; int xin_stride;     // this layer's input rows (frame parity folded into the pointer)
  float* xout; long xout_bstride; long xout_row0; int xout_stride;       // this layer's output rows (cone rows at offsets < 0)
  float* pout;                           // pre-norm rows [b * R + r][512]; row R - 1 of an utterance is the presum row of the chain
  const int* offs; int R;                // cone offsets < 0 (descending) followed by 0; rows per utterance
  int tap_off[3];
};
struct XConeParams {
  int B, L, frame;
  int tail_rows;                         // 1: the LAST layer's cone rows are normalised / gated / stored as well (round 4: the layers behind it run on the chain, xtail_kernel.h)
  XConeLayer lay[5];
  unsigned* bar; unsigned bar_base;      // team barriers: bar[team * 32] counts arrivals since the decode started; value before this launch
  int* err;
  unsigned* done; unsigned done_target;  // teams that finished since the decode started; the team that brings it to done_target ...
  unsigned* sig; unsigned sig_val;       // ... tells the chain's stream that this piece is complete (instead of a stream write operation behind the launch)
  const unsigned* wait; unsigned wait_val;                // then: the NEXT side-stream piece needs the chain's counter at wait_val; the team leaders poll it here
  int* wait_err;                                          //   (instead of a stream wait operation in front of that piece); a time-out raises this word
  long long* ts;                         // measurement (DCTTS_TRACE): workgroup 0 / thread 0 records 100 MHz wall-clock stamps at its phase boundaries
  // Round 5, the folded form (fold != 0): AudioDec C_1 and HC_2 over their cone rows -- the two row operations that used to be launches of their own in front of this
  // one (rowc1_kernel, rowhc2_kernel: 10.4 + 10.3 us per frame for ~3 us of work each) -- are the first two phases of this launch.  A team needs only ITS four
  // utterances' rows, so it meets at its own barrier instead of at two launch boundaries: workgroup grp serves utterance grp / 4 of the team with rows
  // (grp % 4) * 8 + wave + 32 i (eight waves, three rows each), the utterance's shared operands staged in LDS once per workgroup.
  int fold; int pad_;
  RowC1Params rc1; RowHc2Params rhc2;
};

__device__ __forceinline__ bool xcone_barrier(unsigned* bar, int grp, unsigned xcc, unsigned target, int* err, bool go) {
  // caller: all stores of the phase issued; every thread calls this.  Same barrier as xgroup_kernel's: one word per workgroup in the team's line,
  // plain stores, one 64-byte poll -- no read-modify-write on a shared word.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // this thread's stores are in the L2
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const unsigned mine = (target << 4) | xcc;                              // (the writer's XCD rides in the word: see xgroup_kernel.h)
    if (lane == 0) __hip_atomic_store(bar + grp, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // stays in the XCD's L2
    if (go) {
      int spins = 0;
      for (;;) {
        const unsigned v = lane < 16 ? __hip_atomic_load(bar + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : mine;       // sc1 loads: past the L1, served by the L2
        const bool there = (int)((v >> 4) - target) >= 0;
        if (__builtin_amdgcn_ballot_w64(there && (v & 15u) != xcc) != 0ull) { if (lane == 0) atomicOr(err, 8); break; }         // a split team
        if (__builtin_amdgcn_ballot_w64(!there) == 0ull) break;
        if (++spins > (1 << 16) || ((spins & 255) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { if (lane == 0) atomicOr(err, 4); break; }
      }
    }
  }
  __syncthreads();
  return true;
}

// grid: 128 blocks of 512 threads, whatever the batch
// pm_row (folded form): prev_max_attentions of this frame, pm_all + frame * B -- a kernel ARGUMENT, so that the window position, which everything of the row phases
// hangs on, is requested at entry instead of behind the load of the per-frame table
__global__ void __launch_bounds__(512) xcone_kernel(const XConeParams* __restrict__ pp, const int* __restrict__ pm_row) {
  __shared__ __attribute__((aligned(16))) float red[2][2 * 8 * 2 * 4 * 64];  // split-K partial sums of two row tiles, double-buffered: one barrier per pass
  __shared__ int s_go;
  __shared__ int s_xoff[256];            // per local row m of the team (M <= 4 * 64): element offset of its input row t in xin, -1 = the row does not exist (t < 0)
  __shared__ int s_prow[256];            // ... its pre-norm row index in pout
  typedef const __attribute__((address_space(4))) XConeParams CP;
  CP& p = *(CP*)pp;
  // the launch's own parameters in ONE batch of scalar loads (lazily: six dependent batches before the first layer)
  asm volatile("; xcone: parameters, one batch" :: "s"(p.B), "s"(p.L), "s"(p.frame), "s"(p.bar), "s"(p.bar_base), "s"(p.err), "s"(p.ts),
               "s"(p.done), "s"(p.done_target), "s"(p.sig), "s"(p.sig_val), "s"(p.wait), "s"(p.wait_val), "s"(p.wait_err), "s"(p.lay[0].wp));
  const int tid = threadIdx.x;
  const int bx = blockIdx.x & 7, bq = blockIdx.x >> 3;
  // the grid is always 128 workgroups (8 teams): a team takes the utterance groups team, team + 8, ... in turn (xgroup_kernel.h says why)
  const int grp = bq & 15, team = bx;
  if (team * 4 >= p.B) return;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  unsigned* const bar = p.bar + team * 32;
  const unsigned xcc = xg_xcc_id();
  if (tid == 0) s_go = __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;     // an earlier launch of this decode already failed: no more waiting
  unsigned arrived = p.bar_base;
  const int frame = p.frame;
  int nts = 0;
  auto stamp = [&]() { if (p.ts && blockIdx.x == 0 && tid == 0 && nts < 60) { p.ts[100 + (nts > 0)] = clock64(); p.ts[nts++] = wall_clock64(); } };   // ([100], [101]: shader clock at the first / latest stamp)
  stamp();

  for (int b0 = team * 4; b0 < p.B; b0 += 32) {
  const int nb = (p.B - b0 < 4) ? p.B - b0 : 4;
  if (p.fold) {
    int tid_a = tid;                                                     // (opaque per phase, like tid_o below: what is derived from the thread index stays inside its phase)
    asm volatile("; xcone: thread index, opaque (phase A)" : "+v"(tid_a));
    const int lane = tid_a & 63, wave = tid_a >> 6;
    // ---- phase A: AudioDec C_1 over its cone rows (decode3_kernels.h: rowc1_*), phase B: HC_2 over its cone rows + its presum row (rowhc2_*)
    const RowC1Params& ca = pp->rc1; const RowHc2Params& cb = pp->rhc2;
    f32x4* const s_c = reinterpret_cast<f32x4*>(&red[0][0]);           // phase B's shared operands: 18 KB + 18 KB of the 64 KB split-K buffer (nothing else uses it before the first GEMM pass)
    f32x4* const s_v = s_c + 9 * 128;
    f32x4* const s_sh = s_v + 9 * 128;                                  // phase A's: 9 KB behind them
    const int ub = grp >> 2, q4 = grp & 3;
    const bool uok = ub < nb;
    const int b = uok ? b0 + ub : b0;
    const int pm = pm_row[b];
    auto ldq = [](const float* q_) { return *reinterpret_cast<const f32x4*>(q_); };
    __syncthreads();                                                    // (a second round: the previous round's last reads of `red`)
    rowc1_stage(ca, b, pm, s_sh, tid_a, 512);
    {
      f32x4 vq[3], vcq[3]; int tt[3]; bool lv[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int r0 = q4 * 8 + wave + 32 * i;
        const int r = r0 < ca.R ? r0 : ca.R - 1;
        const int t_ = ca.contig ? ca.frame - 1 - r : ca.frame + ca.offs[r];
        lv[i] = uok && r0 < ca.R && t_ >= 0;
        tt[i] = t_ < 0 ? 0 : t_;
        vq[i] = ldq(ca.Qh + ((long)b * ca.q_bstride + ca.q_row0 + tt[i]) * ca.q_stride + lane * 4);
        vcq[i] = ldq(ca.C1Q + ((long)b * ca.c_bstride + ca.c_row0 + tt[i]) * ca.c_stride + lane * 4);
      }
      // phase B's shared operands (36 KB per workgroup, ~1.5 us at an LDS-DMA fill rate of ~25 GB/s per CU) do not depend on phase A: requested here, they land while
      // phase A's rows are finished.  vmcnt counts in issue order, so phase A waits for everything but those requests: 2 x 3 per thread in waves 0 and 1 (1152 pieces over
      // 512 threads), 2 x 2 in the others.
      rowhc2_stage(cb, b, pm, s_c, s_v, tid_a, 512);
      if (wave < 2) asm volatile("s_waitcnt vmcnt(6)" : "+v"(vq[0]), "+v"(vcq[0]), "+v"(vq[1]), "+v"(vcq[1]), "+v"(vq[2]), "+v"(vcq[2]) :: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" : "+v"(vq[0]), "+v"(vcq[0]), "+v"(vq[1]), "+v"(vcq[1]), "+v"(vq[2]), "+v"(vcq[2]) :: "memory");
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 3; ++i) if (lv[i]) rowc1_finish(ca, b, tt[i], pm, vq[i], vcq[i], s_sh, lane);
    }
    arrived += 16u;
    stamp();                                                           // C_1's cone rows done
    int tid_b = tid;
    asm volatile("; xcone: thread index, opaque (phase B)" : "+v"(tid_b));
    const int lane_b = tid_b & 63, wave_b = tid_b >> 6;
    const RowHc2Ln ln = rowhc2_ln(cb, lane_b);
    xcone_barrier(bar, grp, xcc, arrived, p.err, s_go != 0);             // C_1's rows and scalars of the team's utterances are in this XCD's L2 (the barrier drains vmcnt and synchronises: the staged pieces are in LDS)
    stamp();
    {
      // three rows per wave, one at a time: with two rows' operands in flight (2 x 43 registers) the kernel, which lives at the 256-register limit, spills
      // The newest row of the C1Q . W2 cache (time frame - 1) comes from passengers of the chain's launch one piece earlier.  Only HC_2's newest cone row and the
      // presum row read it (taps reach back, never forward): the wave that owns them polls the passengers' counter (bounded) -- not the team (rounds 3-4 polled at the
      // END of rowc1_kernel's launch, where the row was always there; at the head of this launch the leader waited 3.4 us for it with the whole team behind it)
      if (ca.wait && s_go) {
        const int rfirst = q4 * 8 + wave_b;
        // row 0 (t = frame - 1) / the presum row R - 1 (t = frame: its taps read frame - 1 and frame - 2), whichever of this wave's three slots it sits in
        // (R - 1 < 96: the host folds the row phases only then, v3_xcone_table)
        if (rfirst == 0 || (rfirst <= cb.R - 1 && ((cb.R - 1 - rfirst) & 31) == 0)) {
          bool ok = false;
          for (int i = 0; i < (1 << 20) && !ok; ++i) {
            ok = __hip_atomic_load(ca.wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ca.wait_val;
            if (!ok) __builtin_amdgcn_s_sleep(8);
          }
          if (!ok && lane_b == 0) atomicOr(ca.wait_err, 1);
        }
      }
#pragma unroll 1
      for (int i = 0; i < 3; ++i) {
        RowHc2Row cur;
        rowhc2_load(cb, b, q4 * 8 + wave_b + 32 * i, lane_b, cur);
        if (uok && cur.live) rowhc2_finish(cb, b, pm, cur, ln, s_c, s_v, lane_b);
      }
    }
    arrived += 16u;
    stamp();                                                           // HC_2's cone rows done
    xcone_barrier(bar, grp, xcc, arrived, p.err, s_go != 0);             // HC_2's rows: the input of the first GEMM layer below
    stamp();
  }
  // Everything the GEMM layers derive from the thread index is derived HERE, from a value the compiler cannot see through: hoisted to the kernel's entry, the ~40
  // per-lane offsets and 64-bit weight addresses lived in registers across the row phases above and the kernel spilled (600 bytes of scratch; xtail_kernel.h has the same trick)
  int tid_o = tid;
  asm volatile("; xcone: thread index, opaque" : "+v"(tid_o));
  const int lane = tid_o & 63, wave = tid_o >> 6;
  const int arow = lane & 15, aq = lane >> 4, c4 = aq * 4;
  const int etile = wave >> 2, ecol = lane & 15, ej = wave & 3;
  const int pcol = etile * 256 + grp * 16 + ecol;
  // this workgroup's slice of a layer's weights, two column tiles.  Wave w owns the SIX CONSECUTIVE k-groups 6 w .. 6 w + 5 (k = 96 w .. 96 w + 95 of
  // the 768 = 3 taps x 256 channels), not w, w + 8, ...: its six A requests per row then cover 384 contiguous bytes = three whole 128-byte
  // lines.  (With one 64-byte piece per row and request, in-kernel stamps showed a pass's 96 KB of rows taking ~4 us to land: ~25 GB/s per CU.)
  f32x4 bq0[6], bq1[6];
  auto load_w = [&](int layer, f32x4 (&q0)[6], f32x4 (&q1)[6]) {
    const float* wb = p.lay[layer].wp + lane * 4;
    const unsigned w0 = (unsigned)(grp * 2) * 48u * 256u, w1 = w0 + 48u * 256u;
#pragma unroll
    for (int i = 0; i < 6; ++i) { q0[i] = ldv(wb, w0 + (unsigned)(6 * wave + i) * 256u); q1[i] = ldv(wb, w1 + (unsigned)(6 * wave + i) * 256u); }
  };
  load_w(0, bq0, bq1);
  for (int li = 0; li < p.L; ++li) {
    {                                      // ... and a layer's descriptor in one batch (lazily: three dependent batches at the top of every layer, more in the row pass)
      typedef const __attribute__((address_space(4))) XConeLayer CL;
      CL& y = p.lay[li];
      asm volatile("; xcone: a layer's descriptor, one batch" :: "s"(y.bias), "s"(y.g1), "s"(y.b1), "s"(y.g2), "s"(y.b2), "s"(y.xin), "s"(y.xin_bstride), "s"(y.xin_row0),
                   "s"(y.xin_stride), "s"(y.xout), "s"(y.xout_bstride), "s"(y.xout_row0), "s"(y.xout_stride), "s"(y.pout), "s"(y.offs), "s"(y.R),
                   "s"(y.tap_off[0]), "s"(y.tap_off[1]));
    }
    const int R = p.lay[li].R, M = nb * R, ntile = (M + 15) >> 4;
    const float bias = p.lay[li].bias[pcol];
    const float* xin = p.lay[li].xin;
    const long xbs = p.lay[li].xin_bstride, xr0 = p.lay[li].xin_row0; const int xs = p.lay[li].xin_stride;
    const int to0 = p.lay[li].tap_off[0] * xs, to1 = p.lay[li].tap_off[1] * xs;        // (tap 2 is the row itself: causal)
    float* pout = p.lay[li].pout;
    const int xsafe = (int)(((long)b0 * xbs + xr0) * xs);
    // ---- row tables of the layer (the two integer divisions and the offset-table read happen once per row, not once per tile and lane)
    if (tid < 256) {
      int xo = -1, pr = 0;
      if (tid < M) {
        const int bl = tid / R, r = tid - bl * R;
        const int t = frame + p.lay[li].offs[r];
        if (t >= 0) xo = (int)(((long)(b0 + bl) * xbs + xr0 + t) * xs);
        pr = ((b0 + bl) * R + r) | ((r == R - 1) ? (1 << 30) : 0);                       // bit 30: presum row (its centre tap belongs to the chain)
      }
      s_xoff[tid] = xo; s_prow[tid] = pr;
    }
    __syncthreads();
    // A fragments of a row tile, RAW: lane (arow, aq) requests row tile * 16 + arow, channels c .. c + 3 of k-group (wave, i).  Rows that do
    // not exist / a presum row's centre tap must read as zero: the request goes to a readable address and `fix_a` zeroes the registers --
    // LATER, where the values are consumed: a select right behind a load is a use, and the wait for it would sit in front of the MFMAs
    // the load is supposed to hide behind (first version: vmcnt(5) .. vmcnt(0) right after the six requests, 2.7 us per tile instead of 1.3).
    auto load_a = [&](int tile, f32x4 (&a)[6], int& flags) {
      const int m = tile * 16 + arow;
      const int mi = m < 256 ? m : 255;
      const int xo = s_xoff[mi];
      const bool ok = m < M && xo >= 0;
      flags = (ok ? 1 : 0) | (((s_prow[mi] >> 30) & 1) ? 2 : 0);
      const float* xrow = xin + (ok ? xo : xsafe) + c4;                       // (a row that does not exist reads this team's first utterance, time 0: the taps reach back
                                                                              //  into that utterance's own zero rows, never in front of the buffer)
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int g = 6 * wave + i, tap = g >> 4;                              // wave-uniform
        const int toff = (tap == 0) ? to0 : ((tap == 1) ? to1 : 0);
        a[i] = *reinterpret_cast<const f32x4*>(xrow + toff + 16 * (g & 15));
      }
    };
    auto fix_a = [&](f32x4 (&a)[6], const f32x4 (&raw)[6], int flags) {
#pragma unroll
      for (int i = 0; i < 6; ++i) a[i] = (!(flags & 1) || ((flags & 2) && ((6 * wave + i) >> 4) == 2)) ? z4 : raw[i];
    };
    // Two row tiles per pass: their 96 MFMAs per wave go back to back, then ONE barrier and one fixed-order reduction for both (in-kernel stamps:
    // with one tile per pass a tile cost 2.9 us, 1.3 of them MFMA; the rest -- LDS round trip, barrier, address set-up -- is paid per pass).
    // Round 4: software-pipelined by one pass -- the split-K reduction of pass k - 1 (LDS reads, adds, the pre-norm stores) sits in the same straight-line
    // code as the MFMAs of pass k, so the matrix pipe works while the other wave of the SIMD (and this wave's own vector instructions) reduce: with
    // "MFMAs, barrier, reduce" per pass a pass took 4.5-5 us, 2.56 of them MFMA on a SIMD shared by two waves (stamps).  Still one barrier per pass: pass k
    // writes red[k & 1] while pass k - 1 is read from red[(k - 1) & 1], and pass k + 1 overwrites that copy only behind barrier k.
    f32x4 a0[6], a1[6], n0[6], n1[6];
    int f0 = 0, f1 = 0;
    load_a(0, n0, f0); fix_a(a0, n0, f0);
    if (ntile > 1) { load_a(1, n1, f1); fix_a(a1, n1, f1); }
    auto reduce_store = [&](const int tile) {                                  // finish the pass that started at row tile `tile`
      const float* rb = red[(tile >> 1) & 1];
      float v0 = bias, v1 = bias;
#pragma unroll
      for (int w = 0; w < 8; ++w) { v0 += rb[((w * 2 + etile) * 4 + ej) * 64 + lane]; v1 += rb[4096 + ((w * 2 + etile) * 4 + ej) * 64 + lane]; }
      const int me = tile * 16 + aq * 4 + ej;                                 // the rows this thread finishes
      if (me < M && s_xoff[me] >= 0) pout[(long)(s_prow[me] & 0x3fffffff) * 512 + pcol] = v0;
      if (tile + 1 < ntile && me + 16 < M && s_xoff[me + 16] >= 0) pout[(long)(s_prow[me + 16] & 0x3fffffff) * 512 + pcol] = v1;
    };
    for (int tile = 0; tile < ntile; tile += 2) {
      const bool two = tile + 1 < ntile;                                      // uniform
      if (tile + 2 < ntile) load_a(tile + 2, n0, f0);                         // the next pass's rows are in flight during this pass's MFMAs
      if (tile + 3 < ntile) load_a(tile + 3, n1, f1);
      f32x4 acc0 = z4, acc1 = z4, acc2 = z4, acc3 = z4;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i][e], bq0[i][e], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i][e], bq1[i][e], acc1, 0, 0, 0);
        }
      }
      if (tile > 0) reduce_store(tile - 2);                                   // (between the two tiles' MFMAs: nothing here depends on them)
      if (two) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i][e], bq0[i][e], acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i][e], bq1[i][e], acc3, 0, 0, 0);
          }
        }
      }
      float* rb = red[(tile >> 1) & 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        rb[((wave * 2 + 0) * 4 + j) * 64 + lane] = acc0[j]; rb[((wave * 2 + 1) * 4 + j) * 64 + lane] = acc1[j];
        rb[4096 + ((wave * 2 + 0) * 4 + j) * 64 + lane] = acc2[j]; rb[4096 + ((wave * 2 + 1) * 4 + j) * 64 + lane] = acc3[j];
      }
      lds_barrier();                                                           // LDS only: the next pass's loads keep flying
      fix_a(a0, n0, f0); fix_a(a1, n1, f1);
    }
    reduce_store((ntile - 1) & ~1);                                            // the last pass
    // ---- the team's pre-norm rows of this layer are complete
    stamp();                                                                   // contraction done
    arrived += 16u;
    xcone_barrier(bar, grp, xcc, arrived, p.err, s_go != 0);
    stamp();                                                                   // barrier passed
    const bool lastl = li + 1 == p.L;
    if (lastl && !p.tail_rows) break;                                          // the last layer only leaves its presum rows (unless its cone rows are somebody's input)
    if (!lastl) load_w(li + 1, bq0, bq1);                                      // the next layer's slice lands while this layer's rows are normalised
    // ---- layer-norm / gate / highway mix of the cone rows (offsets < 0): one wave per row, the team's 128 waves in turn
    const int Rb = R - 1;
    if (Rb > 0) {
      float* xout = p.lay[li].xout;
      const long obs = p.lay[li].xout_bstride, or0 = p.lay[li].xout_row0; const int os = p.lay[li].xout_stride;
      const int c = lane * 4;
      // the layer's layer-norm parameters once, with the first row's requests (inside norm_hc_regs they were a second round trip per row)
      const float4 g1 = ld4(p.lay[li].g1 + c), be1 = ld4(p.lay[li].b1 + c), g2 = ld4(p.lay[li].g2 + c), be2 = ld4(p.lay[li].b2 + c);
      for (int q = grp * 8 + wave; q < nb * Rb; q += 128) {
        const int bl = q / Rb, r = q - bl * Rb;
        const int m = bl * R + r;
        const int xo = s_xoff[m];
        if (xo < 0) continue;                                                  // wave-uniform: the row does not exist yet (t < 0)
        const long prow = (long)(s_prow[m] & 0x3fffffff);
        const float4 h1 = ld4(pout + prow * 512 + c), h2 = ld4(pout + prow * 512 + 256 + c);
        const float4 xr = ld4(xin + xo + c);                                   // the layer's own input row (modules.py:171,193)
        const float4 o = norm_hc_vals(h1, h2, xr, g1, be1, g2, be2);
        const int t = (int)((long)xo / xs - ((long)(b0 + bl) * xbs + xr0));
        *reinterpret_cast<float4*>(xout + ((long)(b0 + bl) * obs + or0 + t) * os + c) = o;
      }
    }
    stamp();                                                                   // row pass done
    arrived += 16u;
    xcone_barrier(bar, grp, xcc, arrived, p.err, s_go != 0);
    stamp();                                                                   // barrier passed
  }
  }                                        // next utterance group of this team (the barrier count simply runs on)
  // ---- the team's rows are in this XCD's L2 (every team-mate has passed the last barrier behind its stores): write them back, count the
  // team, and let the last team publish the piece to the chain's stream, which polls `sig` in its next launch (chain3_kernel / xgroup_kernel: wait2)
  if (p.done && grp == 0 && tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned old = __hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // sc1: the teams are on different XCDs, this one must not stay in an L2
    if (old + 1u == p.done_target) __hip_atomic_store(p.sig, p.sig_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (p.wait && s_go) {
      // the chain writes that value when its piece `frame` STARTS, i.e. before it waits for this launch: no cycle; normally it is there already
      bool ok = false;
      for (int i = 0; i < (1 << 20) && !ok; ++i) {
        ok = __hip_atomic_load(p.wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= p.wait_val;
        if (!ok) __builtin_amdgcn_s_sleep(2);                                 // (eight pollers: the next piece of this stream starts when they see it)
      }
      if (!ok) atomicOr(p.wait_err, 1);
    }
  }
}

}  // namespace dctts
