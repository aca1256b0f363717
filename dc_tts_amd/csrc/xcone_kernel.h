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
  int fold; int U;                       // U: utterances per team and round (xgroup_kernel.h: XGroupParams::U; 0 = 4)
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
  // Round 6: a layer's weight slice lives in LDS (96 KB: the workgroup's two 16-column tiles x 48 k-groups, in the packing's own order: a k-group of a tile is the
  // 1 KB one ds_read_b128 per lane fetches), and a wave contracts WHOLE row tiles against it -- see the GEMM below.  `aux`: the row phases' staged operands (45 KB),
  // later the second K halves' partial sums of a layer with four row tiles or fewer (8 KB).
  __shared__ __attribute__((aligned(16))) float wlds[2 * 48 * 256];
  __shared__ __attribute__((aligned(16))) float aux[45 * 256];
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
  const int U = p.U ? p.U : 4;
  if (team * U >= p.B) return;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  unsigned* const bar = p.bar + team * 32;
  const unsigned xcc = xg_xcc_id();
  if (tid == 0) s_go = __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;     // an earlier launch of this decode already failed: no more waiting
  unsigned arrived = p.bar_base;
  const int frame = p.frame;
  int nts = 0;
  auto stamp = [&]() { if (p.ts && blockIdx.x == 0 && tid == 0 && nts < 60) { p.ts[100 + (nts > 0)] = clock64(); p.ts[nts++] = wall_clock64(); } };   // ([100], [101]: shader clock at the first / latest stamp)
  stamp();

  // a layer's weight slice, global memory -> LDS without a register in between (global_load_lds_dwordx4: a wave's 64 lanes x 16 bytes land at consecutive LDS
  // addresses behind the wave-uniform base).  The slice is contiguous in the packing: tiles 2 grp and 2 grp + 1, 48 k-groups of 1 KB each.
  // (a wave's 12 one-KB pieces as two halves: the first layer's slice is requested half in each row phase -- a wave's requests return in order and every team
  //  barrier drains them, so a whole slice behind phase A's operands held that phase's barrier for ~4 us)
  auto load_slice = [&](const int layer, const int wv, const int k0 = 0, const int k1 = 12) {
    const float* src = p.lay[layer].wp + (unsigned)(grp * 2) * 48u * 256u + (unsigned)(tid & 63) * 4u;
    const int w12 = __builtin_amdgcn_readfirstlane(wv) * 12;
#pragma unroll
    for (int k = 0; k < 12; ++k)
      if (k >= k0 && k < k1)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (w12 + k) * 256), (__attribute__((address_space(3))) void*)&wlds[(w12 + k) * 256], 16, 0, 0);
  };
  for (int b0 = team * U; b0 < p.B; b0 += 8 * U) {
  const int nb = (p.B - b0 < U) ? p.B - b0 : U;
  if (p.fold) {
    int tid_a = tid;                                                     // (opaque per phase, like tid_o below: what is derived from the thread index stays inside its phase)
    asm volatile("; xcone: thread index, opaque (phase A)" : "+v"(tid_a));
    const int lane = tid_a & 63, wave = tid_a >> 6;
    // ---- phase A: AudioDec C_1 over its cone rows (decode3_kernels.h: rowc1_*), phase B: HC_2 over its cone rows + its presum row (rowhc2_*)
    const RowC1Params& ca = pp->rc1; const RowHc2Params& cb = pp->rhc2;
    f32x4* const s_c = reinterpret_cast<f32x4*>(&aux[0]);              // phase B's shared operands: 18 KB + 18 KB (nothing else uses `aux` before the first GEMM layer)
    f32x4* const s_v = s_c + 9 * 128;
    f32x4* const s_sh = s_v + 9 * 128;                                  // phase A's: 9 KB behind them
    // (U utterances share the team's 16 workgroups: 16 / U of them per utterance, each wave a row every 128 / U rows -- U = 4: rows (grp % 4) * 8 + wave + 32 i,
    //  three per wave; U = 1: one row per wave of all sixteen workgroups)
    const int wpu = 16 / U, ub = grp / wpu, q4 = grp - ub * wpu, rstep = 8 * wpu;
    const bool uok = ub < nb;
    const int b = uok ? b0 + ub : b0;
    const int pm = pm_row[b];
    auto ldq = [](const float* q_) { return *reinterpret_cast<const f32x4*>(q_); };
    __syncthreads();                                                    // (a second round: the previous round's last reads of `aux`)
    rowc1_stage(ca, b, pm, s_sh, tid_a, 512);
    {
      f32x4 vq[3], vcq[3]; int tt[3]; bool lv[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int r0 = q4 * 8 + wave + rstep * i;
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
      // ... and neither does the first GEMM layer's weight slice (round 6: 96 KB per workgroup into `wlds`, 12 one-KB pieces per wave): its first half is requested
      // behind them and drained by this phase's barrier, the second half at the start of phase B
      load_slice(0, wave, 0, 6);
      if (wave < 2) asm volatile("s_waitcnt vmcnt(12)" : "+v"(vq[0]), "+v"(vcq[0]), "+v"(vq[1]), "+v"(vcq[1]), "+v"(vq[2]), "+v"(vcq[2]) :: "memory");
      else asm volatile("s_waitcnt vmcnt(10)" : "+v"(vq[0]), "+v"(vcq[0]), "+v"(vq[1]), "+v"(vcq[1]), "+v"(vq[2]), "+v"(vcq[2]) :: "memory");
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
        if (rfirst == 0 || (rfirst <= cb.R - 1 && (cb.R - 1 - rfirst) % rstep == 0 && (cb.R - 1 - rfirst) / rstep < 3)) {
          bool ok = false;
          for (int i = 0; i < (1 << 20) && !ok; ++i) {
            ok = __hip_atomic_load(ca.wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ca.wait_val;
            if (!ok) __builtin_amdgcn_s_sleep(8);
          }
          if (!ok && lane_b == 0) atomicOr(ca.wait_err, 1);
        }
      }
      // The slice's second half: requested HERE, in straight-line code in front of the rows, drained by the barrier behind this phase.  NOT inside the row loop
      // below (behind the last row's operands, where it would cost that row nothing): in that place -- one iteration of a loop the compiler keeps rolled -- the
      // FIRST decodes at a new geometry came out wrong in a quarter of the cases (tools/flaky_probe.py: 7 of 32, with the slice re-requested in front of the
      // GEMM as well, i.e. not through the LDS copy), none in 32 with the request outside the loop or whole in phase A.
      load_slice(0, wave_b, 6, 12);
#pragma unroll 1
      for (int i = 0; i < 3; ++i) {
        RowHc2Row cur;
        rowhc2_load(cb, b, q4 * 8 + wave_b + rstep * i, lane_b, cur);
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
  const int ecol = lane & 15;
  // Round 6, the GEMM layers.  Rounds 3-5 kept the workgroup's weight slice in REGISTERS, split K over the eight waves and reduced every pass of two row tiles
  // through LDS behind a barrier: a pass took ~5.1 us of which 2.56 were MFMA (stamps: HC_3's 12 row tiles 30.6 us, 0.47 of the matrix pipe of the CUs the kernel
  // owns).  Now the slice is in LDS (`wlds`, see load_slice) and a wave contracts a WHOLE row tile x the workgroup's 32 columns x all of K = 768 on its own: A from
  // global memory (the XCD's L2) through a ring of eight k-groups in registers, B as two ds_read_b128 per k-group (1 KB each = four MFMAs' operands; 32 LDS cycles
  // per 256 cycles of matrix pipe over the four SIMDs), no split-K exchange, no barrier inside a layer's contraction.  A layer with four row tiles or fewer (HC_4;
  // small batches) splits every tile's K into its two halves over a pair of waves, so that eight waves have work: the second half goes through LDS once.
  // Summation order of a pre-norm value, whatever the form: (bias + sum over k-groups 0 .. 23) + sum over k-groups 24 .. 47, each sum one MFMA accumulator chain
  // -- results do not depend on how many utterances a team serves (shard-invariant, bitwise).
  if (!p.fold) { __syncthreads(); load_slice(0, wave); }      // (the split launch form: nothing in front of the GEMM layers to hide the request behind)
  float* const wl = wlds + lane * 4;
  for (int li = 0; li < p.L; ++li) {
    {                                      // ... and a layer's descriptor in one batch (lazily: three dependent batches at the top of every layer, more in the row pass)
      typedef const __attribute__((address_space(4))) XConeLayer CL;
      CL& y = p.lay[li];
      asm volatile("; xcone: a layer's descriptor, one batch" :: "s"(y.bias), "s"(y.g1), "s"(y.b1), "s"(y.g2), "s"(y.b2), "s"(y.xin), "s"(y.xin_bstride), "s"(y.xin_row0),
                   "s"(y.xin_stride), "s"(y.xout), "s"(y.xout_bstride), "s"(y.xout_row0), "s"(y.xout_stride), "s"(y.pout), "s"(y.offs), "s"(y.R),
                   "s"(y.tap_off[0]), "s"(y.tap_off[1]));
    }
    const int R = p.lay[li].R, M = nb * R, ntile = (M + 15) >> 4;
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // (the split launch form's first layer: this wave's pieces of the weight slice are in LDS; otherwise drained by the barrier above)
    __syncthreads();
    // ---- the contraction.  unit = (row tile, K half): 24 k-groups of 16 channels.  The layer's ntile = 8 a + b row tiles: wave w takes the tiles w, w + 8, ... below
    //      8 a, both halves of each (finished from registers); of the b left over, when b <= 4, a PAIR of waves takes a tile -- wave w half w & 1 of tile 8 a + (w >> 1),
    //      the second half goes through LDS once -- so that HC_3's 12 tiles are three units for every wave (two waves busy on every SIMD to the end) and a layer of
    //      four tiles or fewer (HC_4, HC_5; small batches) still has work for up to eight waves; b > 4: one tile per wave again.
    //      A request: lane (arow, aq) reads row tile * 16 + arow, channels 4 aq .. 4 aq + 3 of the
    //      k-group (a row that does not exist reads a readable address and its result is never stored: MFMA rows are independent; a presum row's centre tap must
    //      contribute zero: selected where the values are consumed).
    const int ta = ntile >> 3, tb = ntile & 7;
    const bool split = tb > 0 && tb <= 4;                                                      // (uniform per workgroup)
    const int nu_tail = tb == 0 ? 0 : (split ? (wave < 2 * tb ? 1 : 0) : (wave < tb ? 2 : 0));
    const int nu = 2 * ta + nu_tail;                                                           // this wave's units
    const float bias0 = p.lay[li].bias[grp * 16 + ecol], bias1 = p.lay[li].bias[256 + grp * 16 + ecol];      // this lane's gate / info column
    float* const part = aux;                                                                   // split tiles: [tile b][2 tiles of columns][4][64]
    auto unit_tile = [&](int u) { return u < 2 * ta ? wave + 8 * (u >> 1) : 8 * ta + (split ? (wave >> 1) : wave); };
    auto unit_half = [&](int u) { return (u >= 2 * ta && split) ? (wave & 1) : (u & 1); };
    // (BYTE offsets from the uniform base xin, 32 bits: the request then takes the scalar-base + 32-bit-lane-offset form; with element offsets every request carried a
    //  64-bit shift-and-add, the product could overflow 32 bits for all the compiler knows)
    auto ldb = [](const float* base, unsigned boff) { return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + boff); };
    auto row_off = [&](int tile, int& flag2) {                  // byte offset of this lane's A row (+ its channel quad) in xin
      const int m = tile * 16 + arow;
      const int mi = m < 256 ? m : 255;
      const int xo = s_xoff[mi];
      const bool ok = m < M && xo >= 0;
      flag2 = (ok && ((s_prow[mi] >> 30) & 1)) ? 1 : 0;
      return (unsigned)((ok ? xo : xsafe) + c4) * 4u;
    };
    // chunk = eight consecutive k-groups (never across a tap: 24 h + 8 c is a multiple of 8): its tap and first channel group
    auto chunk_off = [&](int h, int c) { const int g = 24 * h + 8 * c, tap = g >> 4; return (unsigned)(((tap == 0) ? to0 : ((tap == 1) ? to1 : 0)) + 16 * (g & 15)) * 4u; };
    f32x4 acc0 = z4, acc1 = z4, s0 = z4, s1 = z4;
    if (nu > 0) {
      f32x4 ring[8];
      int fl_cur = 0, fl_nxt = 0;
      unsigned off_cur = row_off(unit_tile(0), fl_cur), off_nxt = off_cur;
      {
        const unsigned o = off_cur + chunk_off(unit_half(0), 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) ring[i] = ldb(xin, o + 64u * i);
      }
      if (p.ts && li == 0 && blockIdx.x == 0 && lane == 0) p.ts[60 + wave * 5] = wall_clock64();                      // (first requests out)
      for (int u = 0; u < nu; ++u) {
        const int tile = unit_tile(u), h = unit_half(u);
        const bool more = u + 1 < nu;
        const int hn = more ? unit_half(u + 1) : h;
        if (more) off_nxt = row_off(unit_tile(u + 1), fl_nxt); else { off_nxt = off_cur; fl_nxt = fl_cur; }      // (the last unit re-requests its own first chunk: no branch around a load)
        acc0 = z4; acc1 = z4;
        // (software-pipelined by hand and pinned: the B operands of k-group n + 1 are requested from LDS in front of the MFMAs of k-group n, the A slot is refilled
        //  behind them, and a scheduling barrier keeps the compiler from sinking either request to its use -- left alone it served the ring one load at a time.
        //  What the form reaches: tools/micro/mfma_feed_lab.hip -- 38 cycles per MFMA per SIMD against 32 with register operands, one or two waves per SIMD alike; in
        //  this kernel ~40 (per-wave unit stamps, DCTTS_TRACE).  Spreading the three requests behind single MFMAs (sched_group_barrier) gives 34.7 in the lab and nothing
        //  here; the requests' coalescing is not it either: with every quad of lanes on one 64-byte line, and with every request on the same line, the layer took as long)
        f32x4 b0 = *reinterpret_cast<const f32x4*>(wl + (24 * h) * 256), b1 = *reinterpret_cast<const f32x4*>(wl + (48 + 24 * h) * 256);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int g0 = 24 * h + 8 * c;
          const bool zero2 = (g0 >= 32) && fl_cur;                                            // the centre tap of a presum row belongs to the chain
          const unsigned onext = (c < 2) ? off_cur + chunk_off(h, c + 1) : off_nxt + chunk_off(hn, 0);
          const float* wg = wl + g0 * 256;
          const float* wgn = (c < 2) ? wg + 8 * 256 : wl + (24 * hn) * 256;                  // the k-group behind this chunk's last one: the next chunk's / the next unit's first
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float* wn = (i < 7) ? wg + (i + 1) * 256 : wgn;
            const f32x4 nb0 = *reinterpret_cast<const f32x4*>(wn), nb1 = *reinterpret_cast<const f32x4*>(wn + 48 * 256);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 a = ring[i];
            if (zero2) a = z4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b0[e], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b1[e], acc1, 0, 0, 0);
            }
            ring[i] = ldb(xin, onext + 64u * i);                                              // the slot is refilled behind the MFMAs that read it: eight k-groups ahead
            __builtin_amdgcn_sched_barrier(0);
            b0 = nb0; b1 = nb1;
          }
        }
        if (!(split && u >= 2 * ta)) {                                                        // a tile this wave finishes on its own
          if (h == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { s0[j] = bias0 + acc0[j]; s1[j] = bias1 + acc1[j]; }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int me = tile * 16 + aq * 4 + j;                                            // D layout of the 16 x 16 tile: lane (column ecol, row group aq), register j = row 4 aq + j
              const int mi = me < 256 ? me : 255;
              if (me < M && s_xoff[mi] >= 0) {
                float* o = pout + (long)(s_prow[mi] & 0x3fffffff) * 512 + grp * 16 + ecol;
                o[0] = s0[j] + acc0[j]; o[256] = s1[j] + acc1[j];
              }
            }
          }
        }
        off_cur = off_nxt; fl_cur = fl_nxt;
        if (p.ts && li == 0 && blockIdx.x == 0 && lane == 0 && u < 4) p.ts[61 + wave * 5 + u] = wall_clock64();      // (DCTTS_TRACE: every wave's unit ends in the first GEMM layer)
      }
    }
    if (split) {                                                                               // the tiles a pair of waves shares: (bias + first half) + second half, as everywhere
      const int tb_i = wave >> 1, tile = 8 * ta + tb_i;
      if (nu_tail > 0 && (wave & 1)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { part[((tb_i * 2 + 0) * 4 + j) * 64 + lane] = acc0[j]; part[((tb_i * 2 + 1) * 4 + j) * 64 + lane] = acc1[j]; }
      }
      lds_barrier();
      if (nu_tail > 0 && !(wave & 1)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int me = tile * 16 + aq * 4 + j;
          const int mi = me < 256 ? me : 255;
          if (me < M && s_xoff[mi] >= 0) {
            float* o = pout + (long)(s_prow[mi] & 0x3fffffff) * 512 + grp * 16 + ecol;
            o[0] = (bias0 + acc0[j]) + part[((tb_i * 2 + 0) * 4 + j) * 64 + lane];
            o[256] = (bias1 + acc1[j]) + part[((tb_i * 2 + 1) * 4 + j) * 64 + lane];
          }
        }
      }
    }
    // ---- the team's pre-norm rows of this layer are complete
    stamp();                                                                   // contraction done
    arrived += 16u;
    xcone_barrier(bar, grp, xcc, arrived, p.err, s_go != 0);
    stamp();                                                                   // barrier passed
    const bool lastl = li + 1 == p.L;
    if (lastl && !p.tail_rows) break;                                          // the last layer only leaves its presum rows (unless its cone rows are somebody's input)
    // ---- layer-norm / gate / highway mix of the cone rows (offsets < 0): one wave per row, the team's 128 waves in turn.  The next layer's weight slice is requested
    //      BEHIND the wave's first row's operands (a wave's requests return in order: in front of them the row would wait for 12 KB of weights): every wave is past
    //      the barrier behind this layer's last LDS read, and the barrier behind the row pass drains the request.
    const int Rb = R - 1;
    {
      float* xout = p.lay[li].xout;
      const long obs = p.lay[li].xout_bstride, or0 = p.lay[li].xout_row0; const int os = p.lay[li].xout_stride;
      const int c = lane * 4;
      // the layer's layer-norm parameters once, with the first row's requests (inside norm_hc_regs they were a second round trip per row)
      const float4 g1 = ld4(p.lay[li].g1 + c), be1 = ld4(p.lay[li].b1 + c), g2 = ld4(p.lay[li].g2 + c), be2 = ld4(p.lay[li].b2 + c);
      const int nrows = nb * Rb;
      auto row_of = [&](int q, int& bl, int& xo, long& prow) {       // row q of the pass: its utterance, input offset (-1: does not exist), pre-norm row
        const int qq = q < nrows ? q : 0;
        bl = Rb > 0 ? qq / Rb : 0;
        const int m = bl * R + (qq - bl * Rb);
        xo = (q < nrows) ? s_xoff[m] : -1;
        prow = (long)(s_prow[m] & 0x3fffffff);
      };
      int q = grp * 8 + wave, bl, xo; long prow;
      row_of(q, bl, xo, prow);
      float4 h1 = ld4(pout + prow * 512 + c), h2 = ld4(pout + prow * 512 + 256 + c);      // (a row that does not exist: a readable row, never stored)
      float4 xr = ld4(xin + (xo >= 0 ? xo : xsafe) + c);                                    // the layer's own input row (modules.py:171,193)
      if (!lastl) load_slice(li + 1, wave);
      for (;;) {
        if (xo >= 0) {                                                                       // wave-uniform
          const float4 o = norm_hc_vals(h1, h2, xr, g1, be1, g2, be2);
          const int t = (int)((long)xo / xs - ((long)(b0 + bl) * xbs + xr0));
          *reinterpret_cast<float4*>(xout + ((long)(b0 + bl) * obs + or0 + t) * os + c) = o;
        }
        q += 128;
        if (q >= nrows) break;
        row_of(q, bl, xo, prow);
        h1 = ld4(pout + prow * 512 + c); h2 = ld4(pout + prow * 512 + 256 + c);
        xr = ld4(xin + (xo >= 0 ? xo : xsafe) + c);
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
