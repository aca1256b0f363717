// xtail_kernel.h -- the TAIL of AudioDec for one decode frame on the chain's stream (round 4): the last three highway layers HC_5 .. HC_7
// (networks.py:175-191) over the FEW rows of the dependency cone they need, then the seven k = 1 layers of xmlp_kernel.h, as one launch in team form.
//
// Why.  Exact parity with synthesize.py:47-54 makes every frame re-evaluate AudioDec's dependency cone with the frame's attention window
// (85 / 83 / 45 / 15 / 5 / 3 / 1 rows per utterance for C_1 .. HC_7).  Rounds 2-3 put ALL cone rows at offsets < 0 on the side stream and only the
// newest row of every layer on the chain; with the team kernels the side stream became the longer one (99 against ~72 us of chain work per frame once
// the k = 1 layers ran in team form), and 23 of its 99 us were the three LAST layers, whose cone is 5, 3 and 1 rows per utterance: latency
// (barrier + row pass + a cold start per layer), not arithmetic.  Those rows fit the chain's MFMA tiles for free -- a team owns four utterances,
// i.e. 4 of the 16 rows of a tile -- so here the chain computes them itself:
//     HC_5 (dilation 27):  rows t .. t-4 of every utterance, K = 768 (all three taps): inputs are HC_4's rows at (t - r) - {0, 27, 54}:
//                          the 14 cone rows the side stream's xcone_kernel leaves (its LAST layer now) + the chain's own newest row
//     HC_6 (dilation 1):   rows t .. t-2, inputs = the five HC_5 rows above
//     HC_7 (dilation 1):   row t, inputs = the three HC_6 rows above
// and the side stream stops behind HC_4.  Nothing about the arithmetic changes: every row is the same contraction, two layer-norms, gate and
// highway mix (modules.py:183-193) as in xcone_kernel / xgroup_kernel; rows that were "presum + centre tap" are now one K = 768 contraction.
//
// Merged form (np == 3, the default since the middle of round 4).  The launch in front of this one was xgroup_kernel's AudioDec run: the NEWEST row of HC_2,
// HC_3 and HC_4 (K = 256: the centre tap; the older taps come from the side stream as a presum), whose last row this kernel then re-read from memory.  Those
// three layers now run in FRONT of the cone layers in this launch -- same arithmetic (xgroup_kernel's layer loop: M = 4 rows, A operand in registers, compact
// rebuild) -- so a chain piece was two launches instead of three (round 5: ONE, xchain_kernel at the end of this file), the row stays in LDS, the first cone layer's weight slice (96 KB per workgroup) is in
// flight while the newest-row layers wait for each other, and the cone rows are staged between them.  The launch then also carries what the AudioDec launch
// carried: the chain's "piece complete" signal, the wait for the side stream, and the passenger workgroups (xgroup_kernel.h).  86.5 -> 83.1 us per frame;
// with the next cone layer's slice pulled into the L2 one layer ahead (one dword per line: the slice then arrives in ~0.7 us instead of 2-3) 82.1.
//
// Team form exactly as xgroup_kernel.h / xmlp_kernel.h: 16 workgroups on one XCD own four utterances, workgroup `grp` owns a (gate, info) pair of
// 16-column tiles, K is split over the 8 waves (six consecutive k-groups of 16 each), partial sums meet in LDS in a fixed order, pre-norm slices +
// partial layer-norm statistics are published with plain stores, the team passes its flag-word barrier, and EVERY workgroup rebuilds all rows of
// the layer's output (M <= 20 rows x 256 channels) into its own LDS, where the next layer's A operand is read from.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "xmlp_kernel.h"

namespace dctts {

struct XTailHc {
  const float* wp;                       // 16-column tiles, all three taps: [tile = 2 grp + h][48 k-groups][lane][4]
  const float* bias;                     // [512]
  const float* g1; const float* b1; const float* g2; const float* b2;
  int nin, nout, ts, pad_;               // rows per utterance in / out; the input row of (output row r, tap) is r + (2 - tap) * ts
};
struct XTailP {                          // a newest-row layer of the merged form
  const float* wp;                       // centre tap, 16-column tiles: [tile = 2 grp + h][16 k-groups][lane][4]
  const float* presum; int presum_bs; int pad_;     // bias + older taps of row b at presum + b * presum_bs (written by the side stream)
  const float* g1; const float* b1; const float* g2; const float* b2;
};
struct XTailParams {
  XMlpParams m;                          // the k = 1 layers; m.P0 / stats0 / g1 .. b2 / res describe the producer of the FIRST highway layer's newest input row (HC_4)
  XTailHc hc[3]; int nh; int nin0;       // highway layers; input rows per utterance of the first one
  int frame; int U;                      // utterances per team and round (xgroup_kernel.h: XGroupParams::U; 0 = 4); the frame t of the newest row: output row r of a layer is time t - r, and rows in front of t = 0 do not exist -- they are the ZERO padding
                                         // of the next layer's input (modules.py:173-177 pads every layer's input), not a layer evaluated on padding
  const float* xin; long xin_bs; int xin_stride; int pad0;   // the first layer's other input rows come from the side stream's buffer: row of time t of utterance b at xin + b * xin_bs
  int in_off[16];                        // ... + in_off[q] * xin_stride for input row q (in_off[0] == 0: the newest row, rebuilt here)
  float* xch; float* sch;                // exchange for the highway layers: [2][groups][20][512] pre-norm rows, [2][groups][20][16][4] statistics
  int xch_set, sch_set;
  // ---- merged form (np == 3): the NEWEST-ROW layers in front of the cone layers -- AudioDec HC_2 .. HC_4 at row t, what xgroup_kernel's AudioDec run computed one launch
  //      earlier (K = 256: the centre tap; the older taps arrive as the side stream's presum), so that their row stays in LDS (and a chain piece was two launches; round 5: one)
  int np; int pad2;
  const float* pP0; const float* pstats0; const float* pg1; const float* pb1;      // C_1's pre-norm rows [b][256], partial statistics [b][16][4], layer-norm parameters
  XTailP pl[4];                          // (round 6, NP = 4: HC_2 .. HC_5)
  float* xchp; float* schp; int xchp_set, schp_set;       // their exchange: [2][B_pad][512] pre-norm rows, [2][B_pad][16][4] statistics (xgroup_kernel's layout)
  unsigned* sig; unsigned sig_val;                        // first launch of a chain piece: *sig = sig_val ("every earlier piece of this stream is complete")
  const unsigned* wait2; unsigned wait_val;               // ... and the presums + cone rows come from the side stream: poll *wait2 >= wait_val first
  // passengers (xgroup_kernel.h): workgroups behind the 128 team workgroups run independent hbulk_body items on compute units nobody is using
  const SplitParams* ptab; int p_blocks, p_ipl, p_step; int p_count_from;
  unsigned* pdone; unsigned pdone_target; unsigned* psig; unsigned psig_val;
  long long* ts;                         // measurement (DCTTS_TRACE, TS instantiation): workgroup 0 / thread 0 records 100 MHz wall-clock stamps at its phase boundaries
};

template <bool DRAINED = false>          // DRAINED: the caller has waited for its stores itself (and has requests in flight that the barrier need not wait for)
__device__ __forceinline__ void team_barrier(unsigned* bar, int grp, unsigned xcc, unsigned target, int* err, bool go) {
  if constexpr (!DRAINED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's stores are in the L2
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const unsigned me = (target << 4) | xcc;
    if (lane == 0) __hip_atomic_store(bar + grp, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (go) {
      int spins = 0;
      for (;;) {
        const unsigned v = lane < 16 ? __hip_atomic_load(bar + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : me;
        const bool there = (int)((v >> 4) - target) >= 0;
        if (__builtin_amdgcn_ballot_w64(there && (v & 15u) != xcc) != 0ull) { if (lane == 0) atomicOr(err, 2); break; }         // a split team
        if (__builtin_amdgcn_ballot_w64(!there) == 0ull) break;
        if (++spins > (1 << 16) || ((spins & 255) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { if (lane == 0) atomicOr(err, 1); break; }
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ f32x4 ld4_sc1(const float* p) {                  // past the L1, served by this XCD's L2 (the team-mates' plain stores)
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}

constexpr int XT_LDR = 260;              // LDS row stride (floats) of the activation rows
constexpr int XT_MAXM = 20;              // rows a team's layer can have (5 per utterance)

// The kernel's body as a function (returns false for a workgroup that has nothing to do behind it: a passenger, a team without utterances)
// NP (merged form): how many newest-row layers run in front of the cone layers -- 3 (rounds 4-5: HC_2 .. HC_4, then HC_5 .. HC_7 over 5 / 3 / 1 rows) or 4 (round 6:
// HC_2 .. HC_5, then HC_6 / HC_7 over 3 / 1 rows: HC_5's four older cone rows and its presum moved to the side stream's xcone_kernel, which has the time since
// its GEMM layers run from LDS-resident weight slices; the chain's HC_5 drops from a K = 768 contraction over 20 rows with 60 staged input rows to a newest-row layer).
template <bool TS = false, int NP = 3>
__device__ __forceinline__ bool xtail_body(const XTailParams* __restrict__ pp) {
  static_assert(NP == 3 || NP == 4, "three or four newest-row layers");
  constexpr int NH = 6 - NP;             // cone layers behind them
  __shared__ __attribute__((aligned(16))) float lds_rows[(60 + XT_MAXM) * XT_LDR];
  float* const bufA = lds_rows;                          // the first layer's input rows (4 x 15), later the second layer's output (4 x 3)
  float* const bufB = lds_rows + 60 * XT_LDR;            // the first layer's output (4 x 5), later the third layer's (4 x 1)
  __shared__ __attribute__((aligned(16))) float red[2 * 4096];               // split-K partial sums of two row tiles: [tile][wave][h][j][lane]
  __shared__ __attribute__((aligned(16))) float sstat[XT_MAXM * 4];          // per row: mean / rstd of the gate half, of the info half
  __shared__ int s_go;
  __shared__ XMlpLayer s_lay[7];
  __shared__ __attribute__((aligned(16))) float s_xs[8][4 * 32];
  typedef const __attribute__((address_space(4))) XTailParams CP;
  CP& p = *(CP*)pp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x >= 128) {                          // a passenger (merged form only): the row buffers are its scratch (hsplit_smem(32) = 64 KB)
    __shared__ long s_prow[2][32];
    const int nd = p.p_blocks / p.p_ipl, q = (int)blockIdx.x - 128, qd = q / p.p_ipl, item = q - qd * p.p_ipl;
    const int ncount = nd - p.p_count_from, layer = qd < ncount ? p.p_count_from + qd : qd - ncount;      // the counted descriptors are dispatched first
    ConstSplitParams& sp = *((ConstSplitParams*)p.ptab + layer);
    long long t_in = 0;
    if constexpr (TS) t_in = wall_clock64();
    if (layer >= p.p_count_from) {                          // the counted ones (the C1Q . W2 cache's newest row): their K = 256 twins, three descriptors on
      ConstSplitParams& sq = *((ConstSplitParams*)p.ptab + layer + 3);
      hbulk_body<4, ConstSplitParams, 4, true>(sq, p.p_step + sq.step_val, item, p.p_ipl, p.p_ipl, lds_rows, s_prow);
    } else {
      hbulk_body<8, ConstSplitParams, 8, true>(sp, p.p_step + sp.step_val, item, p.p_ipl, p.p_ipl, lds_rows, s_prow);
    }
    if constexpr (TS) {                                    // measurement: when the first and the last passenger ran (slots 56 .. 59 of the stamps)
      if (p.ts && tid == 0 && (q == 0 || q == p.p_blocks - 1)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); p.ts[q == 0 ? 56 : 58] = t_in; p.ts[q == 0 ? 57 : 59] = wall_clock64(); }
    }
    if (p.pdone && layer >= p.p_count_from) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                   // every thread's stores are out
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // ... and written back past this XCD's L2
        const unsigned old = __hip_atomic_fetch_add(p.pdone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (old + 1u == p.pdone_target) __hip_atomic_store(p.psig, p.psig_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return false;
  }
  // merged form: this launch is the first one of a chain piece, so every earlier piece of this stream is complete: say so FIRST -- the side stream's next piece
  // starts from this word (xcone_kernel's team leaders poll it), and since round 4 the side stream is as long as the chain
  if (blockIdx.x == 0 && tid == 0 && p.np && p.sig) __hip_atomic_store(p.sig, p.sig_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int team = (int)blockIdx.x & 7, grp = ((int)blockIdx.x >> 3) & 15;
  const int B = p.m.B;
  const int U = p.U ? p.U : 4;
  if (team * U >= B) return false;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  unsigned* const bar = p.m.bar + team * 32;
  const unsigned xcc = xg_xcc_id();
  const int nl = p.m.nl, nh = p.nh, np = p.np;
  if (tid < (int)(sizeof(XMlpLayer) * 7 / 4)) reinterpret_cast<uint32_t*>(s_lay)[tid] = reinterpret_cast<const uint32_t*>(pp->m.lay)[tid];
  if (tid == 0 && !np) s_go = __hip_atomic_load(p.m.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;      // (merged form: set with the wait for the side stream below)
  // (stamps go to LDS and are copied out at the end: a global store that may be pending makes every later vector-memory wait a full drain)
  __shared__ long long s_ts[TS ? 64 : 1];
  int nts = 0;
  auto stamp = [&]() { if constexpr (TS) { if (blockIdx.x == 0 && tid == 0 && nts < 60) s_ts[nts++] = wall_clock64(); } };
  stamp();

  for (int round = 0, m0 = team * U; m0 < B; ++round, m0 += 8 * U) {
    const int mend = (m0 + U < B) ? m0 + U : B;            // this round's utterances: m0 .. mend - 1 (the four-utterance machinery runs whatever U is)
    // Everything derived from the thread index is derived INSIDE the round, from a value the compiler cannot see through: hoisted out of this loop (which runs
    // once up to B = 32), ~100 loop-invariant LDS / global offsets of all the phases below lived in registers for the whole kernel, and the kernel spilled.
    int tid_r = threadIdx.x;
    asm volatile("" : "+v"(tid_r));
    const int tid = tid_r, lane = tid & 63, wave = tid >> 6;
    const int arow = lane & 15, aq = lane >> 4, c4 = aq * 4;
    const int cr = lane >> 4, cc = lane & 15;
    const int etile = wave >> 2, ej = wave & 3, ecol = lane & 15;
    const int pcol = etile * 256 + grp * 16 + ecol;
    float* const xs = s_xs[wave];
    const unsigned ch0 = (unsigned)(16 * wave + cc), ch1 = ch0 + 128u;
    __syncthreads();                                     // (round 0: s_lay / s_go; later rounds: the previous round's last LDS reads)
    bool team_ok = s_go != 0;                            // (merged form, round 0: set behind the wait for the side stream)
    const unsigned rbase = p.m.bar_base + (unsigned)round * (unsigned)(np + nh + nl) * 16u;
    const int gi = m0 / U;                               // utterance group: its rows of the exchange buffers
    auto bof = [&](int u) { return (m0 + u < mend) ? m0 + u : m0; };      // an utterance slot past the batch repeats the group's first utterance (never stored)

    // ---- request order matters: a wave's loads return in order.  First what is consumed first -- the producer's pre-norm rows / statistics / residual of the
    //      newest input row, then the 59 staged input rows (60 KB per workgroup) -- and only then the weights: the first layer's slice (96 KB per workgroup:
    //      12 KB per wave = six k-groups x two column tiles) lands while the rows are written to LDS and the newest row is rebuilt; the second layer's slice is
    //      requested behind it; the third layer's goes into the first layer's registers once that layer's exchanged rows have landed (a request issued earlier
    //      would sit in front of every later load of the wave and be waited for with it).
    f32x4 wA0[6], wA1[6], wB0[6], wB1[6];
    unsigned wl = ((unsigned)(grp * 2) * 48u + (unsigned)(6 * wave)) * 256u + (unsigned)lane * 4u;      // uniform base + ONE 32-bit offset per lane (+ immediates)
    asm volatile("" : "+v"(wl));                          // opaque per round: hoisted out of the round loop, the 36 request addresses derived from it lived in registers (and spilled) for the whole kernel
    auto load_w = [&](const float* wp, f32x4 (&q0)[6], f32x4 (&q1)[6]) {
#pragma unroll
      for (int i = 0; i < 6; ++i) { q0[i] = ldv(wp, wl + (unsigned)i * 256u); q1[i] = ldv(wp, wl + (48u + (unsigned)i) * 256u); }
    };
    // the same slice for a FOUR-row layer contracted on 4 x 4 x 1 blocks (the last cone layer, round 5): lane (cb, kh, j) takes column 4 (cb & 3) + j of tile cb >> 2
    // for the 16 channels of k-groups 6 w + 3 kh + i, i = 0 .. 2 -- twelve float4, q0[0 .. 5] then q1[0 .. 5] in (i, k / 4) order
    auto load_w4 = [&](const float* wp, f32x4 (&q0)[6], f32x4 (&q1)[6]) {
      const int cb_ = (lane >> 2) & 7, kh_ = lane >> 5;
      const unsigned base = ((unsigned)(grp * 2 + (cb_ >> 2)) * 48u + (unsigned)(6 * wave + 3 * kh_)) * 256u + (unsigned)(((cb_ & 3) * 4 + (lane & 3)) * 4);
#pragma unroll
      for (int k = 0; k < 12; ++k) { const f32x4 v = ldv(wp, base + (unsigned)(k >> 2) * 256u + (unsigned)(k & 3) * 64u); if (k < 6) q0[k] = v; else q1[k - 6] = v; }
    };
    auto load_w3 = [&](auto PC, const float* wp, f32x4 (&q0)[6], f32x4 (&q1)[6]) {      // a third of the slice (k-groups 2 part, 2 part + 1 of both tiles)
      constexpr int part = decltype(PC)::value;
#pragma unroll
      for (int i = 2 * part; i < 2 * part + 2; ++i) { q0[i] = ldv(wp, wl + (unsigned)i * 256u); q1[i] = ldv(wp, wl + (48u + (unsigned)i) * 256u); }
    };
    const int nu = (tid >> 6) & 3;
    const unsigned ncme = (unsigned)(tid & 63) * 4u;
    f32x4 vb[2] = {z4, z4};
    float cbias = 0.f;
    auto load_c0 = [&]() {                               // the first k = 1 layer's slice and bias: requested once the last highway layer's exchanged rows have landed
      const int nkg = p.m.lay[0].nkg;
      const float* wb = p.m.lay[0].wp + lane * 4;
#pragma unroll
      for (int e = 0; e < 2; ++e) { const int kg = wave + 8 * e; vb[e] = ldv(wb, (unsigned)(grp * nkg + (kg < nkg ? kg : nkg - 1)) * 256u); }
      cbias = p.m.lay[0].bias[grp * 16 + (lane & 15)];
    };
    if (np) {
      // ==== merged form: the three newest-row layers first (xgroup_kernel's layer loop, M = 4 rows: A operand in registers, compact rebuild through s_xs)
      const int b_ = m0 + arow;
      const unsigned bb = (arow < 4 && b_ < mend) ? (unsigned)b_ : (unsigned)m0;        // lanes of MFMA rows nobody reads run on the team's first row
      const int erow = aq * 4 + (wave & 3), eb = m0 + erow;
      const bool wr = erow < 4 && eb < mend;
      const unsigned crow_p = (m0 + cr < mend) ? (unsigned)(m0 + cr) : (unsigned)m0;
      // ---- request order: what the first layer needs (C_1's rows, its statistics and layer-norm parameters, the layer's 16-column tiles), then the FIRST CONE LAYER's
      //      slice (96 KB per workgroup, 12 MB per launch out of the Infinity Cache: ~3 us): it lands under the newest-row layers instead of in front of the cone layer
      // (round 5: the contraction on 4 x 4 x 1 MFMA blocks, as in xgroup_kernel.h -- a team's layer has four rows; lane (cb, kh, j): columns 4 cb .. 4 cb + 3 of the
      // workgroup's 32, the 16 channels of k-group wave + 8 kh, row j; the weights come out of the same packing through another lane -> address map)
      const int j4 = lane & 3, cb4 = (lane >> 2) & 7, kh = lane >> 5;
      const unsigned pwoff = ((unsigned)(grp * 2 + (cb4 >> 2)) * 16u + (unsigned)(wave + 8 * kh)) * 256u + (unsigned)(((cb4 & 3) * 4 + j4) * 4);
      float p0c[2], g1c[2], b1c[2];
      f32x4 st0, pw[4];
      {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const unsigned ch = (unsigned)((8 * e + wave) * 16 + cc);
          p0c[e] = p.pP0[crow_p * 256u + ch]; g1c[e] = p.pg1[ch]; b1c[e] = p.pb1[ch];
        }
        st0 = ldv(p.pstats0, crow_p * 64u + (unsigned)(cc * 4));
#pragma unroll
        for (int q = 0; q < 4; ++q) pw[q] = ldv(p.pl[0].wp, pwoff + 64u * q);
      }
      // ---- the stream signal and the wait for the side stream (this is the first launch of a chain piece), while those loads are in flight
      if (tid == 0 && round == 0) {
        int go = __hip_atomic_load(p.m.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;     // an earlier launch of this decode already failed: no more waiting, the decode is reported invalid
        if (p.wait2 && go) {
          bool ok = false;
          for (int i = 0; i < (1 << 20) && !ok; ++i) {                       // bounded: about a second
            ok = __hip_atomic_load(p.wait2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= p.wait_val;
            if (!ok) __builtin_amdgcn_s_sleep(16);
          }
          if (!ok) atomicOr(p.m.err, 16);
        }
        s_go = go;
      }
      __syncthreads();
      team_ok = s_go != 0;
      float addv = 0.f;
      if (wr) {                                            // presum of the first layer: written by the side stream -> read past the L1 / a possibly stale line
        const float* ap = p.pl[0].presum + (unsigned)(eb * p.pl[0].presum_bs) + (unsigned)pcol;
        addv = __hip_atomic_load(ap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (sc0 sc1)
      }
      // The first cone layer's slice (96 KB per workgroup, 12 MB per launch out of the Infinity Cache: ~3 us) is requested in THIRDS, one behind the presum and one
      // behind each of the first two newest-row layers (round 5).  Requests return in order and every team barrier drains the wave's requests: in front of the presum
      // (round 4, with an s_waitcnt vmcnt(0) on it) the first row waited for a slice that nobody needs before the first cone layer; as one batch behind it, the next
      // barrier did.  A third lands within a layer's span.
      load_w3(std::integral_constant<int, 0>{}, p.hc[0].wp, wA0, wA1);
      float xc[2];                                         // the layer's input at (row cr, channels 16 w + cc and 128 + 16 w + cc) = the next rebuild's highway residual
      f32x4 ax[4];                                         // ... and as the contraction's A operand: row j4, the 16 channels of k-group wave + 8 kh
      {
        const float m1 = row16_sum(st0[0]) * (1.0f / 16.0f);
        const float d1 = st0[0] - m1;
        const float r1 = rsqrt_fast(row16_sum(st0[1] + 16.0f * d1 * d1) * (1.0f / 256.0f) + 1e-12f);
#pragma unroll
        for (int e = 0; e < 2; ++e) { xc[e] = (p0c[e] - m1) * r1 * g1c[e] + b1c[e]; xs[cr * 32 + e * 16 + cc] = xc[e]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) ax[q] = *reinterpret_cast<const f32x4*>(&xs[j4 * 32 + kh * 16 + 4 * q]);
      }
      stamp();                                             // first row built (the wait for the side stream is in here)
      auto player = [&](auto GC) {
        constexpr int g = decltype(GC)::value;
        constexpr bool lastp = (g == NP - 1);
        f32x4 acc0 = z4, acc1 = z4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 a = ax[q], b = pw[q];
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], b[0], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], b[1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], b[2], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], b[3], acc1, 0, 0, 0);
        }
        *reinterpret_cast<f32x4*>(&red[wave * 256 + lane * 4]) = acc0 + acc1;      // [lane (cb, kh, j)][row]
        // requests that do not depend on the team: this layer's layer-norm parameters (compact), the next layer's tiles and presum
        float ng1[2], nb1[2], ng2[2], nb2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const unsigned chc = (unsigned)((8 * e + wave) * 16 + cc);
          ng1[e] = p.pl[g].g1[chc]; nb1[e] = p.pl[g].b1[chc]; ng2[e] = p.pl[g].g2[chc]; nb2[e] = p.pl[g].b2[chc];
        }
        float naddv = 0.f;
        if constexpr (!lastp) {
#pragma unroll
          for (int q = 0; q < 4; ++q) pw[q] = ldv(p.pl[g + 1].wp, pwoff + 64u * q);
          if (wr) naddv = p.pl[g + 1].presum[(unsigned)(eb * p.pl[g + 1].presum_bs) + (unsigned)pcol];      // behind the wait for the side stream; never read before in this launch
        }
        __syncthreads();
        float v_ = 0.f;                                      // (row wave & 3, column ecol of tile etile): 8 waves x 2 k-halves, fixed order
        {
          const float* rp = &red[((etile * 4 + (ecol >> 2)) * 4 + (ecol & 3)) * 4 + (wave & 3)];
#pragma unroll
          for (int w = 0; w < 8; ++w) { v_ += rp[w * 256]; v_ += rp[w * 256 + 128]; }
        }
        v_ += addv;
        const float mg = row16_sum(v_) * (1.0f / 16.0f);
        const float dv = v_ - mg;
        const float m2g = row16_sum(dv * dv);
        constexpr int par = g & 1;
        if (wr) {
          p.xchp[(long)par * p.xchp_set + (long)eb * 512 + pcol] = v_;
          if (ecol == 0) { float* so = p.schp + (long)par * p.schp_set + ((long)eb * 16 + grp) * 4 + etile * 2; so[0] = mg; so[1] = m2g; }
        }
        team_barrier(bar, grp, xcc, rbase + (unsigned)(g + 1) * 16u, p.m.err, team_ok);
        float hg[2], hi[2]; f32x4 stc;
        {
          const float* xr = p.xchp + (long)par * p.xchp_set + (long)crow_p * 512 + wave * 16 + cc;   // gate channel 16 w + cc; +128 floats = the second k-group; +256 = info
          const float* sr = p.schp + (long)par * p.schp_set + (long)crow_p * 64 + cc * 4;             // column group cc's partial statistics of the row
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
        {
          const float m1 = row16_sum(stc[0]) * (1.0f / 16.0f), m2 = row16_sum(stc[2]) * (1.0f / 16.0f);
          const float d1 = stc[0] - m1, d2 = stc[2] - m2;
          const float r1 = rsqrt_fast(row16_sum(stc[1] + 16.0f * d1 * d1) * (1.0f / 256.0f) + 1e-12f);
          const float r2 = rsqrt_fast(row16_sum(stc[3] + 16.0f * d2 * d2) * (1.0f / 256.0f) + 1e-12f);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float s_ = sigmoid_fast((hg[e] - m1) * r1 * ng1[e] + nb1[e]);
            xc[e] = s_ * ((hi[e] - m2) * r2 * ng2[e] + nb2[e]) + (1.0f - s_) * xc[e];
            if constexpr (!lastp) xs[cr * 32 + e * 16 + cc] = xc[e];
            else bufA[(cr * p.nin0) * XT_LDR + (8 * e + wave) * 16 + cc] = xc[e];      // HC_4's newest row: input row 0 of the first cone layer
          }
          if constexpr (!lastp) {
#pragma unroll
            for (int q = 0; q < 4; ++q) ax[q] = *reinterpret_cast<const f32x4*>(&xs[j4 * 32 + kh * 16 + 4 * q]);
          }
        }
        addv = naddv;
        stamp();                                           // newest-row layer done
      };
      player(std::integral_constant<int, 0>{});
      load_w3(std::integral_constant<int, 1>{}, p.hc[0].wp, wA0, wA1);
      // ---- the first cone layer's other input rows (the side stream's cone rows of HC_4: never read before in this launch) in two halves: requested behind
      //      a newest-row layer, written to LDS behind the next one (all eight requests at once would not fit beside two cone layers' weight slices)
      f32x4 sv[4];
      auto stage_req = [&](const int k0) {
        const int nin = p.nin0;
        const float* xbase = p.xin - 64 * p.xin_stride;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int it = tid + 512 * (k0 + k);
          const int rowi = (it >> 6) < 4 * nin ? (it >> 6) : 0;
          const int u = rowi / nin, q = rowi - u * nin;
          sv[k] = ldv(xbase, (unsigned)((long)bof(u) * p.xin_bs + (long)(p.in_off[q] + 64) * p.xin_stride) + ncme);
        }
      };
      auto stage_put = [&](const int k0) {                 // (row 0 of every utterance is a placeholder: the last newest-row layer writes it)
        const int nin = p.nin0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int it = tid + 512 * (k0 + k);
          if ((it >> 6) < 4 * nin && (it >> 6) % nin != 0) *reinterpret_cast<f32x4*>(&bufA[(it >> 6) * XT_LDR + ncme]) = sv[k];
        }
      };
      stage_req(0);
      player(std::integral_constant<int, 1>{});
      load_w3(std::integral_constant<int, 2>{}, p.hc[0].wp, wA0, wA1);
      stage_put(0);
      if constexpr (NP == 3) {
        stage_req(4);
        player(std::integral_constant<int, 2>{});
        stage_put(4);
      } else {                                             // (the first cone layer's 4 x 5 input rows fit the first half's four sweeps)
        player(std::integral_constant<int, 2>{});
        player(std::integral_constant<int, 3>{});
      }
      __syncthreads();
    } else {
      // the newest input row of every utterance: producer's pre-norm row + partial statistics + residual (an earlier launch: plain loads)
      f32x4 nst = z4, nhg = z4, nhi = z4, nxr = z4, ng1 = z4, nb1 = z4, ng2 = z4, nb2 = z4;
      {
        const int su = (tid >> 4) & 3, sg = tid & 15;
        nst = ldv(p.m.stats0, (unsigned)bof(su) * 64u + (unsigned)sg * 4u);                        // (threads 0 .. 63 use it)
        const float* pr = p.m.P0 + (long)bof(nu) * p.m.p0_bs;
        nhg = ldv(pr, ncme); nhi = ldv(pr, 256u + ncme); nxr = ldv(p.m.res + (long)bof(nu) * p.m.res_bs, ncme);      // (threads 0 .. 255 use them)
      }
      // ---- stage the first layer's input rows from the side stream's buffer (cone rows of the producing layer at offsets < 0; rows in front of t = 0 are the
      //      buffer's zero rows = the causal padding, modules.py:173-177); row 0 of every utterance is the newest row, rebuilt below
      {
        const int nin = p.nin0;
        const float* xbase = p.xin - 64 * p.xin_stride;       // (64 zero rows sit in front of every utterance: uniform base + a non-negative 32-bit offset per lane)
        f32x4 sv[8];
  #pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int it = tid + 512 * k;
          const int rowi = (it >> 6) < 4 * nin ? (it >> 6) : 0;
          const int u = rowi / nin, q = rowi - u * nin;
          sv[k] = ldv(xbase, (unsigned)((long)bof(u) * p.xin_bs + (long)(p.in_off[q] + 64) * p.xin_stride) + ncme);      // (q == 0: any readable row; overwritten below)
        }
        load_w(p.hc[0].wp, wA0, wA1);
  #pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int it = tid + 512 * k;
          if ((it >> 6) < 4 * nin) *reinterpret_cast<f32x4*>(&bufA[(it >> 6) * XT_LDR + ncme]) = sv[k];
        }
      }
      ng1 = ldv(p.m.g1, ncme); nb1 = ldv(p.m.b1, ncme); ng2 = ldv(p.m.g2, ncme); nb2 = ldv(p.m.b2, ncme);
      if (tid < 64) {
        const int u = tid >> 4, g = tid & 15;
        const float m1 = row16_sum(nst[0]) * (1.0f / 16.0f), m2 = row16_sum(nst[2]) * (1.0f / 16.0f);
        const float d1 = nst[0] - m1, d2 = nst[2] - m2;
        const float r1 = rsqrt_fast(row16_sum(nst[1] + 16.0f * d1 * d1) * (1.0f / 256.0f) + 1e-12f);
        const float r2 = rsqrt_fast(row16_sum(nst[3] + 16.0f * d2 * d2) * (1.0f / 256.0f) + 1e-12f);
        if (g == 0) { sstat[u * 4 + 0] = m1; sstat[u * 4 + 1] = r1; sstat[u * 4 + 2] = m2; sstat[u * 4 + 3] = r2; }
      }
      __syncthreads();                                     // (also: the staged rows are in LDS, so the rebuilt rows below overwrite row 0's placeholders)
      if (tid < 256) {
        const float m1 = sstat[nu * 4 + 0], r1 = sstat[nu * 4 + 1], m2 = sstat[nu * 4 + 2], r2 = sstat[nu * 4 + 3];
        f32x4 o;
  #pragma unroll
        for (int e = 0; e < 4; ++e) { const float s_ = sigmoid_fast((nhg[e] - m1) * r1 * ng1[e] + nb1[e]); o[e] = s_ * ((nhi[e] - m2) * r2 * ng2[e] + nb2[e]) + (1.0f - s_) * nxr[e]; }
        *reinterpret_cast<f32x4*>(&bufA[(nu * p.nin0) * XT_LDR + ncme]) = o;
      }
      __syncthreads();

    }
    stamp();                                               // input rows staged, newest row rebuilt
    // ---- one highway layer: A from `bin` (LDS), B from the registers passed in; out rows rebuilt into `bout`
    auto hlayer = [&](auto NITC, const int h, const float* bin, float* bout, const f32x4 (&bq0)[6], const f32x4 (&bq1)[6], auto&& after_landed) {
      constexpr int NIT = decltype(NITC)::value;           // 512-thread sweeps over the layer's M x 64 four-channel items: 3 for M = 20 (two row tiles), 2 for M = 12, 1 for M = 4
      typedef const __attribute__((address_space(4))) XTailHc CH;
      CH& y = p.hc[h];
      const int nin = y.nin, nout = y.nout, ts = y.ts, M = 4 * nout;
      constexpr bool two = NIT > 2;
      const float bias = y.bias[pcol];
      const unsigned cme = (unsigned)(tid & 63) * 4u;       // the four channels of the rows this thread rebuilds (the same for each of them: 512 is a multiple of 64)
      // The NEXT layer's weight slice (96 KB per workgroup, contiguous: two adjacent 16-column tiles) is pulled into this XCD's L2 now, one dword per 64-byte
      // line: its requests go out behind this layer's hand-off (a request in front of the hand-off's loads would be waited for with them), and out of the
      // Infinity Cache that burst took 2-3 us during which the wave could not issue anything else (stamps: "rows rebuilt"); out of the L2 it takes ~0.7 us.
      // Not earlier than one layer ahead: the side stream's kernels stream through the same L2.
      float pf0 = 0.f, pf1 = 0.f, pf2 = 0.f;
      if (h + 1 < nh) {
        const float* q = p.hc[h + 1].wp + (unsigned)(grp * 2) * 48u * 256u + (unsigned)(wave * 3) * 1024u + (unsigned)lane * 16u;
        pf0 = q[0]; pf1 = q[1024]; pf2 = q[2048];
      }
      constexpr bool four = (NIT == 1);                    // M = 4: one row per utterance -> 4 x 4 x 1 blocks (round 5; bq0 / bq1 then hold load_w4's layout)
      if constexpr (four) {
        const int j4 = lane & 3, kh = lane >> 5;
        f32x4 acc0 = z4, acc1 = z4;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int g = 6 * wave + 3 * kh + i, tap = g >> 4, cg = g & 15;
          const float* ar = &bin[(j4 * nin + (2 - tap) * ts) * XT_LDR + cg * 16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ar + 4 * q);
            const int k = i * 4 + q;
            const f32x4 b = k < 6 ? bq0[k < 6 ? k : 0] : bq1[k < 6 ? 0 : k - 6];
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], b[0], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], b[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], b[2], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], b[3], acc1, 0, 0, 0);
          }
        }
        *reinterpret_cast<f32x4*>(&red[wave * 256 + lane * 4]) = acc0 + acc1;      // [lane (cb, kh, j)][row]
      } else {
      // A fragments: lane (arow, aq) holds row tile * 16 + arow, channels 4 aq .. 4 aq + 3 of k-group 6 w + i
      f32x4 a0[6], a1[6];
      {
        const int ma = arow < M ? arow : 0, mb = (16 + arow < M) ? 16 + arow : 0;
        const int ua = ma / nout, ra = ma - ua * nout, ub = mb / nout, rb_ = mb - ub * nout;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int g = 6 * wave + i, tap = g >> 4, cg = g & 15;
          a0[i] = *reinterpret_cast<const f32x4*>(&bin[(ua * nin + ra + (2 - tap) * ts) * XT_LDR + cg * 16 + c4]);
          if constexpr (two) a1[i] = *reinterpret_cast<const f32x4*>(&bin[(ub * nin + rb_ + (2 - tap) * ts) * XT_LDR + cg * 16 + c4]);
        }
      }
      f32x4 acc0 = z4, acc1 = z4, acc2 = z4, acc3 = z4;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i][e], bq0[i][e], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i][e], bq1[i][e], acc1, 0, 0, 0);
        }
      }
      if constexpr (two) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i][e], bq0[i][e], acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i][e], bq1[i][e], acc3, 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        red[((wave * 2 + 0) * 4 + j) * 64 + lane] = acc0[j]; red[((wave * 2 + 1) * 4 + j) * 64 + lane] = acc1[j];
        if constexpr (two) { red[4096 + ((wave * 2 + 0) * 4 + j) * 64 + lane] = acc2[j]; red[4096 + ((wave * 2 + 1) * 4 + j) * 64 + lane] = acc3[j]; }
      }
      }
      stamp();                                             // A fragments read, MFMAs issued, partial sums written
      // the layer's layer-norm parameters for this thread's channels: requested now (they land during the reduction and the hand-off), used behind it
      const f32x4 g1 = ldv(y.g1, cme), b1 = ldv(y.b1, cme), g2 = ldv(y.g2, cme), b2 = ldv(y.b2, cme);
      __syncthreads();
      float v0 = bias, v1 = bias;
      if constexpr (four) {                                // (row ej, column ecol of tile etile): 8 waves x 2 k-halves, fixed order
        const float* rp = &red[((etile * 4 + (ecol >> 2)) * 4 + (ecol & 3)) * 4 + ej];
#pragma unroll
        for (int w = 0; w < 8; ++w) { v0 += rp[w * 256]; v0 += rp[w * 256 + 128]; }
      } else {
#pragma unroll
        for (int w = 0; w < 8; ++w) { v0 += red[((w * 2 + etile) * 4 + ej) * 64 + lane]; if constexpr (two) v1 += red[4096 + ((w * 2 + etile) * 4 + ej) * 64 + lane]; }
      }
      const int par = h & 1;
      float* const xr_ = p.xch + (long)par * p.xch_set + (long)gi * (XT_MAXM * 512);
      float* const sr_ = p.sch + (long)par * p.sch_set + (long)gi * (XT_MAXM * 64);
      {
        const int me = aq * 4 + ej;                          // the row of tile 0 this lane finishes; tile 1: me + 16
        const float mg0 = row16_sum(v0) * (1.0f / 16.0f), dv0 = v0 - mg0, q0 = row16_sum(dv0 * dv0);
        const float mg1 = row16_sum(v1) * (1.0f / 16.0f), dv1 = v1 - mg1, q1 = row16_sum(dv1 * dv1);
        if (me < M) {
          xr_[me * 512 + pcol] = v0;
          if (ecol == 0) { float* so = sr_ + (me * 16 + grp) * 4 + etile * 2; so[0] = mg0; so[1] = q0; }
        }
        if (two && me + 16 < M) {
          xr_[(me + 16) * 512 + pcol] = v1;
          if (ecol == 0) { float* so = sr_ + ((me + 16) * 16 + grp) * 4 + etile * 2; so[0] = mg1; so[1] = q1; }
        }
      }
      stamp();                                             // slice reduced, statistics, published
      team_barrier(bar, grp, xcc, rbase + (unsigned)(np + h + 1) * 16u, p.m.err, team_ok);
      stamp();                                             // team barrier passed
      // ---- every workgroup rebuilds the layer's M output rows.  ALL requests of the phase in one batch, past the L1: column group (tid & 15)'s partial
      //      statistics of row tid >> 4, and the (gate, info) values of up to three (row, 4-channel) items per thread: item k = tid + 512 k.
      constexpr int nit = NIT;
      f32x4 st, hg0, hi0, hg1 = z4, hi1 = z4, hg2 = z4, hi2 = z4;
      {
        const int ms = (tid >> 4) < M ? (tid >> 4) : M - 1;
        const float* sp = sr_ + (ms * 16 + (tid & 15)) * 4;
        auto ip = [&](int k) { const int it = tid + 512 * k; const int m = (it >> 6) < M ? (it >> 6) : M - 1; return (const float*)(xr_ + m * 512 + cme); };
        const float* q0 = ip(0); const float* q1 = ip(1); const float* q2 = ip(2);
        asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %4, off offset:1024 sc1"
                     : "=&v"(st), "=&v"(hg0), "=&v"(hi0) : "v"(sp), "v"(q0) : "memory");
        if constexpr (nit > 1) asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:1024 sc1" : "=&v"(hg1), "=&v"(hi1) : "v"(q1) : "memory");
        if constexpr (nit > 2) asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:1024 sc1" : "=&v"(hg2), "=&v"(hi2) : "v"(q2) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(st), "+v"(hg0), "+v"(hi0), "+v"(hg1), "+v"(hi1), "+v"(hg2), "+v"(hi2) :: "memory");
      }
      stamp();                                             // exchanged rows + statistics landed
      asm volatile("" :: "v"(pf0), "v"(pf1), "v"(pf2));    // (the prefetched lines' destination registers stay reserved until here: everything has landed)
      after_landed();                                      // (the slot for requests that must not sit in front of the loads above)
      if (tid < 16 * M) {
        const int m = tid >> 4, g = tid & 15;
        const float m1 = row16_sum(st[0]) * (1.0f / 16.0f), m2 = row16_sum(st[2]) * (1.0f / 16.0f);
        const float d1 = st[0] - m1, d2 = st[2] - m2;
        const float r1 = rsqrt_fast(row16_sum(st[1] + 16.0f * d1 * d1) * (1.0f / 256.0f) + 1e-12f);
        const float r2 = rsqrt_fast(row16_sum(st[3] + 16.0f * d2 * d2) * (1.0f / 256.0f) + 1e-12f);
        if (g == 0) { sstat[m * 4 + 0] = m1; sstat[m * 4 + 1] = r1; sstat[m * 4 + 2] = m2; sstat[m * 4 + 3] = r2; }
      }
      __syncthreads();
      auto finish = [&](int k, const f32x4 hg, const f32x4 hi) {
        const int it = tid + 512 * k;
        if (it >= M * 64) return;
        const int m = it >> 6;
        const int u = m / nout, r = m - u * nout;
        const f32x4 xr = *reinterpret_cast<const f32x4*>(&bin[(u * nin + r) * XT_LDR + cme]);
        const float m1 = sstat[m * 4 + 0], r1 = sstat[m * 4 + 1], m2 = sstat[m * 4 + 2], r2 = sstat[m * 4 + 3];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float s_ = sigmoid_fast((hg[e] - m1) * r1 * g1[e] + b1[e]); o[e] = s_ * ((hi[e] - m2) * r2 * g2[e] + b2[e]) + (1.0f - s_) * xr[e]; }
        if (r > p.frame) o = z4;                           // time t - r < 0: causal padding
        *reinterpret_cast<f32x4*>(&bout[m * XT_LDR + cme]) = o;
      };
      finish(0, hg0, hi0);
      if constexpr (nit > 1) finish(1, hg1, hi1);
      if constexpr (nit > 2) finish(2, hg2, hi2);
      __syncthreads();
      stamp();                                             // output rows rebuilt
    };
    // (the host checks the shape this is unrolled for: 5 / 3 / 1 rows per utterance)
    const float* bin_;
    if constexpr (NH == 3) {
      hlayer(std::integral_constant<int, 3>{}, 0, bufA, bufB, wA0, wA1, [&]() { load_w(p.hc[1].wp, wB0, wB1); });
      hlayer(std::integral_constant<int, 2>{}, 1, bufB, bufA, wB0, wB1, [&]() { load_w4(p.hc[2].wp, wA0, wA1); });
      hlayer(std::integral_constant<int, 1>{}, 2, bufA, bufB, wA0, wA1, [&]() { load_c0(); });
      bin_ = bufB;
    } else {                                               // (host-checked shape: 3 / 1 rows per utterance out of 5 / 3 input rows)
      hlayer(std::integral_constant<int, 2>{}, 0, bufA, bufB, wA0, wA1, [&]() { load_w4(p.hc[1].wp, wB0, wB1); });
      hlayer(std::integral_constant<int, 1>{}, 1, bufB, bufA, wB0, wB1, [&]() { load_c0(); });
      bin_ = bufA;
    }
    const float* const bin = bin_;
    // `bin` now holds one row per utterance (row u): the input of the first k = 1 layer

    // ---- the k = 1 layers (xmlp_kernel.h's loop)
    const unsigned crow = (m0 + cr < mend) ? (unsigned)(m0 + cr) : (unsigned)m0;      // (a row slot past the batch repeats the TEAM'S first row: its tags are this team's)
    const bool crow_ok = m0 + cr < mend;
    const int erow = aq * 4 + wave;
    const int eb = m0 + erow;
    const bool wr = wave < 4 && erow < 4 && eb < mend;
    float4 x[2];
    x[0] = *reinterpret_cast<const float4*>(&bin[(arow & 3) * XT_LDR + wave * 16 + c4]);
    x[1] = *reinterpret_cast<const float4*>(&bin[(arow & 3) * XT_LDR + (8 + wave) * 16 + c4]);
    const unsigned cbase = rbase + (unsigned)(np + nh) * 16u;
    for (int l = 0; l < nl; ++l) {
      const int nkg = __builtin_amdgcn_readfirstlane(s_lay[l].nkg), cout = __builtin_amdgcn_readfirstlane(s_lay[l].cout), act = __builtin_amdgcn_readfirstlane(s_lay[l].act);
      const bool last = (l + 1 == nl);
      const bool mine = grp * 16 < cout;
      f32x4 acc = z4;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const f32x4 b = (mine && wave + 8 * e < nkg) ? vb[e] : z4;
        const float4 a = x[e];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[3], acc, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(wave * 4 + j) * 64 + lane] = acc[j];
      float lg[2], lb[2], nbias = 0.f;
      {
        const unsigned i0 = ch0 < (unsigned)cout ? ch0 : 0u, i1 = ch1 < (unsigned)cout ? ch1 : 0u;
        lg[0] = ldg1(s_lay[l].g, i0); lg[1] = ldg1(s_lay[l].g, i1); lb[0] = ldg1(s_lay[l].be, i0); lb[1] = ldg1(s_lay[l].be, i1);
      }
      if (!last) {
        const int nkg2 = __builtin_amdgcn_readfirstlane(s_lay[l + 1].nkg), cout2 = __builtin_amdgcn_readfirstlane(s_lay[l + 1].cout);
        const int tile2 = (grp * 16 < cout2) ? grp : 0;
        const float* wb = s_lay[l + 1].wp + lane * 4;
#pragma unroll
        for (int e = 0; e < 2; ++e) { const int kg = wave + 8 * e; vb[e] = ldg4(wb, (unsigned)(tile2 * nkg2 + (kg < nkg2 ? kg : nkg2 - 1)) * 256u); }
        nbias = ldg1(s_lay[l + 1].bias, (unsigned)(tile2 * 16 + (lane & 15)));
      }
      __syncthreads();
      float v_ = 0.f, mg = 0.f, m2g = 0.f;
      if (wave < 4) {
#pragma unroll
        for (int w = 0; w < 8; ++w) v_ += red[(w * 4 + wave) * 64 + lane];
        v_ += cbias;
        mg = row16_sum(v_) * (1.0f / 16.0f);
        const float dv = v_ - mg;
        m2g = row16_sum(dv * dv);
      }
      const int pc = grp * 16 + (lane & 15);
      if (last && l != p.m.mel_layer) {
        if (wr && mine) {
          p.m.pout[(long)eb * 256 + pc] = v_;
          if ((lane & 15) == 0) { float* so = p.m.stats_out + ((long)eb * 16 + grp) * 4; so[0] = mg; so[1] = m2g; }
        }
        break;
      }
      // ---- hand-off WITHOUT a barrier: every published word travels with the layer's sequence number -- (value, tag) as one 8-byte store, a column group's
      //      statistics as (mean, M2, tag, tag) in one 16-byte store -- and the consumers poll the DATA itself past their L1 until every tag they need is
      //      this layer's.  One L2 round trip instead of "drain the stores, workgroup barrier, flag word, poll, workgroup barrier, load" (0.7 us per layer
      //      less, stamps).  No write-after-read hazard with two parity copies: a workgroup publishes layer l + 2 only behind its reads of layer l + 1,
      //      which exists only once EVERY workgroup has published it, i.e. has finished reading layer l.  The buffers are cleared at the start of a decode
      //      (tags of an earlier decode would repeat).  A team that is not on one XCD never sees its mates' stores: the bounded spin raises the error word.
      const int par = l & 1;
      const unsigned seq = cbase + (unsigned)(l + 1) * 16u;
      const float seqf = __uint_as_float(seq);
      if (wr && mine) {
        *reinterpret_cast<float2*>(&p.m.xch[(long)par * p.m.xch_set + ((long)eb * 256 + pc) * 2]) = make_float2(v_, seqf);
        if ((lane & 15) == 0) *reinterpret_cast<float4*>(&p.m.sch[(long)par * p.m.sch_set + ((long)eb * 16 + grp) * 4]) = make_float4(mg, m2g, seqf, seqf);
      }
      const int ng = cout >> 4;
      float h0, h1, s0, s1;
      {
        const float* xr0 = p.m.xch + (long)par * p.m.xch_set + ((long)crow * 256 + (ch0 < (unsigned)cout ? ch0 : 0u)) * 2;
        const float* xr1 = p.m.xch + (long)par * p.m.xch_set + ((long)crow * 256 + (ch1 < (unsigned)cout ? ch1 : 0u)) * 2;
        const float* sr = p.m.sch + (long)par * p.m.sch_set + ((long)crow * 16 + (cc < ng ? cc : 0)) * 4;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 q0, q1; f32x4 qs;
        int spins = 0;
        for (;;) {
          asm volatile(
              "global_load_dwordx2 %0, %3, off sc1\n\t"
              "global_load_dwordx2 %1, %4, off sc1\n\t"
              "global_load_dwordx4 %2, %5, off sc1\n\t"
              "s_waitcnt vmcnt(0)"
              : "=&v"(q0), "=&v"(q1), "=&v"(qs)
              : "v"(xr0), "v"(xr1), "v"(sr)
              : "memory");
          const bool ok = __float_as_uint(q0[1]) == seq && __float_as_uint(q1[1]) == seq && __float_as_uint(qs[2]) == seq;
          if (__builtin_amdgcn_ballot_w64(!ok) == 0ull || !team_ok) break;
          if (++spins > (1 << 15) || ((spins & 255) == 0 && __hip_atomic_load(p.m.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { if (lane == 0) atomicOr(p.m.err, 1); break; }
        }
        h0 = q0[0]; h1 = q1[0]; s0 = qs[0]; s1 = qs[1];
      }
      {
        const bool gok = cc < ng;
        const float inv_g = 1.0f / (float)ng;
        const float m1 = row16_sum(gok ? s0 : 0.f) * inv_g;
        const float d1 = s0 - m1;
        const float r1 = rsqrt_fast(row16_sum(gok ? s1 + 16.0f * d1 * d1 : 0.f) * (inv_g * (1.0f / 16.0f)) + 1e-12f);
        float y0 = (h0 - m1) * r1 * lg[0] + lb[0], y1 = (h1 - m1) * r1 * lg[1] + lb[1];
        if (l == p.m.mel_layer) {
          const float q0 = sigmoid_fast(y0), q1 = sigmoid_fast(y1);
          if (grp == 0 && crow_ok) {
            if (ch0 < (unsigned)cout) { p.m.logits[(long)crow * p.m.l_bs + ch0] = y0; p.m.ymel[(long)crow * p.m.y_bs + ch0] = q0; }
            if (ch1 < (unsigned)cout) { p.m.logits[(long)crow * p.m.l_bs + ch1] = y1; p.m.ymel[(long)crow * p.m.y_bs + ch1] = q1; }
          }
          y0 = q0; y1 = q1;
        } else if (act == ACT_RELU) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
        xs[cr * 32 + cc] = ch0 < (unsigned)cout ? y0 : 0.f;
        xs[cr * 32 + 16 + cc] = ch1 < (unsigned)cout ? y1 : 0.f;
        x[0] = *reinterpret_cast<const float4*>(&xs[(arow & 3) * 32 + c4]);
        x[1] = *reinterpret_cast<const float4*>(&xs[(arow & 3) * 32 + 16 + c4]);
      }
      cbias = nbias;
      stamp();                                             // k = 1 layer done
      if (last) break;
    }
  }
  if constexpr (TS) { if (p.ts && blockIdx.x == 0 && tid == 0) for (int i = 0; i < nts; ++i) p.ts[i] = s_ts[i]; }
  return true;
}

// grid: 128 blocks of 512 threads, whatever the batch (+ the passengers)
template <bool TS = false, int NP = 3>
__global__ void __launch_bounds__(512) xtail_kernel(const XTailParams* __restrict__ pp) { (void)xtail_body<TS, NP>(pp); }

// Round 5: a chain piece as ONE launch -- xtail_kernel's layers (AudioDec behind C_1, the k = 1 layers around the mel frame), the team's barrier, then xgroup_kernel's
// AudioEnc run of the next frame + attention row + AudioDec C_1.  What the launch boundary between them cost: ~1.7 us of gap + ~3 us until xgroup_kernel's first row was
// built, against one team barrier here (0.9 us): the team's workgroups are where they were, their XCD's L2 has what the k = 1 layers just wrote.  What the boundary
// gave for free and needs care now: (a) the k = 1 layers' last output (AudioEnc C_3's pre-norm rows + statistics) is read by the AudioEnc run's first layer -- behind the
// team barrier, by plain loads of lines this launch has not touched before (the L1 was invalidated at its start); (b) AudioEnc's presums of the next row used to come
// from THIS launch's passengers, workgroups on other XCDs: they now compute the row after that (inputs: rows that are final since the previous piece), so that what
// the AudioEnc run reads was written one launch earlier (decode_host.h: v3_xtail_table).
// Measured (B = 32, one box, A/B): 80.4 -> 78.8 us per frame.  Tried on top and worth nothing: the attention tail's operands requested in front of the AudioEnc
// run's first layer; this frame's XGroupParams entry and the next frame's XTailParams entry touched at the launch's start (first-touch scalar loads).
template <bool TS = false, int NP = 3>
__global__ void __launch_bounds__(512) xchain_kernel(const XTailParams* __restrict__ pt, const XGroupParams* __restrict__ pg) {
  if (!xtail_body<TS, NP>(pt)) return;
  typedef const __attribute__((address_space(4))) XTailParams CP;
  CP& p = *(CP*)pt;
  const int team = (int)blockIdx.x & 7, grp = ((int)blockIdx.x >> 3) & 15;
  const int Uc = p.U ? p.U : 4;
  const int rounds = ((p.m.B + Uc - 1) / Uc + 7) / 8;
  const unsigned target = p.m.bar_base + (unsigned)rounds * (unsigned)(p.np + p.nh + p.m.nl) * 16u + 16u;      // one more meeting than xtail_body's layers (the host counts it)
  const bool go = __hip_atomic_load(p.m.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // the k = 1 layers' last stores are in the L2 ...
  // ... and the team meets while the AudioEnc run's first requests are in flight (weights, layer-norm parameters, history row)
  xgroup_body<TS>(pg, [&]() { team_barrier<true>(p.m.bar + team * 32, grp, xg_xcc_id(), target, p.m.err, go); });
}

}  // namespace dctts
