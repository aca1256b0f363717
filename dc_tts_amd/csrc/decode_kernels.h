// decode_kernels.h -- latency-oriented kernels for one frame of the Text2Mel autoregressive loop
// (synthesize.py:47-54) on gfx950.
//
// Why a second kernel family: in a decode step every AudioEnc layer and the newest-frame rows of the
// AudioDec cone see only B (= 32) rows.  A workgroup that owns all output columns of a row block
// (hconv_kernel.h, the throughput form) then leaves 255 of 256 CUs idle and takes ~75 us per layer.
// Here the output columns are split across workgroups (one 2-tile column group each) and K is split
// across the 8 waves of a workgroup, so a 32-row layer becomes 32 short workgroups.
//
// Layer-norm needs whole rows, so it is DEFERRED.  The GEMM writes pre-norm values P plus, per output row
// and 16-column group, the partial statistics (mean_g, M2_g); whoever consumes a row rebuilds it:
//   * chain layers (the newest frame j): the consumer Chan-combines the 16 partials of each of its centre
//     rows (exact two-pass quality), applies LN (+act) or LN + sigmoid gate + highway mix elementwise to the
//     A fragments it holds in registers, and the first column group materialises the rebuilt rows into the
//     layer's absolute-time history buffer;
//   * bulk layers (cone rows at offsets < 0, independent of frame j): a row kernel (ln_rows_kernel).
//
// hsplit_kernel<MF, TRACE, NG, NT, ONE>:
//                    MF = 32 -> 32 rows x 2 tiles of v_mfma_f32_32x32x2_f32   (bulk cone layers, persistent items; the production
//                               bulk kernel is hbulk_kernel<NG> below: the same arithmetic, software-pipelined across items)
//                    MF = 16 -> 16 rows x 2 tiles of v_mfma_f32_16x16x4_f32   (chain layers; NT = 3 / 1 are the forms specialised
//                               for causal k = 3 / k = 1 layers over 256 channels)
// What the wait-count pass of the compiler needs in order NOT to serialise these kernels' loads (each point cost microseconds
// per launch before it was found): no branch around a load (clamp the address, discard the value), no store -- not even a
// dead debug stamp -- pending next to loads, prefetches refilled after their consumers and pinned with sched_barrier,
// straight-line K loops, no run-time loop with stores between prefetch and use, kernel arguments fetched in one batch.
// A fragments are loaded straight from global memory into registers in MFMA operand layout (no LDS staging:
// no wave shares another wave's K slice); B comes in pre-packed fragment order (one coalesced 1 KiB load per
// wave per four MFMAs); every load of a work item is issued before the first use; the 8 partial accumulators
// are reduced through LDS in a fixed order -> deterministic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "attn_kernels.h"
#include "hconv_kernel.h"

namespace dctts {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { PRO_RAW = 0, PRO_LN_C = 1, PRO_LN_HC = 2, PRO_MEL = 3 };   // PRO_MEL: LN over n_mels + sigmoid = the mel frame (chain: AudioDec C_11 -> AudioEnc C_1)

// How to obtain one 256-channel activation row X[b][t] that exists only as pre-norm values P.
struct RowNorm {
  const float* P; int np;                  // pre-norm rows [prow][np]  (np = 256 for C, 512 for HC)
  const float* g1; const float* b1; const float* g2; const float* b2; int act;
  int ngroups;                             // 16-column statistic groups that exist for a row (16 for 256 channels, 5 for 80)
  const float* res; long res_bstride; long res_row0; int res_stride; long res_set;   // highway residual X_{l-1}[b][t] (HC)
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// uniform base + unsigned 32-bit element offset: selects the SGPR-base / VGPR-offset addressing mode (one VGPR per address)
__device__ __forceinline__ float4 ld4u(const float* base, unsigned off) { return *reinterpret_cast<const float4*>(base + off); }
// latency-critical prologue math: hardware exp / rcp / rsq (about 1 ulp) instead of the long IEEE sequences
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float rsqrt_fast(float x) { return __builtin_amdgcn_rsqf(x); }

// One wave normalises one 256-channel row; lane owns channels 4*lane .. 4*lane+3 (two-pass statistics, DPP sums).
__device__ __forceinline__ float4 norm_c_regs(const RowNorm& n, const float4 x, int lane) {
  const int c = lane * 4;
  const float mean = wave_sum(x.x + x.y + x.z + x.w) * (1.0f / 256.0f);
  const float4 d = make_float4(x.x - mean, x.y - mean, x.z - mean, x.w - mean);
  const float var = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * (1.0f / 256.0f);
  const float rs = 1.0f / sqrtf(var + 1e-12f);
  const float4 g = ld4(n.g1 + c), b = ld4(n.b1 + c);
  float4 y = make_float4(d.x * rs * g.x + b.x, d.y * rs * g.y + b.y, d.z * rs * g.z + b.z, d.w * rs * g.w + b.w);
  if (n.act == ACT_RELU) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
  return y;
}

// (layer-norm parameters as values: a caller that requests them early with its other loads passes them in)
__device__ __forceinline__ float4 norm_hc_vals(const float4 h1, const float4 h2, const float4 xr, const float4 g1, const float4 b1, const float4 g2, const float4 b2) {
  const float m1 = wave_sum(h1.x + h1.y + h1.z + h1.w) * (1.0f / 256.0f);
  const float m2 = wave_sum(h2.x + h2.y + h2.z + h2.w) * (1.0f / 256.0f);
  const float4 d1 = make_float4(h1.x - m1, h1.y - m1, h1.z - m1, h1.w - m1);
  const float4 d2 = make_float4(h2.x - m2, h2.y - m2, h2.z - m2, h2.w - m2);
  const float v1 = wave_sum(d1.x * d1.x + d1.y * d1.y + d1.z * d1.z + d1.w * d1.w) * (1.0f / 256.0f);
  const float v2 = wave_sum(d2.x * d2.x + d2.y * d2.y + d2.z * d2.z + d2.w * d2.w) * (1.0f / 256.0f);
  const float r1 = 1.0f / sqrtf(v1 + 1e-12f), r2 = 1.0f / sqrtf(v2 + 1e-12f);
  float4 o;
  { const float s = sigmoidf_(d1.x * r1 * g1.x + b1.x); o.x = s * (d2.x * r2 * g2.x + b2.x) + (1.0f - s) * xr.x; }
  { const float s = sigmoidf_(d1.y * r1 * g1.y + b1.y); o.y = s * (d2.y * r2 * g2.y + b2.y) + (1.0f - s) * xr.y; }
  { const float s = sigmoidf_(d1.z * r1 * g1.z + b1.z); o.z = s * (d2.z * r2 * g2.z + b2.z) + (1.0f - s) * xr.z; }
  { const float s = sigmoidf_(d1.w * r1 * g1.w + b1.w); o.w = s * (d2.w * r2 * g2.w + b2.w) + (1.0f - s) * xr.w; }
  return o;
}

__device__ __forceinline__ float4 norm_hc_regs(const RowNorm& n, const float4 h1, const float4 h2, const float4 xr, int lane) {
  const int c = lane * 4;
  return norm_hc_vals(h1, h2, xr, ld4(n.g1 + c), ld4(n.b1 + c), ld4(n.g2 + c), ld4(n.b2 + c));
}

__device__ __forceinline__ float4 norm_row_c(const RowNorm& n, long prow, int lane) {
  return norm_c_regs(n, ld4(n.P + prow * n.np + lane * 4), lane);
}

// par = frame parity selecting the residual's buffer copy (0 when the residual buffer is single)
__device__ __forceinline__ float4 norm_row_hc(const RowNorm& n, long prow, int b, int t, int lane, long par = 0) {
  const int c = lane * 4;
  const float4 h1 = ld4(n.P + prow * n.np + c), h2 = ld4(n.P + prow * n.np + 256 + c);
  const float4 xr = ld4(n.res + par * n.res_set + ((long)b * n.res_bstride + n.res_row0 + t) * n.res_stride + c);
  return norm_hc_regs(n, h1, h2, xr, lane);
}

struct SplitParams {
  // ---- row mapping: m in [0,M) -> b = b0 + m / R, r = m % R, t = frame + (offs ? offs[r] : 0); rows with t < 0 are skipped
  int M, R, b0; const int* offs; const int* step; int step_val;   // frame = step_val + (step ? *step : 0)
  int ngroups;                                                   // column groups; work items = row tiles x ngroups, grid-strided
  int tile_rows;                                                 // rows per work item (<= MF): the chain is bound by bytes pulled per CU
                                                                 // (~30 GB/s each), so 8-row items on twice the CUs beat full 16-row tiles
  // ---- centre tap (row t itself): PRO_RAW reads xsrc; PRO_LN_* rebuilds it from pre-norm rows (index b*R + r) using the
  //      producer's per-column-group partial statistics `stats_in` [prow][16 groups][4] = (mean1, M2_1, mean2, M2_2)
  int pro; RowNorm nrm; const float* stats_in;
  float* xmat; long xm_bstride; long xm_row0; int xm_stride; long xm_set;   // where column group 0 materialises the rebuilt row
  float* xmat2; long xm2_bstride; int xm2_stride; int xm2_toff;             // PRO_MEL: the pre-sigmoid logits row, at time t + xm2_toff
  // ---- tap source (absolute-time activation buffer): all taps when PRO_RAW, the non-centre taps otherwise
  const float* xsrc; long xs_bstride; long xs_row0; int xs_stride; long xs_set;
  int ntaps; int tap_off[3]; int cin; int cin_p;
  // ---- weights / output
  const float* wp; const float* bias; int cout; int hc; int np_out;
  float* pout;                                                   // pre-norm rows [b*R + r][np_out]
  float* stats_out;                                              // optional partial statistics of pout (16-row form only)
  // ---- decode v3 (hoisted taps): the chain contracts only the centre tap; everything else arrives as a per-row presum
  const float* presum; int presum_rstride;                       // 16-row form, R == 1: bias + older taps of row b at presum[b * rstride + col] (replaces bias)
  float* raw_out; long raw_bstride; long raw_row0; int raw_stride;   // 16-row form, R == 1: the bare contraction (no bias / presum) -> raw_out[b][t][col]
  int mask_last;                                                 // hbulk_kernel<12>: row r == R-1 of every utterance is a presum row (centre tap contributes 0)
  long abs_bstride; long abs_row0; int abs_toff;                 // hbulk: abs_bstride != 0 -> output row index = b * abs_bstride + abs_row0 + t + abs_toff (a time-indexed cache)
  long long* dbg;                                                // optional: 8 wall-clock (100 MHz) stamps of workgroup 0
  long long* dbg_wg;                                             // optional: (entry, end) stamps of every workgroup (<= 128)
};
// *_set: buffers written by the bulk branch exist twice (frame parity); set stride in floats, 0 = single buffer.

// Kernel arguments arrive through scalar loads from the kernarg segment, and the compiler fetches a ~300-byte parameter struct
// lazily: the chain kernel showed five s_load batches, each followed by s_waitcnt lgkmcnt(0), i.e. five serialised scalar
// memory round trips (~2 us) before its first vector load.  Naming every field as an SGPR input of an empty asm at entry makes
// the compiler fetch the whole struct in one batch.
#define DCTTS_SGPR(x) asm volatile("" ::"s"(x))
template <int NT>
__device__ __forceinline__ void prefetch_params(const SplitParams& p) {
  DCTTS_SGPR(p.M); DCTTS_SGPR(p.b0); DCTTS_SGPR(p.step_val);
  if constexpr (!(NT == 1 || NT == 2 || NT == 4)) { DCTTS_SGPR(p.R); DCTTS_SGPR(p.offs); DCTTS_SGPR(p.step); DCTTS_SGPR(p.ngroups); }
  DCTTS_SGPR(p.tile_rows); DCTTS_SGPR(p.pro);
  DCTTS_SGPR(p.nrm.P); DCTTS_SGPR(p.nrm.np); DCTTS_SGPR(p.nrm.g1); DCTTS_SGPR(p.nrm.b1); DCTTS_SGPR(p.nrm.g2); DCTTS_SGPR(p.nrm.b2);
  DCTTS_SGPR(p.nrm.act); DCTTS_SGPR(p.nrm.res); DCTTS_SGPR(p.nrm.res_bstride); DCTTS_SGPR(p.nrm.res_row0);
  DCTTS_SGPR(p.nrm.res_stride); DCTTS_SGPR(p.nrm.res_set); DCTTS_SGPR(p.stats_in);
  if constexpr (NT == 0 || NT == 4) {     // only the narrow / generic forms can carry the mel prologue; the others are at the SGPR limit already
    DCTTS_SGPR(p.xmat2); DCTTS_SGPR(p.xm2_bstride); DCTTS_SGPR(p.xm2_stride); DCTTS_SGPR(p.xm2_toff); DCTTS_SGPR(p.nrm.ngroups);
  }
  DCTTS_SGPR(p.xmat); DCTTS_SGPR(p.xm_bstride); DCTTS_SGPR(p.xm_row0); DCTTS_SGPR(p.xm_stride); DCTTS_SGPR(p.xm_set);
  DCTTS_SGPR(p.xsrc); DCTTS_SGPR(p.xs_bstride); DCTTS_SGPR(p.xs_row0); DCTTS_SGPR(p.xs_stride); DCTTS_SGPR(p.xs_set);
  if constexpr (NT == 1 || NT == 2 || NT == 4) { DCTTS_SGPR(p.tap_off[0]); }
  else { DCTTS_SGPR(p.ntaps); DCTTS_SGPR(p.tap_off[0]); DCTTS_SGPR(p.tap_off[1]); DCTTS_SGPR(p.tap_off[2]); }
  DCTTS_SGPR(p.cin); DCTTS_SGPR(p.cin_p);
  DCTTS_SGPR(p.wp); DCTTS_SGPR(p.bias); DCTTS_SGPR(p.cout); DCTTS_SGPR(p.hc); DCTTS_SGPR(p.np_out); DCTTS_SGPR(p.pout); DCTTS_SGPR(p.stats_out);
  if constexpr (NT == 1) {
    DCTTS_SGPR(p.presum); DCTTS_SGPR(p.presum_rstride); DCTTS_SGPR(p.raw_out); DCTTS_SGPR(p.raw_bstride); DCTTS_SGPR(p.raw_row0); DCTTS_SGPR(p.raw_stride);
  }
}

// Sum over the four lanes l, l^16, l^32, l^48 (the lanes that share an A-operand row in the 16x16x4 layout), on the
// VALU: gfx950's v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane halves between two registers.
__device__ __forceinline__ float xrow4_sum(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// Chan-combine the 16 per-group partials (mean_g, M2_g over 16 channels each) of one row: exact two-pass quality.
// Each of the row's four lanes holds four groups (st[0..3]); h selects (x,y) = H1 / (z,w) = H2.
__device__ __forceinline__ void combine_stats(const float4 (&st)[4], int h, float& mean, float& rstd, int ngr = 16, int g0 = 0) {
  // ngr < 16 (a row of 16 ngr channels): this lane's groups are g0 .. g0+3; groups >= ngr do not exist (their slots hold zeros)
  const float inv_g = 1.0f / (float)ngr;
  float sm = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) sm += h ? st[g].z : st[g].x;
  mean = xrow4_sum(sm) * inv_g;
  float m2 = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float dm = (h ? st[g].z : st[g].x) - mean;
    const float t = (h ? st[g].w : st[g].y) + 16.0f * dm * dm;
    m2 += (g0 + g < ngr) ? t : 0.f;
  }
  rstd = rsqrt_fast(xrow4_sum(m2) * (inv_g * (1.0f / 16.0f)) + 1e-12f);
}

// TRACE = true compiles the wall-clock stamps in (DCTTS_TRACE).  They must NOT exist in the production instantiation even as
// dead branches: a store that may be pending makes the wait-count pass treat vmcnt as out of order (mixed load / store events)
// and every later wait on a load becomes s_waitcnt vmcnt(0) -- the bulk kernel's weight prefetch ring was serialised by it.
// NG > 0 fixes the k-groups per wave at compile time (K = 64 NG for the 32-row form): the K loop is then straight-line code.
// With a run-time `if (g < KG)` around each step, a prefetch issued inside a conditional block may or may not be followed by
// younger loads, so the only safe wait for it is vmcnt(0) -- the counted waits the ring depends on need unconditional steps.
// NT = 3 (16-row form): three causal taps over 256 channels, centre = tap 2 (every k = 3 layer of AudioEnc / AudioDec).  Wave w's
// k-group i is then tap i >> 1, channels 128 (i & 1) + 16 w: compile-time per i, which removes ~150 select / compare
// instructions from the stretch between kernel entry and the first load.  NT = 1: k = 1 over 256 channels: two k-groups per
// wave instead of six clamped ones (the generic form re-reads the last group four times: 3x the load traffic of such a layer).
// NT = 2: k = 1 over 512 channels (AudioDec C_1: four k-groups per wave).  NT = 4: k = 1 over <= 128 channels (AudioEnc C_1 on
// the 80-channel mel row: one k-group per wave, some waves and columns are padding, so it keeps the clamps).
// ONE (16-row form): a workgroup owns ONE 16-column tile (gate or info) instead of the pair: twice the workgroups, half the
// weight bytes and half the MFMAs per workgroup (the matrix pipe is shared by the two waves of a SIMD: 2 us -> 1 us).
template <int MF, bool TRACE = false, int NG = 0, int NT = 0, bool ONE = false>
__global__ void __launch_bounds__(512) hsplit_kernel(const SplitParams p) {
  constexpr int KGS = (MF == 32) ? 8 : 16;          // k per k-group (4 MFMAs)
  constexpr int NJ = (MF == 32) ? 16 : 4;           // accumulator registers per tile
  constexpr int NGMAX = NG > 0 ? NG : ((MF == 32) ? 12 : (NT == 1 ? 2 : (NT == 2 ? 4 : (NT == 4 ? 1 : 6))));   // k-groups per wave
  constexpr bool FULL = (NT == 1 || NT == 2 || NT == 3);    // every k-group / channel of the form exists: no clamps, no padding
  constexpr int BD = (MF == 32) ? 4 : NGMAX;        // B prefetch ring depth (k-groups); the 16-row form holds all of them
  extern __shared__ __attribute__((aligned(16))) float smem[];     // split-K reduction only
  __shared__ long s_prow[MF];                       // output row index per tile row, -1 = skipped
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  prefetch_params<NT>(p);
  const int wgid = blockIdx.y * gridDim.x + blockIdx.x;
  const bool tr = TRACE && p.dbg && wgid == 0 && tid == 0;
  if constexpr (TRACE) { if (tr) p.dbg[0] = wall_clock64(); if (p.dbg_wg && tid == 0 && wgid < 128) p.dbg_wg[2 * wgid] = wall_clock64(); }
  // CHAINROW: the forms only the newest-frame chain uses (one row per utterance, no offset table, frame index by value, grid =
  // (column groups, row tiles)).  Everything the generic row mapping needs -- a possibly-null offs[] load (its branch made the
  // wait-count pass put s_waitcnt vmcnt(0) in front of the A loads: one full memory round trip per launch, after the weight
  // loads), the device step counter, two integer divisions -- is compiled out.
  constexpr bool CHAINROW = (MF == 16) && (NT == 1 || NT == 2 || NT == 4) && !ONE;
  const int step = CHAINROW ? p.step_val : p.step_val + (p.step ? *p.step : 0);
  const long par = step & 1;
  const int KG = p.ntaps * p.cin_p / KGS;
  const int ntile = CHAINROW ? 1 : (p.M + p.tile_rows - 1) / p.tile_rows;
  const int nitems = CHAINROW ? 1 : ntile * p.ngroups * (ONE ? 2 : 1);
  const int arow = lane & (MF - 1);
  const int aq = (MF == 32) ? (lane >> 5) : (lane >> 4);
  const int c4 = aq * 4;
  const bool ln = (MF == 16) && (p.pro != PRO_RAW);
  const int ctap = (NT == 3) ? 2 : ((p.ntaps == 1) ? 0 : ((p.tap_off[0] == 0) ? 0 : ((p.tap_off[1] == 0) ? 1 : 2)));
  // (tap, first channel) of this lane's fragment of k-group i / g
  auto tap_c = [&](int i, int g, int& tap, int& c) {
    if constexpr (NT == 3) { tap = i >> 1; c = 128 * (i & 1) + 16 * wave + c4; }
    else if constexpr (NT == 1 || NT == 2) { tap = 0; c = 128 * i + 16 * wave + c4; }
    else if constexpr (NT == 4) { tap = 0; c = 16 * g + c4; }
    else { const int k0 = g * KGS; tap = (p.ntaps == 1) ? 0 : (k0 >> 8); c = k0 - tap * p.cin_p + c4; }
  };

  // Persistent over work items: the bulk branch launches fewer workgroups than CUs so that the latency-critical
  // chain branch always finds free CUs; the chain itself has exactly one item per workgroup.
  for (int item = CHAINROW ? 0 : blockIdx.x; item < nitems; item += (MF == 32 ? (int)gridDim.x : nitems)) {
    const int mytile = ONE ? (item & 1) : 0, rest = ONE ? (item >> 1) : item;
    const int tile_x = CHAINROW ? (int)blockIdx.y : rest / p.ngroups, grp = CHAINROW ? (int)blockIdx.x : rest - tile_x * p.ngroups;
    const int m0 = tile_x * p.tile_rows;

    // ---- B fragments: wave w owns k-groups w, w+8, ...; independent of A, so issue first
    const float* wb = p.wp + lane * 4;
    const unsigned w0o = (unsigned)(grp * 2 + mytile) * (unsigned)KG * 256u, w1o = w0o + (unsigned)KG * 256u;
    float4 bq0[BD], bq1[BD];
    // Loads are issued WITHOUT branches around them: a uniform `if (g < KG)` still compiles to a branch, and at every join the
    // wait-count pass falls back to s_waitcnt vmcnt(0) when a register may have a load pending on one path -- the chain kernel
    // then paid ~6 serialised memory round trips per layer (in-kernel stamps: 2.8 us from first to last issue).  k-groups past
    // the end re-read the last one (clamped index); their A fragment is zero, so the duplicate weights contribute nothing.
#pragma unroll
    for (int i = 0; i < BD; ++i) {
      const int g = wave + 8 * i, gc = (FULL || g < KG) ? g : KG - 1;
      bq0[i] = ld4u(wb, w0o + (unsigned)gc * 256u);
      if constexpr (!ONE) bq1[i] = ld4u(wb, w1o + (unsigned)gc * 256u); else bq1[i] = bq0[i];
    }

    // ---- this lane's A row (MFMA A operand: lane -> row lane % MF, k sub-block lane / MF)
    int b = 0, t = 0; long prow = -1; bool valid = false, cmask = false;
    {
      const int m = m0 + arow;
      if (arow < p.tile_rows && m < p.M) {
        if constexpr (CHAINROW) {
          b = p.b0 + m; t = step; prow = b; valid = true;
        } else {
          int bl = m, r = 0;
          if (p.R != 1) { bl = m / p.R; r = m - bl * p.R; }
          b = p.b0 + bl;
          t = step + (p.offs ? p.offs[r] : 0);
          prow = (long)b * p.R + r;
          valid = (t >= 0);
          cmask = (NT == 3) && p.mask_last && (r == p.R - 1);     // v3 presum row: the chain contracts its centre tap
        }
      }
      if (wave == 0 && aq == 0) s_prow[arow] = valid ? prow : -1;
    }
    if constexpr (TRACE) { if (tr) p.dbg[1] = wall_clock64(); }

    // ---- A fragments straight from global memory: every load of the item is in flight before the first use.
    //      Loads are unconditional (skipped rows read row 0 and are zeroed afterwards) so that no exec-mask branches
    //      serialise them; addresses are uniform base + 32-bit offset.  ntaps > 1 implies cin_p == 256 (tap = shift).
    const unsigned xs_row = valid ? (unsigned)(par * p.xs_set + ((long)b * p.xs_bstride + p.xs_row0 + t) * p.xs_stride) : (unsigned)(p.xs_row0 * p.xs_stride);
    const unsigned p_row = valid ? (unsigned)(prow * p.nrm.np) : 0u;
    const unsigned rs_row = valid ? (unsigned)(par * p.nrm.res_set + ((long)b * p.nrm.res_bstride + p.nrm.res_row0 + t) * p.nrm.res_stride) : 0u;
    float4 av[NGMAX];
    float4 h2v[2], rsv[2], g1v[2], b1v[2], g2v[2], b2v[2];   // centre-tap extras of the (at most two) centre k-groups of a wave
    float4 st[4];
    float biasv = 0.f;                                        // MF == 16: this thread's output column, fetched with everything else
    if constexpr (MF == 16) {
      // Branch-free issue (see above).  Addresses that a branch used to skip are redirected to something readable
      // (the tap source's first row / the clamped column) and the value is discarded afterwards.
      const bool hcpro = (p.pro == PRO_LN_HC);
#pragma unroll
      for (int i = 0; i < NGMAX; ++i) {
        const int g = wave + 8 * i, gc = (FULL || g < KG) ? g : KG - 1;
        int tap, c; tap_c(i, gc, tap, c);
        const int toff = (tap == 0) ? p.tap_off[0] : ((tap == 1) ? p.tap_off[1] : p.tap_off[2]);
        const bool centre = ln && tap == ctap;                                  // uniform: a scalar select of the base pointer
        const int cc = (FULL || c < p.cin) ? c : p.cin - 4;                  // pad columns of a narrow input: read in range, zeroed below
        const float* base = centre ? p.nrm.P : p.xsrc;
        const unsigned off = centre ? p_row + (unsigned)cc : xs_row + (unsigned)(toff * p.xs_stride) + (unsigned)cc;
        av[i] = ld4u(base, off);
      }
      // the two centre k-groups of wave w are 16 ctap + w and 16 ctap + 8 + w  (i = 2 ctap + e): channel (8 e + w) 16 + c4
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ce_ = (8 * e + wave) * 16 + c4;
        const unsigned ce = ln ? (unsigned)((FULL || ce_ < p.cin) ? ce_ : p.cin - 4) : 0u;
        g1v[e] = ld4u(ln ? p.nrm.g1 : p.xsrc, ce); b1v[e] = ld4u(ln ? p.nrm.b1 : p.xsrc, ce);
        g2v[e] = ld4u(hcpro ? p.nrm.g2 : p.xsrc, hcpro ? ce : 0u); b2v[e] = ld4u(hcpro ? p.nrm.b2 : p.xsrc, hcpro ? ce : 0u);
        h2v[e] = ld4u(hcpro ? p.nrm.P : p.xsrc, hcpro ? p_row + 256u + ce : 0u);
        rsv[e] = ld4u(hcpro ? p.nrm.res : p.xsrc, hcpro ? rs_row + ce : 0u);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) st[g] = ld4u(ln ? p.stats_in : p.xsrc, (ln && valid) ? (unsigned)(prow * 64) + (unsigned)((aq * 4 + g) * 4) : 0u);
      {
        const int l = tid & 63, tile = ONE ? mytile : (tid >> 8), col = l & 15;
        const int pc = p.hc ? ((grp * MF + col) < p.cout ? tile * p.cout + grp * MF + col : 0)
                            : (((grp * 2 + tile) * MF + col) < p.cout ? (grp * 2 + tile) * MF + col : 0);
        if constexpr (NT == 1) {
          // v3: the per-row presum (bias + the older taps, computed off the critical path) takes the place of the bias.
          // Epilogue element of this thread: row (l >> 4) * 4 + ((tid >> 6) & 3) of the tile; R == 1, so the row index is b.
          const int erow = (l >> 4) * 4 + ((tid >> 6) & 3), em = m0 + erow;
          const bool eok = erow < p.tile_rows && em < p.M;
          const bool ps = p.presum != nullptr;                                       // uniform: scalar select of base and offset
          const unsigned boff = ps ? (eok ? (unsigned)((p.b0 + em) * p.presum_rstride + pc) : 0u) : (unsigned)pc;
          biasv = (ps ? p.presum : p.bias)[boff];
        } else {
          biasv = p.bias[(unsigned)pc];
        }
      }
      if constexpr (TRACE) { if (tr) p.dbg[2] = wall_clock64(); }
      // discard what the redirected loads fetched
#pragma unroll
      for (int i = 0; i < NGMAX; ++i) {
        const int g = wave + 8 * i;
        int tap, c; tap_c(i, g, tap, c);
        const bool centre = ln && tap == ctap;
        if constexpr (FULL) { if (!valid || (NT == 3 && tap == 2 && cmask)) av[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
        else { if (g >= KG || !valid || c >= p.cin) av[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
      }
    } else {
      // same branch-free issue as the 16-row form: clamped k-group / column, value discarded afterwards
#pragma unroll
      for (int i = 0; i < NGMAX; ++i) {
        const int g = wave + 8 * i, gc = g < KG ? g : KG - 1;
        const int k0 = gc * KGS, tap = (p.ntaps == 1) ? 0 : (k0 >> 8), c = k0 - tap * p.cin_p + c4;
        const int toff = (tap == 0) ? p.tap_off[0] : ((tap == 1) ? p.tap_off[1] : p.tap_off[2]);
        const int cc = c < p.cin ? c : p.cin - 4;
        av[i] = ld4u(p.xsrc, xs_row + (unsigned)(toff * p.xs_stride) + (unsigned)cc);
      }
      if constexpr (TRACE) { if (tr) p.dbg[2] = wall_clock64(); }
#pragma unroll
      for (int i = 0; i < NGMAX; ++i) {
        const int g = wave + 8 * i;
        const int k0 = g * KGS, tap = (p.ntaps == 1) ? 0 : (k0 >> 8), c = k0 - tap * p.cin_p + c4;
        if (g >= KG || !valid || c >= p.cin) av[i] = make_float4(0.f, 0.f, 0.f, 0.f);       // pad columns / skipped rows stay zero
      }
    }

    // ---- rebuild the centre-tap values: LN (+ act) or LN + sigmoid gate + highway mix, elementwise given the row statistics
    if constexpr (MF == 16) {
      if (ln) {
        float m1, r1, m2 = 0.f, r2 = 0.f;
        combine_stats(st, 0, m1, r1, FULL ? 16 : p.nrm.ngroups, aq * 4);   // every lane takes part in the cross-lane sums (ln is uniform)
        if (p.pro == PRO_LN_HC) combine_stats(st, 1, m2, r2);
#pragma unroll
        for (int i = 0; i < NGMAX; ++i) {
          const int g = wave + 8 * i;
          if (FULL || g < KG) {
            int tap, c; tap_c(i, g, tap, c);
            if (tap == ctap) {
              const float4 g1 = g1v[i & 1], b1 = b1v[i & 1];
              float4 x = av[i];
              x.x = (x.x - m1) * r1 * g1.x + b1.x; x.y = (x.y - m1) * r1 * g1.y + b1.y;
              x.z = (x.z - m1) * r1 * g1.z + b1.z; x.w = (x.w - m1) * r1 * g1.w + b1.w;
              if (p.pro == PRO_LN_HC) {
                const float4 g2 = g2v[i & 1], b2 = b2v[i & 1];
                const float4 h2 = h2v[i & 1], xr = rsv[i & 1];
                { const float s_ = sigmoid_fast(x.x); x.x = s_ * ((h2.x - m2) * r2 * g2.x + b2.x) + (1.0f - s_) * xr.x; }
                { const float s_ = sigmoid_fast(x.y); x.y = s_ * ((h2.y - m2) * r2 * g2.y + b2.y) + (1.0f - s_) * xr.y; }
                { const float s_ = sigmoid_fast(x.z); x.z = s_ * ((h2.z - m2) * r2 * g2.z + b2.z) + (1.0f - s_) * xr.z; }
                { const float s_ = sigmoid_fast(x.w); x.w = s_ * ((h2.w - m2) * r2 * g2.w + b2.w) + (1.0f - s_) * xr.w; }
              } else if (p.nrm.act == ACT_RELU) {
                x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f);
              } else if (!FULL && p.nrm.act == ACT_SIGMOID) {         // PRO_MEL: x = the logits of the mel frame (networks.py:210)
                if (valid && grp == 0 && mytile == 0 && p.xmat2 && c < p.cin)
                  *reinterpret_cast<float4*>(p.xmat2 + ((long)b * p.xm2_bstride + t + p.xm2_toff) * p.xm2_stride + c) = x;
                x.x = sigmoidf_(x.x); x.y = sigmoidf_(x.y); x.z = sigmoidf_(x.z); x.w = sigmoidf_(x.w);
              }
              if (!FULL && c >= p.cin) x = make_float4(0.f, 0.f, 0.f, 0.f);      // K padding of a narrow input
              if (!valid) x = make_float4(0.f, 0.f, 0.f, 0.f);
              av[i] = x;
              if (valid && grp == 0 && mytile == 0 && p.xmat && (FULL || c < p.cin))
                *reinterpret_cast<float4*>(p.xmat + par * p.xm_set + ((long)b * p.xm_bstride + p.xm_row0 + t) * p.xm_stride + c) = x;
            }
          }
        }
      }
    }
    if constexpr (TRACE) { if (tr) p.dbg[3] = wall_clock64(); }

    // ---- K loop (fully unrolled so the register arrays are statically indexed)
    typedef typename std::conditional<MF == 32, f32x16, f32x4>::type acc_t;
    acc_t acc0, acc1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < NGMAX; ++i) {
      const int g = wave + 8 * i;
      if (NG > 0 || FULL || g < KG) {
        const float4 a = av[i];
        const float4 b0 = bq0[i % BD], b1 = bq1[i % BD];
        if constexpr (MF == 32) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
        } else {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc0, 0, 0, 0); if constexpr (!ONE) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc0, 0, 0, 0); if constexpr (!ONE) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, acc0, 0, 0, 0); if constexpr (!ONE) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, acc0, 0, 0, 0); if constexpr (!ONE) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, acc1, 0, 0, 0);
        }
        // Refill the ring slot AFTER the MFMAs that read it were issued.  Issued before them (as it used to be) the refill is a
        // write-after-read on live MFMA operands: the compiler loaded into a temporary and put s_waitcnt vmcnt(0) right behind
        // every prefetch -- 16 serialised memory round trips per work item in the bulk kernel.
        if (i + BD < NGMAX) {
          const int gn = g + 8 * BD;
          const int gnc = gn < KG ? gn : KG - 1;
          bq0[i % BD] = ld4u(wb, w0o + (unsigned)gnc * 256u); bq1[i % BD] = ld4u(wb, w1o + (unsigned)gnc * 256u);
          // keep the refill HERE: left alone, the machine scheduler sinks it to just before its consumer (4 steps later) to save
          // registers, and the ring degenerates into load -> s_waitcnt vmcnt(0) -> use
          if constexpr (NG > 0) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (TRACE) { if (tr) p.dbg[4] = wall_clock64(); }

    // ---- split-K reduction through LDS: red[wave][tile][j][lane], summed in a fixed order (deterministic)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      smem[((wave * 2 + 0) * NJ + j) * 64 + lane] = acc0[j];
      if constexpr (!ONE) smem[((wave * 2 + 1) * NJ + j) * 64 + lane] = acc1[j];
    }
    __syncthreads();
    if constexpr (TRACE) { if (tr) p.dbg[5] = wall_clock64(); }
    for (int e = tid; e < (ONE ? 1 : 2) * NJ * 64; e += 512) {
      const int l = e & 63, j = (e >> 6) % NJ, ltile = e / (64 * NJ), tile = ONE ? mytile : ltile;
      float v_ = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v_ += smem[((w * 2 + ltile) * NJ + j) * 64 + l];
      int row, col;
      if constexpr (MF == 32) { row = (j & 3) + 8 * (j >> 2) + 4 * (l >> 5); col = l & 31; }
      else                    { row = (l >> 4) * 4 + j;                      col = l & 15; }
      const long orow = s_prow[row];
      int pcol; bool ok;
      if (p.hc) { const int c = grp * MF + col; ok = c < p.cout; pcol = tile * p.cout + c; }
      else      { pcol = (grp * 2 + tile) * MF + col; ok = pcol < p.cout; }
      if constexpr (MF == 16 && NT == 1) {
        if (p.raw_out && ok && orow >= 0) p.raw_out[((long)orow * p.raw_bstride + p.raw_row0 + step) * p.raw_stride + pcol] = v_;
      }
      if constexpr (MF == 16) { if (ok) v_ += biasv; } else { if (ok) v_ += p.bias[pcol]; }
      if (ok && orow >= 0) p.pout[orow * p.np_out + pcol] = v_;
      if constexpr (MF == 16) {
        // partial LN statistics of this 16-column group: a DPP row (16 lanes) holds one output row's 16 columns
        if (p.stats_out) {
          const float mg = row16_sum(ok ? v_ : 0.f) * (1.0f / 16.0f);
          const float dv = ok ? v_ - mg : 0.f;
          const float m2g = row16_sum(dv * dv);
          if (ok && orow >= 0 && col == 0) {
            const int G = p.hc ? grp : grp * 2 + tile;
            float* so = p.stats_out + (orow * 16 + G) * 4 + (p.hc ? tile * 2 : 0);
            so[0] = mg; so[1] = m2g;
          }
        }
      }
    }
    if constexpr (TRACE) { if (tr) p.dbg[6] = wall_clock64(); if (p.dbg_wg && tid == 0 && wgid < 128) p.dbg_wg[2 * wgid + 1] = wall_clock64(); }
    if (MF == 32 && item + (int)gridDim.x < nitems) __syncthreads();      // s_prow / smem are reused by the next item
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// hbulk_kernel: the bulk (cone) branch's contraction, software-pipelined across work items.  Same arithmetic and the same
// summation order as hsplit_kernel<32> (32 rows x one gate/info pair of 32-column tiles, K split over 8 waves, fixed-order
// LDS reduction), PRO_RAW only (its input rows are materialised by ln_rows_kernel).  What differs is the schedule:
//   item n:  K loop | accumulators -> LDS | ISSUE item n+1's row info + A + first weight groups | barrier | epilogue of n
// so the ~1.5 us a work item used to wait for its first loads, and the scalar set-up before them, overlap the reduction and
// the stores of the previous one.  The epilogue issues no loads of its own (its two bias values ride along with the item's
// loads): a load there would have to be waited for with vmcnt(0), i.e. behind everything just issued for the next item.
template <int NG, typename PT>
__device__ __forceinline__ void hbulk_body(const PT& p, const int step, const int item0, const int item_stride, const int nitems,
                                           float* smem, long (*s_prow)[32]) {
  constexpr int MF = 32, NJ = 16, BD = 4, KG = NG * 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long par = step & 1;
  const int arow = lane & 31, c4 = (lane >> 5) * 4;
  const float* wb = p.wp + lane * 4;
  const int ecol = lane & 31;                                  // this thread's output column inside a tile (epilogue)

  float4 av[NG], bq0[BD], bq1[BD];
  unsigned w0o = 0, w1o = 0;
  bool valid = false, cmask = false;
  int grp = 0;
  float bias0 = 0.f, bias1 = 0.f;
  // row info + every first load of `item`; results land in the variables above (they are dead once the K loop is done)
  auto issue = [&](int item, int slot) {
    const int tile_x = item / p.ngroups;
    grp = item - tile_x * p.ngroups;
    const int m0 = tile_x * MF;
    w0o = (unsigned)(grp * 2) * (unsigned)KG * 256u; w1o = w0o + (unsigned)KG * 256u;
#pragma unroll
    for (int i = 0; i < BD; ++i) {
      const unsigned g = (unsigned)(wave + 8 * (i < NG ? i : NG - 1));
      bq0[i] = ld4u(wb, w0o + g * 256u); bq1[i] = ld4u(wb, w1o + g * 256u);
    }
    int b = 0, t = 0; long prow = -1; valid = false; cmask = false;
    const int m = m0 + arow;
    if (m < p.M) {
      int bl = m, r = 0;
      if (p.R != 1) { bl = m / p.R; r = m - bl * p.R; }
      b = p.b0 + bl;
      t = step + (p.offs ? p.offs[r] : 0);
      prow = (long)b * p.R + r;
      valid = (t >= 0);
      cmask = p.mask_last && (r == p.R - 1);                   // presum row: the centre tap is contracted by the chain, not here
      if (p.abs_bstride) prow = (long)b * p.abs_bstride + p.abs_row0 + t + p.abs_toff;
    }
    if (wave == 0 && lane < 32) s_prow[slot][arow] = valid ? prow : -1;
    const unsigned xs_row = valid ? (unsigned)(par * p.xs_set + ((long)b * p.xs_bstride + p.xs_row0 + t) * p.xs_stride) : (unsigned)(p.xs_row0 * p.xs_stride);
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int k0 = (wave + 8 * i) * 8, tap = (p.ntaps == 1) ? 0 : (k0 >> 8), c = k0 - tap * p.cin_p + c4;
      const int toff = (tap == 0) ? p.tap_off[0] : ((tap == 1) ? p.tap_off[1] : p.tap_off[2]);
      av[i] = ld4u(p.xsrc, xs_row + (unsigned)(toff * p.xs_stride) + (unsigned)c);
    }
    {
      const int c0 = p.hc ? grp * MF + ecol : (grp * 2) * MF + ecol, c1 = p.hc ? p.cout + grp * MF + ecol : (grp * 2 + 1) * MF + ecol;
      const bool ok0 = p.hc ? (grp * MF + ecol) < p.cout : c0 < p.cout, ok1 = p.hc ? ok0 : c1 < p.cout;
      bias0 = p.bias[ok0 ? (unsigned)c0 : 0u]; bias1 = p.bias[ok1 ? (unsigned)c1 : 0u];
    }
  };

  int item = item0, slot = 0;
  if (item >= nitems) return;
  issue(item, slot);
  for (;;) {
    const int grp_c = grp;
    const float b0c = bias0, b1c = bias1;
    const unsigned w0c = w0o, w1c = w1o;
    const bool valid_c = valid, cmask_c = cmask;
    typedef float f32x16_ __attribute__((ext_vector_type(16)));
    f32x16_ acc0, acc1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      float4 a = av[i];
      if (!valid_c) a = make_float4(0.f, 0.f, 0.f, 0.f);           // skipped rows (t < 0) contribute nothing
      if (NG == 12 && i >= 8) { if (cmask_c) a = make_float4(0.f, 0.f, 0.f, 0.f); }   // k-groups 64..95 = tap 2 (the centre) of a 3 x 256 layer
      const float4 b0 = bq0[i % BD], b1 = bq1[i % BD];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
      if (i + BD < NG) {                                            // refill the slot after the MFMAs that read it (see hsplit_kernel)
        const unsigned gn = (unsigned)(wave + 8 * (i + BD));
        bq0[i % BD] = ld4u(wb, w0c + gn * 256u); bq1[i % BD] = ld4u(wb, w1c + gn * 256u);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      smem[((wave * 2 + 0) * NJ + j) * 64 + lane] = acc0[j];
      smem[((wave * 2 + 1) * NJ + j) * 64 + lane] = acc1[j];
    }
    // next item's loads go out now; the last item re-issues itself (clamped) so that no branch surrounds the loads
    const int next = item + item_stride;
    const bool more = next < nitems;
    issue(more ? next : item, slot ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    // fully unrolled (4 elements per thread): a run-time loop here gets an s_waitcnt vmcnt(0) in its preheader (it contains
    // stores), which would wait for everything just issued for the next item
#pragma unroll
    for (int q = 0; q < (2 * NJ * 64) / 512; ++q) {
      const int e = tid + 512 * q;
      const int l = e & 63, j = (e >> 6) % NJ, tile = e / (64 * NJ);
      float v_ = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v_ += smem[((w * 2 + tile) * NJ + j) * 64 + l];
      const int row = (j & 3) + 8 * (j >> 2) + 4 * (l >> 5), col = l & 31;
      const long orow = s_prow[slot][row];
      int pcol; bool ok;
      if (p.hc) { const int c = grp_c * MF + col; ok = c < p.cout; pcol = tile * p.cout + c; }
      else      { pcol = (grp_c * 2 + tile) * MF + col; ok = pcol < p.cout; }
      if (ok) v_ += tile ? b1c : b0c;
      if (ok && orow >= 0) p.pout[orow * p.np_out + pcol] = v_;
    }
    if (!more) break;
    __syncthreads();                                              // smem / s_prow[slot] are rewritten by the next round
    item = next; slot ^= 1;
  }
}

template <int NG>
__global__ void __launch_bounds__(512) hbulk_kernel(const SplitParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ long s_prow[2][32];
  prefetch_params<3>(p);
  const int step = p.step_val + (p.step ? *p.step : 0);
  const int nitems = ((p.M + 31) / 32) * p.ngroups;
  hbulk_body<NG, SplitParams>(p, step, blockIdx.x, gridDim.x, nitems, smem, s_prow);
}

// Several independent layers of the same shape in ONE launch (v3: the presums of AudioEnc's ten causal k = 3 layers for the
// next frame).  tab[layer] is a frame-independent descriptor in device memory; the frame index is a kernel argument.
// grid = nlayers * items_per_layer, one item per workgroup.
typedef const __attribute__((address_space(4))) SplitParams ConstSplitParams;   // constant address space: uniform field reads are scalar loads
template <int NG>
__global__ void __launch_bounds__(512) hbulk_group_kernel(const SplitParams* __restrict__ tab, const int items_per_layer, const int step,
                                                          const unsigned* __restrict__ wait, const unsigned wait_val, int* __restrict__ wait_err) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ long s_prow[2][32];
  if (wait) {
    // first launch of a side-stream piece: its input row comes from the chain's stream.  Poll the chain's counter here instead of a
    // stream wait operation in front of the launch (~5 us of command-processor time per frame on the longer stream).  Bounded; a
    // time-out raises the decode's error word (dctts_decode_status).
    if (threadIdx.x == 0) {
      bool ok = false;
      for (int i = 0; i < (1 << 20) && !ok; ++i) {
        ok = __hip_atomic_load(wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= wait_val;
        if (!ok) __builtin_amdgcn_s_sleep(16);
      }
      if (!ok) atomicOr(wait_err, 1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  const int layer = blockIdx.x / items_per_layer, item = blockIdx.x - layer * items_per_layer;
  ConstSplitParams& p = *((ConstSplitParams*)tab + layer);
  hbulk_body<NG, ConstSplitParams>(p, step + p.step_val, item, items_per_layer, items_per_layer, smem, s_prow);      // (step_val: a descriptor's own frame offset)
}

// Row kernel for the bulk branch: X[b][t] = act / gate (LN(P[b*R + r])) for cone rows at offsets < 0.
// grid ceil(M/4), block 256 (wave per row).
struct LnRowsParams {
  int M, R, b0; const int* offs; const int* step; int step_val;
  int Rp;                                  // rows per utterance in P (0 = R; v3: R + 1, the last P row of an utterance is the chain's presum)
  int hc; RowNorm nrm;
  float* x; long x_bstride; long x_row0; int x_stride; long x_set;
};

__global__ void __launch_bounds__(256) ln_rows_kernel(const LnRowsParams p) {
  const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= p.M) return;
  const int bl = m / p.R, r = m - bl * p.R, b = p.b0 + bl;
  const int step = p.step_val + (p.step ? *p.step : 0);
  const long par = step & 1;
  const int t = step + (p.offs ? p.offs[r] : 0);
  if (t < 0) return;
  const long prow = (long)b * (p.Rp ? p.Rp : p.R) + r;
  const float4 x = p.hc ? norm_row_hc(p.nrm, prow, b, t, lane, par) : norm_row_c(p.nrm, prow, lane);
  *reinterpret_cast<float4*>(p.x + par * p.x_set + ((long)b * p.x_bstride + p.x_row0 + t) * p.x_stride + lane * 4) = x;
}

// Row-per-workgroup chain of k=1 conv layers (a per-row MLP): AudioDec C_8..C_11 + sigmoid (mel frame j), then
// AudioEnc C_1..C_3 of frame j+1 on the frame just produced -- seven dependent layers in ONE launch.
// One workgroup = one utterance row.  64 k MACs per layer are trivial, so plain VALU FMAs: wave w owns a K slice,
// lane owns 4 output channels, weights stay in TF layout (Cin, Cout) and stream as coalesced 1 KiB wave loads
// (256 KB per layer through one CU ~ 2 us: the bound); partial sums meet in LDS; layer-norm is done by wave 0
// (lane = 4 channels, DPP sums), so no statistics ever leave the workgroup.
// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier): unlike __syncthreads() it does not drain
// outstanding global loads (vmcnt), so loads issued before it keep flying.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}


}  // namespace dctts
