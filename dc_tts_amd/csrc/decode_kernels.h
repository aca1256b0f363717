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

// ------------------------------------------------------------------------------------------------------------------------
// hbulk_kernel: the bulk (cone) branch's contraction, software-pipelined across work items.  Same arithmetic and the same
// summation order as hsplit_kernel<32> (32 rows x one gate/info pair of 32-column tiles, K split over 8 waves, fixed-order
// LDS reduction), PRO_RAW only (its input rows are materialised by ln_rows_kernel).  What differs is the schedule:
//   item n:  K loop | accumulators -> LDS | ISSUE item n+1's row info + A + first weight groups | barrier | epilogue of n
// so the ~1.5 us a work item used to wait for its first loads, and the scalar set-up before them, overlap the reduction and
// the stores of the previous one.  The epilogue issues no loads of its own (its two bias values ride along with the item's
// loads): a load there would have to be waited for with vmcnt(0), i.e. behind everything just issued for the next item.
// BD_ = weight groups in flight per wave.  4 for the bulk launches (items stream through a workgroup; the next item's first groups are the prefetch); NG -- every
// group requested up front -- for a passenger of the chain's launch (xtail_kernel.h): one item per workgroup, on a compute unit the side stream is waiting for, and
// at 4 in flight the item was two dependent rounds of memory latency long (9 - 11 us for 3.4 us of MFMA).
// ONE: one item per workgroup (item0 only), and no second round of requests to keep the compute unit for.
template <int NG, typename PT, int BD_ = 4, bool ONE = false>
__device__ __forceinline__ void hbulk_body(const PT& p, const int step, const int item0, const int item_stride, const int nitems,
                                           float* smem, long (*s_prow)[32]) {
  constexpr int MF = 32, NJ = 16, BD = BD_ < NG ? BD_ : NG, KG = NG * 8;
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
    const bool more = ONE ? false : next < nitems;
    if constexpr (!ONE) issue(more ? next : item, slot ^ 1);
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
