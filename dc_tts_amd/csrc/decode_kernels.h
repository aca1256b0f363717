// decode_kernels.h -- latency-oriented kernels for one step of the Text2Mel autoregressive loop
// (synthesize.py:47-54) on gfx950.
//
// Why a second kernel family: in a decode step every AudioEnc layer and the tail of the AudioDec cone
// see only B (= 32) rows.  A workgroup that owns all output columns of a row block (hconv_kernel.h, the
// throughput form) then leaves 255 of 256 CUs idle and takes ~75 us per layer.  Here the output columns
// are split across workgroups (one 2-tile column group each) and K is split across the 8 waves of a
// workgroup, so a 32-row layer becomes 16-32 short workgroups.  Layer-norm needs whole rows, so it is
// DEFERRED: the GEMM writes pre-norm values P, and whoever consumes a row normalises it:
//   * chain layers (the newest frame j): the consumer's prologue normalises + gates its 16 centre rows
//     from P (wave per row, two-pass statistics in registers) while staging its A tile into LDS, and the
//     first column group materialises those rows into the layer's absolute-time history buffer;
//   * bulk layers (cone rows at offsets < 0, independent of frame j): a row kernel (ln_rows_kernel).
//
// hsplit_kernel<MF>: MF = 32 -> 32 rows x 2 tiles of v_mfma_f32_32x32x2_f32   (bulk cone layers)
//                    MF = 16 -> 16 rows x 2 tiles of v_mfma_f32_16x16x4_f32   (chain layers)
// A is staged once per workgroup into LDS ([tap][row][Cin+4], conflict-free ds_read_b128); B comes from
// HBM/L2 in MFMA fragment order (one coalesced 1 KiB load per wave per four MFMAs); the 8 partial
// accumulators are reduced through LDS (reusing the A region) in a fixed order -> deterministic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "attn_kernels.h"
#include "hconv_kernel.h"

namespace dctts {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { PRO_RAW = 0, PRO_LN_C = 1, PRO_LN_HC = 2 };

// How to obtain one 256-channel activation row X[b][t] that exists only as pre-norm values P.
struct RowNorm {
  const float* P; int np;                  // pre-norm rows [prow][np]  (np = 256 for C, 512 for HC)
  const float* g1; const float* b1; const float* g2; const float* b2; int act;
  const float* res; long res_bstride; long res_row0; int res_stride;   // highway residual X_{l-1}[b][t] (HC)
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// One wave normalises one 256-channel row; lane owns channels 4*lane .. 4*lane+3.
__device__ __forceinline__ float4 norm_row_c(const RowNorm& n, long prow, int lane) {
  const int c = lane * 4;
  const float4 x = ld4(n.P + prow * n.np + c);
  const float mean = wave_sum(x.x + x.y + x.z + x.w) * (1.0f / 256.0f);
  const float4 d = make_float4(x.x - mean, x.y - mean, x.z - mean, x.w - mean);
  const float var = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * (1.0f / 256.0f);
  const float rs = 1.0f / sqrtf(var + 1e-12f);
  const float4 g = ld4(n.g1 + c), b = ld4(n.b1 + c);
  float4 y = make_float4(d.x * rs * g.x + b.x, d.y * rs * g.y + b.y, d.z * rs * g.z + b.z, d.w * rs * g.w + b.w);
  if (n.act == ACT_RELU) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
  return y;
}

__device__ __forceinline__ float4 norm_row_hc(const RowNorm& n, long prow, int b, int t, int lane) {
  const int c = lane * 4;
  const float4 h1 = ld4(n.P + prow * n.np + c), h2 = ld4(n.P + prow * n.np + 256 + c);
  const float4 xr = ld4(n.res + ((long)b * n.res_bstride + n.res_row0 + t) * n.res_stride + c);
  const float m1 = wave_sum(h1.x + h1.y + h1.z + h1.w) * (1.0f / 256.0f);
  const float m2 = wave_sum(h2.x + h2.y + h2.z + h2.w) * (1.0f / 256.0f);
  const float4 d1 = make_float4(h1.x - m1, h1.y - m1, h1.z - m1, h1.w - m1);
  const float4 d2 = make_float4(h2.x - m2, h2.y - m2, h2.z - m2, h2.w - m2);
  const float v1 = wave_sum(d1.x * d1.x + d1.y * d1.y + d1.z * d1.z + d1.w * d1.w) * (1.0f / 256.0f);
  const float v2 = wave_sum(d2.x * d2.x + d2.y * d2.y + d2.z * d2.z + d2.w * d2.w) * (1.0f / 256.0f);
  const float r1 = 1.0f / sqrtf(v1 + 1e-12f), r2 = 1.0f / sqrtf(v2 + 1e-12f);
  const float4 g1 = ld4(n.g1 + c), b1 = ld4(n.b1 + c), g2 = ld4(n.g2 + c), b2 = ld4(n.b2 + c);
  float4 o;
  { const float s = sigmoidf_(d1.x * r1 * g1.x + b1.x); o.x = s * (d2.x * r2 * g2.x + b2.x) + (1.0f - s) * xr.x; }
  { const float s = sigmoidf_(d1.y * r1 * g1.y + b1.y); o.y = s * (d2.y * r2 * g2.y + b2.y) + (1.0f - s) * xr.y; }
  { const float s = sigmoidf_(d1.z * r1 * g1.z + b1.z); o.z = s * (d2.z * r2 * g2.z + b2.z) + (1.0f - s) * xr.z; }
  { const float s = sigmoidf_(d1.w * r1 * g1.w + b1.w); o.w = s * (d2.w * r2 * g2.w + b2.w) + (1.0f - s) * xr.w; }
  return o;
}

struct SplitParams {
  // ---- row mapping: m in [0,M) -> b = b0 + m / R, r = m % R, t = *step + (offs ? offs[r] : 0); rows with t < 0 are skipped
  int M, R, b0; const int* offs; const int* step;
  // ---- centre tap (row t itself): PRO_RAW reads xsrc; PRO_LN_* rebuilds it from pre-norm rows (index b*R + r)
  int pro; RowNorm nrm;
  float* xmat; long xm_bstride; long xm_row0; int xm_stride;    // where column group 0 materialises the rebuilt row
  // ---- tap source (absolute-time activation buffer): all taps when PRO_RAW, the non-centre taps otherwise
  const float* xsrc; long xs_bstride; long xs_row0; int xs_stride;
  int ntaps; int tap_off[3]; int cin; int cin_p;
  // ---- weights / output
  const float* wp; const float* bias; int cout; int hc; int np_out;
  float* pout;                                                   // pre-norm rows [b*R + r][np_out]
};

template <int MF>
__global__ void __launch_bounds__(512) hsplit_kernel(const SplitParams p) {
  constexpr int BM = MF;
  constexpr int KGS = (MF == 32) ? 8 : 16;          // k per k-group (4 MFMAs)
  constexpr int NJ = (MF == 32) ? 16 : 4;           // accumulator registers per tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_b[BM], s_t[BM];
  __shared__ long s_prow[BM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, grp = blockIdx.y;
  const int LDS_S = p.cin_p + 4;
  const int step = *p.step;

  if (tid < BM) {
    const int m = m0 + tid;
    int b = -1, t = -1; long prow = -1;
    if (m < p.M) {
      const int bl = m / p.R, r = m - bl * p.R;
      b = p.b0 + bl;
      t = step + (p.offs ? p.offs[r] : 0);
      prow = (long)b * p.R + r;
      if (t < 0) b = -1;
    }
    s_b[tid] = b; s_t[tid] = t; s_prow[tid] = prow;
  }
  __syncthreads();

  // ---- stage the A tile: wave per (tap, row)
  for (int idx = wave; idx < p.ntaps * BM; idx += 8) {
    const int tap = idx / BM, row = idx - tap * BM;
    float* dst = smem + (long)idx * LDS_S;
    const int b = s_b[row], t = s_t[row];
    const int toff = (tap == 0) ? p.tap_off[0] : ((tap == 1) ? p.tap_off[1] : p.tap_off[2]);
    if (b < 0) {
      for (int c = lane * 4; c < p.cin_p; c += 256) *reinterpret_cast<float4*>(dst + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else if (toff == 0 && p.pro != PRO_RAW) {
      const float4 x = (p.pro == PRO_LN_C) ? norm_row_c(p.nrm, s_prow[row], lane) : norm_row_hc(p.nrm, s_prow[row], b, t, lane);
      *reinterpret_cast<float4*>(dst + lane * 4) = x;
      if (grp == 0 && p.xmat) *reinterpret_cast<float4*>(p.xmat + ((long)b * p.xm_bstride + p.xm_row0 + t) * p.xm_stride + lane * 4) = x;
    } else {
      const float* src = p.xsrc + ((long)b * p.xs_bstride + p.xs_row0 + t + toff) * p.xs_stride;
      for (int c = lane * 4; c < p.cin_p; c += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < p.cin) v = ld4(src + c);
        *reinterpret_cast<float4*>(dst + c) = v;
      }
    }
  }
  __syncthreads();

  // ---- K loop: wave w owns k-groups w, w+8, ...
  const int KG = p.ntaps * p.cin_p / KGS;
  const float4* w0 = reinterpret_cast<const float4*>(p.wp) + ((long)(grp * 2) * KG) * 64 + lane;
  const float4* w1 = w0 + (long)KG * 64;
  const int arow = (MF == 32) ? (lane & 31) : (lane & 15);
  const int aq = (MF == 32) ? (lane >> 5) : (lane >> 4);
  typedef typename std::conditional<MF == 32, f32x16, f32x4>::type acc_t;
  acc_t acc0, acc1;
#pragma unroll
  for (int j = 0; j < NJ; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
  int g = wave;
  float4 bq0 = make_float4(0.f, 0.f, 0.f, 0.f), bq1 = bq0;
  if (g < KG) { bq0 = w0[(long)g * 64]; bq1 = w1[(long)g * 64]; }
  for (; g < KG; g += 8) {
    const int k0 = g * KGS;
    const int tap = k0 / p.cin_p, koff = k0 - tap * p.cin_p;
    const float4 a = *reinterpret_cast<const float4*>(smem + ((long)tap * BM + arow) * LDS_S + koff + aq * 4);
    const int gn = (g + 8 < KG) ? g + 8 : g;
    const float4 n0 = w0[(long)gn * 64], n1 = w1[(long)gn * 64];
    if constexpr (MF == 32) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq0.x, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq0.y, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq0.z, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq0.w, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq1.w, acc1, 0, 0, 0);
    } else {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq0.x, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq0.y, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq0.z, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq0.w, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq1.w, acc1, 0, 0, 0);
    }
    bq0 = n0; bq1 = n1;
  }
  __syncthreads();                    // every wave is done reading the A tile: reuse it for the reduction

  // red[wave][tile][j][lane]
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    smem[((wave * 2 + 0) * NJ + j) * 64 + lane] = acc0[j];
    smem[((wave * 2 + 1) * NJ + j) * 64 + lane] = acc1[j];
  }
  __syncthreads();
  for (int e = tid; e < 2 * NJ * 64; e += 512) {
    const int l = e & 63, j = (e >> 6) % NJ, tile = e / (64 * NJ);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += smem[((w * 2 + tile) * NJ + j) * 64 + l];
    int row, col;
    if constexpr (MF == 32) { row = (j & 3) + 8 * (j >> 2) + 4 * (l >> 5); col = l & 31; }
    else          { row = (l >> 4) * 4 + j;                      col = l & 15; }
    if (s_b[row] < 0) continue;
    int pcol; bool ok;
    if (p.hc) { const int c = grp * MF + col; ok = c < p.cout; pcol = tile * p.cout + c; }
    else      { pcol = (grp * 2 + tile) * MF + col; ok = pcol < p.cout; }
    if (ok) p.pout[s_prow[row] * p.np_out + pcol] = v + p.bias[pcol];
  }
}

// Row kernel for the bulk branch: X[b][t] = act/ gate (LN(P[b*R + r])) for cone rows at offsets < 0.
// grid ceil(M/4), block 256 (wave per row).
struct LnRowsParams {
  int M, R, b0; const int* offs; const int* step;
  int hc; RowNorm nrm;
  float* x; long x_bstride; long x_row0; int x_stride;
};

__global__ void __launch_bounds__(256) ln_rows_kernel(const LnRowsParams p) {
  const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= p.M) return;
  const int bl = m / p.R, r = m - bl * p.R, b = p.b0 + bl;
  const int t = *p.step + (p.offs ? p.offs[r] : 0);
  if (t < 0) return;
  const long prow = (long)b * p.R + r;
  const float4 x = p.hc ? norm_row_hc(p.nrm, prow, b, t, lane) : norm_row_c(p.nrm, prow, lane);
  *reinterpret_cast<float4*>(p.x + ((long)b * p.x_bstride + p.x_row0 + t) * p.x_stride + lane * 4) = x;
}

// Newest-frame attention (row offset 0): rebuilds Q[j] from AudioEnc's last pre-norm rows, materialises it
// into the Q history, runs the 3-key windowed softmax, writes R[j] and the arg-max for the next step.
// grid ceil(Bg/4), block 256 (wave per utterance).
struct AttnRow0Params {
  int Bg, b0, B; const int* step;
  RowNorm nrm;                                              // Q[j] = gate(LN(P_last[b]))  (R == 1: prow = b)
  float* qhist; long q_bstride; long q_row0; int q_stride;
  const float* K; const float* V; int kv_stride; long kv_bstride; int N, d, win;
  int* pm_all;
  float* rbuf; long r_bstride; long r_row0;
};

__global__ void __launch_bounds__(256) attention_row0_kernel(const AttnRow0Params p) {
  const int lane = threadIdx.x & 63, bl = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bl >= p.Bg) return;
  const int b = p.b0 + bl, j = *p.step, c0 = lane * 4;
  const float4 q = norm_row_hc(p.nrm, (long)b, b, j, lane);
  *reinterpret_cast<float4*>(p.qhist + ((long)b * p.q_bstride + p.q_row0 + j) * p.q_stride + c0) = q;
  const int pm = p.pm_all[(long)j * p.B + b];
  const float scale = 1.0f / sqrtf((float)p.d);
  float lg[3]; float4 vv[3];
  int nk = p.N - pm; if (nk > p.win) nk = p.win;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    lg[k] = -INFINITY; vv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < nk) {
      const long row = (long)b * p.kv_bstride + pm + k;
      const float4 kk = ld4(p.K + row * p.kv_stride + c0);
      vv[k] = ld4(p.V + row * p.kv_stride + c0);
      float a = q.x * kk.x; a = fmaf(q.y, kk.y, a); a = fmaf(q.z, kk.z, a); a = fmaf(q.w, kk.w, a);
      lg[k] = wave_sum(a) * scale;
    }
  }
  float mx = lg[0]; int am = 0;
#pragma unroll
  for (int k = 1; k < 3; ++k) if (lg[k] > mx) { mx = lg[k]; am = k; }
  float e[3], se = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) { e[k] = (k < nk) ? expf(lg[k] - mx) : 0.f; se += e[k]; }
  const float inv = 1.0f / se;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float a = e[k] * inv;
    o.x = fmaf(a, vv[k].x, o.x); o.y = fmaf(a, vv[k].y, o.y); o.z = fmaf(a, vv[k].z, o.z); o.w = fmaf(a, vv[k].w, o.w);
  }
  float* rrow = p.rbuf + ((long)b * p.r_bstride + p.r_row0 + j) * (2 * p.d);
  *reinterpret_cast<float4*>(rrow + c0) = o;
  *reinterpret_cast<float4*>(rrow + p.d + c0) = q;
  if (lane == 0) p.pm_all[(long)(j + 1) * p.B + b] = pm + am;
}

// End of the chain: mel frame j = sigmoid(LN(P_last[b])) over n_mels channels -> S[j+1] (ypad row j+1) and the
// raw logits.  grid ceil(Bg/4), block 256 (wave per utterance).  n_mels <= 128.
struct FinalizeParams {
  int Bg, b0; const int* step;
  const float* P; int np; const float* g; const float* be; int n;
  float* ypad; long y_bstride; long y_row0; int y_stride;       // y_row0 already includes the +1 shift (train.py:51)
  float* logits; long l_bstride; int l_stride;
};

__global__ void __launch_bounds__(256) finalize_kernel(const FinalizeParams p) {
  const int lane = threadIdx.x & 63, bl = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bl >= p.Bg) return;
  const int b = p.b0 + bl, j = *p.step;
  const float* row = p.P + (long)b * p.np;
  const int c0 = lane, c1 = lane + 64;
  const float x0 = (c0 < p.n) ? row[c0] : 0.f, x1 = (c1 < p.n) ? row[c1] : 0.f;
  const float invn = 1.0f / (float)p.n;
  const float mean = wave_sum(x0 + x1) * invn;
  const float d0 = (c0 < p.n) ? x0 - mean : 0.f, d1 = (c1 < p.n) ? x1 - mean : 0.f;
  const float rs = 1.0f / sqrtf(wave_sum(d0 * d0 + d1 * d1) * invn + 1e-12f);
  float* yrow = p.ypad + ((long)b * p.y_bstride + p.y_row0 + j) * p.y_stride;
  float* lrow = p.logits + ((long)b * p.l_bstride + j) * p.l_stride;
  if (c0 < p.n) { const float y = d0 * rs * p.g[c0] + p.be[c0]; lrow[c0] = y; yrow[c0] = sigmoidf_(y); }
  if (c1 < p.n) { const float y = d1 * rs * p.g[c1] + p.be[c1]; lrow[c1] = y; yrow[c1] = sigmoidf_(y); }
}

}  // namespace dctts
