// hconv16_kernel.h -- the 16-row form of hconv_kernel.h (same fusion: conv taps + bias + LayerNorm + act / highway gate),
// built on v_mfma_f32_16x16x4_f32.
//
// Why it exists: a workgroup of hconv_kernel owns 32 rows, so a layer with M rows is M/32 equal work items.  At the bench
// shape the 4T-resolution SSRN layers have 840 items for 256 CUs = 3.28 "rounds": the last round runs on 28 % of the
// chip (82 % tail efficiency).  The host therefore gives the first floor(items / 256) * 256 items to hconv_kernel (exact
// rounds) and the remaining rows to this kernel, whose items are 16 rows = half the time each: 3.5 rounds instead of 4.
// It is also the better form for small M (fewer idle CUs).  A 16x16x4 MFMA feeds half as many rows per B fragment as
// 32x32x2, so this form pulls twice the weight bytes per FLOP -- fine for a tail, not for the bulk of a layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "attn_kernels.h"
#include "hconv_kernel.h"

namespace dctts {

typedef float f32x4_ __attribute__((ext_vector_type(4)));

// Same ConvParams as hconv_kernel; wp must be packed for 16-column tiles ([tile][k-group of 16][lane][4]);
// m_start = first output row of this launch (rows [m_start, M)).
template <int EPI, int NT, int NW>
__global__ void __launch_bounds__(NW * 64) hconv16_kernel(const ConvParams p, const int m_start) {
  constexpr int LDA = 36;
  constexpr int NH = (EPI == EPI_HC) ? 2 : 1;
  static_assert(EPI != EPI_HC || (NT % 2 == 0), "HC tiles come in (gate, info) pairs");
  __shared__ __attribute__((aligned(16))) float As[2][16 * LDA];
  __shared__ float red[NW * 2 * 16];
  __shared__ float tot[2 * 16];
  __shared__ long s_inrow[16];
  __shared__ long s_outrow[16];
  __shared__ long s_out2row[16];
  __shared__ long s_pre[16];             // >= 0: presum row, value = utterance index (decode v3)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int arow = lane & 15, aq = lane >> 4;
  const int m0 = m_start + blockIdx.x * 16;
  const int t_base = p.step ? *p.step : p.t_base_val;

  if (tid < 16) {
    const int m = m0 + tid;
    long inrow = -1, outrow = -1, out2row = -1, pre = -1;
    if (m < p.M) {
      const int b = m / p.R, r = m - b * p.R;
      const int t = t_base + (p.offs ? p.offs[r] : r);
      if (t >= 0) {
        if (p.presum_out && r == p.R - 1) pre = b;
        if (p.gather) {       // embedding lookup (modules.py:13-42): an id outside the table reads row 0 (= the all-zero PAD row), never out of bounds
          const int id = p.gather[m];
          inrow = (id >= 0 && id < p.gather_n) ? id : 0;
        } else {
          inrow = (long)b * p.in_bstride + p.in_row0 + t;
        }
        outrow = (long)b * p.out_bstride + p.out_row0 + (long)t * p.out_tmul + p.out_tadd;
        out2row = (long)b * p.out2_bstride + p.out2_row0 + t;
      }
    }
    s_inrow[tid] = inrow; s_outrow[tid] = outrow; s_out2row[tid] = out2row; s_pre[tid] = pre;
  }
  __syncthreads();

  // ---- A loader: thread (lrow, lc4) moves one float4 per chunk (16 rows x 32 channels)
  const int lrow = (tid >> 3) & 15, lc4 = tid & 7;
  const long my_inrow = s_inrow[lrow];
  const int mask_tap = (s_pre[lrow] >= 0) ? p.ntaps - 1 : -1;        // a presum row leaves its last (centre) tap to the chain
  const int cpt = p.cin_p >> 5;
  const int nch = p.ntaps * cpt;
  const int KG = nch * 2;                // k-groups of 16

  auto load_chunk = [&](int ch) -> float4 {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int tap = ch / cpt;
    const int c = (ch - tap * cpt) * 32 + lc4 * 4;
    if (tid < 128 && my_inrow >= 0 && c < p.cin && tap != mask_tap) {
      const int toff = (tap == 0) ? p.tap_off[0] : ((tap == 1) ? p.tap_off[1] : p.tap_off[2]);
      v = *reinterpret_cast<const float4*>(p.in + (my_inrow + toff) * (long)p.in_stride + c);
    }
    return v;
  };

  const float4* wq[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
    wq[i] = reinterpret_cast<const float4*>(p.wp) + ((long)(wave * NT + i) * KG) * 64 + lane;

  f32x4_ acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  float4 areg = load_chunk(0);
  if (tid < 128) *reinterpret_cast<float4*>(&As[0][lrow * LDA + lc4 * 4]) = areg;
  float4 bcur[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) bcur[i] = wq[i][0];
  __syncthreads();

  for (int ch = 0; ch < nch; ++ch) {
    const bool more = (ch + 1 < nch);
    if (more) areg = load_chunk(ch + 1);
    const float* Ab = As[ch & 1];
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      const int kg = ch * 2 + gq;
      const int kgn = (kg + 1 < KG) ? kg + 1 : kg;
      const float4 a = *reinterpret_cast<const float4*>(&Ab[arow * LDA + gq * 16 + aq * 4]);
      float4 bnext[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) bnext[i] = wq[i][(long)kgn * 64];
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bcur[i].x, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bcur[i].y, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bcur[i].z, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bcur[i].w, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) bcur[i] = bnext[i];
    }
    if (more && tid < 128) *reinterpret_cast<float4*>(&As[(ch + 1) & 1][lrow * LDA + lc4 * 4]) = areg;
    __syncthreads();
  }

  // ===================================================================================== epilogue
  // acc[i][j] = conv output at row aq*4 + j, column arow of 16-wide tile i.
  const int C = p.cout;
  int chan[NT];
  bool cval[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    int ch_, bidx;
    if (EPI == EPI_HC) {
      const int pp = wave * (NT / 2) + (i >> 1);
      ch_ = pp * 16 + arow;
      bidx = (i & 1) * C + ch_;
    } else {
      ch_ = (wave * NT + i) * 16 + arow;
      bidx = ch_;
    }
    chan[i] = ch_;
    cval[i] = ch_ < C;
    const float bv = cval[i] ? p.bias[bidx] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] += bv;
  }

  float mean[NH][4], rstd[NH][4];
  const float invC = 1.0f / (float)C;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float s[NH][4];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) s[h][j] = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int h = (EPI == EPI_HC) ? (i & 1) : 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (pass == 0) s[h][j] += acc[i][j];
        else { const float d = cval[i] ? (acc[i][j] - mean[h][j]) : 0.f; s[h][j] += d * d; }
      }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) s[h][j] = row16_sum(s[h][j]);      // the 16 lanes of a DPP row share aq, i.e. the same 4 rows
    if (arow == 0) {
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[(wave * 2 + h) * 16 + aq * 4 + j] = s[h][j];
    }
    __syncthreads();
    if (tid < 16 * NH) {
      const int h = tid >> 4, r = tid & 15;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[(w * 2 + h) * 16 + r];
      tot[h * 16 + r] = v * invC;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = tot[h * 16 + aq * 4 + j];
        if (pass == 0) mean[h][j] = v; else rstd[h][j] = 1.0f / sqrtf(v + 1e-12f);
      }
    __syncthreads();
  }

  if (EPI == EPI_HC) {
#pragma unroll
    for (int k = 0; k < NT / 2; ++k) {
      const int ch_ = chan[2 * k];
      if (!cval[2 * k]) continue;
      const float g1 = p.g1[ch_], b1 = p.b1[ch_], g2 = p.g2[ch_], b2 = p.b2[ch_];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = aq * 4 + j;
        const long orow = s_outrow[row];
        if (orow < 0) continue;
        if (s_pre[row] >= 0) {                                       // presum row: bias + older taps, un-normalised, for the chain
          float* pr = p.presum_out + s_pre[row] * p.presum_rstride;
          pr[ch_] = acc[2 * k][j]; pr[C + ch_] = acc[2 * k + 1][j];
          continue;
        }
        const float y1 = (acc[2 * k][j] - mean[0][j]) * rstd[0][j] * g1 + b1;
        const float y2 = (acc[2 * k + 1][j] - mean[1][j]) * rstd[1][j] * g2 + b2;
        const float gt = sigmoidf_(y1);
        const float xr = p.in[s_inrow[row] * (long)p.in_stride + ch_];
        p.out[orow * (long)p.out_stride + ch_] = gt * y2 + (1.0f - gt) * xr;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int ch_ = chan[i];
      if (cval[i]) {
        const float g1 = p.g1[ch_], b1 = p.b1[ch_];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = aq * 4 + j;
          const long orow = s_outrow[row];
          if (orow < 0) continue;
          float y = (acc[i][j] - mean[0][j]) * rstd[0][j] * g1 + b1;
          if (p.out2) p.out2[s_out2row[row] * (long)p.out2_stride + ch_] = y;
          if (p.act == ACT_RELU) y = fmaxf(y, 0.f);
          else if (p.act == ACT_SIGMOID) y = sigmoidf_(y);
          p.out[orow * (long)p.out_stride + ch_] = y;
        }
      } else if (ch_ < p.out_zero_to) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long orow = s_outrow[aq * 4 + j];
          if (orow >= 0) p.out[orow * (long)p.out_stride + ch_] = 0.f;
        }
      }
    }
  }
}

inline ConvShape pick_shape16(int epi, int cout) {
  if (epi == EPI_HC) {
    const int tiles = 2 * ((cout + 15) / 16);
    if (tiles <= 64) return {EPI_HC, 8, 8};
    return {EPI_HC, 16, 8};
  }
  const int tiles = (cout + 15) / 16;
  if (tiles <= 64) return {EPI_C, 8, 8};
  return {EPI_C, 6, 11};
}

hipError_t launch_hconv16(const ConvShape& s, const ConvParams& p, int m_start, hipStream_t stream);

}  // namespace dctts
