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
// Round 3: the same treatment as hconv_kernel.h -- three LDS buffers with the one barrier of a 32-channel chunk between its two k-groups, A fragment
// and weight fragments requested ahead (BD k-groups of 16 k; a k-group is NT * 4 MFMAs of 32 cycles), SB = wave-uniform tile bases, bias as the
// accumulators' initial value, parameters and every residual requested before the statistics passes, predicated stores instead of `continue`.
template <int EPI, int NT, int NW, int BD = 1, int SB = 0>
__global__ void __launch_bounds__(NW * 64) hconv16_kernel(const ConvParams p, const int m_start) {
  constexpr int LDA = 36;
  constexpr int NH = (EPI == EPI_HC) ? 2 : 1;
  constexpr int NP = (EPI == EPI_HC) ? NT / 2 : NT;
  static_assert(EPI != EPI_HC || (NT % 2 == 0), "HC tiles come in (gate, info) pairs");
  static_assert(BD == 1 || BD == 2, "the register ring is rotated by the 2 k-groups of a chunk");
  __shared__ __attribute__((aligned(16))) float As[3][16 * LDA];
  __shared__ float red[NW * 2 * 16];
  __shared__ float tot[2][2 * 16];
  __shared__ long s_inrow[16];
  __shared__ long s_outrow[16];
  __shared__ long s_out2row[16];
  __shared__ long s_pre[16];             // >= 0: presum row, value = utterance index (decode v3)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = SB ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
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

  // ---- A loader: thread (lrow, lc4) moves one float4 per chunk (16 rows x 32 channels); threads >= 128 load duplicates
  const int lrow = (tid >> 3) & 15, lc4 = tid & 7;
  const long my_inrow = s_inrow[lrow];
  const int mask_tap = (s_pre[lrow] >= 0) ? p.ntaps - 1 : -1;        // a presum row leaves its last (centre) tap to the chain
  const int cpt = p.cin_p >> 5;
  const int nch = p.ntaps * cpt;
  const int KG = nch * 2;                // k-groups of 16
  const bool row_ok = my_inrow >= 0;
  const long safe_row = p.gather ? 0 : p.in_row0;
  int ltap = 0, lcit = 0;
  auto load_next = [&](bool& ok) -> float4 {     // branch-free (hconv_kernel.h): rows / columns / taps that must read as zero load something readable
    const int c = lcit * 32 + lc4 * 4;
    const int toff = (ltap == 0) ? p.tap_off[0] : ((ltap == 1) ? p.tap_off[1] : p.tap_off[2]);
    ok = row_ok && c < p.cin && ltap != mask_tap;
    const long row = row_ok ? my_inrow + toff : safe_row;
    const float4 v = *reinterpret_cast<const float4*>(p.in + row * (long)p.in_stride + (c < p.cin ? c : 0));
    if (!(ltap == p.ntaps - 1 && lcit == cpt - 1)) { if (++lcit == cpt) { lcit = 0; ++ltap; } }
    return v;
  };

  const float4* wq[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
    wq[i] = reinterpret_cast<const float4*>(p.wp) + ((long)(wave * NT + i) * KG) * 64 + (SB ? 0 : lane);
  const int wl = SB ? lane : 0;

  const int C = p.cout;
  int chan[NT]; bool cval[NT];
  f32x4_ acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    int ch_, bidx;
    if (EPI == EPI_HC) { ch_ = (wave * NP + (i >> 1)) * 16 + arow; bidx = (i & 1) * C + ch_; }
    else { ch_ = (wave * NT + i) * 16 + arow; bidx = ch_; }
    chan[i] = ch_; cval[i] = ch_ < C;
    const float bv = p.bias[cval[i] ? bidx : 0];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = cval[i] ? bv : 0.f;
  }

  bool aok, aok1;
  float4 a0 = load_next(aok);
  float4 a1 = load_next(aok1);
  float4 bq[BD][NT];
#pragma unroll
  for (int d = 0; d < BD; ++d)
#pragma unroll
    for (int i = 0; i < NT; ++i) bq[d][i] = wq[i][(d < KG ? d : KG - 1) * 64 + wl];
  if (!aok) a0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!aok1) a1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 128) {
    *reinterpret_cast<float4*>(&As[0][lrow * LDA + lc4 * 4]) = a0;
    *reinterpret_cast<float4*>(&As[1][lrow * LDA + lc4 * 4]) = a1;
  }
  float4 areg = load_next(aok);
  __syncthreads();
  const int aoff = arow * LDA + aq * 4;
  float4 a = *reinterpret_cast<const float4*>(&As[0][aoff]);
  int cb = 0;
  for (int ch = 0; ch < nch; ++ch) {
    const float* Ab = As[cb];
    const int cb1 = (cb == 2) ? 0 : cb + 1, cb2 = (cb1 == 2) ? 0 : cb1 + 1;
    const float* An = As[cb1];
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      const int kg = ch * 2 + gq;
      const int kgn = (kg + BD < KG) ? kg + BD : KG - 1;
      if (gq == 1) {
        __syncthreads();
        if (!aok) areg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ch + 2 < nch && tid < 128) *reinterpret_cast<float4*>(&As[cb2][lrow * LDA + lc4 * 4]) = areg;
        areg = load_next(aok);
      }
      const float4 an = (gq == 0) ? *reinterpret_cast<const float4*>(&Ab[aoff + 16]) : *reinterpret_cast<const float4*>(&An[aoff]);
      float4 bnext[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) bnext[i] = wq[i][kgn * 64 + wl];
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq[0][i].x, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq[0][i].y, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq[0][i].z, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq[0][i].w, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int d = 0; d + 1 < BD; ++d) bq[d][i] = bq[d + 1][i];
        bq[BD - 1][i] = bnext[i];
      }
      a = an;
    }
    cb = cb1;
  }

  // ===================================================================================== epilogue
  // acc[i][j] = conv output (bias included) at row aq*4 + j, column arow of 16-wide tile i.  Requests first, statistics, then stores.
  float pg1[NP], pb1[NP], pg2[NP], pb2[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int i = (EPI == EPI_HC) ? 2 * k : k;
    const int cs = cval[i] ? chan[i] : 0;
    pg1[k] = p.g1[cs]; pb1[k] = p.b1[cs];
    if (EPI == EPI_HC) { pg2[k] = p.g2[cs]; pb2[k] = p.b2[cs]; }
  }
  float xr[(EPI == EPI_HC) ? NP : 1][4];
  if (EPI == EPI_HC) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long ir = s_inrow[aq * 4 + j];
      const float* rp = p.in + (ir >= 0 ? ir : safe_row) * (long)p.in_stride;
#pragma unroll
      for (int k = 0; k < NP; ++k) xr[k][j] = rp[cval[2 * k] ? chan[2 * k] : 0];
    }
  }

  const float invC = 1.0f / (float)C;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float s[NH][4], mean[NH][4];
    if (pass == 1) {
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) mean[h][j] = tot[0][h * 16 + aq * 4 + j];
    }
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
      v *= invC;
      tot[pass][h * 16 + r] = (pass == 0) ? v : 1.0f / sqrtf(v + 1e-12f);
    }
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = aq * 4 + j;
    const long orow = s_outrow[row];
    const bool ok = orow >= 0;
    const long pre = s_pre[row];
    float* op = p.out + (ok ? orow : 0) * (long)p.out_stride;
    const float r0 = tot[1][row], r1 = tot[1][16 + row];
    const float m0_ = tot[0][row], m1_ = tot[0][16 + row];
    if (EPI == EPI_HC) {
      float* pr = p.presum_out + (pre >= 0 ? pre : 0) * p.presum_rstride;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const float y1 = (acc[2 * k][j] - m0_) * r0 * pg1[k] + pb1[k];
        const float y2 = (acc[2 * k + 1][j] - m1_) * r1 * pg2[k] + pb2[k];
        const float gt = fast_sigmoidf_(y1);
        const float o = gt * y2 + (1.0f - gt) * xr[k][j];
        if (ok && cval[2 * k]) {
          if (pre >= 0) { pr[chan[2 * k]] = acc[2 * k][j]; pr[C + chan[2 * k]] = acc[2 * k + 1][j]; }   // presum row: bias + older taps, un-normalised, for the chain
          else op[chan[2 * k]] = o;
        }
      }
    } else {
      float* op2 = p.out2 ? p.out2 + (ok ? s_out2row[row] : 0) * (long)p.out2_stride : nullptr;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        float y = (acc[k][j] - m0_) * r0 * pg1[k] + pb1[k];
        if (op2 && ok && cval[k]) op2[chan[k]] = y;
        if (p.act == ACT_RELU) y = fmaxf(y, 0.f);
        else if (p.act == ACT_SIGMOID) y = fast_sigmoidf_(y);
        if (ok && (cval[k] || chan[k] < p.out_zero_to)) op[chan[k]] = cval[k] ? y : 0.f;
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
