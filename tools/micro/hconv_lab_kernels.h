// hconv_lab_kernels.h -- experimental copies of hconv_kernel (tools/micro/hconv_lab.hip).  Not product code.
#pragma once
#include <hip/hip_runtime.h>

#include "hconv_kernel.h"

namespace dctts {

// The production epilogue (bias, two-pass layer-norm over the workgroup's rows, gate / activation, store), as a function.
template <int EPI, int NT, int NW>
__device__ __forceinline__ void lab_epilogue(const ConvParams& p, f32x16 (&acc)[NT], const long* s_inrow, const long* s_outrow, const long* s_out2row,
                                             float* red, float* tot) {
  constexpr int NH = (EPI == EPI_HC) ? 2 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int C = p.cout;
  int chan[NT];
  bool cval[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    int ch_, bidx;
    if (EPI == EPI_HC) { const int pp = wave * (NT / 2) + (i >> 1); ch_ = pp * 32 + l31; bidx = (i & 1) * C + ch_; }
    else { ch_ = (wave * NT + i) * 32 + l31; bidx = ch_; }
    chan[i] = ch_; cval[i] = ch_ < C;
    const float bv = cval[i] ? p.bias[bidx] : 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] += bv;
  }
  float mean[NH][16], rstd[NH][16];
  const float invC = 1.0f / (float)C;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float s[NH][16];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 16; ++j) s[h][j] = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int h = (EPI == EPI_HC) ? (i & 1) : 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (pass == 0) s[h][j] += acc[i][j];
        else { const float d = cval[i] ? (acc[i][j] - mean[h][j]) : 0.f; s[h][j] += d * d; }
      }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float v = s[h][j];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
        s[h][j] = v;
      }
    if (l31 == 0) {
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int j = 0; j < 16; ++j) red[(wave * 2 + h) * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhi] = s[h][j];
    }
    __syncthreads();
    if (tid < 32 * NH) {
      const int h = tid >> 5, r = tid & 31;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[(w * 2 + h) * 32 + r];
      tot[h * 32 + r] = v * invC;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float v = tot[h * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhi];
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
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * lhi;
        const long orow = s_outrow[row];
        if (orow < 0) continue;
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
        for (int j = 0; j < 16; ++j) {
          const int row = (j & 3) + 8 * (j >> 2) + 4 * lhi;
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
        for (int j = 0; j < 16; ++j) {
          const int row = (j & 3) + 8 * (j >> 2) + 4 * lhi;
          const long orow = s_outrow[row];
          if (orow >= 0) p.out[orow * (long)p.out_stride + ch_] = 0.f;
        }
      }
    }
  }
}

__device__ __forceinline__ void lab_rows(const ConvParams& p, long* s_inrow, long* s_outrow, long* s_out2row) {
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * 32;
  if (tid < 32) {
    const int m = m0 + tid;
    long inrow = -1, outrow = -1, out2row = -1;
    if (m < p.M) {
      const int b = m / p.R, r = m - b * p.R;
      const int t = r;
      inrow = (long)b * p.in_bstride + p.in_row0 + t;
      outrow = (long)b * p.out_bstride + p.out_row0 + (long)t * p.out_tmul + p.out_tadd;
      out2row = (long)b * p.out2_bstride + p.out2_row0 + t;
    }
    s_inrow[tid] = inrow; s_outrow[tid] = outrow; s_out2row[tid] = out2row;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Ablated copies of the production K loop.  MASK bits: 1 no epilogue, 2 no per-chunk barrier / LDS store, 4 no weight loads in the loop,
// 8 no LDS fragment reads in the loop, 16 no activation loads in the loop.  Results are wrong by construction: timing only.
template <int EPI, int NT, int NW, int MASK>
__global__ void __launch_bounds__(NW * 64) abl_kernel(const ConvParams p) {
  constexpr int LDA = 36;
  __shared__ __attribute__((aligned(16))) float As[2][32 * LDA];
  __shared__ float red[NW * 2 * 32];
  __shared__ float tot[2 * 32];
  __shared__ long s_inrow[32], s_outrow[32], s_out2row[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  lab_rows(p, s_inrow, s_outrow, s_out2row);
  __syncthreads();
  const int lrow = (tid >> 3) & 31, lc4 = tid & 7;
  const long my_inrow = s_inrow[lrow];
  const int cpt = p.cin_p >> 5, nch = p.ntaps * cpt, KG = nch * 4;
  const bool row_ok = my_inrow >= 0;
  const long safe_row = p.in_row0;
  auto load_chunk = [&](int tap, int cit, bool& ok) -> float4 {
    const int c = cit * 32 + lc4 * 4;
    const int toff = (tap == 0) ? p.tap_off[0] : ((tap == 1) ? p.tap_off[1] : p.tap_off[2]);
    ok = row_ok && c < p.cin;
    const long row = row_ok ? my_inrow + toff : safe_row;
    return *reinterpret_cast<const float4*>(p.in + row * (long)p.in_stride + (c < p.cin ? c : 0));
  };
  const float4* wq[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) wq[i] = reinterpret_cast<const float4*>(p.wp) + ((long)(wave * NT + i) * KG) * 64 + lane;
  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bool aok;
  float4 areg = load_chunk(0, 0, aok);
  if (!aok) areg = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 256) *reinterpret_cast<float4*>(&As[0][lrow * LDA + lc4 * 4]) = areg;
  int ntap = 0, ncit = 0;
  float4 bcur[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) bcur[i] = wq[i][0];
  __syncthreads();
  float4 a = *reinterpret_cast<const float4*>(&As[0][l31 * LDA + lhi * 4]);
  for (int ch = 0; ch < nch; ++ch) {
    const bool more = (ch + 1 < nch);
    if (more) { if (++ncit == cpt) { ncit = 0; ++ntap; } }
    if (!(MASK & 16)) { areg = load_chunk(ntap, ncit, aok); __builtin_amdgcn_sched_barrier(0); }
    const float* Ab = As[ch & 1];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int kg = ch * 4 + gq;
      const int kgn = (kg + 1 < KG) ? kg + 1 : KG - 1;
      if (!(MASK & 8)) a = *reinterpret_cast<const float4*>(&Ab[l31 * LDA + gq * 8 + lhi * 4]);
      float4 bnext[NT];
      if (!(MASK & 4)) {
#pragma unroll
        for (int i = 0; i < NT; ++i) bnext[i] = wq[i][(long)kgn * 64];
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bcur[i].x, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bcur[i].y, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bcur[i].z, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bcur[i].w, acc[i], 0, 0, 0);
      if (!(MASK & 4)) {
#pragma unroll
        for (int i = 0; i < NT; ++i) bcur[i] = bnext[i];
      }
    }
    if (!(MASK & 2)) {
      if (!aok) areg = make_float4(0.f, 0.f, 0.f, 0.f);
      if (more && tid < 256) *reinterpret_cast<float4*>(&As[(ch + 1) & 1][lrow * LDA + lc4 * 4]) = areg;
      __syncthreads();
    }
  }
  if (MASK & 1) {
    float s = areg.x + a.x;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 123.456f) p.out[tid] = s;          // keeps the loop alive, practically never stores
    return;
  }
  lab_epilogue<EPI, NT, NW>(p, acc, s_inrow, s_outrow, s_out2row, red, tot);
}

// ------------------------------------------------------------------------------------------------------------------------
// N1: three LDS buffers and ONE barrier per chunk placed in the MIDDLE of the chunk, with nothing that depends on it right behind it:
//   chunk ch:  [k-groups 0, 1 from As[ch % 3]]  barrier  [store chunk ch + 2 into As[(ch + 2) % 3]; request chunk ch + 3]  [k-groups 2, 3]
// As[(ch + 2) % 3] was last read in chunk ch - 1, which every wave has left once it passed this chunk's barrier; the data stored now are read from
// chunk ch + 2 on, behind the barrier of chunk ch + 1.  The A fragment of the next k-group is requested one group ahead (also across the chunk
// boundary: As[(ch + 1) % 3] became visible at this chunk's barrier), so no MFMA waits for an LDS round trip right after a barrier.
template <int EPI, int NT, int NW>
__global__ void __launch_bounds__(NW * 64) n1_kernel(const ConvParams p) {
  constexpr int LDA = 36;
  __shared__ __attribute__((aligned(16))) float As[3][32 * LDA];
  __shared__ float red[NW * 2 * 32];
  __shared__ float tot[2 * 32];
  __shared__ long s_inrow[32], s_outrow[32], s_out2row[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  lab_rows(p, s_inrow, s_outrow, s_out2row);
  __syncthreads();
  const int lrow = (tid >> 3) & 31, lc4 = tid & 7;
  const long my_inrow = s_inrow[lrow];
  const int cpt = p.cin_p >> 5, nch = p.ntaps * cpt, KG = nch * 4;
  const bool row_ok = my_inrow >= 0;
  const long safe_row = p.in_row0;
  // chunk index -> (tap, chunk in tap) without a division: the loader walks forward one chunk at a time
  int ltap = 0, lcit = 0;
  auto load_next = [&](bool& ok) -> float4 {          // loads chunk (ltap, lcit), then advances (clamped at the last chunk)
    const int c = lcit * 32 + lc4 * 4;
    const int toff = (ltap == 0) ? p.tap_off[0] : ((ltap == 1) ? p.tap_off[1] : p.tap_off[2]);
    ok = row_ok && c < p.cin;
    const long row = row_ok ? my_inrow + toff : safe_row;
    const float4 v = *reinterpret_cast<const float4*>(p.in + row * (long)p.in_stride + (c < p.cin ? c : 0));
    if (!(ltap == p.ntaps - 1 && lcit == cpt - 1)) { if (++lcit == cpt) { lcit = 0; ++ltap; } }
    return v;
  };
  const float4* wq[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) wq[i] = reinterpret_cast<const float4*>(p.wp) + ((long)(wave * NT + i) * KG) * 64 + lane;
  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bool aok, aok1;
  float4 a0 = load_next(aok);
  float4 a1 = load_next(aok1);
  if (!aok) a0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!aok1) a1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 256) {
    *reinterpret_cast<float4*>(&As[0][lrow * LDA + lc4 * 4]) = a0;
    if (nch > 1) *reinterpret_cast<float4*>(&As[1][lrow * LDA + lc4 * 4]) = a1;
  }
  float4 areg = load_next(aok);                    // chunk 2 (or a re-read of the last chunk)
  float4 bcur[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) bcur[i] = wq[i][0];
  __syncthreads();
  const int aoff = l31 * LDA + lhi * 4;
  float4 a = *reinterpret_cast<const float4*>(&As[0][aoff]);
  int cb = 0;                                       // ch % 3
  for (int ch = 0; ch < nch; ++ch) {
    const float* Ab = As[cb];
    const int cb1 = (cb == 2) ? 0 : cb + 1, cb2 = (cb1 == 2) ? 0 : cb1 + 1;
    const float* An = As[cb1];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int kg = ch * 4 + gq;
      const int kgn = (kg + 1 < KG) ? kg + 1 : KG - 1;
      if (gq == 2) {
        __syncthreads();
        if (!aok) areg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ch + 2 < nch && tid < 256) *reinterpret_cast<float4*>(&As[cb2][lrow * LDA + lc4 * 4]) = areg;
        areg = load_next(aok);                      // chunk ch + 3
      }
      const float4 an = (gq < 3) ? *reinterpret_cast<const float4*>(&Ab[aoff + (gq + 1) * 8]) : *reinterpret_cast<const float4*>(&An[aoff]);
      float4 bnext[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) bnext[i] = wq[i][(long)kgn * 64];
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bcur[i].x, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bcur[i].y, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bcur[i].z, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bcur[i].w, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) bcur[i] = bnext[i];
      a = an;
    }
    cb = cb1;
  }
  __syncthreads();
  lab_epilogue<EPI, NT, NW>(p, acc, s_inrow, s_outrow, s_out2row, red, tot);
}

}  // namespace dctts
