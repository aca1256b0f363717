// hconv_lab_kernels.h -- experimental copies of hconv_kernel (tools/micro/hconv_lab.hip).  Not product code.
#pragma once
#include <hip/hip_runtime.h>

#include "hconv_kernel.h"

namespace dctts {

// In-kernel stamps (lab only): p.presum_out, when set, is a long long[items][8] table; wave 0 lane 0 of every workgroup records s_memtime there.
__device__ __forceinline__ void lab_stamp(const ConvParams& p, int slot) {
  if (p.presum_out && threadIdx.x == 0) reinterpret_cast<long long*>(p.presum_out)[blockIdx.x * 8 + slot] = (long long)__builtin_amdgcn_s_memtime();
}

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
  lab_stamp(p, 0);
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
  lab_stamp(p, 1);
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
  lab_stamp(p, 2);
  __syncthreads();
  lab_stamp(p, 3);
  lab_epilogue<EPI, NT, NW>(p, acc, s_inrow, s_outrow, s_out2row, red, tot);
  lab_stamp(p, 5);
}

}  // namespace dctts

namespace dctts {

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// (half_sum32 -- the sum over the 32 lanes that share lane >> 5 -- is the production kernel's: hconv_kernel.h)

// N2 = N1's K loop (CHK channels per LDS buffer, GPC = CHK / 8 k-groups per chunk, one barrier in the middle of each chunk, A fragments and
// weight fragments requested one k-group ahead; BD = 2: weight fragments two groups ahead, NO scheduling pin) + an epilogue without load-behind-branch
// chains: bias is the accumulators' initial value, the layer-norm parameters and the first residual rows are requested before the statistics
// passes, rows that do not exist are handled by predicated stores instead of `continue`, sigmoid on v_exp_f32 / v_rcp_f32.
template <int EPI, int NT, int NW, int CHK = 32, int BD = 1>
__global__ void __launch_bounds__(NW * 64) n2_kernel(const ConvParams p) {
  constexpr int LDA = CHK + 4;
  constexpr int GPC = CHK / 8;
  constexpr int LTH = 32 * CHK / 4;          // loader threads: one float4 each per chunk
  constexpr int LC = CHK / 4;
  constexpr int NH = (EPI == EPI_HC) ? 2 : 1;
  static_assert(NW * 64 >= LTH, "not enough threads for the A loader");
  __shared__ __attribute__((aligned(16))) float As[3][32 * LDA];
  __shared__ float red[NW * 2 * 32];
  __shared__ float tot[2][2 * 32];           // [pass][h * 32 + row]: mean, then 1 / sqrt(var + eps)
  __shared__ long s_inrow[32], s_outrow[32], s_out2row[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  lab_rows(p, s_inrow, s_outrow, s_out2row);
  __syncthreads();
  const int lrow = (tid / LC) & 31, lc4 = tid % LC;
  const long my_inrow = s_inrow[lrow];
  const int cpt = p.cin_p / CHK, nch = p.ntaps * cpt, KG = nch * GPC;
  const bool row_ok = my_inrow >= 0;
  const long safe_row = p.in_row0;
  int ltap = 0, lcit = 0;
  auto load_next = [&](bool& ok) -> float4 {
    const int c = lcit * CHK + lc4 * 4;
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

  // channel of tile i inside its LN group; bias = the accumulators' initial value
  const int C = p.cout;
  int chan[NT]; bool cval[NT];
  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    int ch_, bidx;
    if (EPI == EPI_HC) { const int pp = wave * (NT / 2) + (i >> 1); ch_ = pp * 32 + l31; bidx = (i & 1) * C + ch_; }
    else { ch_ = (wave * NT + i) * 32 + l31; bidx = ch_; }
    chan[i] = ch_; cval[i] = ch_ < C;
    const float bv = p.bias[cval[i] ? bidx : 0];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = cval[i] ? bv : 0.f;
  }

  bool aok, aok1;
  float4 a0 = load_next(aok);
  float4 a1 = load_next(aok1);
  float4 bq[BD][NT];
#pragma unroll
  for (int d = 0; d < BD; ++d)
#pragma unroll
    for (int i = 0; i < NT; ++i) bq[d][i] = wq[i][(long)(d < KG ? d : KG - 1) * 64];
  if (!aok) a0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!aok1) a1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < LTH) {
    *reinterpret_cast<float4*>(&As[0][lrow * LDA + lc4 * 4]) = a0;
    if (nch > 1) *reinterpret_cast<float4*>(&As[1][lrow * LDA + lc4 * 4]) = a1;
  }
  float4 areg = load_next(aok);
  __syncthreads();
  const int aoff = l31 * LDA + lhi * 4;
  float4 a = *reinterpret_cast<const float4*>(&As[0][aoff]);
  int cb = 0;
  for (int ch = 0; ch < nch; ++ch) {
    const float* Ab = As[cb];
    const int cb1 = (cb == 2) ? 0 : cb + 1, cb2 = (cb1 == 2) ? 0 : cb1 + 1;
    const float* An = As[cb1];
#pragma unroll
    for (int gq = 0; gq < GPC; ++gq) {
      const int kg = ch * GPC + gq;
      const int kgn = (kg + BD < KG) ? kg + BD : KG - 1;
      if (gq == GPC / 2) {
        __syncthreads();
        if (!aok) areg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ch + 2 < nch && tid < LTH) *reinterpret_cast<float4*>(&As[cb2][lrow * LDA + lc4 * 4]) = areg;
        areg = load_next(aok);
      }
      const float4 an = (gq < GPC - 1) ? *reinterpret_cast<const float4*>(&Ab[aoff + (gq + 1) * 8]) : *reinterpret_cast<const float4*>(&An[aoff]);
      float4 bnext[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) bnext[i] = wq[i][(long)kgn * 64];
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[0][i].x, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[0][i].y, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[0][i].z, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[0][i].w, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        if (BD == 2) { bq[0][i] = bq[BD - 1][i]; bq[BD - 1][i] = bnext[i]; }
        else bq[0][i] = bnext[i];
      }
      a = an;
    }
    cb = cb1;
  }

  // ===================================================================== epilogue
  // acc[i][j]: row (j & 3) + 8 (j >> 2) + 4 lhi, column of tile i / lane l31.  Requests first, then the statistics, then the stores.
  constexpr int NP = (EPI == EPI_HC) ? NT / 2 : NT;
  float pg1[NP], pb1[NP], pg2[NP], pb2[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int i = (EPI == EPI_HC) ? 2 * k : k;
    const int cs = cval[i] ? chan[i] : 0;
    pg1[k] = p.g1[cs]; pb1[k] = p.b1[cs];
    if (EPI == EPI_HC) { pg2[k] = p.g2[cs]; pb2[k] = p.b2[cs]; }
  }
  // residual rows of the highway mix, four rows (j) at a time; the first batch is requested here, behind nothing
  constexpr int JB = 4;
  float xr[2][(EPI == EPI_HC) ? NP : 1][JB];
  auto load_res = [&](int jb, int slot) {
    if (EPI != EPI_HC) return;
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
      const int j = jb * JB + jj;
      const int row = (j & 3) + 8 * (j >> 2) + 4 * lhi;
      const long ir = s_inrow[row];
      const float* rp = p.in + (ir >= 0 ? ir : safe_row) * (long)p.in_stride;
#pragma unroll
      for (int k = 0; k < NP; ++k) xr[slot][k][jj] = rp[cval[2 * k] ? chan[2 * k] : 0];
    }
  };
  load_res(0, 0);

  const float invC = 1.0f / (float)C;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float s[NH][16];
    float mean[NH][16];
    if (pass == 1) {
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int j = 0; j < 16; ++j) mean[h][j] = tot[0][h * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhi];
    }
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
      for (int j = 0; j < 16; ++j) s[h][j] = half_sum32(s[h][j]);
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
      v *= invC;
      tot[pass][h * 32 + r] = (pass == 0) ? v : 1.0f / sqrtf(v + 1e-12f);
    }
    __syncthreads();
  }

  // ---- normalise, gate / activate, store: four rows at a time, the next four rows' residuals in flight
#pragma unroll
  for (int jb = 0; jb < 16 / JB; ++jb) {
    if (jb + 1 < 16 / JB) load_res(jb + 1, (jb + 1) & 1);
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
      const int j = jb * JB + jj;
      const int row = (j & 3) + 8 * (j >> 2) + 4 * lhi;
      const long orow = s_outrow[row];
      const bool ok = orow >= 0;
      float* op = p.out + (ok ? orow : 0) * (long)p.out_stride;
      const float r0 = tot[1][row], r1 = tot[1][32 + row];
      const float m0_ = tot[0][row], m1_ = tot[0][32 + row];
      if (EPI == EPI_HC) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const float y1 = (acc[2 * k][j] - m0_) * r0 * pg1[k] + pb1[k];
          const float y2 = (acc[2 * k + 1][j] - m1_) * r1 * pg2[k] + pb2[k];
          const float gt = fast_sigmoid(y1);
          const float o = gt * y2 + (1.0f - gt) * xr[jb & 1][k][jj];
          if (ok && cval[2 * k]) op[chan[2 * k]] = o;
        }
      } else {
        float* op2 = p.out2 ? p.out2 + (ok ? s_out2row[row] : 0) * (long)p.out2_stride : nullptr;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          float y = (acc[k][j] - m0_) * r0 * pg1[k] + pb1[k];
          if (op2 && ok && cval[k]) op2[chan[k]] = y;
          if (p.act == ACT_RELU) y = fmaxf(y, 0.f);
          else if (p.act == ACT_SIGMOID) y = fast_sigmoid(y);
          if (ok && (cval[k] || chan[k] < p.out_zero_to)) op[chan[k]] = cval[k] ? y : 0.f;
        }
      }
    }
  }
}

}  // namespace dctts

namespace dctts {

// N4 = N2 with wave-uniform tile bases (readfirstlane -> scalar base + one vector offset per load) so that BD = 2 / 4 register sets of weight
// fragments fit; N2 = N1's K loop (CHK channels per LDS buffer, GPC = CHK / 8 k-groups per chunk, one barrier in the middle of each chunk, A fragments and
// weight fragments requested one k-group ahead; BD = 2: weight fragments two groups ahead, NO scheduling pin) + an epilogue without load-behind-branch
// chains: bias is the accumulators' initial value, the layer-norm parameters and the first residual rows are requested before the statistics
// passes, rows that do not exist are handled by predicated stores instead of `continue`, sigmoid on v_exp_f32 / v_rcp_f32.
template <int EPI, int NT, int NW, int CHK = 32, int BD = 1>
__global__ void __launch_bounds__(NW * 64) n4_kernel(const ConvParams p) {
  constexpr int LDA = CHK + 4;
  constexpr int GPC = CHK / 8;
  constexpr int LTH = 32 * CHK / 4;          // loader threads: one float4 each per chunk
  constexpr int LC = CHK / 4;
  constexpr int NH = (EPI == EPI_HC) ? 2 : 1;
  static_assert(NW * 64 >= LTH, "not enough threads for the A loader");
  __shared__ __attribute__((aligned(16))) float As[3][32 * LDA];
  __shared__ float red[NW * 2 * 32];
  __shared__ float tot[2][2 * 32];           // [pass][h * 32 + row]: mean, then 1 / sqrt(var + eps)
  __shared__ long s_inrow[32], s_outrow[32], s_out2row[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  lab_rows(p, s_inrow, s_outrow, s_out2row);
  __syncthreads();
  const int lrow = (tid / LC) & 31, lc4 = tid % LC;
  const long my_inrow = s_inrow[lrow];
  const int cpt = p.cin_p / CHK, nch = p.ntaps * cpt, KG = nch * GPC;
  const bool row_ok = my_inrow >= 0;
  const long safe_row = p.in_row0;
  int ltap = 0, lcit = 0;
  auto load_next = [&](bool& ok) -> float4 {
    const int c = lcit * CHK + lc4 * 4;
    const int toff = (ltap == 0) ? p.tap_off[0] : ((ltap == 1) ? p.tap_off[1] : p.tap_off[2]);
    ok = row_ok && c < p.cin;
    const long row = row_ok ? my_inrow + toff : safe_row;
    const float4 v = *reinterpret_cast<const float4*>(p.in + row * (long)p.in_stride + (c < p.cin ? c : 0));
    if (!(ltap == p.ntaps - 1 && lcit == cpt - 1)) { if (++lcit == cpt) { lcit = 0; ++ltap; } }
    return v;
  };
  const float4* wq[NT];                       // wave-uniform bases: scalar registers
#pragma unroll
  for (int i = 0; i < NT; ++i) wq[i] = reinterpret_cast<const float4*>(p.wp) + ((long)(wave * NT + i) * KG) * 64;

  // channel of tile i inside its LN group; bias = the accumulators' initial value
  const int C = p.cout;
  int chan[NT]; bool cval[NT];
  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    int ch_, bidx;
    if (EPI == EPI_HC) { const int pp = wave * (NT / 2) + (i >> 1); ch_ = pp * 32 + l31; bidx = (i & 1) * C + ch_; }
    else { ch_ = (wave * NT + i) * 32 + l31; bidx = ch_; }
    chan[i] = ch_; cval[i] = ch_ < C;
    const float bv = p.bias[cval[i] ? bidx : 0];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = cval[i] ? bv : 0.f;
  }

  bool aok, aok1;
  float4 a0 = load_next(aok);
  float4 a1 = load_next(aok1);
  float4 bq[BD][NT];
#pragma unroll
  for (int d = 0; d < BD; ++d)
#pragma unroll
    for (int i = 0; i < NT; ++i) bq[d][i] = wq[i][(d < KG ? d : KG - 1) * 64 + lane];
  if (!aok) a0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!aok1) a1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < LTH) {
    *reinterpret_cast<float4*>(&As[0][lrow * LDA + lc4 * 4]) = a0;
    if (nch > 1) *reinterpret_cast<float4*>(&As[1][lrow * LDA + lc4 * 4]) = a1;
  }
  float4 areg = load_next(aok);
  __syncthreads();
  const int aoff = l31 * LDA + lhi * 4;
  float4 a = *reinterpret_cast<const float4*>(&As[0][aoff]);
  int cb = 0;
  for (int ch = 0; ch < nch; ++ch) {
    const float* Ab = As[cb];
    const int cb1 = (cb == 2) ? 0 : cb + 1, cb2 = (cb1 == 2) ? 0 : cb1 + 1;
    const float* An = As[cb1];
#pragma unroll
    for (int gq = 0; gq < GPC; ++gq) {
      const int kg = ch * GPC + gq;
      const int kgn = (kg + BD < KG) ? kg + BD : KG - 1;
      if (gq == GPC / 2) {
        __syncthreads();
        if (!aok) areg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ch + 2 < nch && tid < LTH) *reinterpret_cast<float4*>(&As[cb2][lrow * LDA + lc4 * 4]) = areg;
        areg = load_next(aok);
      }
      const float4 an = (gq < GPC - 1) ? *reinterpret_cast<const float4*>(&Ab[aoff + (gq + 1) * 8]) : *reinterpret_cast<const float4*>(&An[aoff]);
      float4 bnext[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) bnext[i] = wq[i][kgn * 64 + lane];
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[0][i].x, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[0][i].y, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[0][i].z, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[0][i].w, acc[i], 0, 0, 0);
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

  // ===================================================================== epilogue
  // acc[i][j]: row (j & 3) + 8 (j >> 2) + 4 lhi, column of tile i / lane l31.  Requests first, then the statistics, then the stores.
  constexpr int NP = (EPI == EPI_HC) ? NT / 2 : NT;
  float pg1[NP], pb1[NP], pg2[NP], pb2[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int i = (EPI == EPI_HC) ? 2 * k : k;
    const int cs = cval[i] ? chan[i] : 0;
    pg1[k] = p.g1[cs]; pb1[k] = p.b1[cs];
    if (EPI == EPI_HC) { pg2[k] = p.g2[cs]; pb2[k] = p.b2[cs]; }
  }
  // residual rows of the highway mix, four rows (j) at a time; the first batch is requested here, behind nothing
  constexpr int JB = 4;
  float xr[2][(EPI == EPI_HC) ? NP : 1][JB];
  auto load_res = [&](int jb, int slot) {
    if (EPI != EPI_HC) return;
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
      const int j = jb * JB + jj;
      const int row = (j & 3) + 8 * (j >> 2) + 4 * lhi;
      const long ir = s_inrow[row];
      const float* rp = p.in + (ir >= 0 ? ir : safe_row) * (long)p.in_stride;
#pragma unroll
      for (int k = 0; k < NP; ++k) xr[slot][k][jj] = rp[cval[2 * k] ? chan[2 * k] : 0];
    }
  };
  load_res(0, 0);

  const float invC = 1.0f / (float)C;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float s[NH][16];
    float mean[NH][16];
    if (pass == 1) {
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int j = 0; j < 16; ++j) mean[h][j] = tot[0][h * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhi];
    }
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
      for (int j = 0; j < 16; ++j) s[h][j] = half_sum32(s[h][j]);
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
      v *= invC;
      tot[pass][h * 32 + r] = (pass == 0) ? v : 1.0f / sqrtf(v + 1e-12f);
    }
    __syncthreads();
  }

  // ---- normalise, gate / activate, store: four rows at a time, the next four rows' residuals in flight
#pragma unroll
  for (int jb = 0; jb < 16 / JB; ++jb) {
    if (jb + 1 < 16 / JB) load_res(jb + 1, (jb + 1) & 1);
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
      const int j = jb * JB + jj;
      const int row = (j & 3) + 8 * (j >> 2) + 4 * lhi;
      const long orow = s_outrow[row];
      const bool ok = orow >= 0;
      float* op = p.out + (ok ? orow : 0) * (long)p.out_stride;
      const float r0 = tot[1][row], r1 = tot[1][32 + row];
      const float m0_ = tot[0][row], m1_ = tot[0][32 + row];
      if (EPI == EPI_HC) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const float y1 = (acc[2 * k][j] - m0_) * r0 * pg1[k] + pb1[k];
          const float y2 = (acc[2 * k + 1][j] - m1_) * r1 * pg2[k] + pb2[k];
          const float gt = fast_sigmoid(y1);
          const float o = gt * y2 + (1.0f - gt) * xr[jb & 1][k][jj];
          if (ok && cval[2 * k]) op[chan[2 * k]] = o;
        }
      } else {
        float* op2 = p.out2 ? p.out2 + (ok ? s_out2row[row] : 0) * (long)p.out2_stride : nullptr;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          float y = (acc[k][j] - m0_) * r0 * pg1[k] + pb1[k];
          if (op2 && ok && cval[k]) op2[chan[k]] = y;
          if (p.act == ACT_RELU) y = fmaxf(y, 0.f);
          else if (p.act == ACT_SIGMOID) y = fast_sigmoid(y);
          if (ok && (cval[k] || chan[k] < p.out_zero_to)) op[chan[k]] = cval[k] ? y : 0.f;
        }
      }
    }
  }
}


// N5 = N4 with a register budget of 512 / OCC per wave (__launch_bounds__ second argument = waves per SIMD), so that a workgroup of 4 waves
// leaves room for a second workgroup on the CU: one workgroup's prologue / epilogue / stalls overlap the other's MFMAs.
// N4 = N2 with wave-uniform tile bases (readfirstlane -> scalar base + one vector offset per load) so that BD = 2 / 4 register sets of weight
// fragments fit; N2 = N1's K loop (CHK channels per LDS buffer, GPC = CHK / 8 k-groups per chunk, one barrier in the middle of each chunk, A fragments and
// weight fragments requested one k-group ahead; BD = 2: weight fragments two groups ahead, NO scheduling pin) + an epilogue without load-behind-branch
// chains: bias is the accumulators' initial value, the layer-norm parameters and the first residual rows are requested before the statistics
// passes, rows that do not exist are handled by predicated stores instead of `continue`, sigmoid on v_exp_f32 / v_rcp_f32.
template <int EPI, int NT, int NW, int CHK = 32, int BD = 1, int OCC = 2>
__global__ void __launch_bounds__(NW * 64, OCC) n5_kernel(const ConvParams p) {
  constexpr int LDA = CHK + 4;
  constexpr int GPC = CHK / 8;
  constexpr int LTH = 32 * CHK / 4;          // loader threads: one float4 each per chunk
  constexpr int LC = CHK / 4;
  constexpr int NH = (EPI == EPI_HC) ? 2 : 1;
  static_assert(NW * 64 >= LTH, "not enough threads for the A loader");
  __shared__ __attribute__((aligned(16))) float As[3][32 * LDA];
  __shared__ float red[NW * 2 * 32];
  __shared__ float tot[2][2 * 32];           // [pass][h * 32 + row]: mean, then 1 / sqrt(var + eps)
  __shared__ long s_inrow[32], s_outrow[32], s_out2row[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  lab_rows(p, s_inrow, s_outrow, s_out2row);
  __syncthreads();
  const int lrow = (tid / LC) & 31, lc4 = tid % LC;
  const long my_inrow = s_inrow[lrow];
  const int cpt = p.cin_p / CHK, nch = p.ntaps * cpt, KG = nch * GPC;
  const bool row_ok = my_inrow >= 0;
  const long safe_row = p.in_row0;
  int ltap = 0, lcit = 0;
  auto load_next = [&](bool& ok) -> float4 {
    const int c = lcit * CHK + lc4 * 4;
    const int toff = (ltap == 0) ? p.tap_off[0] : ((ltap == 1) ? p.tap_off[1] : p.tap_off[2]);
    ok = row_ok && c < p.cin;
    const long row = row_ok ? my_inrow + toff : safe_row;
    const float4 v = *reinterpret_cast<const float4*>(p.in + row * (long)p.in_stride + (c < p.cin ? c : 0));
    if (!(ltap == p.ntaps - 1 && lcit == cpt - 1)) { if (++lcit == cpt) { lcit = 0; ++ltap; } }
    return v;
  };
  const float4* wq[NT];                       // wave-uniform bases: scalar registers
#pragma unroll
  for (int i = 0; i < NT; ++i) wq[i] = reinterpret_cast<const float4*>(p.wp) + ((long)(wave * NT + i) * KG) * 64;

  // channel of tile i inside its LN group; bias = the accumulators' initial value
  const int C = p.cout;
  int chan[NT]; bool cval[NT];
  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    int ch_, bidx;
    if (EPI == EPI_HC) { const int pp = wave * (NT / 2) + (i >> 1); ch_ = pp * 32 + l31; bidx = (i & 1) * C + ch_; }
    else { ch_ = (wave * NT + i) * 32 + l31; bidx = ch_; }
    chan[i] = ch_; cval[i] = ch_ < C;
    const float bv = p.bias[cval[i] ? bidx : 0];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = cval[i] ? bv : 0.f;
  }

  bool aok, aok1;
  float4 a0 = load_next(aok);
  float4 a1 = load_next(aok1);
  float4 bq[BD][NT];
#pragma unroll
  for (int d = 0; d < BD; ++d)
#pragma unroll
    for (int i = 0; i < NT; ++i) bq[d][i] = wq[i][(d < KG ? d : KG - 1) * 64 + lane];
  if (!aok) a0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!aok1) a1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < LTH) {
    *reinterpret_cast<float4*>(&As[0][lrow * LDA + lc4 * 4]) = a0;
    if (nch > 1) *reinterpret_cast<float4*>(&As[1][lrow * LDA + lc4 * 4]) = a1;
  }
  float4 areg = load_next(aok);
  __syncthreads();
  const int aoff = l31 * LDA + lhi * 4;
  float4 a = *reinterpret_cast<const float4*>(&As[0][aoff]);
  int cb = 0;
  for (int ch = 0; ch < nch; ++ch) {
    const float* Ab = As[cb];
    const int cb1 = (cb == 2) ? 0 : cb + 1, cb2 = (cb1 == 2) ? 0 : cb1 + 1;
    const float* An = As[cb1];
#pragma unroll
    for (int gq = 0; gq < GPC; ++gq) {
      const int kg = ch * GPC + gq;
      const int kgn = (kg + BD < KG) ? kg + BD : KG - 1;
      if (gq == GPC / 2) {
        __syncthreads();
        if (!aok) areg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ch + 2 < nch && tid < LTH) *reinterpret_cast<float4*>(&As[cb2][lrow * LDA + lc4 * 4]) = areg;
        areg = load_next(aok);
      }
      const float4 an = (gq < GPC - 1) ? *reinterpret_cast<const float4*>(&Ab[aoff + (gq + 1) * 8]) : *reinterpret_cast<const float4*>(&An[aoff]);
      float4 bnext[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) bnext[i] = wq[i][kgn * 64 + lane];
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[0][i].x, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[0][i].y, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[0][i].z, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[0][i].w, acc[i], 0, 0, 0);
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

  // ===================================================================== epilogue
  // acc[i][j]: row (j & 3) + 8 (j >> 2) + 4 lhi, column of tile i / lane l31.  Requests first, then the statistics, then the stores.
  constexpr int NP = (EPI == EPI_HC) ? NT / 2 : NT;
  float pg1[NP], pb1[NP], pg2[NP], pb2[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int i = (EPI == EPI_HC) ? 2 * k : k;
    const int cs = cval[i] ? chan[i] : 0;
    pg1[k] = p.g1[cs]; pb1[k] = p.b1[cs];
    if (EPI == EPI_HC) { pg2[k] = p.g2[cs]; pb2[k] = p.b2[cs]; }
  }
  // residual rows of the highway mix, four rows (j) at a time; the first batch is requested here, behind nothing
  constexpr int JB = 4;
  float xr[2][(EPI == EPI_HC) ? NP : 1][JB];
  auto load_res = [&](int jb, int slot) {
    if (EPI != EPI_HC) return;
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
      const int j = jb * JB + jj;
      const int row = (j & 3) + 8 * (j >> 2) + 4 * lhi;
      const long ir = s_inrow[row];
      const float* rp = p.in + (ir >= 0 ? ir : safe_row) * (long)p.in_stride;
#pragma unroll
      for (int k = 0; k < NP; ++k) xr[slot][k][jj] = rp[cval[2 * k] ? chan[2 * k] : 0];
    }
  };
  load_res(0, 0);

  const float invC = 1.0f / (float)C;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float s[NH][16];
    float mean[NH][16];
    if (pass == 1) {
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int j = 0; j < 16; ++j) mean[h][j] = tot[0][h * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhi];
    }
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
      for (int j = 0; j < 16; ++j) s[h][j] = half_sum32(s[h][j]);
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
      v *= invC;
      tot[pass][h * 32 + r] = (pass == 0) ? v : 1.0f / sqrtf(v + 1e-12f);
    }
    __syncthreads();
  }

  // ---- normalise, gate / activate, store: four rows at a time, the next four rows' residuals in flight
#pragma unroll
  for (int jb = 0; jb < 16 / JB; ++jb) {
    if (jb + 1 < 16 / JB) load_res(jb + 1, (jb + 1) & 1);
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
      const int j = jb * JB + jj;
      const int row = (j & 3) + 8 * (j >> 2) + 4 * lhi;
      const long orow = s_outrow[row];
      const bool ok = orow >= 0;
      float* op = p.out + (ok ? orow : 0) * (long)p.out_stride;
      const float r0 = tot[1][row], r1 = tot[1][32 + row];
      const float m0_ = tot[0][row], m1_ = tot[0][32 + row];
      if (EPI == EPI_HC) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const float y1 = (acc[2 * k][j] - m0_) * r0 * pg1[k] + pb1[k];
          const float y2 = (acc[2 * k + 1][j] - m1_) * r1 * pg2[k] + pb2[k];
          const float gt = fast_sigmoid(y1);
          const float o = gt * y2 + (1.0f - gt) * xr[jb & 1][k][jj];
          if (ok && cval[2 * k]) op[chan[2 * k]] = o;
        }
      } else {
        float* op2 = p.out2 ? p.out2 + (ok ? s_out2row[row] : 0) * (long)p.out2_stride : nullptr;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          float y = (acc[k][j] - m0_) * r0 * pg1[k] + pb1[k];
          if (op2 && ok && cval[k]) op2[chan[k]] = y;
          if (p.act == ACT_RELU) y = fmaxf(y, 0.f);
          else if (p.act == ACT_SIGMOID) y = fast_sigmoid(y);
          if (ok && (cval[k] || chan[k] < p.out_zero_to)) op[chan[k]] = cval[k] ? y : 0.f;
        }
      }
    }
  }
}



typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));     // a float4 that is only dword-aligned (user tensors with 1025-float rows)

// N3 = N1's K loop + a TRANSPOSING epilogue.  The MFMA accumulator layout gives a lane one column and 16 rows, so the production epilogue moves
// every output (and every highway residual) as a dword per lane: 128 memory instructions per wave at NT = 8, store-issue bound.  Here the
// normalised statistics are computed in the accumulator layout as before, then each (gate, info) tile pair goes through a wave-private LDS tile
// (16 rows x 32 columns at a time, written conflict-free as dwords, read back conflict-free as one float4 of four consecutive channels per lane),
// so residual loads, parameter loads and stores are dwordx4: a quarter of the memory instructions, each covering whole 128-byte lines.
template <int EPI, int NT, int NW>
__global__ void __launch_bounds__(NW * 64) n3_kernel(const ConvParams p) {
  constexpr int LDA = 36;
  constexpr int NH = (EPI == EPI_HC) ? 2 : 1;
  constexpr int NP = (EPI == EPI_HC) ? NT / 2 : NT;
  __shared__ __attribute__((aligned(16))) float As[3][32 * LDA];
  __shared__ __attribute__((aligned(16))) float Ts[NW][NH][16 * 32];
  __shared__ float red[NW * 2 * 32];
  __shared__ float tot[2][2 * 32];
  __shared__ long s_inrow[32], s_outrow[32], s_out2row[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  lab_stamp(p, 0);
  lab_rows(p, s_inrow, s_outrow, s_out2row);
  __syncthreads();
  const int lrow = (tid >> 3) & 31, lc4 = tid & 7;
  const long my_inrow = s_inrow[lrow];
  const int cpt = p.cin_p >> 5, nch = p.ntaps * cpt, KG = nch * 4;
  const bool row_ok = my_inrow >= 0;
  const long safe_row = p.in_row0;
  int ltap = 0, lcit = 0;
  auto load_next = [&](bool& ok) -> float4 {
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
  float4 bcur[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) bcur[i] = wq[i][0];
  if (!aok) a0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!aok1) a1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 256) {
    *reinterpret_cast<float4*>(&As[0][lrow * LDA + lc4 * 4]) = a0;
    if (nch > 1) *reinterpret_cast<float4*>(&As[1][lrow * LDA + lc4 * 4]) = a1;
  }
  float4 areg = load_next(aok);
  __syncthreads();
  const int aoff = l31 * LDA + lhi * 4;
  float4 a = *reinterpret_cast<const float4*>(&As[0][aoff]);
  int cb = 0;
  lab_stamp(p, 1);
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
        areg = load_next(aok);
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

  // ===================================================================== epilogue
  lab_stamp(p, 2);
  const int C = p.cout;
  // ---- bias (accumulator layout: lane = column)
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    int bidx, ch_;
    if (EPI == EPI_HC) { ch_ = (wave * NP + (i >> 1)) * 32 + l31; bidx = (i & 1) * C + ch_; }
    else { ch_ = (wave * NT + i) * 32 + l31; bidx = ch_; }
    const float bv = (ch_ < C) ? p.bias[bidx] : 0.f;       // columns beyond C hold exactly 0 (zero weights, zero bias)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] += bv;
  }
  // ---- two-pass statistics
  const float invC = 1.0f / (float)C;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float s[NH][16], mean[NH][16];
    if (pass == 1) {
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int j = 0; j < 16; ++j) mean[h][j] = tot[0][h * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhi];
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 16; ++j) s[h][j] = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int h = (EPI == EPI_HC) ? (i & 1) : 0;
      const int ch_ = (EPI == EPI_HC) ? (wave * NP + (i >> 1)) * 32 + l31 : (wave * NT + i) * 32 + l31;
      const bool cv = ch_ < C;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (pass == 0) s[h][j] += acc[i][j];
        else { const float d = cv ? (acc[i][j] - mean[h][j]) : 0.f; s[h][j] += d * d; }
      }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 16; ++j) s[h][j] = half_sum32(s[h][j]);
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
      v *= invC;
      tot[pass][h * 32 + r] = (pass == 0) ? v : 1.0f / sqrtf(v + 1e-12f);
    }
    __syncthreads();
  }

  lab_stamp(p, 4);
  // ---- transposed phase: lane = (row tr + 8 q, four consecutive columns tc .. tc + 3) of a 16-row half tile
  const int tr = lane >> 3, tc = (lane & 7) * 4;
  float* T0 = &Ts[wave][0][0];
  float* T1 = &Ts[wave][NH - 1][0];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    // rows of this half: 16 half + tr + 8 q, q = 0, 1
    long ib[2], ob[2], o2b[2]; bool ok[2]; float m0[2], r0[2], m1[2], r1[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = 16 * half + tr + 8 * q;
      const long ir = s_inrow[row], orow = s_outrow[row];
      ok[q] = orow >= 0;
      ib[q] = (ir >= 0 ? ir : safe_row) * (long)p.in_stride;
      ob[q] = (ok[q] ? orow : 0) * (long)p.out_stride;
      o2b[q] = (ok[q] ? s_out2row[row] : 0) * (long)p.out2_stride;
      m0[q] = tot[0][row]; r0[q] = tot[1][row];
      if (EPI == EPI_HC) { m1[q] = tot[0][32 + row]; r1[q] = tot[1][32 + row]; }
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int cbase = (wave * NP + k) * 32;            // first channel of this tile (pair)
      // the half tile's 8 accumulator registers per lane: j = 8 half .. 8 half + 7  ->  local row (j & 3) + 8 ((j >> 2) & 1) + 4 lhi
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int j = 8 * half + jj;
        const int lr = (jj & 3) + 8 * (jj >> 2) + 4 * lhi;
        if (EPI == EPI_HC) { T0[lr * 32 + l31] = acc[2 * k][j]; T1[lr * 32 + l31] = acc[2 * k + 1][j]; }
        else T0[lr * 32 + l31] = acc[k][j];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int col = cbase + tc;
      const bool fullc = col + 3 < C;
      const int cs = fullc ? col : 0;
      const float4 g1v = *reinterpret_cast<const float4*>(p.g1 + cs), b1v = *reinterpret_cast<const float4*>(p.b1 + cs);
      float4 g2v, b2v;
      if (EPI == EPI_HC) { g2v = *reinterpret_cast<const float4*>(p.g2 + cs); b2v = *reinterpret_cast<const float4*>(p.b2 + cs); }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int lr = tr + 8 * q;
        const float4 h1 = *reinterpret_cast<const float4*>(&T0[lr * 32 + tc]);
        float4 o;
        if (EPI == EPI_HC) {
          const float4 h2 = *reinterpret_cast<const float4*>(&T1[lr * 32 + tc]);
          const f4u xr = *reinterpret_cast<const f4u*>(p.in + ib[q] + cs);
          const float y1x = (h1.x - m0[q]) * r0[q] * g1v.x + b1v.x, y1y = (h1.y - m0[q]) * r0[q] * g1v.y + b1v.y;
          const float y1z = (h1.z - m0[q]) * r0[q] * g1v.z + b1v.z, y1w = (h1.w - m0[q]) * r0[q] * g1v.w + b1v.w;
          const float y2x = (h2.x - m1[q]) * r1[q] * g2v.x + b2v.x, y2y = (h2.y - m1[q]) * r1[q] * g2v.y + b2v.y;
          const float y2z = (h2.z - m1[q]) * r1[q] * g2v.z + b2v.z, y2w = (h2.w - m1[q]) * r1[q] * g2v.w + b2v.w;
          const float gx = fast_sigmoid(y1x), gy = fast_sigmoid(y1y), gz = fast_sigmoid(y1z), gw = fast_sigmoid(y1w);
          o.x = gx * y2x + (1.0f - gx) * xr.x; o.y = gy * y2y + (1.0f - gy) * xr.y;
          o.z = gz * y2z + (1.0f - gz) * xr.z; o.w = gw * y2w + (1.0f - gw) * xr.w;
          if (ok[q] && fullc) *reinterpret_cast<f4u*>(p.out + ob[q] + col) = f4u{o.x, o.y, o.z, o.w};
        } else {
          float y[4] = {(h1.x - m0[q]) * r0[q] * g1v.x + b1v.x, (h1.y - m0[q]) * r0[q] * g1v.y + b1v.y,
                        (h1.z - m0[q]) * r0[q] * g1v.z + b1v.z, (h1.w - m0[q]) * r0[q] * g1v.w + b1v.w};
          if (fullc) {
            if (p.out2 && ok[q]) *reinterpret_cast<f4u*>(p.out2 + o2b[q] + col) = f4u{y[0], y[1], y[2], y[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (p.act == ACT_RELU) y[e] = fmaxf(y[e], 0.f);
              else if (p.act == ACT_SIGMOID) y[e] = fast_sigmoid(y[e]);
            }
            if (ok[q]) *reinterpret_cast<f4u*>(p.out + ob[q] + col) = f4u{y[0], y[1], y[2], y[3]};
          } else if (ok[q]) {                              // the ragged last tile of a 1025-column layer: element by element
            const float hv[4] = {h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int ce = col + e;
              if (ce < C) {
                float ye = (hv[e] - m0[q]) * r0[q] * p.g1[ce] + p.b1[ce];
                if (p.out2) p.out2[o2b[q] + ce] = ye;
                if (p.act == ACT_RELU) ye = fmaxf(ye, 0.f);
                else if (p.act == ACT_SIGMOID) ye = fast_sigmoid(ye);
                p.out[ob[q] + ce] = ye;
              } else if (ce < p.out_zero_to) {
                p.out[ob[q] + ce] = 0.f;
              }
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  lab_stamp(p, 5);
}

}  // namespace dctts
