// train_kernels.h -- kernels of the first TRAINING slice for gfx950: backward of the highway-convolution
// block (modules.py:143-197), the losses of train.py:85-110, the clip + Adam update of train.py:119-131.
//
// Backward of hc, given x and dy (all fp32, channel-last):
//   1. H = conv(x) is recomputed as k plain GEMMs over the zero-padded input rows (the inference kernels never store the
//      pre-norm tensor);
//   2. hc_bwd_rows_kernel: per row, both layer-norms forward and backward, the gate and the highway mix backward, in registers
//      (one wave per row, 4 x NCH channels per lane, DPP reductions): dH (2C), the direct part of dx, and the per-column sums
//      that become d(gamma), d(beta), d(bias);
//   3. dkernel[tap] = x_shifted^T . dH (a GEMM whose K is every row of the batch: split-K partials + a fixed-order sum) and
//      dx += dH_shifted . kernel[tap]^T.
// Zero pad rows before / after every utterance make every tap shift a pointer offset, so the GEMMs are plain (row-major,
// leading dimensions, no im2col): one LDS-tiled fp32 MFMA kernel (128 x 128 x 16 tiles, v_mfma_f32_32x32x2_f32) covers the three
// operand layouts (NN, NT, TN).  Every reduction is two-stage with a fixed order: results are bitwise reproducible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dctts {

// DPP reductions (as in attn_kernels.h, which defines kernels and therefore cannot be included by a second translation unit)
template <int CTRL>
__device__ __forceinline__ float t_dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float t_wave_sum(float v) {
  v = t_dpp_add<0xB1>(v); v = t_dpp_add<0x4E>(v); v = t_dpp_add<0x141>(v); v = t_dpp_add<0x140>(v);     // sum of each 16-lane row in all of its lanes
  return (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48)));
}

typedef float tf32x16 __attribute__((ext_vector_type(16)));
typedef float tf32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------- GEMM
struct GemmParams {
  const float* A; const float* B; float* C;
  int M, N, K;               // C is M x N; the contraction runs over K
  int lda, ldb, ldc;         // floats between consecutive rows of the STORED matrices
  int beta;                  // 0: C = A'B';  1: C += A'B'
  int kchunk;                // split-K: workgroup z contracts k in [z * kchunk, min(K, (z + 1) * kchunk)) into C + z * c_zstride
  long c_zstride;
  int batched;               // 1: z is a batch index instead: A + z * a_zs, B + z * b_zs, C + z * c_zstride, the whole K each
  long a_zs, b_zs;
  int nseg = 1;              // the contraction is the SUM over nseg segments s of (A + s * a_ss)' (B + s * b_ss)', each over the same k range:
  long a_ss = 0, b_ss = 0;   //   the taps of a convolution in one launch (a tap shift is a row offset), no read-modify-write of C between them
  int nsplit = 1;            // batched AND split-K: z = batch * nsplit + split (C + z * c_zstride holds that split's partial)
};

// A' = TA ? A^T : A (A stored M x K, or K x M when TA);  B' = TB ? B^T : B (B stored K x N, or N x K when TB).
// The contiguous extent of every stored matrix and all leading dimensions are multiples of 4 floats (16-byte loads).
// 128 x 128 x 16 tiles, 4 waves of 64 x 64 (2 x 2 MFMA tiles: one LDS read per MFMA), register-prefetched double buffer.
template <bool TA, bool TB>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmParams p) {
  constexpr int BM = 128, BN = 128, BK = 16, LD = 132;
  __shared__ __attribute__((aligned(16))) float As[2][BK][LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int zb = p.batched ? (int)blockIdx.z / p.nsplit : 0, zk = p.batched ? (int)blockIdx.z % p.nsplit : (int)blockIdx.z;
  const int kbeg = zk * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
  const float* pA = p.A + (long)zb * p.a_zs;
  const float* pB = p.B + (long)zb * p.b_zs;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64, l31 = lane & 31, lhi = lane >> 5;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // a tile is 128 x 16 floats = 512 float4: two per thread (i = 0, 1)
  auto load_a = [&](int k0, int i) -> float4 {
    const int idx = tid + i * 256;
    if (TA) { const int k = k0 + (idx >> 5), m = m0 + (idx & 31) * 4; return (k < kend && m < p.M) ? *reinterpret_cast<const float4*>(pA + (long)k * p.lda + m) : z4; }
    const int m = m0 + (idx >> 2), k = k0 + (idx & 3) * 4;
    return (m < p.M && k < kend) ? *reinterpret_cast<const float4*>(pA + (long)m * p.lda + k) : z4;
  };
  auto load_b = [&](int k0, int i) -> float4 {
    const int idx = tid + i * 256;
    if (!TB) { const int k = k0 + (idx >> 5), n = n0 + (idx & 31) * 4; return (k < kend && n < p.N) ? *reinterpret_cast<const float4*>(pB + (long)k * p.ldb + n) : z4; }
    const int n = n0 + (idx >> 2), k = k0 + (idx & 3) * 4;
    return (n < p.N && k < kend) ? *reinterpret_cast<const float4*>(pB + (long)n * p.ldb + k) : z4;
  };
  auto store_t = [&](float (*S)[LD], bool direct, int i, const float4 v) {     // direct: the stored matrix is k-major already
    const int idx = tid + i * 256;
    if (direct) { *reinterpret_cast<float4*>(&S[idx >> 5][(idx & 31) * 4]) = v; return; }
    const int r = idx >> 2, k = (idx & 3) * 4;
    S[k][r] = v.x; S[k + 1][r] = v.y; S[k + 2][r] = v.z; S[k + 3][r] = v.w;
  };
  tf32x16 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[q][j] = 0.f;
  float4 ra[2], rb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { ra[i] = load_a(kbeg, i); rb[i] = load_b(kbeg, i); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { store_t(As[0], TA, i, ra[i]); store_t(Bs[0], !TB, i, rb[i]); }
  __syncthreads();
  int buf = 0, seg = 0;
  for (int k0 = kbeg; k0 < kend || seg + 1 < p.nseg; k0 += BK) {
    if (k0 >= kend) { k0 = kbeg; ++seg; }                               // the tile in LDS is the first one of the next segment
    bool more = k0 + BK < kend;
    int kn = k0 + BK;
    if (!more && seg + 1 < p.nseg) { more = true; kn = kbeg; pA += p.a_ss; pB += p.b_ss; }     // prefetch across the segment boundary
    if (more) {                                                          // in flight while this tile is contracted
#pragma unroll
      for (int i = 0; i < 2; ++i) { ra[i] = load_a(kn, i); rb[i] = load_b(kn, i); }
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a0 = As[buf][kk + lhi][wm + l31], a1 = As[buf][kk + lhi][wm + 32 + l31];
      const float b0 = Bs[buf][kk + lhi][wn + l31], b1 = Bs[buf][kk + lhi][wn + 32 + l31];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { store_t(As[buf ^ 1], TA, i, ra[i]); store_t(Bs[buf ^ 1], !TB, i, rb[i]); }
    }
    __syncthreads();
    buf ^= 1;
  }
  float* C = p.C + (long)blockIdx.z * p.c_zstride;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int col = n0 + wn + (q & 1) * 32 + l31;
    if (col >= p.N) continue;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int row = m0 + wm + (q >> 1) * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhi;      // accumulator layout of v_mfma_f32_32x32x2_f32
      if (row < p.M) { float* c = C + (long)row * p.ldc + col; *c = (p.beta ? *c : 0.f) + acc[q][j]; }
    }
  }
}

// out[i] = sum over z of part[z * zstride + i]  (fixed order)
// blockIdx.y = one of several such sums laid out back to back (part + y * nz * zstride -> out + y * n)
__global__ void sum_partials_kernel(const float* __restrict__ part, int nz, long zstride, long n, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  part += (long)blockIdx.y * nz * zstride; out += (long)blockIdx.y * n;
  float s = 0.f;
  for (int z = 0; z < nz; ++z) s += part[(long)z * zstride + i];
  out[i] = s;
}

// the six column sums of hc_bwd_rows_kernel's per-workgroup partials [nblk][6][C] -> d(g1), d(b1), d(g2), d(b2), d(bias) (2C), fixed order
__global__ void colsum6_kernel(const float* __restrict__ part, int nblk, int C, float* dg1, float* db1, float* dg2, float* db2, float* dbias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 6 * C) return;
  float s = 0.f;
  for (int z = 0; z < nblk; ++z) s += part[(long)z * 6 * C + i];
  const int j = i / C, c = i - j * C;
  float* out = j == 0 ? dg1 : (j == 1 ? db1 : (j == 2 ? dg2 : (j == 3 ? db2 : dbias + (j - 4) * C)));
  out[c] = s;
}

// dst (B, Tp, C) <- src (B, T, C) at rows [off, off + T), zeros elsewhere;  reverse = 1: src (B, T, C) <- dst rows [off, off + T)
__global__ void pad_rows_kernel(float* __restrict__ padded, float* __restrict__ flat, int B, int T, int Tp, int off, int C, int reverse) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // one float4 each
  const int c4 = C / 4;
  if (reverse) {
    if (i >= (long)B * T * c4) return;
    const long row = i / c4; const int c = (int)(i - row * c4);
    const int b = (int)(row / T), t = (int)(row - (long)b * T);
    reinterpret_cast<float4*>(flat)[i] = reinterpret_cast<const float4*>(padded)[((long)b * Tp + off + t) * c4 + c];
    return;
  }
  if (i >= (long)B * Tp * c4) return;
  const long row = i / c4; const int c = (int)(i - row * c4);
  const int b = (int)(row / Tp), tp = (int)(row - (long)b * Tp), t = tp - off;
  reinterpret_cast<float4*>(padded)[i] = (t >= 0 && t < T) ? reinterpret_cast<const float4*>(flat)[((long)b * T + t) * c4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// the same for channel counts that are not multiples of 4: padded rows are Cp = round_up(C, 4) floats wide, the extra columns zero
__global__ void pad_rows_generic_kernel(float* __restrict__ padded, float* __restrict__ flat, int B, int T, int Tp, int off, int C, int Cp, int reverse) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (reverse) {
    if (i >= (long)B * T * C) return;
    const long row = i / C; const int c = (int)(i - row * C);
    const int b = (int)(row / T), t = (int)(row - (long)b * T);
    flat[i] = padded[((long)b * Tp + off + t) * Cp + c];
    return;
  }
  if (i >= (long)B * Tp * Cp) return;
  const long row = i / Cp; const int c = (int)(i - row * Cp);
  const int b = (int)(row / Tp), tp = (int)(row - (long)b * Tp), t = tp - off;
  padded[i] = (t >= 0 && t < T && c < C) ? flat[((long)b * T + t) * C + c] : 0.f;
}

// dst (nmat, Rp, Cp) <- src (nmat, R, C), zero padded
__global__ void pad_matrix_kernel(const float* __restrict__ src, float* __restrict__ dst, int nmat, int R, int C, int Rp, int Cp) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nmat * Rp * Cp) return;
  const int c = (int)(i % Cp); const long q = i / Cp; const int r = (int)(q % Rp), m = (int)(q / Rp);
  dst[i] = (r < R && c < C) ? src[((long)m * R + r) * C + c] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------- hc backward, row part
struct HcBwdRowsParams {
  int B, T, Tp, C;
  int h_off, x_off;                // row offset of t = 0 inside an utterance of the H-aligned / x-aligned padded buffers
  const float* Hp;                 // (B, Tp, 2C) pre-norm WITHOUT bias, H-aligned
  const float* x; const float* dy; // (B, T, C)
  const float* bias; const float* g1; const float* b1; const float* g2; const float* b2;
  float* dHp;                      // (B, Tp, 2C) H-aligned; pad rows stay zero
  float* dxp;                      // (B, Tp, C) x-aligned: receives dy * (1 - gate); the GEMMs accumulate the conv part on top
  float* part;                     // [gridDim.x][6][C]: per-workgroup column sums of d(g1), d(b1), d(g2), d(b2), d(bias[:C]), d(bias[C:])
};

__device__ __forceinline__ float t_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// One wave per row, lane = 4 x NCH channels (C = 256 NCH).  modules.py:143-197 backward; layer-norm as modules.py:45-64 (biased
// two-pass variance, eps 1e-12 inside the root).
template <int NCH>
__global__ void __launch_bounds__(256) hc_bwd_rows_kernel(const HcBwdRowsParams p) {
  __shared__ __attribute__((aligned(16))) float red[4][6][NCH * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = p.C;
  const float invC = 1.0f / (float)C;
  tf32x4 acc[6][NCH];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int q = 0; q < NCH; ++q) acc[j][q] = tf32x4{0.f, 0.f, 0.f, 0.f};
  auto ld = [](const float* q) { return *reinterpret_cast<const tf32x4*>(q); };
  auto hsum = [](const tf32x4 v) { return v[0] + v[1] + v[2] + v[3]; };
  const long rows = (long)p.B * p.T;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const int b = (int)(r / p.T), t = (int)(r - (long)b * p.T);
    const float* H = p.Hp + ((long)b * p.Tp + p.h_off + t) * 2 * C;
    tf32x4 h1[NCH], h2[NCH], xv[NCH], dv[NCH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int c = q * 256 + lane * 4;
      h1[q] = ld(H + c) + ld(p.bias + c); h2[q] = ld(H + C + c) + ld(p.bias + C + c);
      xv[q] = ld(p.x + r * C + c); dv[q] = ld(p.dy + r * C + c);
      s1 += hsum(h1[q]); s2 += hsum(h2[q]);
    }
    const float m1 = t_wave_sum(s1) * invC, m2 = t_wave_sum(s2) * invC;
    float v1 = 0.f, v2 = 0.f;
#pragma unroll
    for (int q = 0; q < NCH; ++q) { h1[q] -= m1; h2[q] -= m2; v1 += hsum(h1[q] * h1[q]); v2 += hsum(h2[q] * h2[q]); }
    const float r1 = 1.0f / sqrtf(t_wave_sum(v1) * invC + 1e-12f), r2 = 1.0f / sqrtf(t_wave_sum(v2) * invC + 1e-12f);
    // forward: n = xhat * gamma + beta; s = sigmoid(n1); y = s n2 + (1 - s) x.   backward: dn2 = dy s; dn1 = dy (n2 - x) s (1 - s)
    tf32x4 dxh1[NCH], dxh2[NCH];
    float a1 = 0.f, a2 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int c = q * 256 + lane * 4;
      const tf32x4 g1 = ld(p.g1 + c), be1 = ld(p.b1 + c), g2 = ld(p.g2 + c), be2 = ld(p.b2 + c);
      h1[q] *= r1; h2[q] *= r2;                                          // xhat
      const tf32x4 n1 = h1[q] * g1 + be1, n2 = h2[q] * g2 + be2;
      tf32x4 s;
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = t_sigmoid(n1[e]);
      const tf32x4 dn2 = dv[q] * s, dn1 = dv[q] * (n2 - xv[q]) * s * (1.0f - s);
      *reinterpret_cast<tf32x4*>(p.dxp + ((long)b * p.Tp + p.x_off + t) * C + c) = dv[q] * (1.0f - s);
      acc[0][q] += dn1 * h1[q]; acc[1][q] += dn1; acc[2][q] += dn2 * h2[q]; acc[3][q] += dn2;
      dxh1[q] = dn1 * g1; dxh2[q] = dn2 * g2;
      a1 += hsum(dxh1[q]); a2 += hsum(dxh2[q]); c1 += hsum(dxh1[q] * h1[q]); c2 += hsum(dxh2[q] * h2[q]);
    }
    const float ma1 = t_wave_sum(a1) * invC, ma2 = t_wave_sum(a2) * invC, mc1 = t_wave_sum(c1) * invC, mc2 = t_wave_sum(c2) * invC;
    float* dH = p.dHp + ((long)b * p.Tp + p.h_off + t) * 2 * C;
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int c = q * 256 + lane * 4;
      const tf32x4 d1 = (dxh1[q] - ma1 - h1[q] * mc1) * r1, d2 = (dxh2[q] - ma2 - h2[q] * mc2) * r2;   // layer-norm backward
      *reinterpret_cast<tf32x4*>(dH + c) = d1; *reinterpret_cast<tf32x4*>(dH + C + c) = d2;
      acc[4][q] += d1; acc[5][q] += d2;
    }
  }
  // column sums of this workgroup: 4 waves -> one row of `part`
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int q = 0; q < NCH; ++q) *reinterpret_cast<tf32x4*>(&red[wave][j][q * 256 + lane * 4]) = acc[j][q];
  __syncthreads();
  for (int i = threadIdx.x; i < 6 * C; i += 256) {
    const int j = i / C, c = i - j * C;
    p.part[((long)blockIdx.x * 6 + j) * C + c] = (red[0][j][c] + red[1][j][c]) + (red[2][j][c] + red[3][j][c]);
  }
}

// ---------------------------------------------------------------------------------------------------------------- conv1d backward, row part
struct CBwdRowsParams {
  int B, T, Tp, C;                 // C = output channels (a multiple of 256)
  int h_off;
  const float* Hp;                 // (B, Tp, C) pre-norm WITHOUT bias, H-aligned
  const float* dy;                 // (B, T, C)
  const float* bias; const float* g; const float* b;
  int act;                         // 0 none, 1 relu, 2 sigmoid (modules.py:136-138)
  float* dHp;                      // (B, Tp, C) H-aligned; pad rows stay zero
  float* part;                     // [gridDim.x][3][C]: column sums of d(gamma), d(beta), d(bias)
};

// modules.py:91-141 backward up to the convolution: y = act(layer_norm(H)).  One wave per row, lane = 4 x NCH channels.
template <int NCH>
__global__ void __launch_bounds__(256) c_bwd_rows_kernel(const CBwdRowsParams p) {
  __shared__ __attribute__((aligned(16))) float red[4][3][NCH * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = p.C;
  const float invC = 1.0f / (float)C;
  tf32x4 acc[3][NCH];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int q = 0; q < NCH; ++q) acc[j][q] = tf32x4{0.f, 0.f, 0.f, 0.f};
  auto ld = [](const float* q) { return *reinterpret_cast<const tf32x4*>(q); };
  auto hsum = [](const tf32x4 v) { return v[0] + v[1] + v[2] + v[3]; };
  const long rows = (long)p.B * p.T;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const int b = (int)(r / p.T), t = (int)(r - (long)b * p.T);
    const float* H = p.Hp + ((long)b * p.Tp + p.h_off + t) * C;
    tf32x4 h[NCH], dxh[NCH];
    float s1 = 0.f;
#pragma unroll
    for (int q = 0; q < NCH; ++q) { const int c = q * 256 + lane * 4; h[q] = ld(H + c) + ld(p.bias + c); s1 += hsum(h[q]); }
    const float m = t_wave_sum(s1) * invC;
    float v1 = 0.f;
#pragma unroll
    for (int q = 0; q < NCH; ++q) { h[q] -= m; v1 += hsum(h[q] * h[q]); }
    const float rs = 1.0f / sqrtf(t_wave_sum(v1) * invC + 1e-12f);
    float a1 = 0.f, c1 = 0.f;
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int c = q * 256 + lane * 4;
      const tf32x4 g = ld(p.g + c), be = ld(p.b + c), dv = ld(p.dy + r * C + c);
      h[q] *= rs;                                                        // xhat
      const tf32x4 n = h[q] * g + be;
      tf32x4 dn = dv;
      if (p.act == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dn[e] = n[e] > 0.f ? dv[e] : 0.f;
      } else if (p.act == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float y = t_sigmoid(n[e]); dn[e] = dv[e] * y * (1.0f - y); }
      }
      acc[0][q] += dn * h[q]; acc[1][q] += dn;
      dxh[q] = dn * g;
      a1 += hsum(dxh[q]); c1 += hsum(dxh[q] * h[q]);
    }
    const float ma = t_wave_sum(a1) * invC, mc = t_wave_sum(c1) * invC;
    float* dH = p.dHp + ((long)b * p.Tp + p.h_off + t) * C;
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int c = q * 256 + lane * 4;
      const tf32x4 d = (dxh[q] - ma - h[q] * mc) * rs;
      *reinterpret_cast<tf32x4*>(dH + c) = d;
      acc[2][q] += d;
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int q = 0; q < NCH; ++q) *reinterpret_cast<tf32x4*>(&red[wave][j][q * 256 + lane * 4]) = acc[j][q];
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += 256) {
    const int j = i / C, c = i - j * C;
    p.part[((long)blockIdx.x * 3 + j) * C + c] = (red[0][j][c] + red[1][j][c]) + (red[2][j][c] + red[3][j][c]);
  }
}

// The same for any width C <= 64 NI (80 mel bins, 1025 linear bins): lane owns channels lane + 64 i; Hp / dHp rows are Cp floats wide.
template <int NI>
__global__ void __launch_bounds__(256) c_bwd_rows_generic_kernel(const CBwdRowsParams p, const int Cp) {
  extern __shared__ float gred[];                  // [4 waves][3][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = p.C;
  const float invC = 1.0f / (float)C;
  float acc[3][NI];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[j][i] = 0.f;
  const long rows = (long)p.B * p.T;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const int b = (int)(r / p.T), t = (int)(r - (long)b * p.T);
    const float* H = p.Hp + ((long)b * p.Tp + p.h_off + t) * Cp;
    float h[NI], dxh[NI];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) { const int c = lane + 64 * i; h[i] = c < C ? H[c] + p.bias[c] : 0.f; s1 += h[i]; }
    const float m = t_wave_sum(s1) * invC;
    float v1 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) { const int c = lane + 64 * i; h[i] = c < C ? h[i] - m : 0.f; v1 += h[i] * h[i]; }
    const float rs = 1.0f / sqrtf(t_wave_sum(v1) * invC + 1e-12f);
    float a1 = 0.f, c1 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      dxh[i] = 0.f;
      if (c < C) {
        const float g = p.g[c], dv = p.dy[r * C + c];
        h[i] *= rs;
        const float n = h[i] * g + p.b[c];
        float dn = dv;
        if (p.act == 1) dn = n > 0.f ? dv : 0.f;
        else if (p.act == 2) { const float y = t_sigmoid(n); dn = dv * y * (1.0f - y); }
        acc[0][i] += dn * h[i]; acc[1][i] += dn;
        dxh[i] = dn * g;
        a1 += dxh[i]; c1 += dxh[i] * h[i];
      }
    }
    const float ma = t_wave_sum(a1) * invC, mc = t_wave_sum(c1) * invC;
    float* dH = p.dHp + ((long)b * p.Tp + p.h_off + t) * Cp;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      if (c < C) { const float d = (dxh[i] - ma - h[i] * mc) * rs; dH[c] = d; acc[2][i] += d; }
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < NI; ++i) { const int c = lane + 64 * i; if (c < C) gred[(wave * 3 + j) * C + c] = acc[j][i]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += 256) {
    const int j = i / C, c = i - j * C;
    p.part[((long)blockIdx.x * 3 + j) * C + c] = (gred[(0 * 3 + j) * C + c] + gred[(1 * 3 + j) * C + c]) + (gred[(2 * 3 + j) * C + c] + gred[(3 * 3 + j) * C + c]);
  }
}

// the three column sums of c_bwd_rows_kernel's partials [nblk][3][C] -> d(gamma), d(beta), d(bias), fixed order
__global__ void colsum3_kernel(const float* __restrict__ part, int nblk, int C, float* dg, float* db, float* dbias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * C) return;
  float s = 0.f;
  for (int z = 0; z < nblk; ++z) s += part[(long)z * 3 * C + i];
  const int j = i / C, c = i - j * C;
  (j == 0 ? dg : (j == 1 ? db : dbias))[c] = s;
}

// ---------------------------------------------------------------------------------------------------------------- forward row parts (training keeps TF-layout weights)
// y = sigmoid(LN(H1)) * LN(H2) + (1 - sigmoid(LN(H1))) * x   (modules.py:189-194), H = Hp + bias.  Wave per row, lane = 4 x NCH channels.
struct HcFwdRowsParams { int B, T, Tp, C, h_off; const float* Hp; const float* x; const float* bias; const float* g1; const float* b1; const float* g2; const float* b2; float* y; };
template <int NCH>
__global__ void __launch_bounds__(256) hc_fwd_rows_kernel(const HcFwdRowsParams p) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (long)p.B * p.T) return;
  const int C = p.C, b = (int)(r / p.T), t = (int)(r - (long)b * p.T);
  const float invC = 1.0f / (float)C;
  auto ld = [](const float* q) { return *reinterpret_cast<const tf32x4*>(q); };
  auto hsum = [](const tf32x4 v) { return v[0] + v[1] + v[2] + v[3]; };
  const float* H = p.Hp + ((long)b * p.Tp + p.h_off + t) * 2 * C;
  tf32x4 h1[NCH], h2[NCH];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int q = 0; q < NCH; ++q) { const int c = q * 256 + lane * 4; h1[q] = ld(H + c) + ld(p.bias + c); h2[q] = ld(H + C + c) + ld(p.bias + C + c); s1 += hsum(h1[q]); s2 += hsum(h2[q]); }
  const float m1 = t_wave_sum(s1) * invC, m2 = t_wave_sum(s2) * invC;
  float v1 = 0.f, v2 = 0.f;
#pragma unroll
  for (int q = 0; q < NCH; ++q) { h1[q] -= m1; h2[q] -= m2; v1 += hsum(h1[q] * h1[q]); v2 += hsum(h2[q] * h2[q]); }
  const float r1 = 1.0f / sqrtf(t_wave_sum(v1) * invC + 1e-12f), r2 = 1.0f / sqrtf(t_wave_sum(v2) * invC + 1e-12f);
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    const int c = q * 256 + lane * 4;
    const tf32x4 n1 = h1[q] * r1 * ld(p.g1 + c) + ld(p.b1 + c), n2 = h2[q] * r2 * ld(p.g2 + c) + ld(p.b2 + c), xv = ld(p.x + r * C + c);
    tf32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float s_ = t_sigmoid(n1[e]); y[e] = s_ * n2[e] + (1.0f - s_) * xv[e]; }
    *reinterpret_cast<tf32x4*>(p.y + r * C + c) = y;
  }
}

// y = act(LN(Hp + bias)) (modules.py:135-138) for any width C <= 1088: lane owns channels lane + 64 i; Hp rows are Cp floats wide.
struct CFwdRowsParams { int B, T, Tp, C, Cp, h_off; const float* Hp; const float* bias; const float* g; const float* b; int act; float* y; };
__global__ void __launch_bounds__(256) c_fwd_rows_kernel(const CFwdRowsParams p) {
  constexpr int NI = 17;
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (long)p.B * p.T) return;
  const int C = p.C, b = (int)(r / p.T), t = (int)(r - (long)b * p.T);
  const float invC = 1.0f / (float)C;
  const float* H = p.Hp + ((long)b * p.Tp + p.h_off + t) * p.Cp;
  float h[NI];
  float s1 = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) { const int c = lane + 64 * i; h[i] = c < C ? H[c] + p.bias[c] : 0.f; s1 += h[i]; }
  const float m = t_wave_sum(s1) * invC;
  float v1 = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) { const int c = lane + 64 * i; h[i] = c < C ? h[i] - m : 0.f; v1 += h[i] * h[i]; }
  const float rs = 1.0f / sqrtf(t_wave_sum(v1) * invC + 1e-12f);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane + 64 * i;
    if (c < C) {
      float n = h[i] * rs * p.g[c] + p.b[c];
      if (p.act == 1) n = fmaxf(n, 0.f); else if (p.act == 2) n = t_sigmoid(n);
      p.y[r * C + c] = n;
    }
  }
}

// tf.layers.dropout(training=True) (modules.py:139,195,245): y = x * keep / (1 - rate).  The keep bit of element i under `key` is a
// counter-based hash (splitmix64(splitmix64(key) + i)), so the backward pass regenerates the same mask from the same key: dx = dy * keep / (1 - rate).
// (TensorFlow's own random stream cannot be reproduced; oracle/train_ref.dropout_mask restates this hash.)
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, unsigned long long key, float rate, float scale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // the key is hashed BEFORE the element counter is added: layer keys differ in their low bits only (+1 per layer), and splitmix64(key + i)
  // would make the masks of consecutive layers one random stream shifted by one element (mask[l+1][i] == mask[l][i+1])
  const float u = (float)(splitmix64(splitmix64(key) + (unsigned long long)i) >> 40) * (1.0f / 16777216.0f);     // 24 uniform bits in [0, 1)
  y[i] = u >= rate ? x[i] * scale : 0.f;
}

__global__ void sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = t_sigmoid(x[i]);
}

// modules.py:13-42 forward: y[i] = table[ids[i]], row 0 reads as zeros (:36-38); ids outside the table read row 0
__global__ void embed_fwd_kernel(const int* __restrict__ ids, const float* __restrict__ table, long n, int vocab, int e, float* __restrict__ y) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * e) return;
  const long r = i / e; const int c = (int)(i - r * e);
  const int id = ids[r];
  y[i] = (id > 0 && id < vocab) ? table[(long)id * e + c] : 0.f;
}

// dst (B, C, R) <- src (B, R, C)^T per batch item (alignments = A^T, networks.py:153)
// (src rows are ld floats apart: the attention matrix keeps a leading dimension rounded up to 4 floats for the GEMMs' 16-byte loads)
__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int R, int C, int ld) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * R * C) return;
  const int r = (int)(i % R); const long q = i / R; const int c = (int)(q % C), b = (int)(q / C);
  dst[i] = src[((long)b * R + r) * ld + c];
}

// dst rows (ld_dst apart) at column offset <- src rows (C wide)
__global__ void scatter_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int ld_dst, long rows, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  const long r = i / C; const int c = (int)(i - r * C);
  dst[r * ld_dst + c] = src[i];
}

// ---------------------------------------------------------------------------------------------------------------- attention backward, row parts
// S (rows, Np) <- softmax over the first N columns of scale * S, in place (networks.py:140,148; training: no mask).  Wave per row.
__global__ void __launch_bounds__(256) softmax_rows_kernel(float* __restrict__ S, long rows, int N, int Np, float scale) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float* s = S + r * Np;
  float mx = -INFINITY;
  for (int n = lane; n < N; n += 64) mx = fmaxf(mx, s[n] * scale);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int n = lane; n < N; n += 64) { const float e = expf(s[n] * scale - mx); s[n] = e; sum += e; }
  const float inv = 1.0f / t_wave_sum(sum);
  for (int n = lane; n < N; n += 64) s[n] *= inv;
  for (int n = N + lane; n < Np; n += 64) s[n] = 0.f;             // pad columns: read (as zeros) by the 16-byte loads of the GEMMs that contract over N
}

// dS = A * (dA + dAl^T - sum_n A (dA + dAl^T)) * scale, in place in dA (rows = B * T; dAl is (B, N, T)).  Wave per row.
__global__ void __launch_bounds__(256) softmax_bwd_rows_kernel(const float* __restrict__ A, float* __restrict__ dA, const float* __restrict__ dAl,
                                                               int B, int T, int N, int Np, float scale) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (long)B * T) return;
  const int b = (int)(r / T), t = (int)(r - (long)b * T);
  const float* a = A + r * Np; float* d = dA + r * Np;
  const float* al = dAl + (long)b * N * T + t;
  float dot = 0.f;
  for (int n = lane; n < N; n += 64) { const float g = d[n] + al[(long)n * T]; d[n] = g; dot += a[n] * g; }
  dot = t_wave_sum(dot);
  for (int n = lane; n < N; n += 64) d[n] = a[n] * (d[n] - dot) * scale;
  for (int n = N + lane; n < Np; n += 64) d[n] = 0.f;
}

// dst (rows, C) <- src (rows, C) taken from a wider row (ld floats apart)
__global__ void copy_cols_kernel(const float* __restrict__ src, int ld, float* __restrict__ dst, long rows, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  const long r = i / C; const int c = (int)(i - r * C);
  dst[i] = src[r * ld + c];
}

// modules.py:13-42 backward: dTable[v] = sum of dy rows whose id is v; row 0 (the zero row of the lookup) receives nothing.
// One workgroup per vocabulary entry, fixed summation order.
__global__ void __launch_bounds__(256) embed_bwd_kernel(const int* __restrict__ ids, const float* __restrict__ dy, long n, int e, float* __restrict__ dT) {
  const int v = blockIdx.x;
  for (int c = threadIdx.x; c < e; c += 256) {
    float s = 0.f;
    if (v != 0) for (long i = 0; i < n; ++i) if (ids[i] == v) s += dy[i * e + c];
    dT[(long)v * e + c] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------- losses
// train.py:87,90 (and :104,107): L1 and sigmoid cross-entropy means over n elements, gradients of their sum.
// part[block][2] = partial sums of |Y - z| and xent(logits, z).
__global__ void __launch_bounds__(256) l1_bd_loss_kernel(const float* __restrict__ Y, const float* __restrict__ logits, const float* __restrict__ z,
                                                         long n, float* __restrict__ dY, float* __restrict__ dlog, float* __restrict__ part) {
  __shared__ float sh[2][4];
  const float invn = 1.0f / (float)n;
  float l1 = 0.f, bd = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = Y[i] - z[i], x = logits[i];
    l1 += fabsf(d);
    dY[i] = (d > 0.f ? invn : (d < 0.f ? -invn : 0.f));
    bd += fmaxf(x, 0.f) - x * z[i] + log1pf(expf(-fabsf(x)));           // tf.nn.sigmoid_cross_entropy_with_logits
    dlog[i] = (t_sigmoid(x) - z[i]) * invn;
  }
  l1 = t_wave_sum(l1); bd = t_wave_sum(bd);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = l1; sh[1][threadIdx.x >> 6] = bd; }
  __syncthreads();
  if (threadIdx.x < 2) part[(long)blockIdx.x * 2 + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

// train.py:93-97 with utils.py:134-140's weights: alignments (B, N, T); part[block] = partial sum of |A W| over the cropped region
__global__ void __launch_bounds__(256) att_loss_kernel(const float* __restrict__ A, int B, int N, int T, int max_N, int max_T, float inv_mask_sum,
                                                       float* __restrict__ dA, float* __restrict__ part) {
  __shared__ float sh[4];
  const long n = (long)B * N * T;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int t = (int)(i % T), nn = (int)((i / T) % N);
    float g = 0.f;
    if (nn < max_N && t < max_T) {
      const float d = (float)t / (float)max_T - (float)nn / (float)max_N;
      const float w = 1.0f - expf(-d * d / (2.0f * 0.2f * 0.2f));        // utils.py:138, g = 0.2
      const float aw = A[i] * w;
      s += fabsf(aw);
      g = (aw > 0.f ? w : (aw < 0.f ? -w : 0.f)) * inv_mask_sum;
    }
    dA[i] = g;
  }
  s = t_wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// out[j] = scale * sum over blocks of part[block * stride + j], j < nout (one workgroup, fixed order)
__global__ void __launch_bounds__(256) finish_loss_kernel(const float* __restrict__ part, int nblk, int stride, int nout, float scale, float* __restrict__ out) {
  __shared__ float sh[4];
  for (int j = 0; j < nout; ++j) {
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 256) s += part[(long)b * stride + j];
    s = t_wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[j] = ((sh[0] + sh[1]) + (sh[2] + sh[3])) * scale;
    __syncthreads();
  }
}

// train.py:119-131: clip_by_value(grad, -1, 1) then tf.train.AdamOptimizer (beta1 0.9, beta2 0.999, eps 1e-8)
__global__ void adam_step_kernel(float* __restrict__ var, const float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v, long n, float lr_t) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = fminf(fmaxf(grad[i], -1.0f), 1.0f);
  const float mi = 0.9f * m[i] + (1.0f - 0.9f) * g, vi = 0.999f * v[i] + (1.0f - 0.999f) * g * g;
  m[i] = mi; v[i] = vi;
  var[i] -= lr_t * mi / (sqrtf(vi) + 1e-8f);
}

// The same update for many variables in ONE launch (a training step updates 209 Text2Mel or 80 SSRN variables): blockIdx.y = variable,
// blockIdx.x = a 16 K-element chunk of it (blocks past the end of a short variable leave at once).
struct AdamItem { float* var; const float* grad; float* m; float* v; long n; };
struct AdamBatch { AdamItem it[64]; };                  // 2.5 KB of kernel arguments: no table upload, nothing to keep alive on the host
__global__ void __launch_bounds__(256) adam_multi_kernel(const AdamBatch items, float lr_t) {
  const AdamItem it = items.it[blockIdx.y];
  const long base = (long)blockIdx.x * 16384;
  if (base >= it.n) return;
  const long end = base + 16384 < it.n ? base + 16384 : it.n;
  for (long i = base + threadIdx.x; i < end; i += 256) {
    const float g = fminf(fmaxf(it.grad[i], -1.0f), 1.0f);
    const float mi = 0.9f * it.m[i] + (1.0f - 0.9f) * g, vi = 0.999f * it.v[i] + (1.0f - 0.999f) * g * g;
    it.m[i] = mi; it.v[i] = vi;
    it.var[i] -= lr_t * mi / (sqrtf(vi) + 1e-8f);
  }
}

}  // namespace dctts
