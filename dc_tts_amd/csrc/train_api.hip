// train_api.hip -- C ABI of the first training slice (include/dctts_train.h): backward of the highway-convolution block,
// the losses of train.py:85-110 with gradients, the clip + Adam update of train.py:119-131.  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/dctts_hip.h"
#include "../../include/dctts_train.h"
#include "train_kernels.h"

using namespace dctts;

int dctts_set_error(int code, const std::string& msg);       // dctts_api.hip: the library's one error channel (dctts_last_error)
#define TFAIL(code, msg) return dctts_set_error((code), (msg))
#define THIP(x)                                                                                      \
  do {                                                                                               \
    hipError_t e__ = (x);                                                                            \
    if (e__ != hipSuccess) return dctts_set_error(DCTTS_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e__)); \
  } while (0)

struct TBuf { void* p = nullptr; size_t bytes = 0; };

// One saved pre-norm tensor of a forward pass (dctts_train_tape): identified by the layer's kernel pointer and geometry
struct TapeEntry { TBuf buf; const float* kernel = nullptr; int B = 0, T = 0, Cin = 0, Ch = 0, k = 0, rate = 0, pl = 0; };

struct dctts_train {
  int device = 0;
  TBuf xp, Hp, dHp, dxp, part, wpart, lpart, wpad, att;
  bool tape_on = false;               // forward passes keep their pre-norm tensors, backward passes consume them in reverse order
  std::vector<TapeEntry> tape; size_t tape_n = 0;
};

namespace {
struct DevScope {                    // restores the caller's current device
  int old = -1; bool ok = false;
  explicit DevScope(int dev) { ok = hipGetDevice(&old) == hipSuccess && hipSetDevice(dev) == hipSuccess; }
  ~DevScope() { if (old >= 0) (void)hipSetDevice(old); }
};
int reserve(TBuf* b, size_t bytes) {
  if (b->bytes >= bytes) return 0;
  (void)hipDeviceSynchronize();
  if (b->p) (void)hipFree(b->p);
  b->p = nullptr; b->bytes = 0;
  if (hipMalloc(&b->p, bytes) != hipSuccess) return dctts_set_error(DCTTS_ERR_HIP, "training workspace: hipMalloc failed");
  b->bytes = bytes;
  return 0;
}
template <bool TA, bool TB>
int gemm(hipStream_t st, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int beta,
         int splits = 1, long zstride = 0, int nseg = 1, long a_ss = 0, long b_ss = 0) {
  GemmParams p{A, B, C, M, N, K, lda, ldb, ldc, beta, K, zstride, 0, 0, 0};
  p.nseg = nseg; p.a_ss = a_ss; p.b_ss = b_ss;
  if (splits > 1) p.kchunk = ((K + splits - 1) / splits + 15) / 16 * 16;
  const int nz = (K + p.kchunk - 1) / p.kchunk;
  hipLaunchKernelGGL((gemm_kernel<TA, TB>), dim3((N + 127) / 128, (M + 127) / 128, nz), dim3(256), 0, st, p);
  if (hipGetLastError() != hipSuccess) return dctts_set_error(DCTTS_ERR_HIP, "gemm launch failed");
  return nz;
}
// nb GEMMs of the same shape (A + b * a_zs, B + b * b_zs), each split over K: the partial of (b, split) lands at C + (b * nz + split) * zstride; returns nz
template <bool TA, bool TB>
int gemm_batched_splitk(hipStream_t st, int nb, const float* A, long a_zs, const float* B, long b_zs, float* C, long zstride,
                        int M, int N, int K, int lda, int ldb, int ldc, int splits) {
  GemmParams p{A, B, C, M, N, K, lda, ldb, ldc, 0, K, zstride, 1, a_zs, b_zs};
  if (splits > 1) p.kchunk = ((K + splits - 1) / splits + 15) / 16 * 16;
  const int nz = (K + p.kchunk - 1) / p.kchunk;
  p.nsplit = nz;
  hipLaunchKernelGGL((gemm_kernel<TA, TB>), dim3((N + 127) / 128, (M + 127) / 128, nb * nz), dim3(256), 0, st, p);
  if (hipGetLastError() != hipSuccess) return dctts_set_error(DCTTS_ERR_HIP, "gemm launch failed");
  return nz;
}
// nb independent GEMMs of the same shape: A + z * a_zs, B + z * b_zs, C + z * c_zs
template <bool TA, bool TB>
int gemm_batched(hipStream_t st, int nb, const float* A, long a_zs, const float* B, long b_zs, float* C, long c_zs, int M, int N, int K, int lda, int ldb, int ldc, int beta) {
  GemmParams p{A, B, C, M, N, K, lda, ldb, ldc, beta, K, c_zs, 1, a_zs, b_zs};
  hipLaunchKernelGGL((gemm_kernel<TA, TB>), dim3((N + 127) / 128, (M + 127) / 128, nb), dim3(256), 0, st, p);
  if (hipGetLastError() != hipSuccess) return dctts_set_error(DCTTS_ERR_HIP, "gemm launch failed");
  return 0;
}
}  // namespace

extern "C" int dctts_train_create(dctts_train** out, int device) {
  if (!out) TFAIL(DCTTS_ERR_ARG, "null argument");
  dctts_train* t = new dctts_train();
  t->device = device;
  DevScope ds(device);
  if (!ds.ok) { delete t; TFAIL(DCTTS_ERR_HIP, "hipSetDevice: no such device"); }
  *out = t;
  return 0;
}

extern "C" int dctts_train_tape(dctts_train* t, int enable) {
  if (!t) TFAIL(DCTTS_ERR_ARG, "null argument");
  t->tape_on = enable != 0;
  t->tape_n = 0;
  return 0;
}

extern "C" int dctts_train_destroy(dctts_train* t) {
  if (!t) return 0;
  DevScope ds(t->device);
  for (TBuf* b : {&t->xp, &t->Hp, &t->dHp, &t->dxp, &t->part, &t->wpart, &t->lpart, &t->wpad, &t->att}) if (b->p) (void)hipFree(b->p);
  for (TapeEntry& e : t->tape) if (e.buf.p) (void)hipFree(e.buf.p);
  delete t;
  return 0;
}

extern "C" size_t dctts_train_device_bytes(const dctts_train* t) {
  if (!t) return 0;
  return t->xp.bytes + t->Hp.bytes + t->dHp.bytes + t->dxp.bytes + t->part.bytes + t->wpart.bytes + t->lpart.bytes + t->wpad.bytes + t->att.bytes +
         [&] { size_t n = 0; for (const TapeEntry& e : t->tape) n += e.buf.bytes; return n; }();
}

namespace {
// Shared scaffolding of the two convolution backward passes.  Cin input channels, Ch pre-norm channels (2C for hc, Cout for conv1d).
// x-aligned buffers hold t = 0 at row pl of an utterance, H-aligned ones at row pr: then for every tap j
//   H_aligned[r] needs x_aligned[r - pr + j*rate]   and   dx_aligned[r] needs dH_aligned[r + pr - j*rate]   (r a flat row index),
// i.e. every shift is a pointer offset, and rows that fall into another utterance's padding read / write zeros.
struct ConvGeom {
  int B, T, Cin, Ch, k, rate, pl, pr, Tp; long R, Rv;
  int Cinp, Chp;                                   // widths of the padded buffers: multiples of 4 floats (16-byte loads); the extra columns hold zeros
  ConvGeom(int B_, int T_, int Cin_, int Ch_, int k_, int rate_, int causal) : B(B_), T(T_), Cin(Cin_), Ch(Ch_), k(k_), rate(rate_) {
    // tf.layers.conv1d tap j reads x[t + j*rate - pl] (modules.py:121-125,173-177): CAUSAL pl = (k-1) rate, SAME pl = total / 2
    const int total = (k - 1) * rate;
    pl = causal ? total : total / 2; pr = total - pl; Tp = pl + T + pr;
    R = (long)B * Tp; Rv = R - pl - pr;            // rows of the padded buffers; rows every GEMM output covers
    Cinp = (Cin + 3) / 4 * 4; Chp = (Ch + 3) / 4 * 4;
  }
  bool padded() const { return Cinp != Cin || Chp != Ch; }
  int splits() const { const int wt = ((Cin + 127) / 128) * ((Ch + 127) / 128); return std::max(1, std::min(64, (512 + wt - 1) / wt)); }
};

// pads x, clears the gradient buffers, recomputes the pre-norm tensor H (without bias) over rows [pr, R - pl) of the H-aligned buffer.
// *kp = the kernel the GEMMs read: the caller's (k, Cin, Ch) or, when a width is not a multiple of 4, a zero-padded (k, Cinp, Chp) copy.
// *hp = where the pre-norm rows are: the scratch buffer, or -- backward pass with the tape on and a matching entry on top -- the tensor
// the forward pass kept (then the k GEMMs are skipped).  fwd: forward pass (keeps a copy when the tape is on).
int conv_prenorm(dctts_train* t, hipStream_t st, const ConvGeom& g, const float* x, const float* kernel, const float** kp, const float** hp, bool fwd) {
  if (reserve(&t->xp, (size_t)g.R * g.Cinp * 4) || reserve(&t->Hp, (size_t)g.R * g.Chp * 4) || reserve(&t->dHp, (size_t)g.R * g.Chp * 4) ||
      reserve(&t->dxp, (size_t)g.R * g.Cinp * 4) || reserve(&t->wpart, (size_t)g.k * g.splits() * g.Cin * g.Ch * 4)) return DCTTS_ERR_HIP;
  float *xp = (float*)t->xp.p, *Hp = (float*)t->Hp.p;
  *kp = kernel;
  if (g.padded()) {
    if (reserve(&t->wpad, (size_t)g.k * g.Cinp * g.Chp * 4)) return DCTTS_ERR_HIP;
    const long n = (long)g.k * g.Cinp * g.Chp;
    hipLaunchKernelGGL(pad_matrix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, kernel, (float*)t->wpad.p, g.k, g.Cin, g.Ch, g.Cinp, g.Chp);
    THIP(hipGetLastError());
    *kp = (const float*)t->wpad.p;
    const long ne = g.R * g.Cinp;
    hipLaunchKernelGGL(pad_rows_generic_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, xp, const_cast<float*>(x), g.B, g.T, g.Tp, g.pl, g.Cin, g.Cinp, 0);
  } else {
    const long n4 = g.R * (g.Cin / 4);
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, xp, const_cast<float*>(x), g.B, g.T, g.Tp, g.pl, g.Cin, 0);
  }
  THIP(hipGetLastError());
  *hp = Hp;
  if (!fwd) {
    THIP(hipMemsetAsync(t->dHp.p, 0, (size_t)g.R * g.Chp * 4, st));
    THIP(hipMemsetAsync(t->dxp.p, 0, (size_t)g.R * g.Cinp * 4, st));
    if (t->tape_on && t->tape_n > 0) {
      const TapeEntry& e = t->tape[t->tape_n - 1];
      if (e.kernel == kernel && e.B == g.B && e.T == g.T && e.Cin == g.Cin && e.Ch == g.Ch && e.k == g.k && e.rate == g.rate && e.pl == g.pl) {
        --t->tape_n; *hp = (const float*)e.buf.p;
        return 0;
      }
      t->tape_n = 0;                                   // out of step with the forward pass: recompute from here on
    }
  }
  {   // all taps in one launch: tap j reads x shifted by j * rate rows and kernel[j]
    const int rc = gemm<false, false>(st, xp, *kp, Hp + (long)g.pr * g.Chp, (int)g.Rv, g.Chp, g.Cinp, g.Cinp, g.Chp, g.Chp, 0,
                                      1, 0, g.k, (long)g.rate * g.Cinp, (long)g.Cinp * g.Chp);
    if (rc < 0) return rc;
  }
  if (fwd && t->tape_on) {
    if (t->tape_n == t->tape.size()) t->tape.emplace_back();
    TapeEntry& e = t->tape[t->tape_n];
    const size_t bytes = (size_t)g.R * g.Chp * 4;
    if (reserve(&e.buf, bytes)) return DCTTS_ERR_HIP;
    THIP(hipMemcpyAsync(e.buf.p, Hp, bytes, hipMemcpyDeviceToDevice, st));
    e.kernel = kernel; e.B = g.B; e.T = g.T; e.Cin = g.Cin; e.Ch = g.Ch; e.k = g.k; e.rate = g.rate; e.pl = g.pl;
    ++t->tape_n;
  }
  return 0;
}

// dkernel[j] = x_shifted^T . dH (K = every row: split-K partials, fixed-order sum);  dx (+)= dH_shifted . kernel[j]^T;  dx out of its padding
int conv_grads(dctts_train* t, hipStream_t st, const ConvGeom& g, const float* kp, float* dkernel, float* dx) {
  float *xp = (float*)t->xp.p, *dHp = (float*)t->dHp.p, *dxp = (float*)t->dxp.p;
  const long nw = (long)g.Cin * g.Ch;
  {   // the taps as ONE batched split-K launch (tap j = x shifted by j * rate rows against the same dH), one fixed-order sum, one dgrad launch
    const int nz = gemm_batched_splitk<true, false>(st, g.k, xp, (long)g.rate * g.Cinp, dHp + (long)g.pr * g.Chp, 0, (float*)t->wpart.p, nw,
                                                    g.Cin, g.Ch, (int)g.Rv, g.Cinp, g.Chp, g.Ch, g.splits());
    if (nz < 0) return nz;
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((nw + 255) / 256), g.k), dim3(256), 0, st, (const float*)t->wpart.p, nz, nw, nw, dkernel);
    THIP(hipGetLastError());
    const int rc = gemm<false, true>(st, dHp + (long)(g.pl + g.pr) * g.Chp, kp, dxp + (long)g.pl * g.Cinp, (int)g.Rv, g.Cinp, g.Chp, g.Chp, g.Chp, g.Cinp, 1,      // += : the row kernel left the direct part of dx there
                                     1, 0, g.k, -(long)g.rate * g.Chp, (long)g.Cinp * g.Chp);
    if (rc < 0) return rc;
  }
  if (g.padded()) {
    const long ne = (long)g.B * g.T * g.Cin;
    hipLaunchKernelGGL(pad_rows_generic_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, dxp, dx, g.B, g.T, g.Tp, g.pl, g.Cin, g.Cinp, 1);
  } else {
    const long m4 = (long)g.B * g.T * (g.Cin / 4);
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((m4 + 255) / 256)), dim3(256), 0, st, dxp, dx, g.B, g.T, g.Tp, g.pl, g.Cin, 1);
  }
  THIP(hipGetLastError());
  return 0;
}

// the row part of a conv1d / conv1d_transpose backward: vector lanes for the widths of the hidden layers, scalar lanes otherwise
int launch_c_rows(dctts_train* t, hipStream_t st, const CBwdRowsParams& q, int Cp, int nblk) {
  const int C = q.C;
  if (C == 256 && Cp == C) hipLaunchKernelGGL((c_bwd_rows_kernel<1>), dim3(nblk), dim3(256), 0, st, q);
  else if (C == 512 && Cp == C) hipLaunchKernelGGL((c_bwd_rows_kernel<2>), dim3(nblk), dim3(256), 0, st, q);
  else if (C == 1024 && Cp == C) hipLaunchKernelGGL((c_bwd_rows_kernel<4>), dim3(nblk), dim3(256), 0, st, q);
  else {
    const size_t lds = (size_t)4 * 3 * C * 4;
    if (C <= 128) hipLaunchKernelGGL((c_bwd_rows_generic_kernel<2>), dim3(nblk), dim3(256), lds, st, q, Cp);
    else hipLaunchKernelGGL((c_bwd_rows_generic_kernel<17>), dim3(nblk), dim3(256), lds, st, q, Cp);
  }
  THIP(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int dctts_train_hc_backward(dctts_train* t, const float* x, const float* dy, const float* kernel, const float* bias,
                                       const float* g1, const float* b1, const float* g2, const float* b2,
                                       int B, int T, int C, int k, int rate, int causal,
                                       float* dx, float* dkernel, float* dbias, float* dg1, float* db1, float* dg2, float* db2, void* stream) {
  if (!t || !x || !dy || !kernel || !bias || !g1 || !b1 || !g2 || !b2 || !dx || !dkernel || !dbias || !dg1 || !db1 || !dg2 || !db2)
    TFAIL(DCTTS_ERR_ARG, "hc_backward: null argument");
  if (B <= 0 || T <= 0 || (C != 256 && C != 512 && C != 1024) || (k != 1 && k != 3) || rate < 1)
    TFAIL(DCTTS_ERR_ARG, "hc_backward: C must be 256, 512 or 1024, k 1 or 3, rate >= 1");
  DevScope ds(t->device);
  if (!ds.ok) TFAIL(DCTTS_ERR_HIP, "hipSetDevice failed");
  hipStream_t st = (hipStream_t)stream;
  const ConvGeom g(B, T, C, 2 * C, k, rate, causal);
  const float *kp = nullptr, *hp = nullptr;
  int rc = conv_prenorm(t, st, g, x, kernel, &kp, &hp, false);
  if (rc) return rc;
  // the row part: dH, the direct part of dx, column sums
  const int nblk = (int)std::min<long>(256, ((long)B * T + 3) / 4);
  if (reserve(&t->part, (size_t)nblk * 6 * C * 4)) return DCTTS_ERR_HIP;
  HcBwdRowsParams q{B, T, g.Tp, C, g.pr, g.pl, hp, x, dy, bias, g1, b1, g2, b2, (float*)t->dHp.p, (float*)t->dxp.p, (float*)t->part.p};
  if (C == 256) hipLaunchKernelGGL((hc_bwd_rows_kernel<1>), dim3(nblk), dim3(256), 0, st, q);
  else if (C == 512) hipLaunchKernelGGL((hc_bwd_rows_kernel<2>), dim3(nblk), dim3(256), 0, st, q);
  else hipLaunchKernelGGL((hc_bwd_rows_kernel<4>), dim3(nblk), dim3(256), 0, st, q);
  THIP(hipGetLastError());
  hipLaunchKernelGGL(colsum6_kernel, dim3((6 * C + 255) / 256), dim3(256), 0, st, (const float*)t->part.p, nblk, C, dg1, db1, dg2, db2, dbias);
  THIP(hipGetLastError());
  return conv_grads(t, st, g, kp, dkernel, dx);
}

extern "C" int dctts_train_conv1d_backward(dctts_train* t, const float* x, const float* dy, const float* kernel, const float* bias,
                                           const float* gamma, const float* beta, int B, int T, int Cin, int Cout, int k, int rate, int causal, int act,
                                           float* dx, float* dkernel, float* dbias, float* dgamma, float* dbeta, void* stream) {
  if (!t || !x || !dy || !kernel || !bias || !gamma || !beta || !dx || !dkernel || !dbias || !dgamma || !dbeta) TFAIL(DCTTS_ERR_ARG, "conv1d_backward: null argument");
  if (B <= 0 || T <= 0 || Cin <= 0 || Cin > 4096 || Cout <= 0 || Cout > 1088 || (k != 1 && k != 3) || rate < 1 || act < 0 || act > 2)
    TFAIL(DCTTS_ERR_ARG, "conv1d_backward: 1 <= Cout <= 1088, k 1 or 3, act 0..2");
  DevScope ds(t->device);
  if (!ds.ok) TFAIL(DCTTS_ERR_HIP, "hipSetDevice failed");
  hipStream_t st = (hipStream_t)stream;
  const ConvGeom g(B, T, Cin, Cout, k, rate, causal);
  const float *kp = nullptr, *hp = nullptr;
  int rc = conv_prenorm(t, st, g, x, kernel, &kp, &hp, false);
  if (rc) return rc;
  const int nblk = (int)std::min<long>(256, ((long)B * T + 3) / 4);
  if (reserve(&t->part, (size_t)nblk * 6 * std::max(Cout, 256) * 4)) return DCTTS_ERR_HIP;
  CBwdRowsParams q{B, T, g.Tp, Cout, g.pr, hp, dy, bias, gamma, beta, act, (float*)t->dHp.p, (float*)t->part.p};
  if ((rc = launch_c_rows(t, st, q, g.Chp, nblk)) != 0) return rc;
  hipLaunchKernelGGL(colsum3_kernel, dim3((3 * Cout + 255) / 256), dim3(256), 0, st, (const float*)t->part.p, nblk, Cout, dgamma, dbeta, dbias);
  THIP(hipGetLastError());
  return conv_grads(t, st, g, kp, dkernel, dx);
}

// modules.py:199-247 backward.  out[2t] = b + x[t] W0^T + x[t-1] W2^T, out[2t+1] = b + x[t] W1^T (W_j = kernel[0][j], Cout x Cin), then layer-norm.
// x rows get ONE zero row in front of every utterance (flat row r = b (T + 1) + 1 + t); the pre-norm / gradient rows live in a buffer
// with 2 (T + 1) rows per utterance so that output row 2t + p of an utterance is flat row 2 r + p: every operand of the six GEMMs is a
// strided view (leading dimension 2 C selects the even or the odd rows).
extern "C" int dctts_train_conv1d_transpose_backward(dctts_train* t, const float* x, const float* dy, const float* kernel, const float* bias,
                                                     const float* gamma, const float* beta, int B, int T, int Cin, int Cout,
                                                     float* dx, float* dkernel, float* dbias, float* dgamma, float* dbeta, void* stream) {
  if (!t || !x || !dy || !kernel || !bias || !gamma || !beta || !dx || !dkernel || !dbias || !dgamma || !dbeta) TFAIL(DCTTS_ERR_ARG, "conv1d_transpose_backward: null argument");
  if (B <= 0 || T <= 0 || Cin <= 0 || (Cin & 3) || Cout <= 0 || (Cout & 3) || Cout > 1088) TFAIL(DCTTS_ERR_ARG, "conv1d_transpose_backward: Cin, Cout multiples of 4, Cout <= 1088");
  DevScope ds(t->device);
  if (!ds.ok) TFAIL(DCTTS_ERR_HIP, "hipSetDevice failed");
  hipStream_t st = (hipStream_t)stream;
  const long R = (long)B * (T + 1);
  const int wt = ((Cin + 127) / 128) * ((Cout + 127) / 128), splits = std::max(1, std::min(64, (512 + wt - 1) / wt));
  if (reserve(&t->xp, (size_t)R * Cin * 4) || reserve(&t->Hp, (size_t)2 * (R + 1) * Cout * 4) || reserve(&t->dHp, (size_t)2 * (R + 1) * Cout * 4) ||
      reserve(&t->dxp, (size_t)R * Cin * 4) || reserve(&t->wpart, (size_t)splits * Cin * Cout * 4)) return DCTTS_ERR_HIP;
  float *xp = (float*)t->xp.p, *Hp = (float*)t->Hp.p, *dHp = (float*)t->dHp.p, *dxp = (float*)t->dxp.p;
  const long n4 = R * (Cin / 4);
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, xp, const_cast<float*>(x), B, T, T + 1, 1, Cin, 0);
  THIP(hipGetLastError());
  THIP(hipMemsetAsync(dHp, 0, (size_t)2 * (R + 1) * Cout * 4, st));
  THIP(hipMemsetAsync(dxp, 0, (size_t)R * Cin * 4, st));
  const long wsz = (long)Cout * Cin;
  const float *W0 = kernel, *W1 = kernel + wsz, *W2 = kernel + 2 * wsz;
  const int M = (int)(R - 1);                       // flat rows r = 1 .. R - 1
  int rc;
  // pre-norm rows (without bias): even rows 2r, odd rows 2r + 1
  if ((rc = gemm<false, true>(st, xp + Cin, W0, Hp + 2L * Cout, M, Cout, Cin, Cin, Cin, 2 * Cout, 0)) < 0) return rc;
  if ((rc = gemm<false, true>(st, xp, W2, Hp + 2L * Cout, M, Cout, Cin, Cin, Cin, 2 * Cout, 1)) < 0) return rc;
  if ((rc = gemm<false, true>(st, xp + Cin, W1, Hp + 3L * Cout, M, Cout, Cin, Cin, Cin, 2 * Cout, 0)) < 0) return rc;
  // layer-norm backward over the 2T output rows of every utterance (output row u of utterance b is flat row 2 b (T + 1) + 2 + u)
  const int nblk = (int)std::min<long>(256, ((long)B * 2 * T + 3) / 4);
  if (reserve(&t->part, (size_t)nblk * 6 * std::max(Cout, 256) * 4)) return DCTTS_ERR_HIP;
  CBwdRowsParams q{B, 2 * T, 2 * (T + 1), Cout, 2, Hp, dy, bias, gamma, beta, 0, dHp, (float*)t->part.p};
  if ((rc = launch_c_rows(t, st, q, Cout, nblk)) != 0) return rc;
  hipLaunchKernelGGL(colsum3_kernel, dim3((3 * Cout + 255) / 256), dim3(256), 0, st, (const float*)t->part.p, nblk, Cout, dgamma, dbeta, dbias);
  THIP(hipGetLastError());
  // dW_j (Cout x Cin) = dH_rows^T . x_rows;  dx[r] = dH[2r] W0 + dH[2r+1] W1 + dH[2r+2] W2
  const float* Aw[3] = {dHp + 2L * Cout, dHp + 3L * Cout, dHp + 2L * Cout};
  const float* Bw[3] = {xp + Cin, xp + Cin, xp};
  for (int j = 0; j < 3; ++j) {
    const int nz = gemm<true, false>(st, Aw[j], Bw[j], (float*)t->wpart.p, Cout, Cin, M, 2 * Cout, Cin, Cin, 0, splits, wsz);
    if (nz < 0) return nz;
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((wsz + 255) / 256)), dim3(256), 0, st, (const float*)t->wpart.p, nz, wsz, wsz, dkernel + (long)j * wsz);
    THIP(hipGetLastError());
  }
  if ((rc = gemm<false, false>(st, dHp + 2L * Cout, W0, dxp + Cin, M, Cin, Cout, 2 * Cout, Cin, Cin, 0)) < 0) return rc;
  if ((rc = gemm<false, false>(st, dHp + 3L * Cout, W1, dxp + Cin, M, Cin, Cout, 2 * Cout, Cin, Cin, 1)) < 0) return rc;
  if ((rc = gemm<false, false>(st, dHp + 4L * Cout, W2, dxp + Cin, M, Cin, Cout, 2 * Cout, Cin, Cin, 1)) < 0) return rc;
  const long m4 = (long)B * T * (Cin / 4);
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((m4 + 255) / 256)), dim3(256), 0, st, dxp, dx, B, T, T + 1, 1, Cin, 1);
  THIP(hipGetLastError());
  return 0;
}

// ---- forward passes on the TF-layout variables a trainer holds (the inference context reads MFMA-packed copies made once at upload)
extern "C" int dctts_train_hc_forward(dctts_train* t, const float* x, const float* kernel, const float* bias, const float* g1, const float* b1,
                                      const float* g2, const float* b2, int B, int T, int C, int k, int rate, int causal, float* y, void* stream) {
  if (!t || !x || !kernel || !bias || !g1 || !b1 || !g2 || !b2 || !y) TFAIL(DCTTS_ERR_ARG, "hc_forward: null argument");
  if (B <= 0 || T <= 0 || (C != 256 && C != 512 && C != 1024) || (k != 1 && k != 3) || rate < 1) TFAIL(DCTTS_ERR_ARG, "hc_forward: C must be 256, 512 or 1024, k 1 or 3, rate >= 1");
  DevScope ds(t->device);
  if (!ds.ok) TFAIL(DCTTS_ERR_HIP, "hipSetDevice failed");
  hipStream_t st = (hipStream_t)stream;
  const ConvGeom g(B, T, C, 2 * C, k, rate, causal);
  const float *kp = nullptr, *hp = nullptr;
  int rc = conv_prenorm(t, st, g, x, kernel, &kp, &hp, true);
  if (rc) return rc;
  HcFwdRowsParams q{B, T, g.Tp, C, g.pr, hp, x, bias, g1, b1, g2, b2, y};
  const unsigned nb = (unsigned)(((long)B * T + 3) / 4);
  if (C == 256) hipLaunchKernelGGL((hc_fwd_rows_kernel<1>), dim3(nb), dim3(256), 0, st, q);
  else if (C == 512) hipLaunchKernelGGL((hc_fwd_rows_kernel<2>), dim3(nb), dim3(256), 0, st, q);
  else hipLaunchKernelGGL((hc_fwd_rows_kernel<4>), dim3(nb), dim3(256), 0, st, q);
  THIP(hipGetLastError());
  return 0;
}

extern "C" int dctts_train_conv1d_forward(dctts_train* t, const float* x, const float* kernel, const float* bias, const float* gamma, const float* beta,
                                          int B, int T, int Cin, int Cout, int k, int rate, int causal, int act, float* y, void* stream) {
  if (!t || !x || !kernel || !bias || !gamma || !beta || !y) TFAIL(DCTTS_ERR_ARG, "conv1d_forward: null argument");
  if (B <= 0 || T <= 0 || Cin <= 0 || Cin > 4096 || Cout <= 0 || Cout > 1088 || (k != 1 && k != 3) || rate < 1 || act < 0 || act > 2) TFAIL(DCTTS_ERR_ARG, "conv1d_forward: 1 <= Cout <= 1088, k 1 or 3, act 0..2");
  DevScope ds(t->device);
  if (!ds.ok) TFAIL(DCTTS_ERR_HIP, "hipSetDevice failed");
  hipStream_t st = (hipStream_t)stream;
  const ConvGeom g(B, T, Cin, Cout, k, rate, causal);
  const float *kp = nullptr, *hp = nullptr;
  int rc = conv_prenorm(t, st, g, x, kernel, &kp, &hp, true);
  if (rc) return rc;
  CFwdRowsParams q{B, T, g.Tp, Cout, g.Chp, g.pr, hp, bias, gamma, beta, act, y};
  hipLaunchKernelGGL(c_fwd_rows_kernel, dim3((unsigned)(((long)B * T + 3) / 4)), dim3(256), 0, st, q);
  THIP(hipGetLastError());
  return 0;
}

extern "C" int dctts_train_conv1d_transpose_forward(dctts_train* t, const float* x, const float* kernel, const float* bias, const float* gamma, const float* beta,
                                                    int B, int T, int Cin, int Cout, float* y, void* stream) {
  if (!t || !x || !kernel || !bias || !gamma || !beta || !y) TFAIL(DCTTS_ERR_ARG, "conv1d_transpose_forward: null argument");
  if (B <= 0 || T <= 0 || Cin <= 0 || (Cin & 3) || Cout <= 0 || (Cout & 3) || Cout > 1088) TFAIL(DCTTS_ERR_ARG, "conv1d_transpose_forward: Cin, Cout multiples of 4, Cout <= 1088");
  DevScope ds(t->device);
  if (!ds.ok) TFAIL(DCTTS_ERR_HIP, "hipSetDevice failed");
  hipStream_t st = (hipStream_t)stream;
  const long R = (long)B * (T + 1);
  if (reserve(&t->xp, (size_t)R * Cin * 4) || reserve(&t->Hp, (size_t)2 * (R + 1) * Cout * 4)) return DCTTS_ERR_HIP;
  float *xp = (float*)t->xp.p, *Hp = (float*)t->Hp.p;
  const long n4 = R * (Cin / 4);
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, xp, const_cast<float*>(x), B, T, T + 1, 1, Cin, 0);
  THIP(hipGetLastError());
  const long wsz = (long)Cout * Cin;
  const int M = (int)(R - 1);
  int rc;
  if ((rc = gemm<false, true>(st, xp + Cin, kernel, Hp + 2L * Cout, M, Cout, Cin, Cin, Cin, 2 * Cout, 0)) < 0) return rc;
  if ((rc = gemm<false, true>(st, xp, kernel + 2 * wsz, Hp + 2L * Cout, M, Cout, Cin, Cin, Cin, 2 * Cout, 1)) < 0) return rc;
  if ((rc = gemm<false, true>(st, xp + Cin, kernel + wsz, Hp + 3L * Cout, M, Cout, Cin, Cin, Cin, 2 * Cout, 0)) < 0) return rc;
  CFwdRowsParams q{B, 2 * T, 2 * (T + 1), Cout, Cout, 2, Hp, bias, gamma, beta, 0, y};
  hipLaunchKernelGGL(c_fwd_rows_kernel, dim3((unsigned)(((long)B * 2 * T + 3) / 4)), dim3(256), 0, st, q);
  THIP(hipGetLastError());
  return 0;
}

extern "C" int dctts_train_embed_forward(dctts_train* t, const int32_t* ids, const float* table, long long n, int vocab, int e, float* y, void* stream) {
  if (!t || !ids || !table || !y || n <= 0 || vocab <= 0 || e <= 0) TFAIL(DCTTS_ERR_ARG, "embed_forward: bad argument");
  DevScope ds(t->device);
  if (!ds.ok) TFAIL(DCTTS_ERR_HIP, "hipSetDevice failed");
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)((n * e + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const int*)ids, table, (long)n, vocab, e, y);
  THIP(hipGetLastError());
  return 0;
}

extern "C" int dctts_train_dropout(dctts_train* t, const float* x, float* y, long long n, uint64_t key, float rate, void* stream) {
  if (!t || !x || !y || n <= 0 || rate < 0.f || rate >= 1.f) TFAIL(DCTTS_ERR_ARG, "dropout: bad argument");
  DevScope ds(t->device);
  if (!ds.ok) TFAIL(DCTTS_ERR_HIP, "hipSetDevice failed");
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n, (unsigned long long)key, rate, 1.0f / (1.0f - rate));
  THIP(hipGetLastError());
  return 0;
}

extern "C" int dctts_train_sigmoid(dctts_train* t, const float* x, float* y, long long n, void* stream) {
  if (!t || !x || !y || n <= 0) TFAIL(DCTTS_ERR_ARG, "sigmoid: bad argument");
  DevScope ds(t->device);
  if (!ds.ok) TFAIL(DCTTS_ERR_HIP, "hipSetDevice failed");
  hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n);
  THIP(hipGetLastError());
  return 0;
}

// networks.py:126-155 forward, training form: R = [softmax(Q K^T / sqrt(d)) V ; Q] (B, T, 2d), alignments (B, N, T)
extern "C" int dctts_train_attention_forward(dctts_train* t, const float* Q, const float* K, const float* V, int B, int T, int N, int d,
                                             float* R, float* alignments, void* stream) {
  if (!t || !Q || !K || !V || !R || !alignments) TFAIL(DCTTS_ERR_ARG, "attention_forward: null argument");
  if (B <= 0 || T <= 0 || N <= 0 || d <= 0 || (d & 3)) TFAIL(DCTTS_ERR_ARG, "attention_forward: d must be a multiple of 4");
  DevScope ds(t->device);
  if (!ds.ok) TFAIL(DCTTS_ERR_HIP, "hipSetDevice failed");
  hipStream_t st = (hipStream_t)stream;
  const long rows = (long)B * T;
  const int Np = (N + 3) & ~3;             // leading dimension of the attention matrix (the GEMMs load 16 bytes at a time): any N, softmax over the true N
  if (reserve(&t->att, (size_t)2 * rows * Np * 4)) return DCTTS_ERR_HIP;
  float* A = (float*)t->att.p;
  int rc;
  if ((rc = gemm_batched<false, true>(st, B, Q, (long)T * d, K, (long)N * d, A, (long)T * Np, T, N, d, d, d, Np, 0))) return rc;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, A, rows, N, Np, 1.0f / std::sqrt((float)d));
  THIP(hipGetLastError());
  if ((rc = gemm_batched<false, false>(st, B, A, (long)T * Np, V, (long)N * d, R, (long)T * 2 * d, T, d, N, Np, d, 2 * d, 0))) return rc;
  hipLaunchKernelGGL(scatter_cols_kernel, dim3((unsigned)((rows * d + 255) / 256)), dim3(256), 0, st, Q, R + d, 2 * d, rows, d);
  THIP(hipGetLastError());
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((rows * N + 255) / 256)), dim3(256), 0, st, (const float*)A, alignments, B, T, N, Np);
  THIP(hipGetLastError());
  return 0;
}

// networks.py:126-155 backward, training form (no monotonic mask): A = softmax(Q K^T / sqrt(d)), R = [A V ; Q], alignments = A^T.
extern "C" int dctts_train_attention_backward(dctts_train* t, const float* Q, const float* K, const float* V, const float* dR, const float* dAl,
                                              int B, int T, int N, int d, float* dQ, float* dK, float* dV, void* stream) {
  if (!t || !Q || !K || !V || !dR || !dAl || !dQ || !dK || !dV) TFAIL(DCTTS_ERR_ARG, "attention_backward: null argument");
  if (B <= 0 || T <= 0 || N <= 0 || d <= 0 || (d & 3)) TFAIL(DCTTS_ERR_ARG, "attention_backward: d must be a multiple of 4");
  DevScope ds(t->device);
  if (!ds.ok) TFAIL(DCTTS_ERR_HIP, "hipSetDevice failed");
  hipStream_t st = (hipStream_t)stream;
  const long rows = (long)B * T;
  const int Np = (N + 3) & ~3;             // (as in the forward pass)
  if (reserve(&t->att, (size_t)2 * rows * Np * 4)) return DCTTS_ERR_HIP;
  float* A = (float*)t->att.p; float* dA = A + rows * Np;
  const float scale = 1.0f / std::sqrt((float)d);
  int rc;
  // A = softmax(Q K^T * scale)
  if ((rc = gemm_batched<false, true>(st, B, Q, (long)T * d, K, (long)N * d, A, (long)T * Np, T, N, d, d, d, Np, 0))) return rc;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, A, rows, N, Np, scale);
  THIP(hipGetLastError());
  // dA = dR[:, :, :d] V^T (+ dAl^T in the row kernel);  dV = A^T dR[:, :, :d]
  if ((rc = gemm_batched<false, true>(st, B, dR, (long)T * 2 * d, V, (long)N * d, dA, (long)T * Np, T, N, d, 2 * d, d, Np, 0))) return rc;
  if ((rc = gemm_batched<true, false>(st, B, A, (long)T * Np, dR, (long)T * 2 * d, dV, (long)N * d, N, d, T, Np, 2 * d, d, 0))) return rc;
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const float*)A, dA, dAl, B, T, N, Np, scale);
  THIP(hipGetLastError());
  // dQ = dS K + dR[:, :, d:];  dK = dS^T Q      (the 1 / sqrt(d) is folded into dS)
  hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)((rows * d + 255) / 256)), dim3(256), 0, st, dR + d, 2 * d, dQ, rows, d);
  THIP(hipGetLastError());
  if ((rc = gemm_batched<false, false>(st, B, dA, (long)T * Np, K, (long)N * d, dQ, (long)T * d, T, d, N, Np, d, d, 1))) return rc;
  if ((rc = gemm_batched<true, false>(st, B, dA, (long)T * Np, Q, (long)T * d, dK, (long)N * d, N, d, T, Np, d, d, 0))) return rc;
  return 0;
}

extern "C" int dctts_train_embed_backward(dctts_train* t, const int32_t* ids, const float* dy, long long n, int vocab, int e, float* dtable, void* stream) {
  if (!t || !ids || !dy || !dtable || n <= 0 || vocab <= 0 || e <= 0) TFAIL(DCTTS_ERR_ARG, "embed_backward: bad argument");
  DevScope ds(t->device);
  if (!ds.ok) TFAIL(DCTTS_ERR_HIP, "hipSetDevice failed");
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(vocab), dim3(256), 0, (hipStream_t)stream, (const int*)ids, dy, (long)n, e, dtable);
  THIP(hipGetLastError());
  return 0;
}

static int loss_blocks(long n) { return (int)std::min<long>(1024, (n + 2047) / 2048); }

extern "C" int dctts_train_ssrn_losses(dctts_train* t, const float* Z, const float* Z_logits, const float* mags, long long n,
                                       float* losses, float* dZ, float* dlogits, void* stream) {
  if (!t || !Z || !Z_logits || !mags || !losses || !dZ || !dlogits || n <= 0) TFAIL(DCTTS_ERR_ARG, "losses: bad argument");
  DevScope ds(t->device);
  if (!ds.ok) TFAIL(DCTTS_ERR_HIP, "hipSetDevice failed");
  hipStream_t st = (hipStream_t)stream;
  const int nb = loss_blocks(n);
  if (reserve(&t->lpart, (size_t)3 * 1024 * 4)) return DCTTS_ERR_HIP;
  hipLaunchKernelGGL(l1_bd_loss_kernel, dim3(nb), dim3(256), 0, st, Z, Z_logits, mags, (long)n, dZ, dlogits, (float*)t->lpart.p);
  THIP(hipGetLastError());
  hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(256), 0, st, (const float*)t->lpart.p, nb, 2, 2, 1.0f / (float)n, losses);
  THIP(hipGetLastError());
  return 0;
}

extern "C" int dctts_train_text2mel_losses(dctts_train* t, const float* Y, const float* Y_logits, const float* mels, const float* alignments,
                                           int B, int T, int n_mels, int N, int max_N, int max_T,
                                           float* losses, float* dY, float* dlogits, float* dA, void* stream) {
  if (!t || !alignments || !dA || !losses || B <= 0 || T <= 0 || n_mels <= 0 || N <= 0 || max_N <= 0 || max_T <= 0) TFAIL(DCTTS_ERR_ARG, "text2mel losses: bad argument");
  int rc = dctts_train_ssrn_losses(t, Y, Y_logits, mels, (long long)B * T * n_mels, losses, dY, dlogits, stream);
  if (rc) return rc;
  DevScope ds(t->device);
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)B * N * T;
  const int nb = loss_blocks(n);
  const double mask_sum = (double)B * std::min(N, max_N) * std::min(T, max_T);      // train.py:94,96: entries of the cropped A that are not padding
  float* part = (float*)t->lpart.p + 2048;
  hipLaunchKernelGGL(att_loss_kernel, dim3(nb), dim3(256), 0, st, alignments, B, N, T, max_N, max_T, (float)(1.0 / mask_sum), dA, part);
  THIP(hipGetLastError());
  hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(256), 0, st, (const float*)part, nb, 1, 1, (float)(1.0 / mask_sum), losses + 2);
  THIP(hipGetLastError());
  return 0;
}

extern "C" int dctts_train_adam_step(dctts_train* t, float* var, const float* grad, float* m, float* v, long long n, int step, float lr, void* stream) {
  if (!t || !var || !grad || !m || !v || n <= 0 || step < 1) TFAIL(DCTTS_ERR_ARG, "adam: bad argument");
  DevScope ds(t->device);
  if (!ds.ok) TFAIL(DCTTS_ERR_HIP, "hipSetDevice failed");
  const double lr_t = (double)lr * std::sqrt(1.0 - std::pow(0.999, step)) / (1.0 - std::pow(0.9, step));
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, var, grad, m, v, (long)n, (float)lr_t);
  THIP(hipGetLastError());
  return 0;
}

extern "C" int dctts_train_adam_step_multi(dctts_train* t, int count, float* const* vars, const float* const* grads, float* const* ms, float* const* vs,
                                           const long long* ns, int step, float lr, void* stream) {
  if (!t || count <= 0 || !vars || !grads || !ms || !vs || !ns || step < 1) TFAIL(DCTTS_ERR_ARG, "adam (multi): bad argument");
  DevScope ds(t->device);
  if (!ds.ok) TFAIL(DCTTS_ERR_HIP, "hipSetDevice failed");
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < count; ++i)
    if (!vars[i] || !grads[i] || !ms[i] || !vs[i] || ns[i] <= 0) TFAIL(DCTTS_ERR_ARG, "adam (multi): null tensor");
  const double lr_t = (double)lr * std::sqrt(1.0 - std::pow(0.999, step)) / (1.0 - std::pow(0.9, step));
  for (int i0 = 0; i0 < count; i0 += 64) {
    AdamBatch b;
    const int nb = std::min(64, count - i0);
    long nmax = 0;
    for (int i = 0; i < nb; ++i) {
      b.it[i] = AdamItem{vars[i0 + i], grads[i0 + i], ms[i0 + i], vs[i0 + i], (long)ns[i0 + i]};
      nmax = std::max(nmax, (long)ns[i0 + i]);
    }
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)((nmax + 16383) / 16384), (unsigned)nb), dim3(256), 0, st, b, (float)lr_t);
  }
  THIP(hipGetLastError());
  return 0;
}
