// decode3_kernels.h -- row kernels of the round-2 decode ("v3") on gfx950.
//
// Two algebraic moves take work off the latency-critical chain of a decode step (synthesize.py:47-54) without changing what is
// computed (tests/algo_model.incremental_decode_v3 states the data flow in numpy; test_v3_model_equals_reference_loop proves it
// against the restated reference loop in fp64):
//
//  1. AudioDec C_1 (networks.py:167-174) is a k = 1 conv on R = [A.V ; Q] (networks.py:150-151).  With its kernel split by rows,
//     W1 = [W_top ; W_bot],
//         C_1pre[t] = bias + sum_k a_k(t) * VW[p + k] + C1Q[t],       VW[n]  = V[n] . W_top   (once per batch, after TextEnc)
//                                                                     C1Q[t] = Q[t] . W_bot   (once per frame: window-independent)
//     where a_k(t) is the <= 3-key windowed softmax of Q[t] . K[p + k] / 16 (networks.py:140-148).  Re-evaluating C_1 over the
//     84 older cone rows with frame f's window is therefore a ROW operation (rowc1_kernel: 3 dot products, a softmax, 3 axpys,
//     one layer-norm per row) instead of attention rows + an (84 B) x 512 x 256 GEMM + a layer-norm pass.
//  2. A causal k = 3 layer's newest row is  presum + x[t] . W[2]  with  presum = bias + x[t-2d] . W[0] + x[t-d] . W[1]
//     (modules.py:173-187 with the taps written out).  The presum only reads rows that are final one frame earlier (AudioEnc) or
//     that the bulk of the same frame produces (AudioDec cone rows < j), so it is computed on the bulk stream
//     (hbulk_group_kernel / the masked last row of hbulk_kernel<12>) and the chain contracts K = 256 instead of 768.
//
// attnq_kernel is the chain's attention for the newest row: it rebuilds Q[j] from AudioEnc's last pre-norm row, picks the next
// window (arg-max of the post-softmax row, first index on ties) and emits C_1's presum for frame j.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "decode_kernels.h"

namespace dctts {

enum { MAXWIN = 3 };      // attention_win_size the decode kernels are unrolled for (dctts_create rejects anything larger)

// <= MAXWIN windowed attention weights of one query row held as 4 channels per lane (d == 256).  Returns nk; a[k] = 0 for k >= nk.
__device__ __forceinline__ int window_softmax(const float4 q, const float* Krow0, int kv_stride, int c0, int pm, int N, int win, float scale,
                                              float (&a)[MAXWIN], int& am) {
  int nk = N - pm; if (nk > win) nk = win;            // allowed keys pm .. pm + nk - 1 (nk >= 1: pm <= N - 1)
  float lg[MAXWIN];
#pragma unroll
  for (int k = 0; k < MAXWIN; ++k) {
    lg[k] = -INFINITY;
    if (k < nk) {
      const float4 kk = ld4(Krow0 + (long)k * kv_stride + c0);
      float s = q.x * kk.x; s = fmaf(q.y, kk.y, s); s = fmaf(q.z, kk.z, s); s = fmaf(q.w, kk.w, s);
      lg[k] = wave_sum(s) * scale;
    }
  }
  float mx = lg[0];
#pragma unroll
  for (int k = 1; k < MAXWIN; ++k) mx = fmaxf(mx, lg[k]);
  float se = 0.f;
#pragma unroll
  for (int k = 0; k < MAXWIN; ++k) { a[k] = (k < nk) ? expf(lg[k] - mx) : 0.f; se += a[k]; }
  const float inv = 1.0f / se;
  am = 0;
  float best = a[0] * inv; a[0] = best;
#pragma unroll
  for (int k = 1; k < MAXWIN; ++k) { a[k] *= inv; if (a[k] > best) { best = a[k]; am = k; } }   // post-softmax arg-max, first index on ties
  return nk;
}

// ---------------------------------------------------------------------------------------------------------------- bulk: C_1 cone rows
// x1[b][t] = LN(bias + sum_k a_k VW[b][p+k] + C1Q[b][t]) * gamma + beta for the cone rows t = frame + offs[r] (offs < 0).
// grid (ceil(R / 4), B), block 256: one wave per row, lane = 4 channels.
struct RowC1Params {
  int B, R; const int* offs; int frame;
  const float* Qh; long q_bstride; long q_row0; int q_stride;        // AudioEnc history (absolute time), Q = its last layer
  const float* K; int k_stride; const float* VW; int vw_stride; long kv_bstride;   // rows (b * kv_bstride + n)
  const float* C1Q; long c_bstride; long c_row0; int c_stride;
  const float* bias; const float* g; const float* be;
  int N, d, win; const int* pm_all;
  float* x; long x_bstride; long x_row0; int x_stride; long x_set;   // AudioDec C_1 output rows, parity copy frame & 1
};

__global__ void __launch_bounds__(256) rowc1_kernel(const RowC1Params p) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (r >= p.R) return;
  const int t = p.frame + p.offs[r];
  if (t < 0) return;
  const int c0 = lane * 4;
  const int pm = p.pm_all[(long)p.frame * p.B + b];
  const float4 q = ld4(p.Qh + ((long)b * p.q_bstride + p.q_row0 + t) * p.q_stride + c0);
  const float4 cq = ld4(p.C1Q + ((long)b * p.c_bstride + p.c_row0 + t) * p.c_stride + c0);
  const float4 bi = ld4(p.bias + c0), g = ld4(p.g + c0), be = ld4(p.be + c0);
  float a[MAXWIN]; int am;
  const long kv0 = (long)b * p.kv_bstride + pm;
  const int nk = window_softmax(q, p.K + kv0 * p.k_stride, p.k_stride, c0, pm, p.N, p.win, 1.0f / sqrtf((float)p.d), a, am);
  float4 y = make_float4(bi.x + cq.x, bi.y + cq.y, bi.z + cq.z, bi.w + cq.w);
#pragma unroll
  for (int k = 0; k < MAXWIN; ++k) {
    if (k < nk) {
      const float4 v = ld4(p.VW + (kv0 + k) * p.vw_stride + c0);
      y.x = fmaf(a[k], v.x, y.x); y.y = fmaf(a[k], v.y, y.y); y.z = fmaf(a[k], v.z, y.z); y.w = fmaf(a[k], v.w, y.w);
    }
  }
  const float mean = wave_sum(y.x + y.y + y.z + y.w) * (1.0f / 256.0f);
  const float4 dv = make_float4(y.x - mean, y.y - mean, y.z - mean, y.w - mean);
  const float var = wave_sum(dv.x * dv.x + dv.y * dv.y + dv.z * dv.z + dv.w * dv.w) * (1.0f / 256.0f);
  const float rs = 1.0f / sqrtf(var + 1e-12f);
  const float4 o = make_float4(dv.x * rs * g.x + be.x, dv.y * rs * g.y + be.y, dv.z * rs * g.z + be.z, dv.w * rs * g.w + be.w);
  *reinterpret_cast<float4*>(p.x + (long)(p.frame & 1) * p.x_set + ((long)b * p.x_bstride + p.x_row0 + t) * p.x_stride + c0) = o;
}

// ---------------------------------------------------------------------------------------------------------------- chain: attention row j
// Q[j] = gate(LN(AudioEnc last pre-norm row)) -> Q history; window of frame j -> weights a_k, arg-max -> window of frame j+1;
// presum[b][:] = bias + sum_k a_k VW[b][p+k]  (AudioDec C_1's presum for frame j; the chain adds Q[j] . W_bot).
// grid ceil(B / 4), block 256: one wave per utterance.
struct AttnQParams {
  int B; int frame;
  RowNorm nrm;                                                      // Q[j] from P_last[b] (R == 1: prow = b)
  float* qhist; long q_bstride; long q_row0; int q_stride;
  const float* K; int k_stride; const float* VW; int vw_stride; long kv_bstride;
  const float* bias;
  int N, d, win; int* pm_all;
  float* presum;                                                    // [B][d]
};

__global__ void __launch_bounds__(256) attnq_kernel(const AttnQParams p) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= p.B) return;
  const int j = p.frame, c0 = lane * 4;
  const float4 q = norm_row_hc(p.nrm, (long)b, b, j, lane);
  *reinterpret_cast<float4*>(p.qhist + ((long)b * p.q_bstride + p.q_row0 + j) * p.q_stride + c0) = q;
  const int pm = p.pm_all[(long)j * p.B + b];
  float a[MAXWIN]; int am;
  const long kv0 = (long)b * p.kv_bstride + pm;
  const int nk = window_softmax(q, p.K + kv0 * p.k_stride, p.k_stride, c0, pm, p.N, p.win, 1.0f / sqrtf((float)p.d), a, am);
  float4 y = ld4(p.bias + c0);
#pragma unroll
  for (int k = 0; k < MAXWIN; ++k) {
    if (k < nk) {
      const float4 v = ld4(p.VW + (kv0 + k) * p.vw_stride + c0);
      y.x = fmaf(a[k], v.x, y.x); y.y = fmaf(a[k], v.y, y.y); y.z = fmaf(a[k], v.z, y.z); y.w = fmaf(a[k], v.w, y.w);
    }
  }
  *reinterpret_cast<float4*>(p.presum + (long)b * p.d + c0) = y;
  if (lane == 0) p.pm_all[(long)(j + 1) * p.B + b] = pm + am;       // max_attentions[:, j] (synthesize.py:54)
}

}  // namespace dctts
