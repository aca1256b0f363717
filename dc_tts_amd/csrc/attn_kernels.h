// attn_kernels.h -- dot-product attention of the Text2Mel path (networks.py:126-155) for gfx950.
//
// Two kernels:
//   attention_full_kernel    the boundary function Attention(Q,K,V,...) over all (T, N): logits,
//                            monotonic mask (networks.py:142-147), softmax, argmax, [A.V ; Q], alignments.
//   attention_window_kernel  the decode-step form: after masking only keys p <= n < min(p+3, N)
//                            survive (exp(-2^32 - max) underflows to exactly 0 in fp32), so the
//                            softmax is evaluated over <= 3 logits for the rows AudioDec's cone needs.
// Both are tiny next to the conv stacks (0.4 % of the MACs); they are wavefront-shuffle kernels,
// not MFMA ("wavefront shuffles where it is not a real GEMM").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dctts {

// DPP reductions (no LDS crossbar): quad_perm / row_half_mirror / row_mirror leave the sum of each 16-lane row in
// all of its lanes; the four row sums are then combined through v_readlane.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);     // row_half_mirror
  v = dpp_add<0x140>(v);     // row_mirror
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  return (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48)));
}

// grid (T, B), block 256.  Q (B,T,d) / K,V (B,N,d) with arbitrary row strides (floats).
struct AttnFullParams {
  const float* Q; int q_stride; long q_bstride;     // floats per row / rows per batch
  const float* K; int k_stride; long k_bstride;
  const float* V; int v_stride; long v_bstride;
  int T, N, d;
  int monotonic; const int* prev_max; int win;
  float* R;          // (B,T,2d) contiguous, may be null
  float* align;      // (B,N,T) contiguous, may be null
  long long* maxatt; // (B,T) int64, may be null
};

__global__ void __launch_bounds__(256) attention_full_kernel(const AttnFullParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* q = sm;               // d
  float* lg = sm + p.d;        // N
  __shared__ float s_red[8];
  __shared__ int s_arg;
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* qrow = p.Q + ((long)b * p.q_bstride + t) * p.q_stride;
  for (int c = tid; c < p.d; c += 256) q[c] = qrow[c];
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)p.d);       // tf.rsqrt(tf.to_float(hp.d)), networks.py:140
  const int pm = p.monotonic ? p.prev_max[b] : 0;
  for (int n = tid; n < p.N; n += 256) {
    const float* krow = p.K + ((long)b * p.k_bstride + n) * p.k_stride;
    float a = 0.f;
    for (int c = 0; c < p.d; c += 4) {
      const float4 kv = *reinterpret_cast<const float4*>(krow + c);
      a = fmaf(q[c], kv.x, a); a = fmaf(q[c + 1], kv.y, a); a = fmaf(q[c + 2], kv.z, a); a = fmaf(q[c + 3], kv.w, a);
    }
    a *= scale;
    if (p.monotonic) {
      // key_masks: n < p ; reverse_masks: n >= p + win  (sequence_mask(max_N - win - p)[:, ::-1])
      const bool masked = (n < pm) || (n >= pm + p.win);
      if (masked) a = -4294967296.0f;                 // float32(-2**32 + 1)
    }
    lg[n] = a;
  }
  __syncthreads();
  // (value desc, index asc) arg-max reduction over the workgroup; result in s_red[0] / s_arg
  __shared__ float s_mx[4]; __shared__ int s_am[4];
  auto block_argmax = [&](float mx, int am) {
    for (int o = 32; o >= 1; o >>= 1) {
      const float ov = __shfl_xor(mx, o); const int oi = __shfl_xor(am, o);
      if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
    }
    if ((tid & 63) == 0) { s_mx[tid >> 6] = mx; s_am[tid >> 6] = am; }
    __syncthreads();
    if (tid == 0) {
      float m = s_mx[0]; int a = s_am[0];
      for (int w = 1; w < 4; ++w) if (s_mx[w] > m || (s_mx[w] == m && s_am[w] < a)) { m = s_mx[w]; a = s_am[w]; }
      s_red[0] = m; s_arg = a;
    }
    __syncthreads();
  };
  float mx = -INFINITY; int am = 0x7fffffff;
  for (int n = tid; n < p.N; n += 256) { const float v = lg[n]; if (v > mx) { mx = v; am = n; } }
  block_argmax(mx, am);
  mx = s_red[0];
  float se = 0.f;
  for (int n = tid; n < p.N; n += 256) { const float e = expf(lg[n] - mx); lg[n] = e; se += e; }
  se = wave_sum(se);
  __syncthreads();
  if ((tid & 63) == 0) s_red[4 + (tid >> 6)] = se;
  __syncthreads();
  const float inv = 1.0f / (s_red[4] + s_red[5] + s_red[6] + s_red[7]);
  // tf.argmax runs on the POST-softmax row (networks.py:148-149), first index on ties: softmax is only weakly monotone in fp32
  float pmx = -INFINITY; int pam = 0x7fffffff;
  for (int n = tid; n < p.N; n += 256) {
    const float a = lg[n] * inv; lg[n] = a;
    if (a > pmx) { pmx = a; pam = n; }
    if (p.align) p.align[((long)b * p.N + n) * p.T + t] = a;
  }
  __syncthreads();
  block_argmax(pmx, pam);
  if (tid == 0 && p.maxatt) p.maxatt[(long)b * p.T + t] = (long long)s_arg;
  if (!p.R) return;                                  // alignments / arg-max only (the decode's `alignments` output)
  float* rrow = p.R + ((long)b * p.T + t) * (2 * p.d);
  for (int c = tid; c < p.d; c += 256) {
    float acc = 0.f;
    for (int n = 0; n < p.N; ++n) {
      const float a = lg[n];
      if (a != 0.f) acc = fmaf(a, p.V[((long)b * p.v_bstride + n) * p.v_stride + c], acc);
    }
    rrow[c] = acc;
    rrow[p.d + c] = q[c];
  }
}

// Decode-step windowed attention.  grid (ceil(R/4), B), block 256 = 4 waves, one wave per row.
// Row r of the table maps to absolute time t = *step + offs[r]; rows with t < 0 are skipped.
// Writes R rows into the absolute-time buffer rbuf (B, pad+T, 2d) and, for offs[r] == 0,
// the arg-max key into pm_next[b] (= max_attentions[:, j], synthesize.py:54).
struct AttnWinParams {
  const float* Qh; long q_bstride; long q_row0; int q_stride;      // AudioEnc history (absolute time)
  const float* K; const float* V; int kv_stride; long kv_bstride;  // (B,N,*) rows
  int N, d, win;
  const int* step; int step_val; const int* offs; int R;
  const int* pm_all;     // (T+1, B) int32: row j = prev_max_attentions fed at step j
  int B;
  float* rbuf; long r_bstride; long r_row0; long r_set;             // row stride 2d; r_set: parity set stride (0 = single)
};

__global__ void __launch_bounds__(256) attention_window_kernel(const AttnWinParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave, b = blockIdx.y;
  if (r >= p.R) return;
  const int j = p.step_val + (p.step ? *p.step : 0);
  const int t = j + p.offs[r];
  if (t < 0) return;
  const int pm = p.pm_all[(long)j * p.B + b];
  const float* qrow = p.Qh + ((long)b * p.q_bstride + p.q_row0 + t) * p.q_stride;
  const int c0 = lane * 4;                       // d == 256: one float4 per lane
  const float4 q = *reinterpret_cast<const float4*>(qrow + c0);
  const float scale = 1.0f / sqrtf((float)p.d);
  float lg[3]; float4 vv[3];
  int nk = p.N - pm; if (nk > p.win) nk = p.win;  // allowed keys pm .. pm+nk-1  (nk >= 1 since pm <= N-1)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    lg[k] = -INFINITY; vv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < nk) {
      const long row = (long)b * p.kv_bstride + pm + k;
      const float4 kk = *reinterpret_cast<const float4*>(p.K + row * p.kv_stride + c0);
      vv[k] = *reinterpret_cast<const float4*>(p.V + row * p.kv_stride + c0);
      float a = q.x * kk.x; a = fmaf(q.y, kk.y, a); a = fmaf(q.z, kk.z, a); a = fmaf(q.w, kk.w, a);
      lg[k] = wave_sum(a) * scale;
    }
  }
  float mx = lg[0];
#pragma unroll
  for (int k = 1; k < 3; ++k) mx = fmaxf(mx, lg[k]);
  float e[3], se = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) { e[k] = (k < nk) ? expf(lg[k] - mx) : 0.f; se += e[k]; }
  const float inv = 1.0f / se;
  // tf.argmax of the POST-softmax row, first index on ties (networks.py:148-149)
  int am = 0; float pbest = e[0] * inv;
#pragma unroll
  for (int k = 1; k < 3; ++k) { const float a = e[k] * inv; if (a > pbest) { pbest = a; am = k; } }
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float a = e[k] * inv;
    o.x = fmaf(a, vv[k].x, o.x); o.y = fmaf(a, vv[k].y, o.y); o.z = fmaf(a, vv[k].z, o.z); o.w = fmaf(a, vv[k].w, o.w);
  }
  float* rrow = p.rbuf + (long)(j & 1) * p.r_set + ((long)b * p.r_bstride + p.r_row0 + t) * (2 * p.d);
  *reinterpret_cast<float4*>(rrow + c0) = o;
  *reinterpret_cast<float4*>(rrow + p.d + c0) = q;
  if (p.offs[r] == 0 && lane == 0) {
    const_cast<int*>(p.pm_all)[(long)(j + 1) * p.B + b] = pm + am;
  }
}

__global__ void step_inc_kernel(int* step) { if (threadIdx.x == 0 && blockIdx.x == 0) *step += 1; }

// pm_all rows 1..T (T+1,B) int32 -> max_attentions (B,T) int64
__global__ void traj_to_i64_kernel(const int* pm_all, long long* out, int B, int T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * T) { const int b = i / T, t = i - b * T; out[i] = (long long)pm_all[(long)(t + 1) * B + b]; }
}


// The LAST kernel of every decode.  xerr / werr: this decode's error words (team kernels; in-kernel waits), inject: the debug hook's bits.
// dstat: the context's sticky status block -- [0] OR of error bits over every decode since the last report, [1] failed decodes, [2] whether THIS
// decode failed (rewritten by every decode; poison_if_failed_kernel reads it behind the SSRN pass of dctts_synthesize).  A failed decode's outputs
// are overwritten with NaN / -1: whatever consumes them cannot mistake them for results (the reference's sess.run raises instead of returning).
__global__ void __launch_bounds__(256) decode_finish_kernel(const int* xerr, const int* werr, int inject, int* dstat, float* Y, long ny,
                                                            long long* mx, long nmx, float* al, long nal) {
  int e = inject;
  if (xerr) e |= *xerr;
  if (werr && *werr) e |= 32;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    dstat[2] = e != 0;
    if (e) { atomicOr(dstat, e); atomicAdd(dstat + 1, 1); }
  }
  if (e == 0) return;
  const float qnan = __int_as_float(0x7fc00000);
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x, step = (long)gridDim.x * blockDim.x;
  for (long i = i0; i < ny; i += step) Y[i] = qnan;
  if (mx) for (long i = i0; i < nmx; i += step) mx[i] = -1;
  if (al) for (long i = i0; i < nal; i += step) al[i] = qnan;
}
__global__ void __launch_bounds__(256) poison_if_failed_kernel(const int* dstat, float* Z, long nz) {
  if (dstat[2] == 0) return;
  const float qnan = __int_as_float(0x7fc00000);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nz; i += (long)gridDim.x * blockDim.x) Z[i] = qnan;
}

}  // namespace dctts
