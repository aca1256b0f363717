// decode3_kernels.h -- row kernels of the round-2 decode ("v3") on gfx950.
//
// Two algebraic moves take work off the latency-critical chain of a decode step (synthesize.py:47-54) without changing what is
// computed (oracle/incremental_ref.incremental_decode_v3 states the data flow in numpy; test_v3_model_equals_reference_loop proves it
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

__device__ __forceinline__ f32x4 ldv(const float* base, unsigned off) { return *reinterpret_cast<const f32x4*>(base + off); }

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

// The same on key rows that are already in registers (requested up front with clamped, always-valid addresses: a load behind `k < nk` is waited for
// at the branch's join, one dependent round trip per key).  Same arithmetic in the same order as window_softmax.
__device__ __forceinline__ int window_softmax_regs(const float4 q, const f32x4 (&kr)[MAXWIN], int pm, int N, int win, float scale, float (&a)[MAXWIN], int& am) {
  int nk = N - pm; if (nk > win) nk = win;
  float lg[MAXWIN];
#pragma unroll
  for (int k = 0; k < MAXWIN; ++k) {
    float s_ = q.x * kr[k][0]; s_ = fmaf(q.y, kr[k][1], s_); s_ = fmaf(q.z, kr[k][2], s_); s_ = fmaf(q.w, kr[k][3], s_);
    const float l = wave_sum(s_) * scale;
    lg[k] = (k < nk) ? l : -INFINITY;
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
// grid (ceil(R / ROWC1_NW), B), block ROWC1_NW waves: one wave per row, lane = 4 channels.
struct RowC1Params {
  int B, R; const int* offs; int frame;
  const float* Qh; long q_bstride; long q_row0; int q_stride;        // AudioEnc history (absolute time), Q = its last layer
  const float* K; int k_stride; const float* VW; int vw_stride; long kv_bstride;   // rows (b * kv_bstride + n)
  const float* C1Q; long c_bstride; long c_row0; int c_stride;
  const float* bias; const float* g; const float* be;
  int N, d, win; const int* pm_all;
  float* x; long x_bstride; long x_row0; int x_stride; long x_set;   // AudioDec C_1 output rows, parity copy frame & 1
  float* scal; long s_bstride; long s_row0;                          // per row: (mean, rstd of the pre-norm row, a_0, a_1, a_2, -, -, -) for rowhc2_kernel
  // the launch behind this one (rowhc2_kernel) needs the newest C1Q . W2 row, computed by passengers of the chain's xgroup_kernel launch:
  // ONE thread of this launch polls their counter before it exits (bounded; a time-out raises wait_err), so the stream cannot go on before
  const unsigned* wait; unsigned wait_val; int* wait_err;
  int contig;                        // the offset table is -1, -2, ..., -R (checked on the host): xcone_kernel's folded form computes t = frame - 1 - r instead of loading offs[r]
};

constexpr int ROWC1_NW = 4;              // rows (waves) per workgroup.  Measured with 16: 14.7 instead of 10.4 us per launch -- the launch lives in what the chain's and the
                                         // passengers' big workgroups leave free on the CUs, and small workgroups fit there (same for ROWHC2_NW: 12 rows 12.7 against 10.4 us)
// The three pieces of a C_1 cone row, shared by rowc1_kernel (a launch of its own: one workgroup = four rows of one utterance) and by xcone_kernel's folded form
// (round 5: the row kernels as the first phases of the side stream's ONE launch; a workgroup = eight waves x three rows of one of its team's utterances):
//   rowc1_stage   what the rows of an utterance share -- the window's K and V.W rows, C_1's bias and layer-norm parameters (9 KB) -- global memory -> LDS
//                 (global_load_lds_dwordx4, no register in between; the caller waits for vmcnt(0) and synchronises)
//   rowc1_finish  one row from its Q and C1Q operands and the staged pieces: 3 dot products, softmax, 3 axpys, layer-norm; stores the row and its scalars
__device__ __forceinline__ void rowc1_stage(const RowC1Params& p, const int b, const int pm, f32x4* s_sh, const int tid, const int nthreads) {
  for (int it = tid; it < 576; it += nthreads) {                        // (wave-uniform trip count: 576 sixteen-byte pieces = 9 waves' worth, nthreads a multiple of 64)
    const int piece = it >> 6;
    int n = pm + (piece % 3); if (n > p.N - 1) n = p.N - 1;             // keys beyond the window: clamped into the utterance, weight exactly 0
    const float* src = piece < 3 ? p.K + ((long)b * p.kv_bstride + n) * p.k_stride
                     : piece < 6 ? p.VW + ((long)b * p.kv_bstride + n) * p.vw_stride
                     : piece == 6 ? p.bias : (piece == 7 ? p.g : p.be);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (it & 63) * 4), (__attribute__((address_space(3))) void*)&s_sh[it & ~63], 16, 0, 0);
  }
}
__device__ __forceinline__ void rowc1_finish(const RowC1Params& p, const int b, const int t, const int pm, const f32x4 vq, const f32x4 vcq, const f32x4* s_sh, const int lane) {
  auto f4 = [](const f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); };
  const int c0 = lane * 4;
  const f32x4 vbi = s_sh[6 * 64 + lane], vg = s_sh[7 * 64 + lane], vbe = s_sh[8 * 64 + lane];
  f32x4 kr[MAXWIN], vr[MAXWIN];
#pragma unroll
  for (int k = 0; k < MAXWIN; ++k) { kr[k] = s_sh[k * 64 + lane]; vr[k] = s_sh[(3 + k) * 64 + lane]; }
  const float4 q = f4(vq), cq = f4(vcq), bi = f4(vbi), g = f4(vg), be = f4(vbe);
  float a[MAXWIN]; int am;
  const int nk = window_softmax_regs(q, kr, pm, p.N, p.win, 1.0f / sqrtf((float)p.d), a, am);
  (void)nk;
  float4 y = make_float4(bi.x + cq.x, bi.y + cq.y, bi.z + cq.z, bi.w + cq.w);
#pragma unroll
  for (int k = 0; k < MAXWIN; ++k) {                                  // a[k] == 0 beyond the window
    y.x = fmaf(a[k], vr[k][0], y.x); y.y = fmaf(a[k], vr[k][1], y.y); y.z = fmaf(a[k], vr[k][2], y.z); y.w = fmaf(a[k], vr[k][3], y.w);
  }
  const float mean = wave_sum(y.x + y.y + y.z + y.w) * (1.0f / 256.0f);
  const float4 dv = make_float4(y.x - mean, y.y - mean, y.z - mean, y.w - mean);
  const float var = wave_sum(dv.x * dv.x + dv.y * dv.y + dv.z * dv.z + dv.w * dv.w) * (1.0f / 256.0f);
  const float rs = 1.0f / sqrtf(var + 1e-12f);
  const float4 o = make_float4(dv.x * rs * g.x + be.x, dv.y * rs * g.y + be.y, dv.z * rs * g.z + be.z, dv.w * rs * g.w + be.w);
  *reinterpret_cast<float4*>(p.x + (long)(p.frame & 1) * p.x_set + ((long)b * p.x_bstride + p.x_row0 + t) * p.x_stride + c0) = o;
  if (p.scal && lane == 0) {
    float* sc = p.scal + ((long)b * p.s_bstride + p.s_row0 + t) * 8;
    *reinterpret_cast<float4*>(sc) = make_float4(mean, rs, a[0], a[1]);
    sc[4] = a[2];
  }
}
__global__ void __launch_bounds__(ROWC1_NW * 64) rowc1_kernel(const RowC1Params p) {
  // Round 4: what a workgroup's four rows share -- 9 of the 11 KB a row asked for -- goes global memory -> LDS once per workgroup; a row's own requests are its Q and C1Q rows.
  __shared__ f32x4 s_sh[9 * 64];           // [K rows of keys 0..2 | V.W rows of keys 0..2 | bias | gamma | beta][256 floats]
  const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.y;
  const int r0 = __builtin_amdgcn_readfirstlane(blockIdx.x * ROWC1_NW + (tid >> 6));    // wave-uniform
  const int r = r0 < p.R ? r0 : p.R - 1;                                                 // (a wave past the table computes the last row again and stores nothing)
  const int t_ = p.frame + p.offs[r];
  const bool live = r0 < p.R && t_ >= 0;
  const int t = t_ < 0 ? 0 : t_;
  const int c0 = lane * 4;
  const int pm = p.pm_all[(long)p.frame * p.B + b];
  auto ldq = [](const float* q_) { return *reinterpret_cast<const f32x4*>(q_); };
  rowc1_stage(p, b, pm, s_sh, tid, ROWC1_NW * 64);
  f32x4 vq = ldq(p.Qh + ((long)b * p.q_bstride + p.q_row0 + t) * p.q_stride + c0);
  f32x4 vcq = ldq(p.C1Q + ((long)b * p.c_bstride + p.c_row0 + t) * p.c_stride + c0);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(vq), "+v"(vcq) :: "memory");               // the pieces are in LDS (LDS-DMA counts as vector memory), the row's operands have landed
  __syncthreads();
  if (live) rowc1_finish(p, b, t, pm, vq, vcq, s_sh, lane);
  if (p.wait && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    bool ok = false;
    for (int i = 0; i < (1 << 20) && !ok; ++i) {
      ok = __hip_atomic_load(p.wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= p.wait_val;
      if (!ok) __builtin_amdgcn_s_sleep(8);
    }
    if (!ok) atomicOr(p.wait_err, 1);
  }
}

// ---------------------------------------------------------------------------------------------------------------- bulk: HC_2 cone rows
// AudioDec HC_2 (networks.py:175-182, the first highway layer; causal k = 3, dilation 1) over its cone rows, WITHOUT a GEMM.
// Its input x1[t'] = (pre[t'] - m) r gamma1 + beta1 is affine in C_1's pre-norm row pre[t'] = b1 + sum_k a_k VW[p+k] + C1Q[t'], so with
// Wt_q = diag(gamma1) W2[q]:
//     x1[t'] . W2[q] = beta1 . W2[q] + r ( b1 . Wt_q + sum_k a_k (VW[p+k] . Wt_q) + C1Q[t'] . Wt_q - m 1^T Wt_q )
// where VWW[n][q] = VW[n] . Wt_q is computed once per batch and C1QW[t'][q] = C1Q[t'] . Wt_q once per frame (one new row): the
// largest GEMM of the cone (82 rows x 768 x 512 per utterance and frame, 45 % of the bulk FLOPs) becomes ~15 vector FMAs per tap.
// The same wave then finishes the layer: layer-norm of both halves, sigmoid gate, highway mix with x1[t] (modules.py:188-193).
// Row r of the table is a cone row (offset < 0) or, last, the chain's presum row (offset 0: taps -2 and -1 only, stored un-normalised).
// grid (ceil(R / ROWHC2_NW), B), block ROWHC2_NW waves: one wave per row; lane = channels 4 lane .. 4 lane + 3 of H1 and of H2.
struct RowHc2Params {
  int B, R; const int* offs; int frame; int tap_off[3];
  const float* scal; long s_bstride; long s_row0;                    // rowc1_kernel's per-row scalars
  const float* VWW; long kv_bstride;                                 // [(b * kv_bstride + n)][3][512]
  const float* C1QW; long c_bstride; long c_row0;                    // [(b * c_bstride + c_row0 + t)][3][512]
  const float* consts;                                               // [3 taps][3: beta1.W2, b1.Wt, 1^T Wt][512]
  const float* bias; const float* g1; const float* b1; const float* g2; const float* b2;   // HC_2's own bias and H1 / H2 layer-norm parameters
  const float* x1; long x1_bstride; long x1_row0; int x1_stride; long x1_set;   // C_1 output rows (highway residual)
  float* x2; long x2_bstride; long x2_row0; int x2_stride; long x2_set;        // HC_2 output rows
  float* presum; long presum_rstride;                                // presum row of utterance b at presum + b * presum_rstride
  int N, win; const int* pm_all;
  int contig;                                                        // the offset table is -1, ..., -(R - 1), 0 (checked on the host): t of row r without the dependent load of offs[r]
};

// Straight-line on purpose: a load behind a uniform branch is waited for at the join (`s_waitcnt vmcnt(0)`), which made the taps' and keys' loads
// three to four dependent round trips in the branchy form (13.8 us per launch).  Everything a row needs is requested in one batch with clamped,
// always-valid addresses; a tap that does not exist (t' < 0, or the centre tap of the presum row) is dropped by a select, a key beyond the
// window has weight exactly 0 (window_softmax).
// Round 4: the operands that do NOT depend on the row -- the nine constant rows (18 KB) and the window's nine V.W.W rows of the utterance (18 KB) --
// are staged in LDS once per workgroup (its four rows belong to one utterance) instead of being requested by every wave: 17 KB instead of 44 KB of
// requests per row, and 100 instead of 198 registers (two more waves per SIMD).  The launch is latency-bound (one round trip per wave, 2.6 passes at two
// waves per SIMD) and sits on the side stream, which bounds the decode frame since the chain's AudioDec layers became one launch: 11.8 us -> see DESIGN.md 2d.
// The arithmetic (operands, order of the fused multiply-adds) is unchanged.
constexpr int ROWHC2_NW = 4;             // rows (waves) per workgroup: see ROWC1_NW
// The pieces of an HC_2 cone row, shared by rowhc2_kernel and xcone_kernel's folded form (see rowc1_stage): stage the utterance's shared operands, request one row's
// own operands, finish the row.  The arithmetic (operands, order of the fused multiply-adds) is the same wherever a row is computed.
struct RowHc2Row { f32x4 w1[3], w2[3], s4[3], xr; float a2[3]; bool ok[3]; int t; bool live, pre_row; };
struct RowHc2Ln { f32x4 g1, b1, g2, b2, h1, h2; };      // HC_2's layer-norm parameters and its bias (the initial value of a row's two halves): the same for every row
__device__ __forceinline__ void rowhc2_stage(const RowHc2Params& p, const int b, const int pm, f32x4* s_c, f32x4* s_v, const int tid, const int nthreads) {
  // 2 x 1152 sixteen-byte pieces, global memory -> LDS without a register in between (a wave's 64 pieces land at consecutive LDS addresses behind the wave-uniform base)
  for (int it = tid; it < 1152; it += nthreads) {                       // (wave-uniform: 1152 = 18 waves' worth)
    const int piece = it >> 7, k = piece / 3, q = piece - 3 * k;
    int n = pm + k; if (n > p.N - 1) n = p.N - 1;                       // keys clamped into the utterance: their weight is 0
    const float* gv = p.VWW + ((long)b * p.kv_bstride + n) * 1536 + q * 512 + (it & 127) * 4;
    const int wbase = it & ~63;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.consts + it * 4), (__attribute__((address_space(3))) void*)&s_c[wbase], 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gv, (__attribute__((address_space(3))) void*)&s_v[wbase], 16, 0, 0);
  }
}
__device__ __forceinline__ RowHc2Ln rowhc2_ln(const RowHc2Params& p, const int lane) {
  auto ldq = [](const float* q) { return *reinterpret_cast<const f32x4*>(q); };
  const int c0 = lane * 4;
  RowHc2Ln l; l.g1 = ldq(p.g1 + c0); l.b1 = ldq(p.b1 + c0); l.g2 = ldq(p.g2 + c0); l.b2 = ldq(p.b2 + c0);
  l.h1 = ldq(p.bias + c0); l.h2 = ldq(p.bias + 256 + c0);
  return l;
}
__device__ __forceinline__ void rowhc2_load(const RowHc2Params& p, const int b, const int r0, const int lane, RowHc2Row& o) {
  auto ldq = [](const float* q) { return *reinterpret_cast<const f32x4*>(q); };
  const int r = r0 < p.R ? r0 : p.R - 1;                                                 // (a wave past the table computes the last row again and stores nothing)
  o.t = p.contig ? (r == p.R - 1 ? p.frame : p.frame - 1 - r) : p.frame + p.offs[r];
  o.live = r0 < p.R && o.t >= 0;
  const int tq = o.t < 0 ? 0 : o.t;
  o.pre_row = (r == p.R - 1);
  const int c0 = lane * 4;
  const long par = p.frame & 1;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int tp = tq + p.tap_off[q];
    o.ok[q] = tp >= 0 && !(o.pre_row && q == 2);                      // causal zero padding / the chain contracts the centre tap
    const int tc = tp < 0 ? 0 : tp;
    const float* sc = p.scal + ((long)b * p.s_bstride + p.s_row0 + tc) * 8;
    o.s4[q] = ldq(sc); o.a2[q] = sc[4];
    const float* cw = p.C1QW + ((long)b * p.c_bstride + p.c_row0 + tc) * 1536 + q * 512;
    o.w1[q] = ldq(cw + c0); o.w2[q] = ldq(cw + 256 + c0);
  }
  o.xr = ldq(p.x1 + par * p.x1_set + ((long)b * p.x1_bstride + p.x1_row0 + tq) * p.x1_stride + c0);
}
__device__ __forceinline__ void rowhc2_finish(const RowHc2Params& p, const int b, const int pm, const RowHc2Row& o, const RowHc2Ln& ln, const f32x4* s_c, const f32x4* s_v, const int lane) {
  const int c0 = lane * 4;
  const long par = p.frame & 1;
  f32x4 h1 = ln.h1, h2 = ln.h2;
  int nk = p.N - pm; if (nk > p.win) nk = p.win;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const float m = o.s4[q][0], rs = o.s4[q][1];
    const float a[3] = {o.s4[q][2], nk > 1 ? o.s4[q][3] : 0.f, nk > 2 ? o.a2[q] : 0.f};    // a key beyond the window has weight 0
    const f32x4 e1 = s_c[(q * 3 + 0) * 128 + lane], e2 = s_c[(q * 3 + 0) * 128 + 64 + lane];       // beta1 . W2[q]
    const f32x4 u1 = s_c[(q * 3 + 1) * 128 + lane], u2 = s_c[(q * 3 + 1) * 128 + 64 + lane];       // b1 . Wt_q
    const f32x4 cs1 = s_c[(q * 3 + 2) * 128 + lane], cs2 = s_c[(q * 3 + 2) * 128 + 64 + lane];     // 1^T Wt_q
    f32x4 v1[3], v2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { v1[k] = s_v[(k * 3 + q) * 128 + lane]; v2[k] = s_v[(k * 3 + q) * 128 + 64 + lane]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float x1 = fmaf(-m, cs1[i], u1[i]) + o.w1[q][i];
      float x2 = fmaf(-m, cs2[i], u2[i]) + o.w2[q][i];
#pragma unroll
      for (int k = 0; k < 3; ++k) { x1 = fmaf(a[k], v1[k][i], x1); x2 = fmaf(a[k], v2[k][i], x2); }
      const float d1 = fmaf(rs, x1, e1[i]), d2 = fmaf(rs, x2, e2[i]);
      h1[i] += o.ok[q] ? d1 : 0.f; h2[i] += o.ok[q] ? d2 : 0.f;        // a select, not a factor: the operands of a dropped tap may be anything
    }
  }
  if (o.pre_row) {
    float* pr = p.presum + (long)b * p.presum_rstride;
    *reinterpret_cast<f32x4*>(pr + c0) = h1; *reinterpret_cast<f32x4*>(pr + 256 + c0) = h2;
    return;
  }
  auto f4 = [](const f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); };
  const float4 out = norm_hc_vals(f4(h1), f4(h2), f4(o.xr), f4(ln.g1), f4(ln.b1), f4(ln.g2), f4(ln.b2));
  *reinterpret_cast<float4*>(p.x2 + par * p.x2_set + ((long)b * p.x2_bstride + p.x2_row0 + o.t) * p.x2_stride + c0) = out;
}
__global__ void __launch_bounds__(ROWHC2_NW * 64) rowhc2_kernel(const RowHc2Params p) {
  __shared__ f32x4 s_c[9 * 128];          // consts: [tap q][beta1.W2 | b1.Wt | 1^T Wt][512 floats]
  __shared__ f32x4 s_v[9 * 128];          // V.W.W rows of the window: [key k][tap q][512 floats]
  const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.y;
  const int r0 = __builtin_amdgcn_readfirstlane(blockIdx.x * ROWHC2_NW + (tid >> 6));   // wave-uniform: the row's table entries are scalar loads
  const int pm = p.pm_all[(long)p.frame * p.B + b];
  rowhc2_stage(p, b, pm, s_c, s_v, tid, ROWHC2_NW * 64);
  RowHc2Row o;
  rowhc2_load(p, b, r0, lane, o);
  RowHc2Ln ln = rowhc2_ln(p, lane);
  // every request above is out before anything is used (left alone, the scheduler sinks each load into the code that consumes it)
#pragma unroll
  for (int q = 0; q < 3; ++q) asm volatile("; rowhc2: a tap's row operands" : "+v"(o.w1[q]), "+v"(o.w2[q]), "+v"(o.s4[q]), "+v"(o.a2[q]));
  asm volatile("; rowhc2: row operands" : "+v"(ln.h1), "+v"(ln.h2), "+v"(o.xr), "+v"(ln.g1), "+v"(ln.b1), "+v"(ln.g2), "+v"(ln.b2));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the pieces are in LDS (LDS-DMA counts as vector memory)
  __syncthreads();
  if (o.live) rowhc2_finish(p, b, pm, o, ln, s_c, s_v, lane);
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
  const int lane = threadIdx.x & 63;
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));      // wave-uniform: the window position is a scalar load
  if (b >= p.B) return;
  const int j = p.frame, c0 = lane * 4;
  const int pm = p.pm_all[(long)j * p.B + b];
  auto ldq = [](const float* q_) { return *reinterpret_cast<const f32x4*>(q_); };
  auto f4 = [](const f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); };
  // one batch of requests: the window's K / VW rows depend only on the window position (known since the previous frame), not on Q[j]
  const RowNorm& n = p.nrm;
  f32x4 vh1 = ldq(n.P + (long)b * n.np + c0), vh2 = ldq(n.P + (long)b * n.np + 256 + c0);
  f32x4 vxr = ldq(n.res + ((long)b * n.res_bstride + n.res_row0 + j) * n.res_stride + c0);
  f32x4 lg1 = ldq(n.g1 + c0), lb1 = ldq(n.b1 + c0), lg2 = ldq(n.g2 + c0), lb2 = ldq(n.b2 + c0), vbias = ldq(p.bias + c0);
  f32x4 kr[MAXWIN], vr[MAXWIN];
#pragma unroll
  for (int k = 0; k < MAXWIN; ++k) {
    int nn = pm + k; if (nn > p.N - 1) nn = p.N - 1;                  // beyond the window: clamped into the utterance, weight exactly 0
    kr[k] = ldq(p.K + ((long)b * p.kv_bstride + nn) * p.k_stride + c0);
    vr[k] = ldq(p.VW + ((long)b * p.kv_bstride + nn) * p.vw_stride + c0);
  }
  asm volatile("; attnq: all operands requested" : "+v"(vh1), "+v"(vh2), "+v"(vxr), "+v"(lg1), "+v"(lb1), "+v"(lg2), "+v"(lb2), "+v"(vbias),
               "+v"(kr[0]), "+v"(kr[1]), "+v"(kr[2]), "+v"(vr[0]), "+v"(vr[1]), "+v"(vr[2]));
  const float4 q = norm_hc_vals(f4(vh1), f4(vh2), f4(vxr), f4(lg1), f4(lb1), f4(lg2), f4(lb2));
  *reinterpret_cast<float4*>(p.qhist + ((long)b * p.q_bstride + p.q_row0 + j) * p.q_stride + c0) = q;
  float a[MAXWIN]; int am;
  (void)window_softmax_regs(q, kr, pm, p.N, p.win, 1.0f / sqrtf((float)p.d), a, am);
  float4 y = f4(vbias);
#pragma unroll
  for (int k = 0; k < MAXWIN; ++k) {
    y.x = fmaf(a[k], vr[k][0], y.x); y.y = fmaf(a[k], vr[k][1], y.y); y.z = fmaf(a[k], vr[k][2], y.z); y.w = fmaf(a[k], vr[k][3], y.w);
  }
  *reinterpret_cast<float4*>(p.presum + (long)b * p.d + c0) = y;
  if (lane == 0) p.pm_all[(long)(j + 1) * p.B + b] = pm + am;       // max_attentions[:, j] (synthesize.py:54)
}

// ---------------------------------------------------------------------------------------------------------------- chain: one layer, newest row
// chain3_kernel<PRO, HCOUT>: the v3 chain layer over 256 input channels -- out_pre[b][:] = add[b][:] + x[b][:] . W for the newest row of
// every utterance, x rebuilt from the producing layer's pre-norm row (deferred layer-norm, see decode_kernels.h).  Same arithmetic
// as hsplit_kernel<16, ., ., 1> (8 rows x one pair of 16-column tiles per workgroup, K = 256 split over 8 waves, 16x16x4 MFMA,
// fixed-order LDS reduction, partial statistics out), rewritten for the one shape the v3 chain has so that nothing stands between
// kernel entry and the loads: prologue kind and output kind are template parameters (a run-time `if (ln)` became a branch, and the
// compiler parked the first loads' s_waitcnt in front of it: one memory round trip before the other 18 loads went out), every
// address is base + b * stride with the frame offset folded into the base on the host (no offset table, no divisions), the grid is
// (column groups, row tiles), and the parameter block is small enough for one scalar-load batch.
struct Chain3Params {
  int B;
  const float* P; int p_bs;          // PRO_LN_*: pre-norm row of utterance b at P + b * p_bs;  PRO_RAW: the input row itself
  const float* stats;                // PRO_LN_*: [b][16][4] per-16-column-group partial statistics of the P rows
  const float* res; int res_bs;      // PRO_LN_HC: highway residual row at res + b * res_bs
  const float* g1; const float* b1; const float* g2; const float* b2;
  int relu;                          // PRO_LN_C: the producing layer ends in a ReLU
  float* xm; int xm_bs;              // the rebuilt row -> xm + b * xm_bs (written by column group 0); nullptr = not kept
  const float* wp;                   // [tile][16 k-groups][lane][4]
  const float* add; int add_bs;      // epilogue addend per (b, column): presum row at add + b * add_bs, or the bias vector when add_bs == 0
  float* pout; int np_out;           // pre-norm output rows [b][np_out]
  float* stats_out;                  // [b][16][4]
  float* raw; int raw_bs;            // the bare contraction -> raw + b * raw_bs (AudioDec C_1 keeps Q[j] . W_bot); nullptr = none
  int cout;                          // output channels of an LN group (HCOUT: 256 gate + 256 info; else the row width: 256 or 80)
  const float* xt; int xt_bs;        // TAP2: the previous time step's input row (tap -1 of a dilation-1 layer) at xt + b * xt_bs
  long long* ts;                     // TS instantiation only (measurement): 8 wall-clock stamps per workgroup
  unsigned* sig; unsigned sig_val;   // first launch of a chain piece: *sig = sig_val ("every earlier piece is complete": the bulk stream waits for it)
  unsigned wait_val; int* gate_err;  // value wait2 waits for; error word raised when the bounded wait gives up (dctts_decode_status)
  const unsigned* wait2;             // first launch of a chain piece: only the epilogue addend comes from the other
                                     // stream, so every other load is issued first, then *wait2 >= wait_val is polled, then the addend is read past
                                     // the (possibly stale) L2 with an sc1 load: no stream operation and no cache invalidate on the chain's stream
};

// TAP2: a dilation-1 AudioEnc layer.  Its tap -1 reads the row the chain produced one frame earlier, which no presum computed
// ahead of the chain piece can contain, so that tap is contracted here as well (K = 512: k-groups 0..15 = tap -1 from the
// history row, 16..31 = the rebuilt centre row); only the oldest tap arrives through the presum.
template <int PRO, bool HCOUT, bool TAP2 = false, bool TS = false>
__global__ void __launch_bounds__(512) chain3_kernel(const Chain3Params p) {
  long long t_in = 0, t_issued = 0, t_landed = 0, t_mfma = 0, t_sync = 0;
  if constexpr (TS) t_in = wall_clock64();
  constexpr unsigned NKG = TAP2 ? 32u : 16u, KC = TAP2 ? 16u : 0u;        // k-groups per tile; first k-group of the centre tap
  __shared__ __attribute__((aligned(16))) float red[8 * 2 * 4 * 64];       // split-K reduction [wave][tile][j][lane]
  // the whole parameter block in ONE scalar-load batch (left alone the compiler fetches fields lazily, next to their first use)
  DCTTS_SGPR(p.B); DCTTS_SGPR(p.P); DCTTS_SGPR(p.p_bs); DCTTS_SGPR(p.stats); DCTTS_SGPR(p.res); DCTTS_SGPR(p.res_bs);
  DCTTS_SGPR(p.g1); DCTTS_SGPR(p.b1); DCTTS_SGPR(p.g2); DCTTS_SGPR(p.b2); DCTTS_SGPR(p.relu); DCTTS_SGPR(p.xm); DCTTS_SGPR(p.xm_bs);
  DCTTS_SGPR(p.wp); DCTTS_SGPR(p.add); DCTTS_SGPR(p.add_bs); DCTTS_SGPR(p.pout); DCTTS_SGPR(p.np_out); DCTTS_SGPR(p.stats_out);
  DCTTS_SGPR(p.raw); DCTTS_SGPR(p.raw_bs); DCTTS_SGPR(p.cout); DCTTS_SGPR(p.sig); DCTTS_SGPR(p.sig_val); DCTTS_SGPR(p.wait2);
  if constexpr (TAP2) { DCTTS_SGPR(p.xt); DCTTS_SGPR(p.xt_bs); }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = blockIdx.x, m0 = blockIdx.y * 8;
  const int arow = lane & 15, aq = lane >> 4, c4 = aq * 4;
  const int b = m0 + arow;
  const bool valid = arow < 8 && b < p.B;
  const unsigned bb = valid ? (unsigned)b : 0u;                             // rows that do not exist read utterance 0 and are zeroed
  // ---- every load of the launch, issued back to back: B fragments of this wave's two k-groups (channels 16 w .. and 128 + 16 w ..)
  //      for both tiles, then what the prologue needs, then the epilogue addend
  const float* wb = p.wp + lane * 4;
  const unsigned w0 = (unsigned)(grp * 2) * NKG * 256u, w1 = w0 + NKG * 256u;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 vb0[2], vb1[2], va[2] = {z4, z4}, vh2[2] = {z4, z4}, vrs[2] = {z4, z4}, vg1[2] = {z4, z4}, vb1_[2] = {z4, z4}, vg2[2] = {z4, z4}, vb2_[2] = {z4, z4},
        vst[4] = {z4, z4, z4, z4};
  f32x4 vtb0[2], vtb1[2], vta[2] = {z4, z4};                                // TAP2: tap -1 weights and history-row fragments
#pragma unroll
  for (int e = 0; e < 2; ++e) { vb0[e] = ldv(wb, w0 + (KC + (unsigned)(wave + 8 * e)) * 256u); vb1[e] = ldv(wb, w1 + (KC + (unsigned)(wave + 8 * e)) * 256u); }
  if constexpr (TAP2) {
#pragma unroll
    for (int e = 0; e < 2; ++e) { vtb0[e] = ldv(wb, w0 + (unsigned)(wave + 8 * e) * 256u); vtb1[e] = ldv(wb, w1 + (unsigned)(wave + 8 * e) * 256u); }
  }
  // The A side is loaded by the lanes of real rows only (8 of the MFMA's 16 rows are padding): a launch's time tracks the bytes
  // its workgroups pull through their CU's one load path (~185 KB with every lane loading, stamps: the last wave's data lands
  // ~1 us after the first's), and the padding lanes' share was half of it.
  if (valid) {
  if constexpr (TAP2) {
#pragma unroll
    for (int e = 0; e < 2; ++e) vta[e] = ldv(p.xt, bb * (unsigned)p.xt_bs + (unsigned)((8 * e + wave) * 16 + c4));
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const unsigned ch = (unsigned)((8 * e + wave) * 16 + c4);
    va[e] = ldv(p.P, bb * (unsigned)p.p_bs + ch);
    if constexpr (PRO != PRO_RAW) { vg1[e] = ldv(p.g1, ch); vb1_[e] = ldv(p.b1, ch); }
    if constexpr (PRO == PRO_LN_HC) {
      vh2[e] = ldv(p.P, bb * (unsigned)p.p_bs + 256u + ch);
      vrs[e] = ldv(p.res, bb * (unsigned)p.res_bs + ch);
      vg2[e] = ldv(p.g2, ch); vb2_[e] = ldv(p.b2, ch);
    }
  }
  if constexpr (PRO != PRO_RAW) {
#pragma unroll
    for (int g = 0; g < 4; ++g) vst[g] = ldv(p.stats, bb * 64u + (unsigned)((aq * 4 + g) * 4));
  }
  }
  // epilogue addend of this thread's output element: row (lane >> 4) * 4 + (wave & 3) of the tile, tile wave >> 2, column lane & 15
  const int erow = aq * 4 + (wave & 3), etile = wave >> 2, ecol = lane & 15;
  const int eb = m0 + erow;
  const bool eok_row = erow < 8 && eb < p.B;
  int pcol; bool ok;
  if constexpr (HCOUT) { const int c = grp * 16 + ecol; ok = c < p.cout; pcol = etile * p.cout + c; }
  else                 { pcol = (grp * 2 + etile) * 16 + ecol; ok = pcol < p.cout; }
  float addv = 0.f;
  if (p.wait2) {
    // The bulk stream's kernels that wrote this frame's presum rows have completed and released them (their stream wrote the
    // counter after them); this launch may have started earlier, so its L2 can still hold the lines of two frames ago: sc1 load.
    if (tid == 0) {
      // bounded (~1 s): a time-out raises the error word (dctts_decode_status reports it) instead of hanging the queue, and every
      // later wait of the same decode gives up at once, so a decode that cannot make progress ends in seconds, not minutes
      bool okw = p.gate_err && __hip_atomic_load(p.gate_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      for (int i = 0; i < (1 << 20) && !okw; ++i) {
        okw = __hip_atomic_load(p.wait2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= p.wait_val;
        if (!okw) __builtin_amdgcn_s_sleep(4);
      }
      if (!okw && p.gate_err) atomicOr(p.gate_err, 1);
    }
    __syncthreads();
    if (eok_row && ok) {
      const float* ap = p.add + (unsigned)(eb * p.add_bs) + (unsigned)pcol;
      asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(addv) : "v"(ap) : "memory");
    }
  } else if (eok_row && ok) addv = p.add[(unsigned)(eb * p.add_bs) + (unsigned)pcol];
  // Pin: ONE asm statement that consumes every loaded vector.  All of them must have been issued before it and nothing that uses
  // them can start before it, so the launch pays one memory round trip (the addend rides along: waited for later, its s_waitcnt
  // vmcnt(0) would also wait for the row stores issued in between -- stores count in vmcnt on gfx9).  (Without it the scheduler sinks each load next to its
  // first use to save registers -- statistics, wait, combine, gamma, wait, ... weights last: ~6 serialised round trips; a
  // sched_barrier does not help, the arithmetic simply moves above it.)
  if constexpr (TS) t_issued = wall_clock64();
#define C3_PIN_BASE "+v"(vb0[0]), "+v"(vb0[1]), "+v"(vb1[0]), "+v"(vb1[1]), "+v"(va[0]), "+v"(va[1]), "+v"(addv)
#define C3_PIN_LN   "+v"(vg1[0]), "+v"(vg1[1]), "+v"(vb1_[0]), "+v"(vb1_[1]), "+v"(vst[0]), "+v"(vst[1]), "+v"(vst[2]), "+v"(vst[3])
#define C3_PIN_HC   "+v"(vh2[0]), "+v"(vh2[1]), "+v"(vrs[0]), "+v"(vrs[1]), "+v"(vg2[0]), "+v"(vg2[1]), "+v"(vb2_[0]), "+v"(vb2_[1])
#define C3_PIN_T2   "+v"(vtb0[0]), "+v"(vtb0[1]), "+v"(vtb1[0]), "+v"(vtb1[1]), "+v"(vta[0]), "+v"(vta[1])
  if constexpr (PRO == PRO_LN_HC && TAP2) asm volatile("; chain3: all loads in flight" : C3_PIN_BASE, C3_PIN_LN, C3_PIN_HC, C3_PIN_T2);
  else if constexpr (PRO == PRO_LN_HC)    asm volatile("; chain3: all loads in flight" : C3_PIN_BASE, C3_PIN_LN, C3_PIN_HC);
  else if constexpr (PRO == PRO_LN_C && TAP2) asm volatile("; chain3: all loads in flight" : C3_PIN_BASE, C3_PIN_LN, C3_PIN_T2);
  else if constexpr (PRO == PRO_LN_C)     asm volatile("; chain3: all loads in flight" : C3_PIN_BASE, C3_PIN_LN);
  else { static_assert(PRO != PRO_RAW || !TAP2, "a raw-input layer has no tap -1 form"); asm volatile("; chain3: all loads in flight" : C3_PIN_BASE); }
#undef C3_PIN_BASE
#undef C3_PIN_LN
#undef C3_PIN_HC
#undef C3_PIN_T2
  // This launch runs, so every earlier launch of the stream has completed and released its stores: say so to the other stream.
  // (A stream write-value packet after the previous piece says the same ~6 us of command-processor time later.)
  if (p.sig && (blockIdx.x | blockIdx.y) == 0 && tid == 0) __hip_atomic_store(p.sig, p.sig_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if constexpr (TS) t_landed = wall_clock64();
  auto f4 = [](const f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); };
  float4 bq0[2], bq1[2], av[2], h2v[2], rsv[2], g1v[2], b1v[2], g2v[2], b2v[2], st[4];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    bq0[e] = f4(vb0[e]); bq1[e] = f4(vb1[e]); av[e] = f4(va[e]);
    if constexpr (PRO != PRO_RAW) { g1v[e] = f4(vg1[e]); b1v[e] = f4(vb1_[e]); }
    if constexpr (PRO == PRO_LN_HC) { h2v[e] = f4(vh2[e]); rsv[e] = f4(vrs[e]); g2v[e] = f4(vg2[e]); b2v[e] = f4(vb2_[e]); }
  }
  if constexpr (PRO != PRO_RAW) {
#pragma unroll
    for (int g = 0; g < 4; ++g) st[g] = f4(vst[g]);
  }

  // ---- rebuild x (LN, + ReLU, or + sigmoid gate + highway mix) in the A-fragment registers
  if constexpr (PRO != PRO_RAW) {
    float m1, r1, m2 = 0.f, r2 = 0.f;
    combine_stats(st, 0, m1, r1);
    if constexpr (PRO == PRO_LN_HC) combine_stats(st, 1, m2, r2);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float4 x = av[e];
      const float4 g1 = g1v[e], b1 = b1v[e];
      x.x = (x.x - m1) * r1 * g1.x + b1.x; x.y = (x.y - m1) * r1 * g1.y + b1.y;
      x.z = (x.z - m1) * r1 * g1.z + b1.z; x.w = (x.w - m1) * r1 * g1.w + b1.w;
      if constexpr (PRO == PRO_LN_HC) {
        const float4 g2 = g2v[e], b2 = b2v[e], h2 = h2v[e], xr = rsv[e];
        { const float s_ = sigmoid_fast(x.x); x.x = s_ * ((h2.x - m2) * r2 * g2.x + b2.x) + (1.0f - s_) * xr.x; }
        { const float s_ = sigmoid_fast(x.y); x.y = s_ * ((h2.y - m2) * r2 * g2.y + b2.y) + (1.0f - s_) * xr.y; }
        { const float s_ = sigmoid_fast(x.z); x.z = s_ * ((h2.z - m2) * r2 * g2.z + b2.z) + (1.0f - s_) * xr.z; }
        { const float s_ = sigmoid_fast(x.w); x.w = s_ * ((h2.w - m2) * r2 * g2.w + b2.w) + (1.0f - s_) * xr.w; }
      } else {
        if (p.relu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
      }
      av[e] = x;
    }
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) if (!valid) av[e] = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- 2 k-groups x 4 MFMAs x 2 tiles (TAP2: the previous row's two k-groups first)
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (TAP2) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      f32x4 a = vta[e];
      if (!valid) a = f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 b0 = vtb0[e], b1 = vtb1[e];
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b0[i], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b1[i], acc1, 0, 0, 0); }
    }
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float4 a = av[e], b0 = bq0[e], b1 = bq1[e];
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, acc1, 0, 0, 0);
  }
  if constexpr (TS) { asm volatile("" : "+v"(acc0), "+v"(acc1)); t_mfma = wall_clock64(); }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[((wave * 2 + 0) * 4 + j) * 64 + lane] = acc0[j]; red[((wave * 2 + 1) * 4 + j) * 64 + lane] = acc1[j]; }
  // the rebuilt row is only needed by later launches (highway residual / history): store it off the path to the barrier
  if constexpr (PRO != PRO_RAW) {
    if (p.xm && grp == 0 && valid) {
#pragma unroll
      for (int e = 0; e < 2; ++e) *reinterpret_cast<float4*>(p.xm + (long)b * p.xm_bs + (8 * e + wave) * 16 + c4) = av[e];
    }
  }
  __syncthreads();
  if constexpr (TS) t_sync = wall_clock64();
  // ---- fixed-order reduction over the 8 waves: thread (wave, lane) owns element j = wave & 3 of tile wave >> 2, lane's (row, column)
  float v_ = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) v_ += red[((w * 2 + etile) * 4 + (wave & 3)) * 64 + lane];
  const bool wr = ok && eok_row;
  if (p.raw && wr) p.raw[(long)eb * p.raw_bs + pcol] = v_;
  if (ok) v_ += addv;
  if (wr) p.pout[(long)eb * p.np_out + pcol] = v_;
  // partial LN statistics of this 16-column group: a DPP row (16 lanes) holds one output row's 16 columns
  const float mg = row16_sum(ok ? v_ : 0.f) * (1.0f / 16.0f);
  const float dv = ok ? v_ - mg : 0.f;
  const float m2g = row16_sum(dv * dv);
  if (wr && ecol == 0) {
    const int G = HCOUT ? grp : grp * 2 + etile;
    float* so = p.stats_out + ((long)eb * 16 + G) * 4 + (HCOUT ? etile * 2 : 0);
    so[0] = mg; so[1] = m2g;
  }
  if constexpr (TS) {
    if (p.ts && lane == 0) {
      long long* o = p.ts + (long)(blockIdx.y * gridDim.x + blockIdx.x) * 32;
      if (wave == 0) { o[0] = t_in; o[1] = t_issued; o[2] = t_landed; o[3] = t_mfma; o[4] = t_sync; o[5] = wall_clock64(); }
      o[8 + wave] = t_in; o[16 + wave] = t_landed; o[24 + wave] = t_mfma;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- chain: the k = 1 layers around the mel frame
// mlp_rows_kernel: AudioDec C_8 .. C_11 (+ sigmoid = mel frame j, networks.py:192-210) and AudioEnc C_1 .. C_3 of frame j+1
// (networks.py:82-105) -- seven dependent k = 1 layers -- in ONE launch, split by ROWS: a workgroup owns R (= 2) utterances and every
// column, so layer-norm is local and nothing is exchanged between workgroups; the price is that each workgroup streams every layer's
// whole weight matrix (256 KB) through its CU.  That is 1.9-2.1 us per layer (tools/micro/cu_stream.hip: 123-135 GB/s per CU, flat
// from 1 to 32 workgroups) against 5.3 us for a column-split launch, so 7 launches (36 us per frame) become one of ~18 us.
// (Round 1's rowmlp_kernel had the same idea and measured 8 us per layer: its loads sat behind uniform branches.)
// R = rows per workgroup: the FMAs of a layer cost ~0.45 us per row on a SIMD's two waves (packed fp32 FMAs issue at half the rate
// the peak suggests), the weight stream ~2.1 us whatever R is, so R = 2 (16 workgroups at B = 32) keeps the layer load-bound.
// Per layer: wave w contracts k in [w cin/8, (w+1) cin/8) for all 4 rows and 4 columns per lane with plain fp32 FMAs (the fp32
// MFMA rate IS the vector rate, and at 4 rows there is no tile to fill) -- weights in TF layout (Cin, Cout), one coalesced 1 KB
// row slice per wave load, all of a wave's loads in flight before the first FMA; partial sums meet in LDS in a fixed order; waves
// 0..3 then finish one row each (bias, two-pass layer-norm over the row, activation) and leave it in LDS as the next layer's input.
struct MlpLayer { const float* w; const float* bias; const float* g; const float* be; int cin, cout, relu, pad_; };
struct MlpRowsParams {
  int B, frame, nlayers, mel_layer;          // mel_layer: index of the layer whose layer-norm output is the mel logits row (-1: none)
  RowNorm nrm; long par;                     // first input row = gate(LN(P[b])) mixed with the residual row (an HC producer; parity copy `par`)
  MlpLayer L[7];
  float* ypad; long y_bstride; long y_row; int y_stride;      // sigmoid(logits) -> ypad[b][y_row] (the +1 shift of train.py:51 folded into y_row)
  float* logits; long l_bstride; long l_row; int l_stride;
  float* pout; float* stats_out;             // the LAST layer (when it is not the mel layer): pre-norm rows [b][cout] + per-16-column partial statistics [b][16][4]
  long long* ts;                             // TS instantiation only (measurement): wall-clock stamps of workgroup 0, wave 0
};

template <int R, bool TS = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) mlp_rows_kernel(const MlpRowsParams* __restrict__ pp, long long* ts) {
  long long tst[32]; int nts = 0;
  auto stampt = [&]() { if constexpr (TS) { if (nts < 32) tst[nts++] = wall_clock64(); } };
  stampt();
  typedef const __attribute__((address_space(4))) MlpRowsParams CP;
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  CP& p = *(CP*)pp;
  // what the first rows and the first layer need, in ONE batch of scalar loads (lazily they arrive as five dependent batches, ~1.5 us)
  asm volatile("; mlp_rows: parameters, one batch"
               :: "s"(p.B), "s"(p.frame), "s"(p.nlayers), "s"(p.mel_layer), "s"(p.par),
                  "s"(p.nrm.P), "s"(p.nrm.np), "s"(p.nrm.g1), "s"(p.nrm.b1), "s"(p.nrm.g2), "s"(p.nrm.b2),
                  "s"(p.nrm.res), "s"(p.nrm.res_bstride), "s"(p.nrm.res_row0), "s"(p.nrm.res_stride), "s"(p.nrm.res_set),
                  "s"(p.L[0].w), "s"(p.L[0].bias), "s"(p.L[0].g), "s"(p.L[0].be), "s"(p.L[0].cin), "s"(p.L[0].cout), "s"(p.L[0].relu));
  __shared__ __attribute__((aligned(16))) float xs[R * 256];
  __shared__ __attribute__((aligned(16))) float red[8 * R * 256];
  __shared__ int s_desc[7 * 12];             // the layer descriptors, copied once: a scalar load from the (cold, per-frame) parameter
                                             // block costs a memory round trip, and every layer needed two or three in sequence
  static_assert(sizeof(MlpLayer) == 48, "MlpLayer is copied to LDS as 12 dwords");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * R, c0 = lane * 4;
  if (tid < 7 * 12) s_desc[tid] = reinterpret_cast<const int*>(&pp->L[0])[tid];
  typedef const __attribute__((address_space(1))) float* gptr;   // GLOBAL pointers: rebuilt from integers they would be generic, and the flat loads
                                                                 // a generic pointer gets also count in lgkmcnt -- the LDS-only barrier then waits for them
  struct LDesc { gptr w; gptr bias; gptr g; gptr be; int cin, cout, relu; };
  auto rfl = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto desc = [&](int l) {                   // layer 0 straight from the parameter block (nothing is in LDS yet), the others from LDS
    LDesc d;
    if (l == 0) { d.w = (gptr)p.L[0].w; d.bias = (gptr)p.L[0].bias; d.g = (gptr)p.L[0].g; d.be = (gptr)p.L[0].be; d.cin = p.L[0].cin; d.cout = p.L[0].cout; d.relu = p.L[0].relu; return d; }
    const int* q = &s_desc[l * 12];
    auto ptr = [&](int i) { return (gptr)(((unsigned long long)(unsigned)rfl(q[2 * i + 1]) << 32) | (unsigned)rfl(q[2 * i])); };
    d.w = ptr(0); d.bias = ptr(1); d.g = ptr(2); d.be = ptr(3); d.cin = rfl(q[8]); d.cout = rfl(q[9]); d.relu = rfl(q[10]);
    return d;
  };

  // A layer's loads: this wave's k slice of the weights (32 row slices of 1 KB, every load unconditional: rows past the slice
  // re-read an existing row, columns past cout read column 0) + the row-finishing phase's parameters.  They are issued one layer
  // AHEAD and INTERLEAVED with the current layer's FMAs -- each pair of register slots is refilled right after the FMAs that read
  // it -- because ISSUING a wave's 35 loads is what takes the time (stamps: eight waves x 35 loads through the CU's one address
  // path = ~2 us, whatever the rows per workgroup; issued in a burst after the FMAs they delayed the barrier by exactly that).
  // The pins at the top of the next layer are where they must have landed (see chain3_kernel).
  f32x8 w8[16]; f32x4 pbias, pg, pbe;
  struct Slice { gptr wp; int k0, nk, cin, cout; };
  auto slice_of = [&](const LDesc& d) {
    Slice sl; sl.cin = d.cin; sl.cout = d.cout;
    const int kw = (((d.cin >> 3) + 3) >> 2) << 2;                       // k slice per wave, a multiple of 4 (32 for cin 256, 12 for cin 80: aligned LDS reads)
    sl.k0 = wave * kw;
    int nk = d.cin - sl.k0; sl.nk = nk < 0 ? 0 : (nk > kw ? kw : nk);
    sl.wp = d.w + ((c0 < d.cout) ? c0 : 0);
    return sl;
  };
  typedef const __attribute__((address_space(1))) f32x4* gv4;
  auto load_pair = [&](const Slice& sl, int u) {                         // rows u, u+1 of the slice -> one 8-register slot
    int ka = sl.k0 + (u < sl.nk ? u : 0), kb = sl.k0 + (u + 1 < sl.nk ? u + 1 : 0);
    ka = ka < sl.cin ? ka : sl.cin - 1; kb = kb < sl.cin ? kb : sl.cin - 1;
    const f32x4 a = *(gv4)(sl.wp + (size_t)ka * sl.cout), b = *(gv4)(sl.wp + (size_t)kb * sl.cout);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto load_params = [&](const LDesc& d) {
    const int cc = (c0 < d.cout) ? c0 : 0;
    pbias = *(gv4)(d.bias + cc); pg = *(gv4)(d.g + cc); pbe = *(gv4)(d.be + cc);
  };
  LDesc cur = desc(0);
  {
    const Slice s0 = slice_of(cur);
#pragma unroll
    for (int u = 0; u < 32; u += 2) w8[u >> 1] = load_pair(s0, u);
    load_params(cur);
  }
  // ---- first input rows: waves 0..3 rebuild one row each from the highway producer's pre-norm row (full-row statistics: the row is ours)
  if (wave < R) {
    const int b = b0 + wave;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b < p.B) {
      RowNorm n; n.P = p.nrm.P; n.np = p.nrm.np; n.g1 = p.nrm.g1; n.b1 = p.nrm.b1; n.g2 = p.nrm.g2; n.b2 = p.nrm.b2; n.act = p.nrm.act; n.ngroups = p.nrm.ngroups;
      n.res = p.nrm.res; n.res_bstride = p.nrm.res_bstride; n.res_row0 = p.nrm.res_row0; n.res_stride = p.nrm.res_stride; n.res_set = p.nrm.res_set;
      x = norm_row_hc(n, (long)b, b, p.frame, lane, p.par);
    }
    *reinterpret_cast<float4*>(&xs[wave * 256 + c0]) = x;
  }
  __syncthreads();
  stampt();
  const int nlayers = p.nlayers, mel_layer = p.mel_layer;
  for (int l = 0; l < nlayers; ++l) {
    const int cin = cur.cin, cout = cur.cout, relu = cur.relu;
    const int kw = (((cin >> 3) + 3) >> 2) << 2, k0 = wave * kw;
    int nk = cin - k0; nk = nk < 0 ? 0 : (nk > kw ? kw : nk);          // rows of the slice that exist (cin 80: waves 0..5 have 12, wave 6 has 8, wave 7 none)
    const bool colok = c0 < cout;
    const LDesc nxt = desc(l + 1 < nlayers ? l + 1 : l);      // the last layer re-reads itself (harmless)
    const Slice sn = slice_of(nxt);
    // Pins in four chunks of eight rows: all of the slice's loads were issued a layer ago (in order), so the first chunk's FMAs can
    // run while the later rows are still landing.  With ONE pin the last wave to be served waited for its whole slice (the CU's load
    // path delivers the eight waves' 280 KB over ~2 us) before its first FMA: 4.8 us per layer, stamps.
    asm volatile("; mlp_rows: parameters + rows 0-7 landed" : "+v"(pbias), "+v"(pg), "+v"(pbe), "+v"(w8[0]), "+v"(w8[1]), "+v"(w8[2]), "+v"(w8[3]) :: "memory");
    stampt();
    const f32x4 cbias = pbias, cg = pg, cbe = pbe;
    f32x4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u4 = 0; u4 < 8; ++u4) {
      if (u4 == 2) asm volatile("; mlp_rows: rows 8-15 landed" : "+v"(w8[4]), "+v"(w8[5]), "+v"(w8[6]), "+v"(w8[7]) :: "memory");
      if (u4 == 4) asm volatile("; mlp_rows: rows 16-23 landed" : "+v"(w8[8]), "+v"(w8[9]), "+v"(w8[10]), "+v"(w8[11]) :: "memory");
      if (u4 == 6) asm volatile("; mlp_rows: rows 24-31 landed" : "+v"(w8[12]), "+v"(w8[13]), "+v"(w8[14]), "+v"(w8[15]) :: "memory");
      f32x4 xv[R];
#pragma unroll
      for (int r = 0; r < R; ++r) xv[r] = *reinterpret_cast<const f32x4*>(&xs[r * 256 + (4 * u4 < nk ? k0 + 4 * u4 : 252)]);   // broadcast read, 16-byte aligned; past this wave's slice (cin 80 only): columns 252..255, which are zero then
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // rows past this wave's slice (cin 80 only) read x from columns that hold exact zeros (a finished row is written as 256
        // columns, zeros beyond its width) and re-read finite weights, so those products vanish: one address select per four rows
        // instead of a mask multiply per product (the masks were a third of the VALU work)
        const int u = 4 * u4 + i;
        const f32x8 wp8 = w8[u >> 1];
        const f32x4 wq = (u & 1) ? __builtin_shufflevector(wp8, wp8, 4, 5, 6, 7) : __builtin_shufflevector(wp8, wp8, 0, 1, 2, 3);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] += xv[r][i] * wq;
      }
      // refill the two slots these FMAs just read with the next layer's rows (not earlier: both slices would be live at once)
      asm volatile("" ::: "memory");
      w8[2 * u4] = load_pair(sn, 4 * u4); w8[2 * u4 + 1] = load_pair(sn, 4 * u4 + 2);
    }
    load_params(nxt);
    stampt();
    cur = nxt;
#pragma unroll
    for (int r = 0; r < R; ++r) *reinterpret_cast<f32x4*>(&red[(wave * R + r) * 256 + c0]) = acc[r];
    lds_barrier();             // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the next layer's weights (2 us per layer, stamps)
    stampt();
    // ---- waves 0..3: one row each -- reduce the 8 partial sums in a fixed order, bias, layer-norm, activation
    if (wave < R) {
      const int b = b0 + wave;
      const bool rowok = b < p.B;
      f32x4 y = {0.f, 0.f, 0.f, 0.f};
      if (colok) {
        y = cbias;
#pragma unroll
        for (int w = 0; w < 8; ++w) y += *reinterpret_cast<const f32x4*>(&red[(w * R + wave) * 256 + c0]);
      }
      const bool lastl = (l + 1 == nlayers);
      if (lastl && l != mel_layer) {
        // hand the pre-norm row + its per-16-column partial statistics to the column-split kernel that follows (chain3_kernel<LN_C>)
        const float qs = dpp_add<0x4E>(dpp_add<0xB1>(y[0] + y[1] + y[2] + y[3]));       // sum over the 4 lanes of a 16-column group
        const float mg = qs * (1.0f / 16.0f);
        const f32x4 dq = {y[0] - mg, y[1] - mg, y[2] - mg, y[3] - mg};
        const float m2 = dpp_add<0x4E>(dpp_add<0xB1>(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3]));
        if (rowok && colok) {
          *reinterpret_cast<f32x4*>(p.pout + (long)b * cout + c0) = y;
          if ((lane & 3) == 0) { float* so = p.stats_out + ((long)b * 16 + (lane >> 2)) * 4; so[0] = mg; so[1] = m2; }
        }
      } else {
        const float invn = 1.0f / (float)cout;
        const float mean = wave_sum(colok ? y[0] + y[1] + y[2] + y[3] : 0.f) * invn;
        const f32x4 dv = colok ? f32x4{y[0] - mean, y[1] - mean, y[2] - mean, y[3] - mean} : f32x4{0.f, 0.f, 0.f, 0.f};
        const float rs = 1.0f / sqrtf(wave_sum(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2] + dv[3] * dv[3]) * invn + 1e-12f);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (colok) {
          o = dv * rs * cg + cbe;
          if (relu) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
          if (l == mel_layer) {                                    // logits of mel frame j (networks.py:202-209); Y[j] = sigmoid (:210)
            if (rowok) *reinterpret_cast<f32x4*>(p.logits + ((long)b * p.l_bstride + p.l_row) * p.l_stride + c0) = o;
            o = f32x4{sigmoidf_(o[0]), sigmoidf_(o[1]), sigmoidf_(o[2]), sigmoidf_(o[3])};
            if (rowok) *reinterpret_cast<f32x4*>(p.ypad + ((long)b * p.y_bstride + p.y_row) * p.y_stride + c0) = o;
          }
        }
        *reinterpret_cast<f32x4*>(&xs[wave * 256 + c0]) = o;      // columns >= cout are zero: never read (the next layer's K is cout)
      }
    }
    lds_barrier();
    stampt();
  }
  if constexpr (TS) { if (ts && blockIdx.x == 0 && tid == 0) for (int i = 0; i < 32; ++i) ts[i] = i < nts ? tst[i] : 0; }
}

}  // namespace dctts
