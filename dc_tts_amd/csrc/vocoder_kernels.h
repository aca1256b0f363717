// vocoder_kernels.h -- the reference's vocoder tail (utils.py:67-114: spectrogram2wav / griffin_lim / invert_spectrogram)
// as HIP kernels for gfx950.  They follow the published librosa 0.6 stft / istft / effects.trim algorithms (the test oracle restates them on the CPU).
//
// Griffin-Lim is 50 x (istft -> stft -> keep the phase).  With n_fft = 2048, hop = 275, win = 1102 this is FFT + streaming
// work (no MFMA).  Per iteration two launches:
//   gl_iter_wave_kernel   ONE WAVE PER FRAME: window * y -> 2048-point real FFT (a 1024-point complex FFT held in the wave's
//                         registers, fft_wave.h) -> est / max(1e-8,|est|) * magnitude -> inverse FFT -> * window -> the 1102
//                         samples under the window, stored per frame (`fr`, (B, F, FRS)).  The spectrum never touches HBM.
//   ola_kernel            librosa.istft's overlap-add as a GATHER (<= 5 frames per output sample, summed in frame order,
//                         divided by the window sum-square): deterministic, no atomics -> y (B, L).
// Around them:
//   istft_frames_kernel   first pass (real, zero-phase spectrum) and the X_best debug path: 256-thread radix-4 LDS FFT.
//   stft_phase_kernel     X_best (B, F, 1025) complex out, for tests (the fused kernel never materialises it).
//   deemph_kernel         scipy.signal.lfilter([1],[1,-a]) (utils.py:89) as a blocked linear-recurrence scan.
//   frame_power_kernel / trim_bounds_kernel   librosa.effects.trim's [start, end) (utils.py:92).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft_wave.h"

namespace dctts {

constexpr int VOC_NFFT = 2048;
constexpr int VOC_M = 1024;           // complex FFT length
constexpr int VOC_BINS = 1025;
constexpr int VOC_THREADS = 256;      // = VOC_M / 4 : one radix-4 butterfly per thread per pass

struct VocGeom {
  int F;            // frames per utterance
  int L;            // samples per utterance = hop * (F - 1)
  int hop, win, lpad;   // lpad = (n_fft - win) / 2 : first sample of a frame under the (centre-padded) window
  int frs;          // floats per stored frame (>= win)
  const float* window;     // (n_fft) padded periodic Hann
  const float* wss;        // (n_fft + hop*(F-1)) overlap-added squared window
  const float2* w1024;     // exp(-2 pi i m / 1024), m < 1024
  const float2* w1024i;    // the same table through a second pointer: stops the compiler from keeping ~40 twiddles live (or
                           // spilled) between the forward and the inverse transform of gl_iter_wave_kernel
  const float2* w2048;     // exp(-2 pi i k / 2048), k <= 1024
  float tiny;
  int dmax;         // ceil(win / hop): frames f-dmax .. f+dmax can overlap frame f's window
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a * conj(b)

// 1024-point complex FFT, Stockham radix-4, data in s[0] on entry, result in s[1] on exit.  INV: e^{+...}, unnormalised.
template <bool INV>
__device__ __forceinline__ void fft1024(float2 (*s)[VOC_M], const float2* __restrict__ w1024, int tid) {
  constexpr int T = VOC_M / 4;
  int src = 0;
#pragma unroll
  for (int p = 1; p < VOC_M; p <<= 2) {
    const float2* in = s[src];
    float2* out = s[src ^ 1];
    const int k = tid & (p - 1);
    const int j = ((tid - k) << 2) + k;
    const int m = k * (VOC_M / (4 * p));
    float2 u0 = in[tid], u1 = in[tid + T], u2 = in[tid + 2 * T], u3 = in[tid + 3 * T];
    if (p > 1) {
      const float2 t1 = w1024[m], t2 = w1024[2 * m], t3 = w1024[3 * m];
      if (INV) { u1 = cmulc(u1, t1); u2 = cmulc(u2, t2); u3 = cmulc(u3, t3); }
      else     { u1 = cmul(u1, t1);  u2 = cmul(u2, t2);  u3 = cmul(u3, t3); }
    }
    const float2 v0 = make_float2(u0.x + u2.x, u0.y + u2.y), v1 = make_float2(u0.x - u2.x, u0.y - u2.y);
    const float2 v2 = make_float2(u1.x + u3.x, u1.y + u3.y);
    const float2 d = make_float2(u1.x - u3.x, u1.y - u3.y);
    const float2 v3 = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);     // d * (+i) / d * (-i)
    out[j] = make_float2(v0.x + v2.x, v0.y + v2.y);
    out[j + p] = make_float2(v1.x + v3.x, v1.y + v3.y);
    out[j + 2 * p] = make_float2(v0.x - v2.x, v0.y - v2.y);
    out[j + 3 * p] = make_float2(v1.x - v3.x, v1.y - v3.y);
    src ^= 1;
    __syncthreads();
  }
}

// One sample of librosa.istft's output BEFORE the centre trim: index m in [0, n_fft + hop*(F-1)).
__device__ __forceinline__ float ola_sample(const VocGeom& g, const float* __restrict__ fr_b, int m) {
  int g_hi = (m - g.lpad) / g.hop;                       // m - g*hop >= lpad
  if (m < g.lpad) return 0.f;
  if (g_hi > g.F - 1) g_hi = g.F - 1;
  int lo_num = m - (g.lpad + g.win - 1);                 // m - g*hop <= lpad + win - 1
  int g_lo = lo_num <= 0 ? 0 : (lo_num + g.hop - 1) / g.hop;
  float acc = 0.f;
  for (int q = g_lo; q <= g_hi; ++q) acc = acc + fr_b[(long)q * g.frs + (m - q * g.hop - g.lpad)];
  const float ws = g.wss[m];
  return ws > g.tiny ? acc / ws : acc;
}

// grid (F, B).  X: (B, F, 1025) complex, or nullptr with spec (B, F, 1025) real (first iteration: zero phase).
__global__ void __launch_bounds__(VOC_THREADS) istft_frames_kernel(const VocGeom g, const float2* __restrict__ X,
                                                                   const float* __restrict__ spec, float* __restrict__ fr) {
  __shared__ float2 s[2][VOC_M];
  const int tid = threadIdx.x, f = blockIdx.x, b = blockIdx.y;
  const long row = ((long)b * g.F + f) * VOC_BINS;
  auto ldx = [&](int k) -> float2 {
    float2 v = X ? X[row + k] : make_float2(spec[row + k], 0.f);
    if (k == 0 || k == VOC_M) v.y = 0.f;                 // irfft ignores the imaginary part of DC / Nyquist
    return v;
  };
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = tid + r * VOC_THREADS;
    const float2 a = ldx(k), c = ldx(VOC_M - k);
    const float2 E = make_float2(0.5f * (a.x + c.x), 0.5f * (a.y - c.y));        // (X[k] + conj X[M-k]) / 2
    const float2 D = make_float2(0.5f * (a.x - c.x), 0.5f * (a.y + c.y));        // (X[k] - conj X[M-k]) / 2
    const float2 O = cmulc(D, g.w2048[k]);                                        // * e^{+2 pi i k / N}
    s[0][k] = make_float2(E.x - O.y, E.y + O.x);                                  // E + i O
  }
  __syncthreads();
  fft1024<true>(s, g.w1024, tid);
  float* out = fr + ((long)b * g.F + f) * g.frs;
  const float inv = 1.0f / (float)VOC_M;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = tid + r * VOC_THREADS;
    const float2 z = s[1][n];
    const int i0 = 2 * n - g.lpad, i1 = i0 + 1;
    if (i0 >= 0 && i0 < g.win) out[i0] = (z.x * inv) * g.window[2 * n];
    if (i1 >= 0 && i1 < g.win) out[i1] = (z.y * inv) * g.window[2 * n + 1];
  }
}

// grid (F, B).  Xout = spec * est / max(1e-8, |est|), est = stft(istft(X_prev)) evaluated from the stored frames.
__global__ void __launch_bounds__(VOC_THREADS) stft_phase_kernel(const VocGeom g, const float* __restrict__ fr,
                                                                 const float* __restrict__ spec, float2* __restrict__ Xout) {
  __shared__ float2 s[2][VOC_M];
  const int tid = threadIdx.x, f = blockIdx.x, b = blockIdx.y;
  const float* fr_b = fr + (long)b * g.F * g.frs;
  auto sample = [&](int i) -> float {                     // windowed input sample i of frame f (librosa.stft, center=True)
    if (i < g.lpad || i >= g.lpad + g.win) return 0.f;
    int n = f * g.hop + i - VOC_NFFT / 2;                 // index into istft's (trimmed) output, reflect-padded
    if (n < 0) n = -n; else if (n >= g.L) n = 2 * (g.L - 1) - n;
    return ola_sample(g, fr_b, n + VOC_NFFT / 2) * g.window[i];
  };
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = tid + r * VOC_THREADS;
    s[0][n] = make_float2(sample(2 * n), sample(2 * n + 1));
  }
  __syncthreads();
  fft1024<false>(s, g.w1024, tid);
  const long row = ((long)b * g.F + f) * VOC_BINS;
  for (int k = tid; k <= VOC_M; k += VOC_THREADS) {
    const float2 a = s[1][k & (VOC_M - 1)], c = s[1][(VOC_M - k) & (VOC_M - 1)];
    const float2 E = make_float2(0.5f * (a.x + c.x), 0.5f * (a.y - c.y));
    const float2 D = make_float2(0.5f * (a.x - c.x), 0.5f * (a.y + c.y));
    const float2 Dw = cmul(D, g.w2048[k]);                                         // * e^{-2 pi i k / N}
    const float2 est = make_float2(E.x + Dw.y, E.y - Dw.x);                        // E + Dw / i
    const float mag = fmaxf(1e-8f, hypotf(est.x, est.y));
    const float sp = spec[row + k];
    Xout[row + k] = make_float2(sp * (est.x / mag), sp * (est.y / mag));
  }
}

// The production form of one Griffin-Lim iteration (utils.py:100-103 + the windowed irfft of the next istft):
//   frames_out[f] = window * irfft(spec[f] * phase(rfft(window * y_pad[f hop : f hop + n_fft])))
// ONE WAVE PER FRAME (fft_wave.h): no workgroup barrier, the spectrum row never leaves registers / the wave's 8.7 KB exchange
// buffer.  `y` is the overlap-added signal of the previous pass (ola_kernel) -- gathering it straight from the frames cost
// ~4x the loads and ~2 k VALU instructions per frame, and this kernel is VALU-issue bound.  Workgroup w runs on XCD w % 8, so
// items are dealt out to give each XCD a contiguous run of frames (neighbouring frames share 3/4 of their input in that L2).
__device__ __forceinline__ void gl_iter_wave_body(const VocGeom& g, const float* __restrict__ y,
                                                  const float* __restrict__ spec, float* __restrict__ fr_out,
                                                  int n_items, int per_xcd) {
  __shared__ float2 ex[FW_EX];
  const int lane = threadIdx.x;
  const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || item >= n_items) return;
  const int b = item / g.F, f = item - b * g.F;
  const float* yb = y + (long)b * g.L;
  const int n0 = f * g.hop - VOC_NFFT / 2;                  // signal index of frame sample 0 (librosa.stft, center=True)
  const int q_lo = g.lpad >> 7, q_hi = (g.lpad + g.win - 1) >> 7;     // lane-strided slots 128 q .. 128 q + 127 that touch the window
  float2 v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    v[q] = make_float2(0.f, 0.f);
    if (q >= q_lo && q <= q_hi) {                            // uniform
      const int i0 = 2 * (lane + 64 * q);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = i0 + h;
        float x = 0.f;
        if (i >= g.lpad && i < g.lpad + g.win) {
          int n = n0 + i;
          if (n < 0) n = -n; else if (n >= g.L) n = 2 * (g.L - 1) - n;      // np.pad(mode="reflect")
          x = yb[(unsigned)n] * g.window[(unsigned)i];
        }
        if (h == 0) v[q].x = x; else v[q].y = x;
      }
    }
  }
  fw_passA_store<false>(v, ex, lane); __syncthreads();
  fw_passB_load<false>(v, ex, lane, g.w1024); __syncthreads();
  fw_passB_store(v, ex, lane); __syncthreads();
  fw_passC_load<false>(v, ex, lane, g.w1024); __syncthreads();          // v[r] = Z[lane + 64 r]
#pragma unroll
  for (int r = 0; r < 16; ++r) ex[fw_base(lane) + 68 * r] = v[r];
  if (lane == 0) ex[FW_N + FW_N / 16] = v[0];             // Z[M] = Z[0], where lane 0's "partner lane 64" formula looks for it
  __syncthreads();
  const float2* exp_ = ex + fw_base(64 - lane);            // Z[M - k], k = lane + 64 r, sits at fw_base(64 - lane) + 68 (15 - r)
  const float* sp_row = spec + ((long)b * g.F + f) * VOC_BINS;
  // Post-twist (Z -> est), phase projection and the inverse transform's pre-twist, all from the pair (Z[k], Z[M-k]):
  //   est[k] = E + Dw/i,  est[M-k] = conj(E) + conj(Dw)/i   with E = (a + conj c)/2, Dw = W^k (a - conj c)/2
  // so the lane that owns k rebuilds X_best[M-k] itself and no second exchange is needed.
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int k = lane + 64 * r;
    const float2 a = v[r], c = exp_[68 * (15 - r)];
    const float2 w = g.w2048[(unsigned)k];
    const float2 E = make_float2(0.5f * (a.x + c.x), 0.5f * (a.y - c.y));
    const float2 D = make_float2(0.5f * (a.x - c.x), 0.5f * (a.y + c.y));
    const float2 Dw = fw_mul(D, w);
    const float2 e1 = make_float2(E.x + Dw.y, E.y - Dw.x);                 // est[k]
    const float2 e2 = make_float2(E.x - Dw.y, -(E.y + Dw.x));              // est[M-k]
    // spec / max(1e-8, |est|) as spec * min(rsq(|est|^2), 1e8): v_rsq_f32 is ~1 ulp, an IEEE sqrt + divide costs ~25 instructions
    const float s1 = sp_row[(unsigned)k] * fminf(__builtin_amdgcn_rsqf(fmaf(e1.x, e1.x, e1.y * e1.y)), 1e8f);
    const float s2 = sp_row[(unsigned)(VOC_M - k)] * fminf(__builtin_amdgcn_rsqf(fmaf(e2.x, e2.x, e2.y * e2.y)), 1e8f);
    const float2 x1 = make_float2(s1 * e1.x, k == 0 ? 0.f : s1 * e1.y);    // X_best[k]   (irfft drops Im of DC / Nyquist)
    const float2 x2 = make_float2(s2 * e2.x, k == 0 ? 0.f : s2 * e2.y);    // X_best[M-k]
    const float2 E2 = make_float2(0.5f * (x1.x + x2.x), 0.5f * (x1.y - x2.y));
    const float2 D2 = make_float2(0.5f * (x1.x - x2.x), 0.5f * (x1.y + x2.y));
    const float2 O2 = fw_mulc(D2, w);
    v[r] = make_float2(E2.x - O2.y, E2.y + O2.x);
  }
  __syncthreads();
  fw_passA_store<true>(v, ex, lane); __syncthreads();
  fw_passB_load<true>(v, ex, lane, g.w1024i); __syncthreads();
  fw_passB_store(v, ex, lane); __syncthreads();
  fw_passC_load<true>(v, ex, lane, g.w1024i);
  float* out = fr_out + ((long)b * g.F + f) * g.frs;
  const float inv = 1.0f / (float)VOC_M;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = lane + 64 * r;
    const int i0 = 2 * n - g.lpad, i1 = i0 + 1;
    if (i0 >= 0 && i0 < g.win) out[(unsigned)i0] = (v[r].x * inv) * g.window[(unsigned)(2 * n)];
    if (i1 >= 0 && i1 < g.win) out[(unsigned)i1] = (v[r].y * inv) * g.window[(unsigned)(2 * n + 1)];
  }
}
// Register budget: 168 VGPRs (3 waves / SIMD) holds the lane program without spills; the 128-VGPR build (4 waves / SIMD)
// spills ~34 dwords.  Both are kept so the trade can be measured (DCTTS_VOC_WAVE = 1 | 2).
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
gl_iter_wave_kernel(const VocGeom g, const float* __restrict__ y, const float* __restrict__ spec, float* __restrict__ fr_out,
                    int n_items, int per_xcd) { gl_iter_wave_body(g, y, spec, fr_out, n_items, per_xcd); }
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
gl_iter_wave4_kernel(const VocGeom g, const float* __restrict__ y, const float* __restrict__ spec, float* __restrict__ fr_out,
                     int n_items, int per_xcd) { gl_iter_wave_body(g, y, spec, fr_out, n_items, per_xcd); }

// grid (ceil(L / 256), B): y (B, L) = istft(X)[n_fft/2 : -n_fft/2] from the stored frames.  The frames that can reach a block of
// 256 consecutive samples are a block-uniform range (scalar divisions, <= dmax + 2 of them); each lane just tests its index.
__global__ void __launch_bounds__(256) ola_kernel(const VocGeom g, const float* __restrict__ fr, float* __restrict__ y) {
  const int b = blockIdx.y;
  const int m0 = blockIdx.x * 256 + VOC_NFFT / 2;           // first y_full index of the block (uniform)
  const int lo_num = m0 - (g.lpad + g.win - 1);
  int q_lo = lo_num <= 0 ? 0 : (lo_num + g.hop - 1) / g.hop;
  int q_hi = (m0 + 255 - g.lpad) / g.hop;
  if (q_hi > g.F - 1) q_hi = g.F - 1;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= g.L) return;
  const int m = n + VOC_NFFT / 2;
  const float* fr_b = fr + (long)b * g.F * g.frs;
  float acc = 0.f;
  for (int q = q_lo; q <= q_hi; ++q) {                       // ascending frame order, as librosa.istft accumulates
    const int idx = m - q * g.hop - g.lpad;
    if (idx >= 0 && idx < g.win) acc = acc + fr_b[(long)q * g.frs + idx];
  }
  const float ws = g.wss[m];
  y[(long)b * g.L + n] = ws > g.tiny ? acc / ws : acc;
}

// utils.py:79-86: spec = (10 ** ((clip(mag,0,1) * max_db - max_db + ref_db) * 0.05)) ** power
__global__ void denorm_kernel(const float* __restrict__ mag, float* __restrict__ spec, long n, float max_db, float ref_db, float power) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = fminf(fmaxf(mag[i], 0.f), 1.f) * max_db - max_db + ref_db;
  v = powf(10.0f, v * 0.05f);
  spec[i] = powf(v, power);
}

// y[n] = x[n] + a * y[n-1].  grid (ceil(L / DE_CHUNK), B), 256 threads x DE_RUN samples; each workgroup re-runs DE_WARM samples
// before its chunk from a zero state (a^1024 = 3e-14: far below fp32 resolution of the carried state).
constexpr int DE_RUN = 20, DE_CHUNK = 4096, DE_WARM = 256 * DE_RUN - DE_CHUNK;
__global__ void __launch_bounds__(256) deemph_kernel(const float* __restrict__ x, float* __restrict__ y, int L, float a) {
  __shared__ float sA[256], sB[256];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int c0 = blockIdx.x * DE_CHUNK;
  const int s0 = c0 - DE_WARM + tid * DE_RUN;            // first sample of this thread's run (may be negative: zeros)
  const float* xb = x + (long)b * L;
  float loc[DE_RUN];
  float prev = 0.f;
#pragma unroll
  for (int i = 0; i < DE_RUN; ++i) {
    const int n = s0 + i;
    const float v = (n >= 0 && n < L) ? xb[n] : 0.f;
    prev = v + a * prev;
    loc[i] = prev;
  }
  float ad = 1.f;
#pragma unroll
  for (int i = 0; i < DE_RUN; ++i) ad *= a;              // a ^ DE_RUN
  float A = ad, Bv = prev;                               // state_out = Bv + A * state_in
  sA[tid] = A; sB[tid] = Bv;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    float pa = 1.f, pb = 0.f;
    const bool has = tid >= o;
    if (has) { pa = sA[tid - o]; pb = sB[tid - o]; }
    __syncthreads();
    if (has) { Bv = Bv + A * pb; A = A * pa; sA[tid] = A; sB[tid] = Bv; }
    __syncthreads();
  }
  const float carry = tid > 0 ? sB[tid - 1] : 0.f;
  float* yb = y + (long)b * L;
  float pw = a;
#pragma unroll
  for (int i = 0; i < DE_RUN; ++i) {
    const int n = s0 + i;
    if (n >= c0 && n < L) yb[n] = loc[i] + pw * carry;
    pw *= a;
  }
}

// librosa.feature.rmse ** 2 (center=True, reflect): grid (n_tf, B), block 256.  pw (B, n_tf).
__global__ void __launch_bounds__(256) frame_power_kernel(const float* __restrict__ y, float* __restrict__ pw, int L, int n_tf,
                                                          int flen, int fhop) {
  __shared__ float red[256];
  const int tid = threadIdx.x, j = blockIdx.x, b = blockIdx.y;
  const float* yb = y + (long)b * L;
  float acc = 0.f;
  for (int i = tid; i < flen; i += 256) {
    int n = j * fhop + i - flen / 2;
    if (n < 0) n = -n; else if (n >= L) n = 2 * (L - 1) - n;
    const float v = yb[n];
    acc += v * v;
  }
  red[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) pw[(long)b * n_tf + j] = red[0] / (float)flen;
}

// librosa.effects.trim: frames with 10 log10(max(1e-10, p)) - 10 log10(max(1e-10, max p)) > -top_db; grid (B), block 256.
__global__ void __launch_bounds__(256) trim_bounds_kernel(const float* __restrict__ pw, int* __restrict__ bounds, int L, int n_tf,
                                                          int fhop, float top_db) {
  __shared__ float smax[256];
  __shared__ int sfirst[256], slast[256];
  const int tid = threadIdx.x, b = blockIdx.x;
  const float* p = pw + (long)b * n_tf;
  float mx = 0.f;
  for (int j = tid; j < n_tf; j += 256) mx = fmaxf(mx, p[j]);
  smax[tid] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) smax[tid] = fmaxf(smax[tid], smax[tid + o]); __syncthreads(); }
  const float ref = 10.0f * log10f(fmaxf(1e-10f, smax[0]));
  int first = 0x7fffffff, last = -1;
  for (int j = tid; j < n_tf; j += 256) {
    const float db = 10.0f * log10f(fmaxf(1e-10f, p[j])) - ref;
    if (db > -top_db) { if (j < first) first = j; if (j > last) last = j; }
  }
  sfirst[tid] = first; slast[tid] = last;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { sfirst[tid] = min(sfirst[tid], sfirst[tid + o]); slast[tid] = max(slast[tid], slast[tid + o]); }
    __syncthreads();
  }
  if (tid == 0) {
    int s = 0, e = 0;
    if (slast[0] >= 0) { s = sfirst[0] * fhop; e = (slast[0] + 1) * fhop; if (e > L) e = L; }
    bounds[2 * b] = s; bounds[2 * b + 1] = e;
  }
}

}  // namespace dctts
