// vocoder_kernels.h -- the reference's vocoder tail (utils.py:67-114: spectrogram2wav / griffin_lim / invert_spectrogram)
// as HIP kernels for gfx950.  librosa.stft / istft are restated in oracle/vocoder_ref.py; these kernels follow that.
//
// Griffin-Lim is 50 x (istft -> stft -> keep the phase).  With n_fft = 2048, hop = 275, win = 1102 this is FFT + streaming
// work (HBM / LDS bound, no MFMA):  one workgroup owns one frame of one utterance and runs the 2048-point real transform
// as a 1024-point complex Stockham radix-4 FFT in LDS (5 passes, one butterfly per thread per pass).
//
//   istft_frames_kernel   X_best row (1025 complex) -> Hermitian pre-twist -> IFFT -> * window -> the 1102 samples under
//                         the window, stored per frame (`fr`, (B, F, FRS)); nothing is overlap-added with atomics.
//   stft_phase_kernel     gathers its 1102 windowed input samples straight from `fr` (overlap-add of <= 5 frames in frame
//                         order / window sum-square, reflect padding at the ends = librosa center=True) -> FFT ->
//                         post-twist -> est / max(1e-8,|est|) * magnitude -> X_best row.  The time signal never exists in
//                         HBM between the two transforms.
//   ola_kernel            final istft: the same gather, written out as the waveform.
//   deemph_kernel         scipy.signal.lfilter([1],[1,-a]) (utils.py:89) as a blocked linear-recurrence scan.
//   frame_power_kernel / trim_bounds_kernel   librosa.effects.trim's [start, end) (utils.py:92).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
  const float2* w2048;     // exp(-2 pi i k / 2048), k <= 1024
  float tiny;
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

// One whole Griffin-Lim iteration per launch (utils.py:100-103 + the istft of the next pass): frames_out[f] =
// window * irfft(spec * phase(stft(istft(frames_in))))[f].  The spectrum row lives only in LDS; `fr_in` / `fr_out` ping-pong.
__global__ void __launch_bounds__(VOC_THREADS) gl_iter_kernel(const VocGeom g, const float* __restrict__ fr_in,
                                                              const float* __restrict__ spec, float* __restrict__ fr_out) {
  __shared__ float2 s[2][VOC_M];
  __shared__ float2 s_nyq;
  const int tid = threadIdx.x, f = blockIdx.x, b = blockIdx.y;
  const float* fr_b = fr_in + (long)b * g.F * g.frs;
  auto sample = [&](int i) -> float {
    if (i < g.lpad || i >= g.lpad + g.win) return 0.f;
    int n = f * g.hop + i - VOC_NFFT / 2;
    if (n < 0) n = -n; else if (n >= g.L) n = 2 * (g.L - 1) - n;
    return ola_sample(g, fr_b, n + VOC_NFFT / 2) * g.window[i];
  };
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = tid + r * VOC_THREADS;
    s[0][n] = make_float2(sample(2 * n), sample(2 * n + 1));
  }
  __syncthreads();
  fft1024<false>(s, g.w1024, tid);                       // Z in s[1]
  const long row = ((long)b * g.F + f) * VOC_BINS;
  for (int k = tid; k <= VOC_M; k += VOC_THREADS) {
    const float2 a = s[1][k & (VOC_M - 1)], c = s[1][(VOC_M - k) & (VOC_M - 1)];
    const float2 E = make_float2(0.5f * (a.x + c.x), 0.5f * (a.y - c.y));
    const float2 D = make_float2(0.5f * (a.x - c.x), 0.5f * (a.y + c.y));
    const float2 Dw = cmul(D, g.w2048[k]);
    const float2 est = make_float2(E.x + Dw.y, E.y - Dw.x);
    const float mag = fmaxf(1e-8f, hypotf(est.x, est.y));
    const float sp = spec[row + k];
    float2 xb = make_float2(sp * (est.x / mag), sp * (est.y / mag));
    if (k == 0 || k == VOC_M) xb.y = 0.f;
    if (k == VOC_M) s_nyq = xb; else s[0][k] = xb;        // X_best row -> s[0] (+ Nyquist)
  }
  __syncthreads();
  float2 a4[4], c4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = tid + r * VOC_THREADS;
    a4[r] = s[0][k];
    c4[r] = (k == 0) ? s_nyq : s[0][VOC_M - k];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = tid + r * VOC_THREADS;
    const float2 a = a4[r], c = c4[r];
    const float2 E = make_float2(0.5f * (a.x + c.x), 0.5f * (a.y - c.y));
    const float2 D = make_float2(0.5f * (a.x - c.x), 0.5f * (a.y + c.y));
    const float2 O = cmulc(D, g.w2048[k]);
    s[0][k] = make_float2(E.x - O.y, E.y + O.x);
  }
  __syncthreads();
  fft1024<true>(s, g.w1024, tid);
  float* out = fr_out + ((long)b * g.F + f) * g.frs;
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

// grid (ceil(L / 256), B): y (B, L) = istft(X)[n_fft/2 : -n_fft/2] from the stored frames.
__global__ void __launch_bounds__(256) ola_kernel(const VocGeom g, const float* __restrict__ fr, float* __restrict__ y) {
  const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (n >= g.L) return;
  y[(long)b * g.L + n] = ola_sample(g, fr + (long)b * g.F * g.frs, n + VOC_NFFT / 2);
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
