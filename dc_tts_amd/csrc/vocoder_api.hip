// vocoder_api.hip -- host side of the vocoder tail (include/dctts_hip.h: dctts_vocoder_*, dctts_spectrogram2wav,
// dctts_griffin_lim).  Replaces utils.py:67-114 as called per utterance from synthesize.py:61-64.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <utility>
#include <mutex>
#include <vector>

#include "../../include/dctts_hip.h"
#include "api_common.h"
#include "vocoder_kernels.h"

using namespace dctts;

#define VHIPCHK(x)                                                                                         \
  do {                                                                                                     \
    hipError_t e__ = (x);                                                                                  \
    if (e__ != hipSuccess)                                                                                 \
      return dctts_set_error(DCTTS_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e__) + " @" + std::to_string(__LINE__)); \
  } while (0)
#define VCHK(x) do { int r__ = (x); if (r__ != 0) return r__; } while (0)

struct VBuf { void* p = nullptr; size_t bytes = 0; };

struct dctts_vocoder {
  dctts_vocoder_config cfg;
  int device = 0;
  int lpad = 0, frs = 0;
  float *window = nullptr, *wss = nullptr;
  float2 *w1024 = nullptr, *w2048 = nullptr;
  int wss_frames = 0;
  int wave_kernel = 1;     // register budget of the iteration kernel: 1 = 3 waves/SIMD without spills, 2 = 4 waves/SIMD (DCTTS_VOC_WAVE)
  VBuf spec, X, fr0, fr1, yraw, pw;
  bool prof = false;                                  // HIP events around every gl_iter_wave launch (dctts_vocoder_prof_*)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
  // One set of scratch buffers per handle: calls from different streams are ordered on the device (a call waits for the completion event of the last call that
  // came from another stream), the host side is serialised by a mutex (round 5; same rule as a dctts_ctx's use groups)
  std::mutex mu; hipEvent_t done = nullptr; hipStream_t last = nullptr; bool used = false;
};

static int voc_acquire(dctts_vocoder* v, hipStream_t st) {
  if (v->used) VHIPCHK(hipStreamWaitEvent(st, v->done, 0));      // (always: a destroyed stream's handle can come back as a new stream's)
  return 0;
}
static int voc_release(dctts_vocoder* v, hipStream_t st) {
  if (!v->done) VHIPCHK(hipEventCreateWithFlags(&v->done, hipEventDisableTiming));
  VHIPCHK(hipEventRecord(v->done, st));
  v->last = st; v->used = true;
  return 0;
}

// A call's hold on the handle's scratch: the completion event is recorded on every way out once anything may have been enqueued (an early error return
// must not leave launches on the shared buffers that the next call from another stream does not wait for).
struct VocGuard {
  dctts_vocoder* v; hipStream_t st; bool held = false;
  VocGuard(dctts_vocoder* v_, hipStream_t st_) : v(v_), st(st_) {}
  int acquire() { const int rc = voc_acquire(v, st); held = (rc == 0); return rc; }
  int release() { held = false; return voc_release(v, st); }
  ~VocGuard() { if (held) (void)voc_release(v, st); }
};

static int vgrow(VBuf& b, size_t bytes) {
  if (b.bytes >= bytes) return 0;
  if (b.p) { VHIPCHK(hipDeviceSynchronize()); VHIPCHK(hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
  VHIPCHK(hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  return 0;
}

extern "C" int dctts_vocoder_create(dctts_vocoder** out, int device, const dctts_vocoder_config* cfg) {
  if (!out || !cfg) return dctts_set_error(DCTTS_ERR_ARG, "null argument");
  if (cfg->n_fft != VOC_NFFT) return dctts_set_error(DCTTS_ERR_ARG, "the FFT kernels are specialised for n_fft == 2048");
  if (cfg->win_length < 2 || cfg->win_length > cfg->n_fft || cfg->hop_length < 1 || cfg->hop_length > cfg->win_length)
    return dctts_set_error(DCTTS_ERR_ARG, "need 1 <= hop_length <= win_length <= n_fft");
  if (cfg->n_iter < 0 || cfg->trim_frame_length < 2 || cfg->trim_hop_length < 1)
    return dctts_set_error(DCTTS_ERR_ARG, "bad n_iter / trim geometry");
  VHIPCHK(hipSetDevice(device));
  dctts_vocoder* v = new dctts_vocoder();
  v->cfg = *cfg; v->device = device;
  if (const char* e = getenv("DCTTS_VOC_WAVE")) v->wave_kernel = atoi(e);
  v->lpad = (cfg->n_fft - cfg->win_length) / 2;                      // librosa.util.pad_center
  v->frs = (cfg->win_length + 3) & ~3;
  // periodic Hann (scipy get_window('hann', win, fftbins=True)), evaluated in double, rounded once
  std::vector<float> w(VOC_NFFT, 0.f);
  for (int n = 0; n < cfg->win_length; ++n)
    w[v->lpad + n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * (double)n / (double)cfg->win_length));
  std::vector<float2> t1(VOC_M), t2(VOC_M + 1);
  for (int m = 0; m < VOC_M; ++m) t1[m] = make_float2((float)std::cos(2.0 * M_PI * m / VOC_M), (float)-std::sin(2.0 * M_PI * m / VOC_M));
  for (int k = 0; k <= VOC_M; ++k) t2[k] = make_float2((float)std::cos(2.0 * M_PI * k / VOC_NFFT), (float)-std::sin(2.0 * M_PI * k / VOC_NFFT));
  VHIPCHK(hipMalloc((void**)&v->window, w.size() * sizeof(float)));
  VHIPCHK(hipMalloc((void**)&v->w1024, t1.size() * sizeof(float2)));
  VHIPCHK(hipMalloc((void**)&v->w2048, t2.size() * sizeof(float2)));
  VHIPCHK(hipMemcpy(v->window, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
  VHIPCHK(hipMemcpy(v->w1024, t1.data(), t1.size() * sizeof(float2), hipMemcpyHostToDevice));
  VHIPCHK(hipMemcpy(v->w2048, t2.data(), t2.size() * sizeof(float2), hipMemcpyHostToDevice));
  *out = v;
  return 0;
}

extern "C" int dctts_vocoder_destroy(dctts_vocoder* v) {
  if (!v) return 0;
  (void)hipSetDevice(v->device);
  (void)hipDeviceSynchronize();
  for (auto& e : v->prof_ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (VBuf* b : {&v->spec, &v->X, &v->fr0, &v->fr1, &v->yraw, &v->pw}) if (b->p) (void)hipFree(b->p);
  if (v->done) (void)hipEventDestroy(v->done);
  if (v->window) (void)hipFree(v->window);
  if (v->wss) (void)hipFree(v->wss);
  if (v->w1024) (void)hipFree(v->w1024);
  if (v->w2048) (void)hipFree(v->w2048);
  delete v;
  return 0;
}

// istft's divisor for F frames: the squared window overlap-added in frame order, in fp32 (librosa window_sumsquare).
static int ensure_wss(dctts_vocoder* v, int F) {
  if (v->wss_frames == F) return 0;
  const int hop = v->cfg.hop_length, win = v->cfg.win_length;
  std::vector<float> w2(VOC_NFFT, 0.f);
  for (int n = 0; n < win; ++n) {
    const float w = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * (double)n / (double)win));
    w2[v->lpad + n] = w * w;
  }
  std::vector<float> acc((size_t)VOC_NFFT + (size_t)hop * (F - 1), 0.f);
  for (int i = 0; i < F; ++i) {
    float* a = acc.data() + (size_t)i * hop;
    for (int n = v->lpad; n < v->lpad + win; ++n) a[n] = a[n] + w2[n];
  }
  VHIPCHK(hipDeviceSynchronize());
  if (v->wss) { VHIPCHK(hipFree(v->wss)); v->wss = nullptr; }
  VHIPCHK(hipMalloc((void**)&v->wss, acc.size() * sizeof(float)));
  VHIPCHK(hipMemcpy(v->wss, acc.data(), acc.size() * sizeof(float), hipMemcpyHostToDevice));
  v->wss_frames = F;
  return 0;
}

static int geom(dctts_vocoder* v, int B, int F, VocGeom* g) {
  if (B < 1 || F < 1) return dctts_set_error(DCTTS_ERR_ARG, "B and F must be positive");
  const long L = (long)v->cfg.hop_length * (F - 1);
  if (L <= VOC_NFFT / 2) return dctts_set_error(DCTTS_ERR_ARG, "too few frames: reflect padding needs hop*(F-1) > n_fft/2 (librosa raises too)");
  if (L > 0x3fffffff) return dctts_set_error(DCTTS_ERR_ARG, "utterance too long");
  VCHK(ensure_wss(v, F));
  g->F = F; g->L = (int)L; g->hop = v->cfg.hop_length; g->win = v->cfg.win_length; g->lpad = v->lpad; g->frs = v->frs;
  g->window = v->window; g->wss = v->wss; g->w1024 = v->w1024; g->w1024i = v->w1024; g->w2048 = v->w2048;
  g->tiny = std::numeric_limits<float>::min();
  g->dmax = (v->cfg.win_length + v->cfg.hop_length - 1) / v->cfg.hop_length;
  return 0;
}

// utils.py:96-106 on device buffers.  spec (B,F,1025) magnitudes -> y (B, L).
static int run_griffin_lim(dctts_vocoder* v, const VocGeom& g, const float* spec, int B, int n_iter, float2* X_best,
                           float* y, hipStream_t st) {
  VCHK(vgrow(v->fr0, (size_t)B * g.F * g.frs * sizeof(float)));
  float* fr = (float*)v->fr0.p;
  const dim3 grid(g.F, B), ogrid((g.L + 255) / 256, B);
  const int n_items = B * g.F, per_xcd = (n_items + 7) / 8;
  hipLaunchKernelGGL(istft_frames_kernel, grid, dim3(VOC_THREADS), 0, st, g, (const float2*)nullptr, spec, fr);
  hipLaunchKernelGGL(ola_kernel, ogrid, dim3(256), 0, st, g, (const float*)fr, y);
  for (int it = 0; it < n_iter; ++it) {
    if (X_best && it == n_iter - 1) {      // the caller wants the last X_best: un-fused pair for this iteration
      hipLaunchKernelGGL(stft_phase_kernel, grid, dim3(VOC_THREADS), 0, st, g, (const float*)fr, spec, X_best);
      VCHK(vgrow(v->fr1, (size_t)B * g.F * g.frs * sizeof(float)));
      hipLaunchKernelGGL(istft_frames_kernel, grid, dim3(VOC_THREADS), 0, st, g, (const float2*)X_best, spec, (float*)v->fr1.p);
      hipLaunchKernelGGL(ola_kernel, ogrid, dim3(256), 0, st, g, (const float*)v->fr1.p, y);
      break;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (v->prof) { VHIPCHK(hipEventCreate(&e0)); VHIPCHK(hipEventCreate(&e1)); VHIPCHK(hipEventRecord(e0, st)); }
    if (v->wave_kernel == 2)
      hipLaunchKernelGGL(gl_iter_wave4_kernel, dim3(8 * per_xcd), dim3(64), 0, st, g, (const float*)y, spec, fr, n_items, per_xcd);
    else
      hipLaunchKernelGGL(gl_iter_wave_kernel, dim3(8 * per_xcd), dim3(64), 0, st, g, (const float*)y, spec, fr, n_items, per_xcd);
    if (v->prof) { VHIPCHK(hipEventRecord(e1, st)); v->prof_ev.emplace_back(e0, e1); }
    hipLaunchKernelGGL(ola_kernel, ogrid, dim3(256), 0, st, g, (const float*)fr, y);
  }
  VHIPCHK(hipGetLastError());
  return 0;
}

extern "C" int dctts_griffin_lim(dctts_vocoder* v, const float* spec, int B, int F, int n_iter, float* y, float* X_best, void* stream) {
  if (!v || !spec || !y) return dctts_set_error(DCTTS_ERR_ARG, "null argument");
  if (n_iter < 0) return dctts_set_error(DCTTS_ERR_ARG, "n_iter < 0");
  if (X_best && n_iter == 0) return dctts_set_error(DCTTS_ERR_ARG, "X_best needs n_iter >= 1");
  VHIPCHK(hipSetDevice(v->device));
  std::lock_guard<std::mutex> lk(v->mu);
  hipStream_t st = (hipStream_t)stream;
  VocGuard guard(v, st);
  VCHK(guard.acquire());
  VocGeom g;
  VCHK(geom(v, B, F, &g));
  VCHK(run_griffin_lim(v, g, spec, B, n_iter, (float2*)X_best, y, st));
  return guard.release();
}

extern "C" int dctts_spectrogram2wav(dctts_vocoder* v, const float* mag, int B, int F, float* wav, int32_t* bounds, void* stream) {
  if (!v || !mag || !wav) return dctts_set_error(DCTTS_ERR_ARG, "null argument");
  VHIPCHK(hipSetDevice(v->device));
  std::lock_guard<std::mutex> lk(v->mu);
  hipStream_t st = (hipStream_t)stream;
  VocGuard guard(v, st);
  VCHK(guard.acquire());
  VocGeom g;
  VCHK(geom(v, B, F, &g));
  if (bounds && g.L <= v->cfg.trim_frame_length / 2) return dctts_set_error(DCTTS_ERR_ARG, "utterance shorter than the trim frame's reflect padding");      // (before anything is launched)
  const long n = (long)B * F * VOC_BINS;
  VCHK(vgrow(v->spec, (size_t)n * sizeof(float)));
  VCHK(vgrow(v->yraw, (size_t)B * g.L * sizeof(float)));
  float* spec = (float*)v->spec.p; float* yraw = (float*)v->yraw.p;
  hipLaunchKernelGGL(denorm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mag, spec, n, v->cfg.max_db, v->cfg.ref_db, v->cfg.power);
  VCHK(run_griffin_lim(v, g, spec, B, v->cfg.n_iter, nullptr, yraw, st));
  hipLaunchKernelGGL(deemph_kernel, dim3((g.L + DE_CHUNK - 1) / DE_CHUNK, B), dim3(256), 0, st, (const float*)yraw, wav, g.L, v->cfg.preemphasis);
  if (bounds) {
    const int flen = v->cfg.trim_frame_length, fhop = v->cfg.trim_hop_length;
    const int n_tf = 1 + g.L / fhop;
    VCHK(vgrow(v->pw, (size_t)B * n_tf * sizeof(float)));
    hipLaunchKernelGGL(frame_power_kernel, dim3(n_tf, B), dim3(256), 0, st, (const float*)wav, (float*)v->pw.p, g.L, n_tf, flen, fhop);
    hipLaunchKernelGGL(trim_bounds_kernel, dim3(B), dim3(256), 0, st, (const float*)v->pw.p, (int*)bounds, g.L, n_tf, fhop, v->cfg.trim_top_db);
  }
  VHIPCHK(hipGetLastError());
  return guard.release();
}

extern "C" int dctts_vocoder_prof_enable(dctts_vocoder* v, int enable) {
  if (!v) return dctts_set_error(DCTTS_ERR_ARG, "null handle");
  v->prof = enable != 0;
  return 0;
}

extern "C" int dctts_vocoder_prof_collect(dctts_vocoder* v, int* launches, double* total_ms) {
  if (!v || !launches || !total_ms) return dctts_set_error(DCTTS_ERR_ARG, "null argument");
  double tot = 0; int n = 0;
  for (auto& e : v->prof_ev) {
    VHIPCHK(hipEventSynchronize(e.second));
    float ms = 0.f;
    VHIPCHK(hipEventElapsedTime(&ms, e.first, e.second));
    tot += ms; ++n;
    (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
  }
  v->prof_ev.clear();
  *launches = n; *total_ms = tot;
  return 0;
}

extern "C" size_t dctts_vocoder_device_bytes(const dctts_vocoder* v) {
  if (!v) return 0;
  return v->spec.bytes + v->X.bytes + v->fr0.bytes + v->fr1.bytes + v->yraw.bytes + v->pw.bytes;
}
