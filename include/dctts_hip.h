/* dctts_hip.h -- C ABI of the MI355X-native DC-TTS synthesis path (libdctts_hip.so).
 *
 * The reference (Kyubyong/dc_tts) has no FFI / plugin registry; the seam it offers is the Python
 * function surface of networks.py, called from train.py:55,58,64,68,77 and driven by
 * synthesize.py:47-57.  Each entry point below replaces one of those call sites and is what a
 * binding (ctypes / cgo / JNI) would target; the Python host layer dc_tts_amd/networks.py is one
 * such binding and mirrors the reference's names and argument order.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer on the context's GPU unless it says "host";
 *   - tensors are channel-last, contiguous: (B, time, C) float32; ids/indices int32 in,
 *     max_attentions int64 out (tf.argmax default), exactly as the reference graph;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are stream-ordered
 *     and never synchronise the device, except where noted (weight upload; dropping the workspace
 *     cache once it outgrows dctts_set_workspace_limit);
 *   - streams and threads (round 5): any entry point may be called from any stream and any host thread.
 *     Calls of one context that share scratch memory are ORDERED on the device whatever streams they come
 *     from -- TextEnc calls with each other and with decodes, decodes with each other, SSRN calls with each
 *     other (a later call waits for the earlier one's completion event) -- so results never depend on what
 *     else is in flight; calls of DIFFERENT kinds on different streams overlap (SSRN of batch n beside the
 *     decode of batch n + 1).  The host side of a context is serialised by a mutex (enqueue only);
 *   - return value 0 = ok, negative = dctts_status; nothing throws across the ABI;
 *   - only training=False (inference) semantics exist: dropout (modules.py:139,195,245) is identity.
 */
#ifndef DCTTS_HIP_H
#define DCTTS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dctts_ctx dctts_ctx;

typedef enum {
  DCTTS_OK = 0,
  DCTTS_ERR_ARG = -1,       /* bad pointer / shape / mode (the reference would raise from TF) */
  DCTTS_ERR_HIP = -2,       /* a HIP runtime call failed: see dctts_last_error() */
  DCTTS_ERR_WEIGHTS = -3,   /* missing / mis-shaped variable, or weights not finalized */
  DCTTS_ERR_STATE = -4
} dctts_status;

/* hyperparams.py:19,14,29-32,38-39 -- the constants the kernels are specialised on */
typedef struct {
  int vocab_size;   /* len(hp.vocab) = 32 */
  int e;            /* 128 */
  int d;            /* 256 */
  int c;            /* 512 */
  int n_mels;       /* 80 */
  int n_linear;     /* 1 + n_fft/2 = 1025 */
  int max_N;        /* 180 */
  int attention_win_size; /* 3 */
} dctts_config;

/* Lifetime.  One context per (process, GPU); thread-safe per context (see "streams and threads" above). */
int dctts_create(dctts_ctx** out, int device, const dctts_config* cfg);
int dctts_destroy(dctts_ctx* ctx);
const char* dctts_last_error(void);

/* Weights: replaces the implicit tf.get_variable / tf.layers variables restored by name in
 * synthesize.py:32-40.  `name` is the TF variable name, e.g.
 * "Text2Mel/AudioEnc/HC_7/conv1d/kernel"; `data` is a HOST float32 array in TF layout
 * (conv kernel (k,Cin,Cout); conv2d_transpose kernel (1,k,Cout,Cin)).  finalize packs every
 * kernel into MFMA fragment order and uploads (synchronises the device). */
int dctts_weights_set(dctts_ctx* ctx, const char* name, const float* data, const int64_t* shape, int ndim);
int dctts_weights_finalize(dctts_ctx* ctx);

/* networks.py:14  TextEnc(L) -> (K, V).  L (B,N) int32; K,V (B,N,d). */
int dctts_textenc_fwd(dctts_ctx* ctx, const int32_t* L, int B, int N, float* K, float* V, void* stream);
/* networks.py:73  AudioEnc(S) -> Q.  S (B,T,n_mels); Q (B,T,d).  Causal. */
int dctts_audioenc_fwd(dctts_ctx* ctx, const float* S, int B, int T, float* Q, void* stream);
/* networks.py:126 Attention(Q,K,V,mononotic_attention,prev_max_attentions) -> (R, alignments, max_attentions).
 * Q (B,T,d); K,V (B,N,d); prev_max (B,) int32 (required iff monotonic); R (B,T,2d);
 * alignments (B,N,T) or NULL; max_attentions (B,T) int64 or NULL.  Monotonic mode requires
 * N == max_N (the mask is built from hp.max_N, networks.py:142-143). */
int dctts_attention_fwd(dctts_ctx* ctx, const float* Q, const float* K, const float* V, int B, int T, int N,
                        int monotonic, const int32_t* prev_max, float* R, float* alignments,
                        int64_t* max_attentions, void* stream);
/* networks.py:157 AudioDec(R) -> (logits, Y).  R (B,T,2d); logits,Y (B,T,n_mels). */
int dctts_audiodec_fwd(dctts_ctx* ctx, const float* R, int B, int T, float* logits, float* Y, void* stream);
/* networks.py:214 SSRN(Y) -> (logits, Z).  Y (B,T,n_mels); logits,Z (B,4T,n_linear); logits may be NULL. */
int dctts_ssrn_fwd(dctts_ctx* ctx, const float* Y, int B, int T, float* logits, float* Z, void* stream);

/* synthesize.py:45-54: the autoregressive Text2Mel loop for max_T = T steps, as an incremental
 * decoder that is arithmetically the reference's full-recompute loop (TextEnc once, AudioEnc
 * incrementally, windowed attention + the 85-row AudioDec dependency cone re-evaluated with the
 * current window at every step).  L (B,N) int32, N == max_N.  Y (B,T,n_mels) out;
 * max_attentions (B,T) int64 out or NULL (column j = the value fed back as prev_max at step j+1);
 * alignments (B,N,T) out or NULL = `g.alignments` as the loop's LAST sess.run fetches it (synthesize.py:48,
 * networks.py:153): every time row against the window of step T-1.
 * A decode that fails on the device (see dctts_decode_status) overwrites Y / alignments with NaN and
 * max_attentions with -1 before the call's work on `stream` ends. */
int dctts_text2mel_decode(dctts_ctx* ctx, const int32_t* L, int B, int N, int T, float* Y,
                          int64_t* max_attentions, float* alignments, void* stream);
/* synthesize.py:45-57: decode + one SSRN pass.  Z (B,4T,n_linear) (NaN as well after a failed decode). */
int dctts_synthesize(dctts_ctx* ctx, const int32_t* L, int B, int N, int T, float* Y, float* Z,
                     int64_t* max_attentions, float* alignments, void* stream);

/* Status of the decodes issued so far on this context; call after synchronising their stream.  The default decode form runs two
 * kernels per frame whose workgroups wait for each other inside the launch (bounded waits, about a second at most: a stalled
 * stream must not hang the queue).  When a wait gives up -- the GPU is shared with another process's kernels, or a team of
 * workgroups was not placed on one XCD -- that decode is INVALID: its last kernel overwrites its outputs with NaN / -1 and raises a
 * sticky status word on the device that no later decode clears.  This call returns DCTTS_ERR_STATE once for all decodes that
 * failed since the previous call (the message has the count and the error bits) and clears the word; a failure nobody asked
 * about makes the next decode call on the context fail instead of running.  Time-outs leave the team kernels on (three failed
 * reports in a row switch them off); a misplaced team switches them off for good.  dc_tts_amd.Engine.synchronize() calls this;
 * Engine.text2mel / synthesize(check=True) also repeat a failed decode once in the safe form (dctts_decode_safe_once). */
int dctts_decode_status(dctts_ctx* ctx);

/* The NEXT decode on this context (only that one) runs in the form that cannot time out: one launch per layer, and the two
 * decode streams meet through stream wait / write operations instead of bounded in-kernel waits (~1.5x the frame time).  The
 * persistent settings -- dctts_set_team_kernels, a switch-off by dctts_decode_status -- are left exactly as they were.  This is
 * what a caller does after dctts_decode_status reported a failed decode (the GPU is shared with somebody else's kernels). */
int dctts_decode_safe_once(dctts_ctx* ctx);

/* 1 (default): runs of dependent layers of a decode frame are ONE launch whose workgroups meet inside an XCD's L2
 * (csrc/xgroup_kernel.h, xcone_kernel.h); 0: one launch per layer (no hand-offs between the workgroups of a launch; ~1.3x the frame time). */
int dctts_set_team_kernels(dctts_ctx* ctx, int enable);

/* Decode launch mode: 0 (default) = every launch eager; 1 = the side-stream (bulk) work of each frame is one hipGraph launch,
 * the latency-critical chain launches stay eager.  Measured on MI355X: a graph launch costs ~10 us of start-up on a path that is
 * ~100 us per frame, so the eager form is the faster one and the default. */
int dctts_set_decode_graph(dctts_ctx* ctx, int enable);

/* Decode algorithm form (results agree to fp32 re-association; both are the exact-parity incremental decode):
 * 3 = (default) column-split chain kernels that contract only each layer's centre tap on the caller's stream (older taps arrive as
 *     presums computed on a side stream), AudioDec C_1 / HC_2 cone rows as row operations on cached V.W / Q.W products
 *     (csrc/decode3_kernels.h),
 * 0 = fused full-row kernels on one stream with a device-side frame counter (one workgroup per 32-row block): the simplest form, a
 *     different implementation of the same arithmetic, kept as a cross-check.
 * The forms measured and dropped in rounds 1-2 (two-stream split kernels, one row-split launch per chain piece, persistent highway
 * groups, in-kernel gates) are in the git history up to commit 19f2cbe. */
int dctts_set_decode_mode(dctts_ctx* ctx, int mode);

/* Device memory the context holds for the shapes seen so far (weights + workspaces), bytes. */
size_t dctts_device_bytes(const dctts_ctx* ctx);

/* OPT-IN reduced-cost contraction for the two throughput networks; the default (0) is exact fp32 everywhere and is what every number quoted as the
 * metric is measured with.  mode 1: SSRN's convolutions, mode 2: SSRN's and TextEnc's run on the bf16 matrix pipe from SPLIT operands -- every fp32
 * weight and activation is written as hi + mid (two bf16 terms, round-to-nearest) and a product is accumulated in fp32 as hi.hi + hi.mid + mid.hi
 * (v_mfma_f32_32x32x16_bf16: 16x the fp32 matrix rate, three instructions instead of eight per 16 k); what is dropped is <= 2^-16 of |x||w| per product.
 * Bias, layer-norm, gate, activations and the 1025th column stay fp32; AudioEnc / Attention / AudioDec (the decode, whose arg-max is fed back:
 * synthesize.py:52-54) always run in fp32, so with mode 1 the mel output and the attention trajectory are bit-identical to mode 0.  Call it BEFORE
 * dctts_weights_finalize with the highest level wanted (the weights get a second, bf16 packing); afterwards any level up to that one can be selected per call
 * sequence.  Measured error against the float64 oracle: tests/test_gpu_parity.py::test_split_bf16_*, DESIGN.md section 11. */
int dctts_set_split_bf16(dctts_ctx* ctx, int mode);

/* Workspaces and the decode's device tables are cached per geometry (B, T, N) and only grow: a serving loop that alternates
 * between batch shapes pays an allocation the first time a shape is seen and nothing afterwards (no hipDeviceSynchronize, no
 * hipFree on a shape change).  When the cached workspaces exceed `bytes` (default 96 GiB of the 288 GB) the next call drops
 * all of them behind ONE device synchronisation and starts again. */
int dctts_set_workspace_limit(dctts_ctx* ctx, size_t bytes);

/* Test / measurement hooks (per-layer test entry, calibration copy, kernel timing) are declared in
 * dctts_hip_debug.h: they are not part of the drop-in surface. */

/* ------------------------------------------------------------------------------------------------
 * Vocoder tail: utils.py:67-114 `spectrogram2wav` / `griffin_lim` / `invert_spectrogram`, which
 * synthesize.py:61-64 calls once per utterance on the CPU (librosa + scipy).  Here: a batch at a time on the GPU.
 * It owns no weights, so it is its own handle.  Spectrogram tensors stay frame-major (B, F, 1 + n_fft/2) -- the layout
 * dctts_ssrn_fwd writes (F = 4T) -- not librosa's (1 + n_fft/2, F). */
typedef struct dctts_vocoder dctts_vocoder;

/* hyperparams.py:13-24 + the librosa.effects.trim defaults the reference relies on (utils.py:92) */
typedef struct {
  int n_fft;            /* 2048 (the FFT kernels are specialised on it) */
  int hop_length;       /* 275 */
  int win_length;       /* 1102 */
  int n_iter;           /* 50 */
  float power;          /* 1.5 */
  float preemphasis;    /* 0.97 */
  float max_db;         /* 100 */
  float ref_db;         /* 20 */
  float trim_top_db;    /* 60 */
  int trim_frame_length;/* 2048 */
  int trim_hop_length;  /* 512 */
} dctts_vocoder_config;

int dctts_vocoder_create(dctts_vocoder** out, int device, const dctts_vocoder_config* cfg);
int dctts_vocoder_destroy(dctts_vocoder* v);
size_t dctts_vocoder_device_bytes(const dctts_vocoder* v);

/* utils.py:67-94 spectrogram2wav for B utterances.  mag (B,F,1+n_fft/2): network output in [0,1] (Z).
 * wav (B, hop_length*(F-1)) float32: de-normalise -> **power -> Griffin-Lim (n_iter) -> de-emphasis, NOT trimmed;
 * bounds (B,2) int32 or NULL: [start, end) of librosa.effects.trim(wav) per utterance (the reference returns
 * wav[start:end]).  Requires hop_length*(F-1) > n_fft/2 (reflect padding; librosa raises below that too). */
int dctts_spectrogram2wav(dctts_vocoder* v, const float* mag, int B, int F, float* wav, int32_t* bounds, void* stream);

/* utils.py:96-106 griffin_lim on magnitudes.  spec (B,F,1+n_fft/2) (already de-normalised, any non-negative values);
 * y (B, hop_length*(F-1)); X_best (B,F,1+n_fft/2,2) interleaved complex64 or NULL = the spectrogram*phase the last
 * iteration produced (needs n_iter >= 1).  n_iter = 0 gives y = istft(spec). */
int dctts_griffin_lim(dctts_vocoder* v, const float* spec, int B, int F, int n_iter, float* y, float* X_best, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCTTS_HIP_H */
