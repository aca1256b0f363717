/* dctts_train.h -- C ABI of the TRAINING path, MI355X (gfx950).
 *
 * What is here: the forward and backward pass of every building block of networks.py (hc, conv1d, conv1d_transpose, embed, Attention), the losses of
 * train.py:85-110 with their gradients, and the clip + Adam update of train.py:119-131.  A trainer written against the
 * reference would call these where TensorFlow's autodiff / optimizer ran:
 *
 *   reference                                              this library
 *   modules.py:143-197  hc(...) under tf.gradients         dctts_train_hc_backward
 *   modules.py:91-141   conv1d(...) under tf.gradients     dctts_train_conv1d_backward
 *   modules.py:199-247  conv1d_transpose(...)              dctts_train_conv1d_transpose_backward
 *   networks.py:126-155 Attention (training form)          dctts_train_attention_backward
 *   modules.py:13-42    embed                              dctts_train_embed_backward
 *   train.py:87,90,93-97  loss_mels, loss_bd1, loss_att    dctts_train_text2mel_losses
 *   train.py:104,107      loss_mags, loss_bd2              dctts_train_ssrn_losses
 *   train.py:119-131      clip_by_value(-1, 1) + Adam      dctts_train_adam_step   (lr from utils.py:142-145, host side)
 *
 * Conventions are those of dctts_hip.h: extern "C", raw DEVICE pointers (fp32, channel-last (B, time, C)), sizes, a
 * hipStream_t as void*, integer status (0 = ok), dctts_last_error() for the message.  Nothing throws; no torch types.
 * All arithmetic is fp32 (contractions on v_mfma_f32_32x32x2_f32); reductions are two-stage and deterministic (no atomics).
 */
#ifndef DCTTS_TRAIN_H_
#define DCTTS_TRAIN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dctts_train dctts_train;

/* Workspaces are owned by the handle and grow with the largest shape seen. */
int dctts_train_create(dctts_train** out, int device);
int dctts_train_destroy(dctts_train* t);
size_t dctts_train_device_bytes(const dctts_train* t);

/* enable = 1: from now on the hc / conv1d FORWARD passes keep their pre-norm tensors on a tape inside the handle and the matching
 * BACKWARD passes, called in reverse order (as a reverse pass over a network does), consume them instead of recomputing (a third of a
 * backward pass's contraction work).  An entry is matched by the layer's kernel pointer and geometry; out of step, the backward pass
 * silently recomputes.  Every call (enable 0 or 1) empties the tape: call it at the start of each training step. */
int dctts_train_tape(dctts_train* t, int enable);

/* Backward of y = hc(x) (modules.py:143-197: conv1d(k, dilation `rate`, SAME or CAUSAL padding) to 2C channels -> split ->
 * layer-norm(H1), layer-norm(H2) -> sigmoid(H1) * H2 + (1 - sigmoid(H1)) * x, training=False dropout i.e. none).
 *   x, dy, dx            (B, T, C)          C a multiple of 256, k in {1, 3}
 *   kernel, dkernel      (k, C, 2C)         TF variable layout (conv1d/kernel)
 *   bias, dbias          (2C)
 *   g1, b1, g2, b2 and their gradients (C)  H1/gamma, H1/beta, H2/gamma, H2/beta
 * The forward pre-norm tensor is recomputed here (the inference kernels never store it). */
int dctts_train_hc_backward(dctts_train* t, const float* x, const float* dy, const float* kernel, const float* bias,
                            const float* g1, const float* b1, const float* g2, const float* b2,
                            int B, int T, int C, int k, int rate, int causal,
                            float* dx, float* dkernel, float* dbias, float* dg1, float* db1, float* dg2, float* db2, void* stream);

/* Backward of y = conv1d(x) (modules.py:91-141: conv1d(k, dilation `rate`, SAME or CAUSAL) Cin -> Cout, layer-norm, activation).
 *   x, dx (B, T, Cin);  dy (B, T, Cout), Cout <= 1088 (256 / 512 / 1024 take the vector path);  kernel, dkernel (k, Cin, Cout);
 *   bias, gamma, beta and their gradients (Cout);  act: 0 none, 1 relu, 2 sigmoid. */
int dctts_train_conv1d_backward(dctts_train* t, const float* x, const float* dy, const float* kernel, const float* bias,
                                const float* gamma, const float* beta, int B, int T, int Cin, int Cout, int k, int rate, int causal, int act,
                                float* dx, float* dkernel, float* dbias, float* dgamma, float* dbeta, void* stream);

/* Backward of y = conv1d_transpose(x) (modules.py:199-247: tf.layers.conv2d_transpose, kernel (1, 3, Cout, Cin), stride 2, 'same',
 * then layer-norm): x, dx (B, T, Cin); dy (B, 2T, Cout); kernel, dkernel (1, 3, Cout, Cin); Cin, Cout multiples of 4. */
int dctts_train_conv1d_transpose_backward(dctts_train* t, const float* x, const float* dy, const float* kernel, const float* bias,
                                          const float* gamma, const float* beta, int B, int T, int Cin, int Cout,
                                          float* dx, float* dkernel, float* dbias, float* dgamma, float* dbeta, void* stream);

/* Backward of the training-time Attention (networks.py:126-155, mononotic_attention=False): A = softmax(Q K^T / sqrt(d)),
 * R = [A V ; Q], alignments = A^T.  Q, dQ (B, T, d); K, V, dK, dV (B, N, d); dR (B, T, 2d); dAl (B, N, T) = gradient with respect to
 * the returned alignments (the guided-attention loss); N and d multiples of 4.  A is recomputed. */
int dctts_train_attention_backward(dctts_train* t, const float* Q, const float* K, const float* V, const float* dR, const float* dAl,
                                   int B, int T, int N, int d, float* dQ, float* dK, float* dV, void* stream);

/* Backward of embed (modules.py:13-42): dtable (vocab, e) = rows of dy (n, e) summed per id; row 0 (zeroed at lookup) receives none. */
int dctts_train_embed_backward(dctts_train* t, const int32_t* ids, const float* dy, long long n, int vocab, int e, float* dtable, void* stream);

/* Forward passes of the same blocks on the TF-layout variables a trainer holds and updates (the inference context of dctts_hip.h
 * reads MFMA-packed copies made once at upload).  Shapes and arguments as in the matching *_backward; y is the block's output.
 * conv1d's act: 0 none, 1 relu, 2 sigmoid.  attention_forward is the training form (no monotonic mask): R (B, T, 2d) = [A V ; Q],
 * alignments (B, N, T) = A^T.  sigmoid: y = 1 / (1 + exp(-x)) elementwise (Y = sigmoid(logits), networks.py:210,290). */
int dctts_train_hc_forward(dctts_train* t, const float* x, const float* kernel, const float* bias, const float* g1, const float* b1,
                           const float* g2, const float* b2, int B, int T, int C, int k, int rate, int causal, float* y, void* stream);
int dctts_train_conv1d_forward(dctts_train* t, const float* x, const float* kernel, const float* bias, const float* gamma, const float* beta,
                               int B, int T, int Cin, int Cout, int k, int rate, int causal, int act, float* y, void* stream);
int dctts_train_conv1d_transpose_forward(dctts_train* t, const float* x, const float* kernel, const float* bias, const float* gamma, const float* beta,
                                         int B, int T, int Cin, int Cout, float* y, void* stream);
int dctts_train_embed_forward(dctts_train* t, const int32_t* ids, const float* table, long long n, int vocab, int e, float* y, void* stream);
int dctts_train_attention_forward(dctts_train* t, const float* Q, const float* K, const float* V, int B, int T, int N, int d,
                                  float* R, float* alignments, void* stream);
int dctts_train_sigmoid(dctts_train* t, const float* x, float* y, long long n, void* stream);
/* tf.layers.dropout(rate, training=True) (modules.py:139,195,245): y = x * keep / (1 - rate), keep bits from a counter-based hash of
 * (key, element index); the backward pass is the same call on dy with the same key.  x == y (in place) is allowed. */
int dctts_train_dropout(dctts_train* t, const float* x, float* y, long long n, uint64_t key, float rate, void* stream);

/* train.py:85-100.  Y, Y_logits, mels (B, T, n_mels); alignments (B, N, T) as networks.py:153 returns them (N <= max_N,
 * T <= max_T: the reference pads them to (max_N, max_T) with -1 and masks the padding).  losses[3] (device) receives
 * loss_mels, loss_bd1, loss_att; dY / dlogits / dA the gradients of their sum with respect to Y (L1 term), Y_logits
 * (divergence term) and alignments (guided-attention term, weights of utils.py:134-140 with g = 0.2). */
int dctts_train_text2mel_losses(dctts_train* t, const float* Y, const float* Y_logits, const float* mels, const float* alignments,
                                int B, int T, int n_mels, int N, int max_N, int max_T,
                                float* losses, float* dY, float* dlogits, float* dA, void* stream);

/* train.py:102-110.  Z, Z_logits, mags (rows, F) flattened; losses[2] = loss_mags, loss_bd2. */
int dctts_train_ssrn_losses(dctts_train* t, const float* Z, const float* Z_logits, const float* mags, long long n,
                            float* losses, float* dZ, float* dlogits, void* stream);

/* train.py:119-131 on one variable of n elements: g = clip(grad, -1, 1); Adam moments m, v (updated in place);
 * var -= lr * sqrt(1 - beta2^step) / (1 - beta1^step) * m / (sqrt(v) + eps), tf.train.AdamOptimizer defaults
 * beta1 = 0.9, beta2 = 0.999, eps = 1e-8; step is 1-based; lr is utils.py:142-145's schedule, evaluated by the caller. */
int dctts_train_adam_step(dctts_train* t, float* var, const float* grad, float* m, float* v, long long n, int step, float lr, void* stream);
/* The same update applied to `count` variables in one launch per 64 variables (train.py:131's apply_gradients over all trainable variables): vars / grads /
 * ms / vs are HOST arrays of `count` device pointers, ns their element counts.  Results are identical to `count` calls of the above. */
int dctts_train_adam_step_multi(dctts_train* t, int count, float* const* vars, const float* const* grads, float* const* ms, float* const* vs,
                                const long long* ns, int step, float lr, void* stream);

#ifdef __cplusplus
}
#endif
#endif
