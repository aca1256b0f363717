// hconv_kernel.h -- fused 1-D conv (1..3 dilated taps) + bias + LayerNorm (+ highway gate) for gfx950.
//
// One kernel covers every conv-type layer of the DC-TTS synthesis path:
//   C   modules.py:91-141   conv(k=1) + bias -> LN -> act                (EPI_C)
//   HC  modules.py:143-197  conv(k) -> split -> LN,LN -> sigmoid gate -> highway mix   (EPI_HC)
//   D   modules.py:199-247  stride-2 transposed conv as two phase launches of EPI_C
//
// Lowering (MI355X-first, not a port of TF's im2col):
//   * the conv is a sliding-window contraction  out[m, :] = sum_tap  x[row(m)+off_tap, :] . W[tap]
//     over K = ntaps * Cin, evaluated with v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain).
//   * a workgroup owns BM = 32 output rows x ALL output columns, so the layer-norm statistics
//     (per row over C channels, two-pass, eps 1e-12) are reduced inside the workgroup and the
//     pre-norm tensor never touches HBM.  NW waves split the columns, NT 32-wide tiles each.
//   * A (activations): 32 rows x 32 channels per chunk, register-staged into a double-buffered,
//     padded LDS tile (row stride 36 floats -> conflict-free ds_read_b128).
//   * B (weights): pre-packed on the host in MFMA fragment order, so one coalesced 1 KiB
//     global_load_dwordx4 per wave feeds four MFMAs; no LDS round trip for an operand that no
//     other wave of the workgroup shares.  Prefetched one k-group (8 k) ahead.
//   * activation buffers carry zero pad rows, so taps never bounds-check (SAME / CAUSAL padding
//     of modules.py:121-125 is the pad rows themselves).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dctts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_C = 0, EPI_HC = 1 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };

struct ConvParams {
  // ---- input rows
  const float* in;        // activation buffer (or embedding table when gather != nullptr)
  const int* gather;      // optional: input row id per output row m (embedding lookup)
  int gather_n;           // rows of the embedding table (ids outside [0, gather_n) read row 0)
  long in_bstride;        // rows per batch item in the input buffer
  long in_row0;           // row index of t = 0
  int in_stride;          // floats per input row
  int cin;                // readable input channels (multiple of 4)
  int cin_p;              // per-tap K extent (multiple of 32; weights zero beyond the real Cin)
  int ntaps;
  int tap_off[3];         // row offset of each tap relative to row t
  // ---- row mapping: m -> (b, r) -> t
  int M;                  // total output rows = batch * R
  int R;                  // rows per batch item in this launch
  const int* offs;        // optional row-offset table (len R): t = t_base + offs[r]; else t = t_base + r
  const int* step;        // optional device step counter: t_base = *step (decode), else 0
  // ---- weights
  const float* wp;        // packed B fragments [tile][kgroup][lane][4]
  const float* bias;      // natural order (2C for HC, Cout for C)
  const float* g1; const float* b1;   // LN gamma/beta (H1 for HC, 'normalize' for C)
  const float* g2; const float* b2;   // H2 (HC only)
  int cout;               // channels leaving the layer (C for HC)
  // ---- output
  float* out; long out_bstride; long out_row0; int out_stride; int out_tmul; int out_tadd;
  int out_zero_to;        // columns [cout, out_zero_to) of each output row are written as 0
  float* out2; long out2_bstride; long out2_row0; int out2_stride;   // optional pre-activation copy (logits)
  int act;
  // ---- decode v3 (hconv16_kernel only): frame index by value, and presum rows
  int t_base_val;         // t_base when step == nullptr
  float* presum_out;      // when set: row r == R-1 of every utterance is a PRESUM row -- its last tap is not contracted (the decode chain
  long presum_rstride;    //   does that) and bias + the older taps go, un-normalised, to presum_out[b * presum_rstride + column]
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// A weight fragment is requested one k-group (8 k = NT * 4 MFMAs) ahead of its use.  Requesting it two groups ahead from a second register set,
// or pinning the requests in front of the group's MFMAs with a sched_barrier, was measured in round 3 and is 27 % SLOWER (SSRN 12.13 -> 15.45 ms,
// profiles/r03_unverified_pass.txt): the waits the ISA shows are covered by the SIMD's other wave, the pinned issue burst is not.
template <int EPI, int NT, int NW>
__global__ void __launch_bounds__(NW * 64) hconv_kernel(const ConvParams p) {
  constexpr int LDA = 36;
  constexpr int NH = (EPI == EPI_HC) ? 2 : 1;
  static_assert(EPI != EPI_HC || (NT % 2 == 0), "HC tiles come in (gate, info) pairs");
  __shared__ __attribute__((aligned(16))) float As[2][32 * LDA];
  __shared__ float red[NW * 2 * 32];
  __shared__ float tot[2 * 32];
  __shared__ long s_inrow[32];
  __shared__ long s_outrow[32];
  __shared__ long s_out2row[32];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int m0 = blockIdx.x * 32;
  const int t_base = p.step ? *p.step : 0;

  if (tid < 32) {
    const int m = m0 + tid;
    long inrow = -1, outrow = -1, out2row = -1;
    if (m < p.M) {
      const int b = m / p.R, r = m - b * p.R;
      const int t = t_base + (p.offs ? p.offs[r] : r);
      if (t >= 0) {
        if (p.gather) {       // embedding lookup (modules.py:13-42): an id outside the table reads row 0 (= the all-zero PAD row), never out of bounds
          const int id = p.gather[m];
          inrow = (id >= 0 && id < p.gather_n) ? id : 0;
        } else {
          inrow = (long)b * p.in_bstride + p.in_row0 + t;
        }
        outrow = (long)b * p.out_bstride + p.out_row0 + (long)t * p.out_tmul + p.out_tadd;
        out2row = (long)b * p.out2_bstride + p.out2_row0 + t;
      }
    }
    s_inrow[tid] = inrow; s_outrow[tid] = outrow; s_out2row[tid] = out2row;
  }
  __syncthreads();

  // ---- A loader: thread (lrow, lc4) moves one float4 per chunk
  const int lrow = (tid >> 3) & 31, lc4 = tid & 7;
  const long my_inrow = s_inrow[lrow];
  const int cpt = p.cin_p >> 5;          // chunks per tap
  const int nch = p.ntaps * cpt;
  const int KG = nch * 4;                // k-groups of 8

  // Branch-free: a load inside a conditional block makes the wait-count pass fall back to s_waitcnt vmcnt(0) at the join (the
  // weight prefetch behind it then drains twice per chunk).  Rows / columns that must read as zero are redirected to a readable
  // address (row in_row0, column 0) and the value is discarded when it is stored to LDS; threads >= 256 load duplicates.
  const bool row_ok = my_inrow >= 0;
  const long safe_row = p.gather ? 0 : p.in_row0;
  auto load_chunk = [&](int tap, int cit, bool& ok) -> float4 {          // cit = chunk index inside the tap
    const int c = cit * 32 + lc4 * 4;
    const int toff = (tap == 0) ? p.tap_off[0] : ((tap == 1) ? p.tap_off[1] : p.tap_off[2]);
    ok = row_ok && c < p.cin;
    const long row = row_ok ? my_inrow + toff : safe_row;
    return *reinterpret_cast<const float4*>(p.in + row * (long)p.in_stride + (c < p.cin ? c : 0));
  };

  const float4* wq[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
    wq[i] = reinterpret_cast<const float4*>(p.wp) + ((long)(wave * NT + i) * KG) * 64 + lane;

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  bool aok;
  float4 areg = load_chunk(0, 0, aok);
  if (!aok) areg = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 256) *reinterpret_cast<float4*>(&As[0][lrow * LDA + lc4 * 4]) = areg;
  int ntap = 0, ncit = 0;                  // (tap, chunk in tap) of the chunk being prefetched
  float4 bcur[NT];                         // fragments of k-group kg
#pragma unroll
  for (int i = 0; i < NT; ++i) bcur[i] = wq[i][0];
  __syncthreads();

  for (int ch = 0; ch < nch; ++ch) {
    const bool more = (ch + 1 < nch);
    if (more) { if (++ncit == cpt) { ncit = 0; ++ntap; } }       // the last chunk re-reads itself: no branch around the load
    areg = load_chunk(ntap, ncit, aok);
    __builtin_amdgcn_sched_barrier(0);     // pin the prefetch here: the scheduler otherwise sinks it to the LDS store at the end
    const float* Ab = As[ch & 1];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int kg = ch * 4 + gq;
      const int kgn = (kg + 1 < KG) ? kg + 1 : KG - 1;
      const float4 a = *reinterpret_cast<const float4*>(&Ab[l31 * LDA + gq * 8 + lhi * 4]);
      float4 bnext[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) bnext[i] = wq[i][(long)kgn * 64];
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bcur[i].x, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bcur[i].y, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bcur[i].z, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bcur[i].w, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) bcur[i] = bnext[i];
    }
    if (!aok) areg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (more && tid < 256) *reinterpret_cast<float4*>(&As[(ch + 1) & 1][lrow * LDA + lc4 * 4]) = areg;
    __syncthreads();
  }

  // =====================================================================================
  // Epilogue.  acc[i][j] = conv output at row (j&3) + 8*(j>>2) + 4*lhi, column of tile i / lane l31.
  // =====================================================================================
  const int C = p.cout;
  int chan[NT];          // channel index inside its LN group
  bool cval[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    int ch_, bidx;
    if (EPI == EPI_HC) {
      const int pp = wave * (NT / 2) + (i >> 1);
      ch_ = pp * 32 + l31;
      bidx = (i & 1) * C + ch_;
    } else {
      ch_ = (wave * NT + i) * 32 + l31;
      bidx = ch_;
    }
    chan[i] = ch_;
    cval[i] = ch_ < C;
    const float bv = cval[i] ? p.bias[bidx] : 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] += bv;
  }

  float mean[NH][16], rstd[NH][16];
  const float invC = 1.0f / (float)C;
  // ---- pass 1: mean ; pass 2: biased variance about the mean (tf.nn.moments, two-pass)
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float s[NH][16];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 16; ++j) s[h][j] = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int h = (EPI == EPI_HC) ? (i & 1) : 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (pass == 0) {
          s[h][j] += acc[i][j];            // padded columns hold exactly 0
        } else {
          const float d = cval[i] ? (acc[i][j] - mean[h][j]) : 0.f;
          s[h][j] += d * d;
        }
      }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float v = s[h][j];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
        s[h][j] = v;
      }
    if (l31 == 0) {
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int j = 0; j < 16; ++j) red[(wave * 2 + h) * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhi] = s[h][j];
    }
    __syncthreads();
    if (tid < 32 * NH) {
      const int h = tid >> 5, r = tid & 31;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[(w * 2 + h) * 32 + r];
      tot[h * 32 + r] = v * invC;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float v = tot[h * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhi];
        if (pass == 0) mean[h][j] = v; else rstd[h][j] = 1.0f / sqrtf(v + 1e-12f);
      }
    __syncthreads();      // red/tot are reused by the next pass
  }

  // ---- normalise, activate / gate, store
  if (EPI == EPI_HC) {
#pragma unroll
    for (int k = 0; k < NT / 2; ++k) {
      const int ch_ = chan[2 * k];
      if (!cval[2 * k]) continue;
      const float g1 = p.g1[ch_], b1 = p.b1[ch_], g2 = p.g2[ch_], b2 = p.b2[ch_];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * lhi;
        const long orow = s_outrow[row];
        if (orow < 0) continue;
        const float y1 = (acc[2 * k][j] - mean[0][j]) * rstd[0][j] * g1 + b1;
        const float y2 = (acc[2 * k + 1][j] - mean[1][j]) * rstd[1][j] * g2 + b2;
        const float gt = sigmoidf_(y1);
        const float xr = p.in[s_inrow[row] * (long)p.in_stride + ch_];
        p.out[orow * (long)p.out_stride + ch_] = gt * y2 + (1.0f - gt) * xr;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int ch_ = chan[i];
      if (cval[i]) {
        const float g1 = p.g1[ch_], b1 = p.b1[ch_];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int row = (j & 3) + 8 * (j >> 2) + 4 * lhi;
          const long orow = s_outrow[row];
          if (orow < 0) continue;
          float y = (acc[i][j] - mean[0][j]) * rstd[0][j] * g1 + b1;
          if (p.out2) p.out2[s_out2row[row] * (long)p.out2_stride + ch_] = y;
          if (p.act == ACT_RELU) y = fmaxf(y, 0.f);
          else if (p.act == ACT_SIGMOID) y = sigmoidf_(y);
          p.out[orow * (long)p.out_stride + ch_] = y;
        }
      } else if (ch_ < p.out_zero_to) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int row = (j & 3) + 8 * (j >> 2) + 4 * lhi;
          const long orow = s_outrow[row];
          if (orow >= 0) p.out[orow * (long)p.out_stride + ch_] = 0.f;
        }
      }
    }
  }
}

// Host-side launch: picks the (NT, NW) instantiation from the layer's tile count.
// Returns hipSuccess or the launch error.
struct ConvShape { int epi, nt, nw; };

inline ConvShape pick_shape(int epi, int cout) {
  if (epi == EPI_HC) {
    const int tiles = 2 * ((cout + 31) / 32);
    if (tiles <= 16) return {EPI_HC, 2, 8};
    if (tiles <= 32) return {EPI_HC, 4, 8};
    return {EPI_HC, 8, 8};
  }
  const int tiles = (cout + 31) / 32;
  if (tiles <= 4) return {EPI_C, 1, 4};
  if (tiles <= 8) return {EPI_C, 1, 8};
  if (tiles <= 16) return {EPI_C, 2, 8};
  if (tiles <= 32) return {EPI_C, 4, 8};
  return {EPI_C, 3, 11};
}

inline int shape_tiles(const ConvShape& s) { return s.nt * s.nw; }

// tiles < 0: all of ceil(M / 32); otherwise only the first `tiles` 32-row items (the rest belong to hconv16_kernel).
hipError_t launch_hconv(const ConvShape& s, const ConvParams& p, hipStream_t stream, int tiles = -1);

}  // namespace dctts
