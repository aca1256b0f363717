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
//   * A (activations): 32 rows x 32 channels per chunk, register-staged into a triple-buffered,
//     padded LDS tile (row stride 36 floats -> conflict-free ds_read_b128), one mid-chunk barrier.
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
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// Split-bf16 contraction (round 5, OPT-IN: dctts_set_split_bf16; the default and the headline stay exact fp32).  A float is written as two bf16 terms,
// x = hi + mid + r with hi = bf16(x), mid = bf16(x - hi) (both round-to-nearest: v_cvt_pk_bf16_f32; x - hi is exact) and |r| <= 2^-18 |x|, and a product is
// accumulated in fp32 from three matrix instructions  hi.hi + hi.mid + mid.hi  (what is dropped, mid.mid and the r terms, is <= 2^-16 of |x||w|), on
// v_mfma_f32_32x32x16_bf16 at 16x the fp32 matrix rate: 3 x 32 cycles per 16 k against 8 x 64.  The weights are split once at upload, the activations in
// registers behind the LDS read.  Two packed halves of a pair (x0, x1) -> their (hi, mid) words:
__device__ __forceinline__ void split_bf16_pair(float x0, float x1, unsigned& hi, unsigned& mid) {
  const bf16x2_t h = __builtin_convertvector((f32x2_t){x0, x1}, bf16x2_t);
  hi = __builtin_bit_cast(unsigned, h);
  const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xffff0000u);
  const bf16x2_t m = __builtin_convertvector((f32x2_t){x0 - h0, x1 - h1}, bf16x2_t);
  mid = __builtin_bit_cast(unsigned, m);
}

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
  // ---- tap-split row tail (hconv_kernel<..., RAW = 1>, grid (row tiles, taps)): rows m_base .. M - 1, workgroup (x, y) contracts tap y only and
  //      stores its bare partial sums to raw_out[(y * (M - m_base) + m - m_base) * raw_ld + column]; hc_tail_finish_kernel adds the taps and finishes
  int m_base; float* raw_out; int raw_ld;
  float* presum_out;      // when set: row r == R-1 of every utterance is a PRESUM row -- its last tap is not contracted (the decode chain
  long presum_rstride;    //   does that) and bias + the older taps go, un-normalised, to presum_out[b * presum_rstride + column]
  const float* wx;        // hconv_kernel<..., XC = 1>: the weights of output column NT * NW * 32 (= cout - 1), cin_p floats, zero beyond the real Cin
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// sigmoid on v_exp_f32 / v_rcp_f32 (about 1 ulp each; saturates cleanly: exp -> inf gives 0, exp -> 0 gives 1)
__device__ __forceinline__ float fast_sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// sum over the 32 lanes that share (lane >> 5): four DPP steps inside each 16-lane row (quad_perm, quad_perm, row_half_mirror, row_mirror), one
// cross-row exchange
template <int CTRL>
__device__ __forceinline__ float hconv_dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float half_sum32(float v) {
  v = hconv_dpp_add<0xB1>(v); v = hconv_dpp_add<0x4E>(v); v = hconv_dpp_add<0x141>(v); v = hconv_dpp_add<0x140>(v);
  return v + __shfl_xor(v, 16);
}

// Round 3 (tools/micro/hconv_lab.hip; profiles/r03_hconv_lab.txt: every variant timed in turn with a cache-thrashing pass in front of each launch,
// because repeating one kernel back to back keeps its weights cache-resident and re-ranks the variants):
//   * K loop: three LDS buffers and ONE barrier per 32-channel chunk, in the MIDDLE of the chunk, with nothing that depends on it right behind it:
//       chunk ch:  [k-groups 0, 1 from As[ch % 3]]  barrier  [store chunk ch + 2 into As[(ch + 2) % 3]; request chunk ch + 3]  [k-groups 2, 3]
//     As[(ch + 2) % 3] was last read in chunk ch - 1, which every wave has left once it passed this chunk's barrier; what is stored now is read
//     from chunk ch + 2 on, behind the barrier of chunk ch + 1.  The A fragment of the next k-group is requested one group ahead, also across the
//     chunk boundary.  (On its own this is worth nothing measurable; it is what lets the weight requests run BD groups ahead without a drain.)
//   * BD = how many k-groups (8 k each = NT * 4 MFMAs) ahead of its use a weight fragment is requested, each depth a register set of 4 NT VGPRs.
//     BD = 2 is worth 10 % on the 512-channel highway layers (a k-group of NT = 4 is ~0.85 us, less than a miss to HBM); with a scheduling
//     pin in front of the MFMAs (round 2's unmeasured variant) the same depth was 27 % SLOWER: no pins.
//   * SB = 1: the tile bases are wave-uniform (readfirstlane), so every weight request is scalar base + one shared vector offset: 15 address
//     VGPRs less at NT = 8 (what makes the batched epilogue fit in 256 registers there).
//   * epilogue without load-behind-branch chains: bias is the accumulators' initial value; the layer-norm parameters and the first residual
//     rows are requested before the statistics passes; rows that do not exist are handled by predicated stores, not `continue`; the next four
//     rows' residuals are in flight while four rows are finished; sigmoid on v_exp_f32 / v_rcp_f32.  2 % (HC_11/12) to 6 % (512-channel, 1025-column).
//   * RAW = 1 (round 3, the row tail of the big highway layers): what is left after the exact rounds of 32-row items is 0.28 of a round; as 16-row
//     items on hconv16_kernel it cost 0.57 of a round on 144 of the 256 CUs.  Here it is 32-row items x TAPS: workgroup (x, y) contracts tap y only
//     (a third of K), 216 workgroups in one round at a third of an item's time, bare partial sums to HBM; hc_tail_finish_kernel (below) adds the three
//     partials and the bias and finishes the rows (two layer-norms, gate, highway mix).
//   * RAW = 2 (round 4, a highway layer whose 32-row items fill less than three quarters of the CUs -- TextEnc: 180 items on 256 CUs, 205 us per layer whatever
//     the items do): the items are split by COLUMNS.  Workgroup (x, y) of NW = 4 waves x one (gate, info) tile pair each owns 32 rows x channels
//     [128 y, 128 y + 128) of both halves and contracts ALL of K; 720 quarter items run three to a CU at once, so a layer takes 3/4 of an item's time (every
//     split into k equal parts has makespan ceil(180 k / 256) / k: 1 for k = 2 and 3 -- the tap split above --, 3/4 for k = 4).  The pre-norm values go to
//     raw_out (one part); hc_tail_finish_kernel adds the bias and finishes the rows.  Same weight packing: tile (y NW + wave) NT + i of the 8-wave layout.
//   * XC = 1 (round 5, SSRN's 1025-column layers C_13 .. C_16): 33 column tiles on 11 waves sit 3 / 3 / 3 / 2 on the four SIMDs, so every chunk takes 9 tile-times where
//     8.25 would do.  Here the first 1024 columns are 8 waves x 4 tiles (two waves per SIMD, 8 tile-times) and column 1024 is a dot product on the vector ALU: wave w
//     owns rows 4 w .. 4 w + 3, a lane multiplies two channels of a chunk (A from the LDS tile every wave reads anyway, the column's weights from a 4 KB LDS copy) and
//     the 16 lanes of a row are summed with DPP at the end; the value joins the layer-norm statistics in the cross-wave step and is normalised / stored by one lane.
//   * BF = 1 (round 5, opt-in): the contraction on the bf16 matrix pipe from split operands (above).  p.wp then points at the bf16 packing
//     [tile][16-k group][hi | mid][lane][8 bf16] (pack_bw_bf16 in dctts_api.hip): lane l holds k = 16 g + 8 (l >> 5) .. + 7 of column l & 31, as
//     v_mfma_f32_32x32x16_bf16 wants its B operand; the A operand is the same 8 consecutive channels of row l & 31 out of the fp32 LDS tile.  The
//     accumulators come out in the fp32 instruction's layout, so prologue, statistics and epilogue are shared.  NT <= 4 (a 16-k group of weights is
//     8 NT registers per ring slot); the 64-tile layers run as two column halves (RAW = 2) + the finishing pass.
//   * SG (round 6): where the NT + 1 requests of a k-group (NT weight fragments, the next A fragment) sit among its 4 NT MFMAs.  0: wherever the compiler's scheduler
//     puts them (rounds 1-5; a PIN in front of the MFMAs was 27 % slower in round 3).  1 / 2: spread with sched_group_barrier -- one MFMA, the LDS read, then one
//     weight request behind every three (1) / four (2) MFMAs.  tools/micro/mfma_feed_lab.hip: a wave issues in order, so requests clustered in front of a k-group's
//     MFMAs cost matrix-pipe time even with a second wave on the SIMD (38 against 34.7 cycles per MFMA in the decode's 16 x 16 x 4 loop).  Here (hconv_lab, 768 items,
//     same run): HC_11 2449 -> 2260 us with SB = 0 (0.803 -> 0.870 of the fp32 MFMA peak), the 1025-column layers 457 -> 445, HC_8 669 -> 655; C_10 gets slower (262 -> 274).
template <int EPI, int NT, int NW, int BD = 1, int SB = 0, int RAW = 0, int XC = 0, int BF = 0, int SG = 0>
__global__ void __launch_bounds__(NW * 64, (XC == 2) ? 4 : ((NW == 4 && NT == 8 && RAW == 0) ? 2 : 1)) hconv_kernel(const ConvParams p) {      // (XC = 2: the same with registers capped for two workgroups per CU)
  static_assert(XC == 0 || (EPI == EPI_C && RAW == 0 && NW == 8), "the extra column rides in the fused k = 1 form: 8 waves x 4 rows");
  static_assert(BF == 0 || (RAW != 1 && NT <= 4 && (BD == 1 || BD == 2)), "split-bf16: whole-K items, at most four tiles per wave, ring rotated by the two groups of a chunk");
  constexpr bool KPART = (RAW == 1);        // this workgroup contracts a PART of K (grid y = part)
  constexpr bool CPART = (RAW == 2);        // this workgroup owns a PART of the columns (grid y = part)
  constexpr int LDA = 36;
  constexpr int NH = (EPI == EPI_HC) ? 2 : 1;
  constexpr int NP = (EPI == EPI_HC) ? NT / 2 : NT;       // output column tiles per wave
  static_assert(EPI != EPI_HC || (NT % 2 == 0), "HC tiles come in (gate, info) pairs");
  static_assert(BD == 1 || BD == 2 || BD == 4, "the register ring is rotated by the 4 k-groups of a chunk");
  __shared__ __attribute__((aligned(16))) float As[3][32 * LDA];
  __shared__ float red[NW * 2 * 32];
  __shared__ float tot[2][2 * 32];           // [pass][h * 32 + row]: mean, then 1 / sqrt(var + eps)
  __shared__ long s_inrow[32];
  __shared__ long s_outrow[32];
  __shared__ long s_out2row[32];
  __shared__ float xw_s[XC ? 1120 : 1];      // the extra column's weights (cin_p <= 1120)
  __shared__ float xcol_s[XC ? 32 : 1];      // its 32 pre-norm values (bias included)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = SB ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
  const int wv = CPART ? (int)blockIdx.y * NW + wave : wave;      // the wave's position in the layer's column order
  const int l31 = lane & 31, lhi = lane >> 5;
  const int m0 = (RAW ? p.m_base : 0) + blockIdx.x * 32;
  const int t_base = p.step ? *p.step : 0;
  // RAW: workgroup y contracts PART y of K -- tap y of a three-tap layer, or the y-th third of the 32-channel chunks of a k = 1 layer
  const int tap0 = (KPART && p.ntaps == 3) ? (int)blockIdx.y : 0;

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
  if constexpr (XC) { for (int i = tid; i < p.cin_p; i += NW * 64) xw_s[i] = p.wx[i]; }
  __syncthreads();

  // ---- A loader: thread (lrow, lc4) moves one float4 per chunk
  const int lrow = (tid >> 3) & 31, lc4 = tid & 7;
  const long my_inrow = s_inrow[lrow];
  const int cpt = p.cin_p >> 5;          // chunks per tap
  const int cpt3 = (cpt + 2) / 3;
  const int ch0 = (KPART && p.ntaps != 3) ? ((int)blockIdx.y * cpt3 < cpt ? (int)blockIdx.y * cpt3 : cpt - 1) : 0;     // first chunk of this part inside its tap
  const int nch = KPART ? ((p.ntaps == 3) ? cpt : ((cpt - ch0 < cpt3) ? cpt - ch0 : cpt3)) : p.ntaps * cpt;
  const int KG = nch * 4;                // k-groups of 8 this workgroup contracts
  const int KGT = KPART ? p.ntaps * cpt * 4 : KG;     // ... of a packed tile; RAW: this workgroup starts at its part's first k-group
  const int kg0 = KPART ? (tap0 * cpt + ch0) * 4 : 0;

  // Branch-free: a load inside a conditional block makes the wait-count pass fall back to s_waitcnt vmcnt(0) at the join (the
  // weight prefetch behind it then drains twice per chunk).  Rows / columns that must read as zero are redirected to a readable
  // address (row in_row0, column 0) and the value is discarded when it is stored to LDS; threads >= 256 load duplicates.
  const bool row_ok = my_inrow >= 0;
  const long safe_row = p.gather ? 0 : p.in_row0;
  int ltap = 0, lcit = 0;                  // (tap, chunk in tap) the loader is at: it walks forward one chunk per call, clamped at the last chunk
  auto load_next = [&](bool& ok) -> float4 {
    const int c = (KPART ? ch0 + lcit : lcit) * 32 + lc4 * 4;
    const int tq = KPART ? tap0 + ltap : ltap;
    const int toff = (tq == 0) ? p.tap_off[0] : ((tq == 1) ? p.tap_off[1] : p.tap_off[2]);
    ok = row_ok && c < p.cin;
    const long row = row_ok ? my_inrow + toff : safe_row;
    const float4 v = *reinterpret_cast<const float4*>(p.in + row * (long)p.in_stride + (c < p.cin ? c : 0));
    if (KPART) { if (lcit < nch - 1) ++lcit; }
    else if (!(ltap == p.ntaps - 1 && lcit == cpt - 1)) { if (++lcit == cpt) { lcit = 0; ++ltap; } }
    return v;
  };

  const float4* wq[NT];                    // SB: wave-uniform bases (scalar registers) + the lane as part of the index
#pragma unroll
  for (int i = 0; i < NT; ++i)
    wq[i] = reinterpret_cast<const float4*>(p.wp) + ((long)(wv * NT + i) * KGT + kg0) * 64 + (SB ? 0 : lane);
  const int wl = SB ? lane : 0;
  // BF: [tile][16-k group][hi | mid][lane] of 16-byte fragments; nch * 2 groups per tile (whole-K items only)
  const uint4* wb[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
    wb[i] = reinterpret_cast<const uint4*>(p.wp) + (long)(wv * NT + i) * (p.ntaps * cpt * 2) * 128 + (SB ? 0 : lane);
  const int KG16 = nch * 2;

  // channel of tile i inside its layer-norm group; the bias is the accumulators' initial value (columns beyond C: zero weights, zero bias -> exactly 0)
  const int C = p.cout;
  int chan[NT]; bool cval[NT];
  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    int ch_, bidx;
    if (EPI == EPI_HC) { ch_ = (wv * NP + (i >> 1)) * 32 + l31; bidx = (i & 1) * C + ch_; }
    else { ch_ = (wv * NT + i) * 32 + l31; bidx = ch_; }
    chan[i] = ch_; cval[i] = ch_ < C;
    const float bv = RAW ? 0.f : p.bias[cval[i] ? bidx : 0];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = cval[i] ? bv : 0.f;
  }

  bool aok, aok1;
  float4 a0 = load_next(aok);
  float4 a1 = load_next(aok1);
  float4 bq[BD][NT];                       // fragments of k-groups kg .. kg + BD - 1
  uint4 bh[BD][NT], bm[BD][NT];            // BF: the hi / mid fragments of 16-k groups g .. g + BD - 1
  if constexpr (!BF) {
#pragma unroll
    for (int d = 0; d < BD; ++d)
#pragma unroll
      for (int i = 0; i < NT; ++i) bq[d][i] = wq[i][(d < KG ? d : KG - 1) * 64 + wl];
  } else {
#pragma unroll
    for (int d = 0; d < BD; ++d)
#pragma unroll
      for (int i = 0; i < NT; ++i) { const int g = (d < KG16 ? d : KG16 - 1) * 128 + wl; bh[d][i] = wb[i][g]; bm[d][i] = wb[i][g + 64]; }
  }
  if (!aok) a0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!aok1) a1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 256) {
    *reinterpret_cast<float4*>(&As[0][lrow * LDA + lc4 * 4]) = a0;
    *reinterpret_cast<float4*>(&As[1][lrow * LDA + lc4 * 4]) = a1;      // with a single chunk this is a copy of chunk 0 that nothing reads
  }
  float4 areg = load_next(aok);            // chunk 2 (or a re-read of the last chunk)
  __syncthreads();
  const int aoff = l31 * LDA + lhi * 4;
  float4 a = *reinterpret_cast<const float4*>(&As[0][aoff]);
  const int aoffb = l31 * LDA + lhi * 8;   // BF: 8 consecutive channels of row l31 per 16-k group
  float4 af0 = *reinterpret_cast<const float4*>(&As[0][aoffb]), af1 = *reinterpret_cast<const float4*>(&As[0][aoffb + 4]);
  int cb = 0;                              // ch % 3
  const int xoff = (wave * 4 + (lane >> 4)) * LDA + (lane & 15) * 2;      // XC: (row, channel pair) of this lane inside a chunk's tile
  float xacc = 0.f;
  for (int ch = 0; ch < nch; ++ch) {
    const float* Ab = As[cb];
    if constexpr (XC) {                    // chunk ch's tile is complete since the barrier of chunk ch - 1 and is not rewritten before chunk ch + 1's
      const float2 xa = *reinterpret_cast<const float2*>(&Ab[xoff]);
      const float2 xw = *reinterpret_cast<const float2*>(&xw_s[ch * 32 + (lane & 15) * 2]);
      xacc = fmaf(xa.x, xw.x, fmaf(xa.y, xw.y, xacc));
    }
    const int cb1 = (cb == 2) ? 0 : cb + 1, cb2 = (cb1 == 2) ? 0 : cb1 + 1;
    const float* An = As[cb1];
    if constexpr (BF) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {          // the two 16-k groups of the chunk; the barrier sits between them (same staging as the fp32 loop)
        const int kg = ch * 2 + q;
        const int kgn = (kg + BD < KG16) ? kg + BD : KG16 - 1;
        if (q == 1) {
          __syncthreads();
          if (!aok) areg = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ch + 2 < nch && tid < 256) *reinterpret_cast<float4*>(&As[cb2][lrow * LDA + lc4 * 4]) = areg;
          areg = load_next(aok);             // chunk ch + 3
        }
        const float* nx = (q == 0) ? &Ab[aoffb + 16] : &An[aoffb];
        const float4 an0 = *reinterpret_cast<const float4*>(nx), an1 = *reinterpret_cast<const float4*>(nx + 4);
        uint4 bhn[NT], bmn[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) { bhn[i] = wb[i][kgn * 128 + wl]; bmn[i] = wb[i][kgn * 128 + 64 + wl]; }
        uint4 ah, am;
        split_bf16_pair(af0.x, af0.y, ah.x, am.x); split_bf16_pair(af0.z, af0.w, ah.y, am.y);
        split_bf16_pair(af1.x, af1.y, ah.z, am.z); split_bf16_pair(af1.z, af1.w, ah.w, am.w);
        const bf16x8_t ahv = __builtin_bit_cast(bf16x8_t, ah), amv = __builtin_bit_cast(bf16x8_t, am);
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahv, __builtin_bit_cast(bf16x8_t, bh[0][i]), acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahv, __builtin_bit_cast(bf16x8_t, bm[0][i]), acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(amv, __builtin_bit_cast(bf16x8_t, bh[0][i]), acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
#pragma unroll
          for (int d = 0; d + 1 < BD; ++d) { bh[d][i] = bh[d + 1][i]; bm[d][i] = bm[d + 1][i]; }
          bh[BD - 1][i] = bhn[i]; bm[BD - 1][i] = bmn[i];
        }
        af0 = an0; af1 = an1;
      }
    } else {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int kg = ch * 4 + gq;
      const int kgn = (kg + BD < KG) ? kg + BD : KG - 1;
      if (gq == 2) {
        __syncthreads();
        if (!aok) areg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ch + 2 < nch && tid < 256) *reinterpret_cast<float4*>(&As[cb2][lrow * LDA + lc4 * 4]) = areg;
        areg = load_next(aok);             // chunk ch + 3
      }
      const float4 an = (gq < 3) ? *reinterpret_cast<const float4*>(&Ab[aoff + (gq + 1) * 8]) : *reinterpret_cast<const float4*>(&An[aoff]);
      float4 bnext[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) bnext[i] = wq[i][kgn * 64 + wl];
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[0][i].x, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[0][i].y, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[0][i].z, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[0][i].w, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int d = 0; d + 1 < BD; ++d) bq[d][i] = bq[d + 1][i];
        bq[BD - 1][i] = bnext[i];
      }
      a = an;
      if constexpr (SG == 1) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int i = 0; i < NT; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, NT - 1, 0);
      } else if constexpr (SG == 2) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int i = 0; i + 1 < NT; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      } else if constexpr (SG == 3) {        // the requests in the FIRST half of the k-group's MFMAs (two MFMAs apart), nothing behind them
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int i = 0; i < NT; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT - 1, 0);
      }
    }
    }
    cb = cb1;
  }

  if constexpr (RAW) {                      // bare partial sums of this part: [part][row - m_base][raw_ld], natural column order (HC: H1 | H2)
    const long mt = (long)p.M - p.m_base;
    const int tap0_ = CPART ? 0 : (int)blockIdx.y;
    if (KPART && p.ntaps != 3 && (int)blockIdx.y * cpt3 >= cpt) {      // (a k = 1 layer with fewer than three chunk groups: this part is empty -- zeros)
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int col = (EPI == EPI_HC) ? (i & 1) * C + chan[i] : chan[i];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * lhi;
        if (cval[i] && m0 + row < p.M) p.raw_out[((long)tap0_ * mt + (m0 + row - p.m_base)) * p.raw_ld + col] = acc[i][j];
      }
    }
    return;
  }
  // =====================================================================================
  // Epilogue.  acc[i][j] = conv output (bias included) at row (j&3) + 8*(j>>2) + 4*lhi, column of tile i / lane l31.
  // Requests first, then the two statistics passes, then the stores.
  // =====================================================================================
  constexpr int CX = NT * NW * 32;           // XC: index of the extra column
  if constexpr (XC) {
    float v = xacc;
    v = hconv_dpp_add<0xB1>(v); v = hconv_dpp_add<0x4E>(v); v = hconv_dpp_add<0x141>(v); v = hconv_dpp_add<0x140>(v);      // the 16 lanes of a row
    if ((lane & 15) == 0) xcol_s[wave * 4 + (lane >> 4)] = v + p.bias[CX];      // (visible behind the first barrier of the statistics pass)
  }
  float pg1[NP], pb1[NP], pg2[NP], pb2[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int i = (EPI == EPI_HC) ? 2 * k : k;
    const int cs = cval[i] ? chan[i] : 0;
    pg1[k] = p.g1[cs]; pb1[k] = p.b1[cs];
    if (EPI == EPI_HC) { pg2[k] = p.g2[cs]; pb2[k] = p.b2[cs]; }
  }
  // residual rows of the highway mix (modules.py:193), four rows (j) at a time; the first batch is requested here, behind nothing
  constexpr int JB = 4;
  float xr[2][(EPI == EPI_HC) ? NP : 1][JB];
  auto load_res = [&](int jb, int slot) {
    if (EPI != EPI_HC) return;
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
      const int j = jb * JB + jj;
      const int row = (j & 3) + 8 * (j >> 2) + 4 * lhi;
      const long ir = s_inrow[row];
      const float* rp = p.in + (ir >= 0 ? ir : safe_row) * (long)p.in_stride;
#pragma unroll
      for (int k = 0; k < NP; ++k) xr[slot][k][jj] = rp[cval[2 * k] ? chan[2 * k] : 0];
    }
  };
  load_res(0, 0);

  // ---- pass 1: mean ; pass 2: biased variance about the mean (tf.nn.moments, two-pass), eps 1e-12 inside the root
  const float invC = 1.0f / (float)C;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float s[NH][16];
    float mean[NH][16];
    if (pass == 1) {
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int j = 0; j < 16; ++j) mean[h][j] = tot[0][h * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhi];
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 16; ++j) s[h][j] = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int h = (EPI == EPI_HC) ? (i & 1) : 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (pass == 0) s[h][j] += acc[i][j];            // padded columns hold exactly 0
        else { const float d = cval[i] ? (acc[i][j] - mean[h][j]) : 0.f; s[h][j] += d * d; }
      }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 16; ++j) s[h][j] = half_sum32(s[h][j]);
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
      if constexpr (XC) { if (pass == 0) v += xcol_s[r]; else { const float d = xcol_s[r] - tot[0][r]; v += d * d; } }
      v *= invC;
      tot[pass][h * 32 + r] = (pass == 0) ? v : 1.0f / sqrtf(v + 1e-12f);
    }
    __syncthreads();
  }

  // ---- normalise, gate / activate, store: four rows at a time, the next four rows' residuals in flight
#pragma unroll
  for (int jb = 0; jb < 16 / JB; ++jb) {
    if (jb + 1 < 16 / JB) load_res(jb + 1, (jb + 1) & 1);
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
      const int j = jb * JB + jj;
      const int row = (j & 3) + 8 * (j >> 2) + 4 * lhi;
      const long orow = s_outrow[row];
      const bool ok = orow >= 0;
      float* op = p.out + (ok ? orow : 0) * (long)p.out_stride;
      const float r0 = tot[1][row], r1 = tot[1][32 + row];
      const float m0_ = tot[0][row], m1_ = tot[0][32 + row];
      if (EPI == EPI_HC) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const float y1 = (acc[2 * k][j] - m0_) * r0 * pg1[k] + pb1[k];
          const float y2 = (acc[2 * k + 1][j] - m1_) * r1 * pg2[k] + pb2[k];
          const float gt = fast_sigmoidf_(y1);
          const float o = gt * y2 + (1.0f - gt) * xr[jb & 1][k][jj];
          if (ok && cval[2 * k]) op[chan[2 * k]] = o;
        }
      } else {
        float* op2 = p.out2 ? p.out2 + (ok ? s_out2row[row] : 0) * (long)p.out2_stride : nullptr;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          float y = (acc[k][j] - m0_) * r0 * pg1[k] + pb1[k];
          if (op2 && ok && cval[k]) op2[chan[k]] = y;
          if (p.act == ACT_RELU) y = fmaxf(y, 0.f);
          else if (p.act == ACT_SIGMOID) y = fast_sigmoidf_(y);
          if (ok && (cval[k] || chan[k] < p.out_zero_to)) op[chan[k]] = cval[k] ? y : 0.f;
        }
      }
    }
  }
  if constexpr (XC) {                        // column CX and the zero pad columns behind it: 16 threads per row
    const int row = tid >> 4, j = tid & 15;
    const long orow = s_outrow[row];
    if (orow >= 0) {
      float* op = p.out + orow * (long)p.out_stride;
      if (j == 0) {
        float y = (xcol_s[row] - tot[0][row]) * tot[1][row] * p.g1[CX] + p.b1[CX];
        if (p.out2) p.out2[s_out2row[row] * (long)p.out2_stride + CX] = y;
        if (p.act == ACT_RELU) y = fmaxf(y, 0.f);
        else if (p.act == ACT_SIGMOID) y = fast_sigmoidf_(y);
        op[CX] = y;
      } else if (CX + j < p.out_zero_to) op[CX + j] = 0.f;
      if (CX + 16 + j < p.out_zero_to) op[CX + 16 + j] = 0.f;
    }
  }
}

// The second half of the tap-split row tail: one wave per row adds the taps' partial sums and the bias, and finishes the row exactly as the fused
// epilogue does (two-pass layer-norm of both halves, eps 1e-12 inside the root, sigmoid gate on v_exp / v_rcp, highway mix with the layer's input row:
// modules.py:188-193).  grid ceil(rows / 4), block 256; C in {256, 512, 1024}.
__global__ void __launch_bounds__(256) hc_tail_finish_kernel(const ConvParams p, const float* __restrict__ part, const int ntap) {
  const int lane = threadIdx.x & 63;
  const int m = p.m_base + blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= p.M) return;
  const int b = m / p.R, r = m - b * p.R;
  const int t = (p.step ? *p.step : 0) + (p.offs ? p.offs[r] : r);
  if (t < 0) return;
  const long inrow = (long)b * p.in_bstride + p.in_row0 + t;
  const long outrow = (long)b * p.out_bstride + p.out_row0 + (long)t * p.out_tmul + p.out_tadd;
  const int C = p.cout, nq = C >> 8;                              // 256 channels per sweep, 4 per lane
  const long mt = (long)p.M - p.m_base, mr = m - p.m_base;
  float4 h1[4], h2[4], xr[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q < nq) {
      const int c = q * 256 + lane * 4;
      float4 a = *reinterpret_cast<const float4*>(p.bias + c), d = *reinterpret_cast<const float4*>(p.bias + C + c);
      for (int tp = 0; tp < ntap; ++tp) {
        const float* pr = part + ((long)tp * mt + mr) * p.raw_ld;
        const float4 u = *reinterpret_cast<const float4*>(pr + c), v = *reinterpret_cast<const float4*>(pr + C + c);
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; d.x += v.x; d.y += v.y; d.z += v.z; d.w += v.w;
      }
      h1[q] = a; h2[q] = d;
      xr[q] = *reinterpret_cast<const float4*>(p.in + inrow * (long)p.in_stride + c);
    } else { h1[q] = h2[q] = xr[q] = make_float4(0.f, 0.f, 0.f, 0.f); }
  }
  auto wsum = [](float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };
  const float invC = 1.0f / (float)C;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) { s1 += h1[q].x + h1[q].y + h1[q].z + h1[q].w; s2 += h2[q].x + h2[q].y + h2[q].z + h2[q].w; }
  const float m1 = wsum(s1) * invC, m2 = wsum(s2) * invC;
  float v1 = 0.f, v2 = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q < nq) {
      const float4 a = h1[q], d = h2[q];
      v1 += (a.x - m1) * (a.x - m1) + (a.y - m1) * (a.y - m1) + (a.z - m1) * (a.z - m1) + (a.w - m1) * (a.w - m1);
      v2 += (d.x - m2) * (d.x - m2) + (d.y - m2) * (d.y - m2) + (d.z - m2) * (d.z - m2) + (d.w - m2) * (d.w - m2);
    }
  }
  const float r1 = 1.0f / sqrtf(wsum(v1) * invC + 1e-12f), r2 = 1.0f / sqrtf(wsum(v2) * invC + 1e-12f);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q < nq) {
      const int c = q * 256 + lane * 4;
      const float4 g1 = *reinterpret_cast<const float4*>(p.g1 + c), b1 = *reinterpret_cast<const float4*>(p.b1 + c);
      const float4 g2 = *reinterpret_cast<const float4*>(p.g2 + c), b2 = *reinterpret_cast<const float4*>(p.b2 + c);
      float4 o;
      { const float gt = fast_sigmoidf_((h1[q].x - m1) * r1 * g1.x + b1.x); o.x = gt * ((h2[q].x - m2) * r2 * g2.x + b2.x) + (1.0f - gt) * xr[q].x; }
      { const float gt = fast_sigmoidf_((h1[q].y - m1) * r1 * g1.y + b1.y); o.y = gt * ((h2[q].y - m2) * r2 * g2.y + b2.y) + (1.0f - gt) * xr[q].y; }
      { const float gt = fast_sigmoidf_((h1[q].z - m1) * r1 * g1.z + b1.z); o.z = gt * ((h2[q].z - m2) * r2 * g2.z + b2.z) + (1.0f - gt) * xr[q].z; }
      { const float gt = fast_sigmoidf_((h1[q].w - m1) * r1 * g1.w + b1.w); o.w = gt * ((h2[q].w - m2) * r2 * g2.w + b2.w) + (1.0f - gt) * xr[q].w; }
      *reinterpret_cast<float4*>(p.out + outrow * (long)p.out_stride + c) = o;
    }
  }
}

// The same for a k = 1 conv layer (EPI_C) whose row tail was split over thirds of K: bias + the parts, layer-norm over the cout real columns, activation,
// the optional pre-activation copy (logits) and the zero pad columns, exactly as the fused epilogue.  One wave per row; cout <= 1280.
__global__ void __launch_bounds__(256) c_tail_finish_kernel(const ConvParams p, const float* __restrict__ part, const int nparts) {
  const int lane = threadIdx.x & 63;
  const int m = p.m_base + blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= p.M) return;
  const int b = m / p.R, r = m - b * p.R;
  const int t = (p.step ? *p.step : 0) + (p.offs ? p.offs[r] : r);
  if (t < 0) return;
  const long outrow = (long)b * p.out_bstride + p.out_row0 + (long)t * p.out_tmul + p.out_tadd;
  const long out2row = (long)b * p.out2_bstride + p.out2_row0 + t;
  const int C = p.cout;
  const long mt = (long)p.M - p.m_base, mr = m - p.m_base;
  // columns 256 q + 4 lane .. + 3 (16-byte requests; the buffers' rows are padded to a multiple of 32 floats, so a group that straddles cout is readable)
  float4 y[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int c = q * 256 + lane * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
      v = *reinterpret_cast<const float4*>(p.bias + c);
      for (int tp = 0; tp < nparts; ++tp) {
        const float4 u = *reinterpret_cast<const float4*>(part + ((long)tp * mt + mr) * p.raw_ld + c);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      if (c + 1 >= C) v.y = 0.f;
      if (c + 2 >= C) v.z = 0.f;
      if (c + 3 >= C) v.w = 0.f;
    }
    y[q] = v;
  }
  auto wsum = [](float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };
  const float invC = 1.0f / (float)C;
  float s1 = 0.f;
#pragma unroll
  for (int q = 0; q < 5; ++q) s1 += y[q].x + y[q].y + y[q].z + y[q].w;
  const float mean = wsum(s1) * invC;
  float v1 = 0.f;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int c = q * 256 + lane * 4;
    const float d0 = (c < C) ? y[q].x - mean : 0.f, d1 = (c + 1 < C) ? y[q].y - mean : 0.f, d2 = (c + 2 < C) ? y[q].z - mean : 0.f, d3 = (c + 3 < C) ? y[q].w - mean : 0.f;
    v1 += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  }
  const float rs = 1.0f / sqrtf(wsum(v1) * invC + 1e-12f);
  float* op = p.out + outrow * (long)p.out_stride;
  float* op2 = p.out2 ? p.out2 + out2row * (long)p.out2_stride : nullptr;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int c = q * 256 + lane * 4;
    if (c >= C && c >= p.out_zero_to) continue;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), be = g;
    if (c + 3 < C) { g = *reinterpret_cast<const float4*>(p.g1 + c); be = *reinterpret_cast<const float4*>(p.b1 + c); }
    else if (c < C) { g.x = p.g1[c]; be.x = p.b1[c]; if (c + 1 < C) { g.y = p.g1[c + 1]; be.y = p.b1[c + 1]; } if (c + 2 < C) { g.z = p.g1[c + 2]; be.z = p.b1[c + 2]; } }
    const float yv[4] = {y[q].x, y[q].y, y[q].z, y[q].w}, gv[4] = {g.x, g.y, g.z, g.w}, bv[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ce = c + e;
      if (ce < C) {
        float o = (yv[e] - mean) * rs * gv[e] + bv[e];
        if (op2) op2[ce] = o;
        if (p.act == ACT_RELU) o = fmaxf(o, 0.f);
        else if (p.act == ACT_SIGMOID) o = fast_sigmoidf_(o);
        op[ce] = o;
      } else if (ce < p.out_zero_to) {
        op[ce] = 0.f;
      }
    }
  }
}

// Host-side launch: picks the (NT, NW) instantiation from the layer's tile count.
// Returns hipSuccess or the launch error.
struct ConvShape { int epi, nt, nw; };     // the (BD, SB) of a shape is fixed in launch_hconv (dctts_api.hip)

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
// the tap-split row tail of a highway layer with three taps: rows p.m_base .. p.M - 1 (ConvParams::raw_out / raw_ld set); then the finishing pass
hipError_t launch_hconv_tail(const ConvShape& s, const ConvParams& p, hipStream_t stream);
// a three-tap 512-channel highway layer as quarter-column items (RAW = 2) + the finishing pass: rows p.m_base .. p.M - 1
hipError_t launch_hconv_cols(const ConvShape& s, const ConvParams& p, hipStream_t stream);

}  // namespace dctts
