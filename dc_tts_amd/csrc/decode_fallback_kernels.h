// decode_fallback_kernels.h -- the decode kernels that only the FALLBACK forms launch (round 5 housekeeping: they used to sit in decode_kernels.h between the
// shipping ones).  hsplit_kernel: one column-split launch per layer -- what a decode runs on after a failed team hand-off or with DCTTS_XGROUP=0 / DCTTS_XCONE=0 /
// dctts_decode_safe_once (chain layers, MF = 16), and run_split's small-item path; ln_rows_kernel: the layer-norm / gate row pass behind a cone GEMM when xcone_kernel is off.
// The default decode (xtail_kernel -> xgroup_kernel | rowc1_kernel -> rowhc2_kernel -> xcone_kernel) launches none of them.  Shared helpers (RowNorm, SplitParams,
// combine_stats, hbulk_body ...) stay in decode_kernels.h.
#pragma once
#include "decode_kernels.h"

namespace dctts {

// TRACE = true compiles the wall-clock stamps in (DCTTS_TRACE).  They must NOT exist in the production instantiation even as
// dead branches: a store that may be pending makes the wait-count pass treat vmcnt as out of order (mixed load / store events)
// and every later wait on a load becomes s_waitcnt vmcnt(0) -- the bulk kernel's weight prefetch ring was serialised by it.
// NG > 0 fixes the k-groups per wave at compile time (K = 64 NG for the 32-row form): the K loop is then straight-line code.
// With a run-time `if (g < KG)` around each step, a prefetch issued inside a conditional block may or may not be followed by
// younger loads, so the only safe wait for it is vmcnt(0) -- the counted waits the ring depends on need unconditional steps.
// NT = 3 (16-row form): three causal taps over 256 channels, centre = tap 2 (every k = 3 layer of AudioEnc / AudioDec).  Wave w's
// k-group i is then tap i >> 1, channels 128 (i & 1) + 16 w: compile-time per i, which removes ~150 select / compare
// instructions from the stretch between kernel entry and the first load.  NT = 1: k = 1 over 256 channels: two k-groups per
// wave instead of six clamped ones (the generic form re-reads the last group four times: 3x the load traffic of such a layer).
// NT = 2: k = 1 over 512 channels (AudioDec C_1: four k-groups per wave).  NT = 4: k = 1 over <= 128 channels (AudioEnc C_1 on
// the 80-channel mel row: one k-group per wave, some waves and columns are padding, so it keeps the clamps).
// ONE (16-row form): a workgroup owns ONE 16-column tile (gate or info) instead of the pair: twice the workgroups, half the
// weight bytes and half the MFMAs per workgroup (the matrix pipe is shared by the two waves of a SIMD: 2 us -> 1 us).
template <int MF, bool TRACE = false, int NG = 0, int NT = 0, bool ONE = false>
__global__ void __launch_bounds__(512) hsplit_kernel(const SplitParams p) {
  constexpr int KGS = (MF == 32) ? 8 : 16;          // k per k-group (4 MFMAs)
  constexpr int NJ = (MF == 32) ? 16 : 4;           // accumulator registers per tile
  constexpr int NGMAX = NG > 0 ? NG : ((MF == 32) ? 12 : (NT == 1 ? 2 : (NT == 2 ? 4 : (NT == 4 ? 1 : 6))));   // k-groups per wave
  constexpr bool FULL = (NT == 1 || NT == 2 || NT == 3);    // every k-group / channel of the form exists: no clamps, no padding
  constexpr int BD = (MF == 32) ? 4 : NGMAX;        // B prefetch ring depth (k-groups); the 16-row form holds all of them
  extern __shared__ __attribute__((aligned(16))) float smem[];     // split-K reduction only
  __shared__ long s_prow[MF];                       // output row index per tile row, -1 = skipped
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  prefetch_params<NT>(p);
  const int wgid = blockIdx.y * gridDim.x + blockIdx.x;
  const bool tr = TRACE && p.dbg && wgid == 0 && tid == 0;
  if constexpr (TRACE) { if (tr) p.dbg[0] = wall_clock64(); if (p.dbg_wg && tid == 0 && wgid < 128) p.dbg_wg[2 * wgid] = wall_clock64(); }
  // CHAINROW: the forms only the newest-frame chain uses (one row per utterance, no offset table, frame index by value, grid =
  // (column groups, row tiles)).  Everything the generic row mapping needs -- a possibly-null offs[] load (its branch made the
  // wait-count pass put s_waitcnt vmcnt(0) in front of the A loads: one full memory round trip per launch, after the weight
  // loads), the device step counter, two integer divisions -- is compiled out.
  constexpr bool CHAINROW = (MF == 16) && (NT == 1 || NT == 2 || NT == 4) && !ONE;
  const int step = CHAINROW ? p.step_val : p.step_val + (p.step ? *p.step : 0);
  const long par = step & 1;
  const int KG = p.ntaps * p.cin_p / KGS;
  const int ntile = CHAINROW ? 1 : (p.M + p.tile_rows - 1) / p.tile_rows;
  const int nitems = CHAINROW ? 1 : ntile * p.ngroups * (ONE ? 2 : 1);
  const int arow = lane & (MF - 1);
  const int aq = (MF == 32) ? (lane >> 5) : (lane >> 4);
  const int c4 = aq * 4;
  const bool ln = (MF == 16) && (p.pro != PRO_RAW);
  const int ctap = (NT == 3) ? 2 : ((p.ntaps == 1) ? 0 : ((p.tap_off[0] == 0) ? 0 : ((p.tap_off[1] == 0) ? 1 : 2)));
  // (tap, first channel) of this lane's fragment of k-group i / g
  auto tap_c = [&](int i, int g, int& tap, int& c) {
    if constexpr (NT == 3) { tap = i >> 1; c = 128 * (i & 1) + 16 * wave + c4; }
    else if constexpr (NT == 1 || NT == 2) { tap = 0; c = 128 * i + 16 * wave + c4; }
    else if constexpr (NT == 4) { tap = 0; c = 16 * g + c4; }
    else { const int k0 = g * KGS; tap = (p.ntaps == 1) ? 0 : (k0 >> 8); c = k0 - tap * p.cin_p + c4; }
  };

  // Persistent over work items: the bulk branch launches fewer workgroups than CUs so that the latency-critical
  // chain branch always finds free CUs; the chain itself has exactly one item per workgroup.
  for (int item = CHAINROW ? 0 : blockIdx.x; item < nitems; item += (MF == 32 ? (int)gridDim.x : nitems)) {
    const int mytile = ONE ? (item & 1) : 0, rest = ONE ? (item >> 1) : item;
    const int tile_x = CHAINROW ? (int)blockIdx.y : rest / p.ngroups, grp = CHAINROW ? (int)blockIdx.x : rest - tile_x * p.ngroups;
    const int m0 = tile_x * p.tile_rows;

    // ---- B fragments: wave w owns k-groups w, w+8, ...; independent of A, so issue first
    const float* wb = p.wp + lane * 4;
    const unsigned w0o = (unsigned)(grp * 2 + mytile) * (unsigned)KG * 256u, w1o = w0o + (unsigned)KG * 256u;
    float4 bq0[BD], bq1[BD];
    // Loads are issued WITHOUT branches around them: a uniform `if (g < KG)` still compiles to a branch, and at every join the
    // wait-count pass falls back to s_waitcnt vmcnt(0) when a register may have a load pending on one path -- the chain kernel
    // then paid ~6 serialised memory round trips per layer (in-kernel stamps: 2.8 us from first to last issue).  k-groups past
    // the end re-read the last one (clamped index); their A fragment is zero, so the duplicate weights contribute nothing.
#pragma unroll
    for (int i = 0; i < BD; ++i) {
      const int g = wave + 8 * i, gc = (FULL || g < KG) ? g : KG - 1;
      bq0[i] = ld4u(wb, w0o + (unsigned)gc * 256u);
      if constexpr (!ONE) bq1[i] = ld4u(wb, w1o + (unsigned)gc * 256u); else bq1[i] = bq0[i];
    }

    // ---- this lane's A row (MFMA A operand: lane -> row lane % MF, k sub-block lane / MF)
    int b = 0, t = 0; long prow = -1; bool valid = false, cmask = false;
    {
      const int m = m0 + arow;
      if (arow < p.tile_rows && m < p.M) {
        if constexpr (CHAINROW) {
          b = p.b0 + m; t = step; prow = b; valid = true;
        } else {
          int bl = m, r = 0;
          if (p.R != 1) { bl = m / p.R; r = m - bl * p.R; }
          b = p.b0 + bl;
          t = step + (p.offs ? p.offs[r] : 0);
          prow = (long)b * p.R + r;
          valid = (t >= 0);
          cmask = (NT == 3) && p.mask_last && (r == p.R - 1);     // v3 presum row: the chain contracts its centre tap
        }
      }
      if (wave == 0 && aq == 0) s_prow[arow] = valid ? prow : -1;
    }
    if constexpr (TRACE) { if (tr) p.dbg[1] = wall_clock64(); }

    // ---- A fragments straight from global memory: every load of the item is in flight before the first use.
    //      Loads are unconditional (skipped rows read row 0 and are zeroed afterwards) so that no exec-mask branches
    //      serialise them; addresses are uniform base + 32-bit offset.  ntaps > 1 implies cin_p == 256 (tap = shift).
    const unsigned xs_row = valid ? (unsigned)(par * p.xs_set + ((long)b * p.xs_bstride + p.xs_row0 + t) * p.xs_stride) : (unsigned)(p.xs_row0 * p.xs_stride);
    const unsigned p_row = valid ? (unsigned)(prow * p.nrm.np) : 0u;
    const unsigned rs_row = valid ? (unsigned)(par * p.nrm.res_set + ((long)b * p.nrm.res_bstride + p.nrm.res_row0 + t) * p.nrm.res_stride) : 0u;
    float4 av[NGMAX];
    float4 h2v[2], rsv[2], g1v[2], b1v[2], g2v[2], b2v[2];   // centre-tap extras of the (at most two) centre k-groups of a wave
    float4 st[4];
    float biasv = 0.f;                                        // MF == 16: this thread's output column, fetched with everything else
    if constexpr (MF == 16) {
      // Branch-free issue (see above).  Addresses that a branch used to skip are redirected to something readable
      // (the tap source's first row / the clamped column) and the value is discarded afterwards.
      const bool hcpro = (p.pro == PRO_LN_HC);
#pragma unroll
      for (int i = 0; i < NGMAX; ++i) {
        const int g = wave + 8 * i, gc = (FULL || g < KG) ? g : KG - 1;
        int tap, c; tap_c(i, gc, tap, c);
        const int toff = (tap == 0) ? p.tap_off[0] : ((tap == 1) ? p.tap_off[1] : p.tap_off[2]);
        const bool centre = ln && tap == ctap;                                  // uniform: a scalar select of the base pointer
        const int cc = (FULL || c < p.cin) ? c : p.cin - 4;                  // pad columns of a narrow input: read in range, zeroed below
        const float* base = centre ? p.nrm.P : p.xsrc;
        const unsigned off = centre ? p_row + (unsigned)cc : xs_row + (unsigned)(toff * p.xs_stride) + (unsigned)cc;
        av[i] = ld4u(base, off);
      }
      // the two centre k-groups of wave w are 16 ctap + w and 16 ctap + 8 + w  (i = 2 ctap + e): channel (8 e + w) 16 + c4
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ce_ = (8 * e + wave) * 16 + c4;
        const unsigned ce = ln ? (unsigned)((FULL || ce_ < p.cin) ? ce_ : p.cin - 4) : 0u;
        g1v[e] = ld4u(ln ? p.nrm.g1 : p.xsrc, ce); b1v[e] = ld4u(ln ? p.nrm.b1 : p.xsrc, ce);
        g2v[e] = ld4u(hcpro ? p.nrm.g2 : p.xsrc, hcpro ? ce : 0u); b2v[e] = ld4u(hcpro ? p.nrm.b2 : p.xsrc, hcpro ? ce : 0u);
        h2v[e] = ld4u(hcpro ? p.nrm.P : p.xsrc, hcpro ? p_row + 256u + ce : 0u);
        rsv[e] = ld4u(hcpro ? p.nrm.res : p.xsrc, hcpro ? rs_row + ce : 0u);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) st[g] = ld4u(ln ? p.stats_in : p.xsrc, (ln && valid) ? (unsigned)(prow * 64) + (unsigned)((aq * 4 + g) * 4) : 0u);
      {
        const int l = tid & 63, tile = ONE ? mytile : (tid >> 8), col = l & 15;
        const int pc = p.hc ? ((grp * MF + col) < p.cout ? tile * p.cout + grp * MF + col : 0)
                            : (((grp * 2 + tile) * MF + col) < p.cout ? (grp * 2 + tile) * MF + col : 0);
        if constexpr (NT == 1) {
          // v3: the per-row presum (bias + the older taps, computed off the critical path) takes the place of the bias.
          // Epilogue element of this thread: row (l >> 4) * 4 + ((tid >> 6) & 3) of the tile; R == 1, so the row index is b.
          const int erow = (l >> 4) * 4 + ((tid >> 6) & 3), em = m0 + erow;
          const bool eok = erow < p.tile_rows && em < p.M;
          const bool ps = p.presum != nullptr;                                       // uniform: scalar select of base and offset
          const unsigned boff = ps ? (eok ? (unsigned)((p.b0 + em) * p.presum_rstride + pc) : 0u) : (unsigned)pc;
          biasv = (ps ? p.presum : p.bias)[boff];
        } else {
          biasv = p.bias[(unsigned)pc];
        }
      }
      if constexpr (TRACE) { if (tr) p.dbg[2] = wall_clock64(); }
      // discard what the redirected loads fetched
#pragma unroll
      for (int i = 0; i < NGMAX; ++i) {
        const int g = wave + 8 * i;
        int tap, c; tap_c(i, g, tap, c);
        const bool centre = ln && tap == ctap;
        if constexpr (FULL) { if (!valid || (NT == 3 && tap == 2 && cmask)) av[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
        else { if (g >= KG || !valid || c >= p.cin) av[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
      }
    } else {
      // same branch-free issue as the 16-row form: clamped k-group / column, value discarded afterwards
#pragma unroll
      for (int i = 0; i < NGMAX; ++i) {
        const int g = wave + 8 * i, gc = g < KG ? g : KG - 1;
        const int k0 = gc * KGS, tap = (p.ntaps == 1) ? 0 : (k0 >> 8), c = k0 - tap * p.cin_p + c4;
        const int toff = (tap == 0) ? p.tap_off[0] : ((tap == 1) ? p.tap_off[1] : p.tap_off[2]);
        const int cc = c < p.cin ? c : p.cin - 4;
        av[i] = ld4u(p.xsrc, xs_row + (unsigned)(toff * p.xs_stride) + (unsigned)cc);
      }
      if constexpr (TRACE) { if (tr) p.dbg[2] = wall_clock64(); }
#pragma unroll
      for (int i = 0; i < NGMAX; ++i) {
        const int g = wave + 8 * i;
        const int k0 = g * KGS, tap = (p.ntaps == 1) ? 0 : (k0 >> 8), c = k0 - tap * p.cin_p + c4;
        if (g >= KG || !valid || c >= p.cin) av[i] = make_float4(0.f, 0.f, 0.f, 0.f);       // pad columns / skipped rows stay zero
      }
    }

    // ---- rebuild the centre-tap values: LN (+ act) or LN + sigmoid gate + highway mix, elementwise given the row statistics
    if constexpr (MF == 16) {
      if (ln) {
        float m1, r1, m2 = 0.f, r2 = 0.f;
        combine_stats(st, 0, m1, r1, FULL ? 16 : p.nrm.ngroups, aq * 4);   // every lane takes part in the cross-lane sums (ln is uniform)
        if (p.pro == PRO_LN_HC) combine_stats(st, 1, m2, r2);
#pragma unroll
        for (int i = 0; i < NGMAX; ++i) {
          const int g = wave + 8 * i;
          if (FULL || g < KG) {
            int tap, c; tap_c(i, g, tap, c);
            if (tap == ctap) {
              const float4 g1 = g1v[i & 1], b1 = b1v[i & 1];
              float4 x = av[i];
              x.x = (x.x - m1) * r1 * g1.x + b1.x; x.y = (x.y - m1) * r1 * g1.y + b1.y;
              x.z = (x.z - m1) * r1 * g1.z + b1.z; x.w = (x.w - m1) * r1 * g1.w + b1.w;
              if (p.pro == PRO_LN_HC) {
                const float4 g2 = g2v[i & 1], b2 = b2v[i & 1];
                const float4 h2 = h2v[i & 1], xr = rsv[i & 1];
                { const float s_ = sigmoid_fast(x.x); x.x = s_ * ((h2.x - m2) * r2 * g2.x + b2.x) + (1.0f - s_) * xr.x; }
                { const float s_ = sigmoid_fast(x.y); x.y = s_ * ((h2.y - m2) * r2 * g2.y + b2.y) + (1.0f - s_) * xr.y; }
                { const float s_ = sigmoid_fast(x.z); x.z = s_ * ((h2.z - m2) * r2 * g2.z + b2.z) + (1.0f - s_) * xr.z; }
                { const float s_ = sigmoid_fast(x.w); x.w = s_ * ((h2.w - m2) * r2 * g2.w + b2.w) + (1.0f - s_) * xr.w; }
              } else if (p.nrm.act == ACT_RELU) {
                x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f);
              } else if (!FULL && p.nrm.act == ACT_SIGMOID) {         // PRO_MEL: x = the logits of the mel frame (networks.py:210)
                if (valid && grp == 0 && mytile == 0 && p.xmat2 && c < p.cin)
                  *reinterpret_cast<float4*>(p.xmat2 + ((long)b * p.xm2_bstride + t + p.xm2_toff) * p.xm2_stride + c) = x;
                x.x = sigmoidf_(x.x); x.y = sigmoidf_(x.y); x.z = sigmoidf_(x.z); x.w = sigmoidf_(x.w);
              }
              if (!FULL && c >= p.cin) x = make_float4(0.f, 0.f, 0.f, 0.f);      // K padding of a narrow input
              if (!valid) x = make_float4(0.f, 0.f, 0.f, 0.f);
              av[i] = x;
              if (valid && grp == 0 && mytile == 0 && p.xmat && (FULL || c < p.cin))
                *reinterpret_cast<float4*>(p.xmat + par * p.xm_set + ((long)b * p.xm_bstride + p.xm_row0 + t) * p.xm_stride + c) = x;
            }
          }
        }
      }
    }
    if constexpr (TRACE) { if (tr) p.dbg[3] = wall_clock64(); }

    // ---- K loop (fully unrolled so the register arrays are statically indexed)
    typedef typename std::conditional<MF == 32, f32x16, f32x4>::type acc_t;
    acc_t acc0, acc1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < NGMAX; ++i) {
      const int g = wave + 8 * i;
      if (NG > 0 || FULL || g < KG) {
        const float4 a = av[i];
        const float4 b0 = bq0[i % BD], b1 = bq1[i % BD];
        if constexpr (MF == 32) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
        } else {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc0, 0, 0, 0); if constexpr (!ONE) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc0, 0, 0, 0); if constexpr (!ONE) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, acc0, 0, 0, 0); if constexpr (!ONE) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, acc0, 0, 0, 0); if constexpr (!ONE) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, acc1, 0, 0, 0);
        }
        // Refill the ring slot AFTER the MFMAs that read it were issued.  Issued before them (as it used to be) the refill is a
        // write-after-read on live MFMA operands: the compiler loaded into a temporary and put s_waitcnt vmcnt(0) right behind
        // every prefetch -- 16 serialised memory round trips per work item in the bulk kernel.
        if (i + BD < NGMAX) {
          const int gn = g + 8 * BD;
          const int gnc = gn < KG ? gn : KG - 1;
          bq0[i % BD] = ld4u(wb, w0o + (unsigned)gnc * 256u); bq1[i % BD] = ld4u(wb, w1o + (unsigned)gnc * 256u);
          // keep the refill HERE: left alone, the machine scheduler sinks it to just before its consumer (4 steps later) to save
          // registers, and the ring degenerates into load -> s_waitcnt vmcnt(0) -> use
          if constexpr (NG > 0) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (TRACE) { if (tr) p.dbg[4] = wall_clock64(); }

    // ---- split-K reduction through LDS: red[wave][tile][j][lane], summed in a fixed order (deterministic)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      smem[((wave * 2 + 0) * NJ + j) * 64 + lane] = acc0[j];
      if constexpr (!ONE) smem[((wave * 2 + 1) * NJ + j) * 64 + lane] = acc1[j];
    }
    __syncthreads();
    if constexpr (TRACE) { if (tr) p.dbg[5] = wall_clock64(); }
    for (int e = tid; e < (ONE ? 1 : 2) * NJ * 64; e += 512) {
      const int l = e & 63, j = (e >> 6) % NJ, ltile = e / (64 * NJ), tile = ONE ? mytile : ltile;
      float v_ = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v_ += smem[((w * 2 + ltile) * NJ + j) * 64 + l];
      int row, col;
      if constexpr (MF == 32) { row = (j & 3) + 8 * (j >> 2) + 4 * (l >> 5); col = l & 31; }
      else                    { row = (l >> 4) * 4 + j;                      col = l & 15; }
      const long orow = s_prow[row];
      int pcol; bool ok;
      if (p.hc) { const int c = grp * MF + col; ok = c < p.cout; pcol = tile * p.cout + c; }
      else      { pcol = (grp * 2 + tile) * MF + col; ok = pcol < p.cout; }
      if constexpr (MF == 16 && NT == 1) {
        if (p.raw_out && ok && orow >= 0) p.raw_out[((long)orow * p.raw_bstride + p.raw_row0 + step) * p.raw_stride + pcol] = v_;
      }
      if constexpr (MF == 16) { if (ok) v_ += biasv; } else { if (ok) v_ += p.bias[pcol]; }
      if (ok && orow >= 0) p.pout[orow * p.np_out + pcol] = v_;
      if constexpr (MF == 16) {
        // partial LN statistics of this 16-column group: a DPP row (16 lanes) holds one output row's 16 columns
        if (p.stats_out) {
          const float mg = row16_sum(ok ? v_ : 0.f) * (1.0f / 16.0f);
          const float dv = ok ? v_ - mg : 0.f;
          const float m2g = row16_sum(dv * dv);
          if (ok && orow >= 0 && col == 0) {
            const int G = p.hc ? grp : grp * 2 + tile;
            float* so = p.stats_out + (orow * 16 + G) * 4 + (p.hc ? tile * 2 : 0);
            so[0] = mg; so[1] = m2g;
          }
        }
      }
    }
    if constexpr (TRACE) { if (tr) p.dbg[6] = wall_clock64(); if (p.dbg_wg && tid == 0 && wgid < 128) p.dbg_wg[2 * wgid + 1] = wall_clock64(); }
    if (MF == 32 && item + (int)gridDim.x < nitems) __syncthreads();      // s_prow / smem are reused by the next item
  }
}

// Row kernel for the bulk branch: X[b][t] = act / gate (LN(P[b*R + r])) for cone rows at offsets < 0.
// grid ceil(M/4), block 256 (wave per row).
struct LnRowsParams {
  int M, R, b0; const int* offs; const int* step; int step_val;
  int Rp;                                  // rows per utterance in P (0 = R; v3: R + 1, the last P row of an utterance is the chain's presum)
  int hc; RowNorm nrm;
  float* x; long x_bstride; long x_row0; int x_stride; long x_set;
};

__global__ void __launch_bounds__(256) ln_rows_kernel(const LnRowsParams p) {
  const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= p.M) return;
  const int bl = m / p.R, r = m - bl * p.R, b = p.b0 + bl;
  const int step = p.step_val + (p.step ? *p.step : 0);
  const long par = step & 1;
  const int t = step + (p.offs ? p.offs[r] : 0);
  if (t < 0) return;
  const long prow = (long)b * (p.Rp ? p.Rp : p.R) + r;
  const float4 x = p.hc ? norm_row_hc(p.nrm, prow, b, t, lane, par) : norm_row_c(p.nrm, prow, lane);
  *reinterpret_cast<float4*>(p.x + par * p.x_set + ((long)b * p.x_bstride + p.x_row0 + t) * p.x_stride + lane * 4) = x;
}

}  // namespace dctts
