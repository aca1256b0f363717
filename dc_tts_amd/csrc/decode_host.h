// decode_host.h -- the host side of the decode (synthesize.py:45-54): workspaces, the launch plans of both decode forms (mode 3: two streams,
// incremental; mode 0: one stream, plain), their device-side tables, and the C ABI entry points dctts_text2mel_decode / dctts_synthesize /
// dctts_decode_status / dctts_set_decode_{graph,mode}.  Included by dctts_api.hip (one translation unit: everything here uses its context
// struct and helpers); not a header for anybody else.
#pragma once

// ------------------------------------------------------------------------------------------------ decode (synthesize.py:45-54)
struct DecodeWs {
  View kv, ypad, rbuf, logits; std::vector<View> ae, ad; int* pm_all; int* step;
  std::vector<float*> se, sd;          // per-column-group partial LN statistics of pe / pd: [B][16][4]
  std::vector<float*> pe, pd, pb;      // pre-norm rows: AudioEnc chain [B][np], AudioDec chain [B][np], AudioDec bulk [B*Rb][np]
  // v3
  float* vw = nullptr;                 // V . W_top  [B*N][d]
  View c1q;                            // Q[t] . W_bot, absolute time
  std::vector<float*> pse; long pse_set = 0;   // AudioEnc presums per k=3 layer, two parity copies [2][B][2d] (bias + older taps of row f in copy f & 1)
  std::vector<float*> pb3; long pb3_set[16] = {0};   // AudioDec cone pre-norm rows + presum row, two parity copies: [2][B*(Rb+1)][np]
  float* ps0 = nullptr;                // AudioDec C_1 presum [B][d] (attnq_kernel)
  float* vww = nullptr;                // VW . diag(gamma1) W2[q]  [B*N][3][2d]  (rowhc2_kernel)
  View c1qw;                           // C1Q[t] . diag(gamma1) W2[q], absolute time, stride 3 * 2d
  View scal;                           // rowc1_kernel's per-row scalars, absolute time, stride 8
};

static int decode_ws(dctts_ctx* c, int B, int N, int T, DecodeWs* w, hipStream_t st) {
  ws_select(c, "dec.", geom("dec", B, T, N), st);      // (captured graphs are keyed by their own geometry strings and rebuilt where they are used)
  const int d = c->cfg.d, nm = c->cfg.n_mels;
  const long rows = PAD + T + 2;
  // ypad row (PAD + t) holds S[t] = Y[t-1]  (train.py:51); row PAD is the zero frame fed at t = 0
  CHK(ws_view(c, "dec.ypad", B, rows, PAD, nm, &w->ypad));
  CHK(ws_view2(c, "dec.rbuf", B, rows, PAD, 2 * d, &w->rbuf));
  CHK(ws_view(c, "dec.logits", B, T, 0, nm, &w->logits));
  w->ae.resize(c->audioenc.size()); w->ad.resize(c->audiodec.size());
  void* p;
  for (size_t i = 0; i < w->ae.size(); ++i) CHK(ws_view(c, "dec.ae" + std::to_string(i), B, rows, PAD, d, &w->ae[i]));
  for (size_t i = 0; i + 1 < w->ad.size(); ++i) {
    // layers whose cone reaches rows < j are rewritten by the bulk stream one frame ahead: two parity copies
    if (c->cone_len[i] > 1) CHK(ws_view2(c, "dec.ad" + std::to_string(i), B, rows, PAD, d, &w->ad[i]));
    else CHK(ws_view(c, "dec.ad" + std::to_string(i), B, rows, PAD, d, &w->ad[i]));
  }
  w->pe.resize(c->audioenc.size()); w->pd.resize(c->audiodec.size()); w->pb.resize(c->audiodec.size());
  w->se.resize(c->audioenc.size()); w->sd.resize(c->audiodec.size());
  for (size_t i = 0; i < w->se.size(); ++i) { CHK(ws_get(c, "dec.se" + std::to_string(i), (size_t)B * 64 * sizeof(float), &p)); w->se[i] = (float*)p; }
  for (size_t i = 0; i < w->sd.size(); ++i) { CHK(ws_get(c, "dec.sd" + std::to_string(i), (size_t)B * 64 * sizeof(float), &p)); w->sd[i] = (float*)p; }
  for (size_t i = 0; i < w->pe.size(); ++i) { const int np = c->audioenc[i].hc ? 2 * d : c->audioenc[i].cout; CHK(ws_get(c, "dec.pe" + std::to_string(i), (size_t)B * np * sizeof(float), &p)); w->pe[i] = (float*)p; }
  for (size_t i = 0; i < w->pd.size(); ++i) {
    const int np = c->audiodec[i].hc ? 2 * d : c->audiodec[i].cout;
    CHK(ws_get(c, "dec.pd" + std::to_string(i), (size_t)B * np * sizeof(float), &p)); w->pd[i] = (float*)p;
    const int Rb = c->cone_len[i] - 1;
    w->pb[i] = nullptr;
    if (Rb > 0) { CHK(ws_get(c, "dec.pb" + std::to_string(i), (size_t)B * Rb * np * sizeof(float), &p)); w->pb[i] = (float*)p; }
  }
  CHK(ws_get(c, "dec.pm", (size_t)(T + 2) * B * sizeof(int), &p)); w->pm_all = (int*)p;
  CHK(ws_get(c, "dec.step", 256, &p)); w->step = (int*)p;
  if (c->decode_mode == 3 || c->decode_mode == 4) {
    CHK(ws_get(c, "dec.vw", (size_t)B * N * d * sizeof(float), &p)); w->vw = (float*)p;
    CHK(ws_view(c, "dec.c1q", B, rows, PAD, d, &w->c1q));
    CHK(ws_get(c, "dec.ps0", (size_t)B * d * sizeof(float), &p)); w->ps0 = (float*)p;
    CHK(ws_get(c, "dec.vww", (size_t)B * N * 6 * d * sizeof(float), &p)); w->vww = (float*)p;
    CHK(ws_view(c, "dec.c1qw", B, rows, PAD, 6 * d, &w->c1qw));
    CHK(ws_view(c, "dec.scal", B, rows, PAD, 8, &w->scal));
    w->pse.assign(c->audioenc.size(), nullptr);
    for (size_t i = 0; i < w->pse.size(); ++i)
      if (c->audioenc[i].wpp) { CHK(ws_get(c, "dec.pse" + std::to_string(i), (size_t)2 * B * 2 * d * sizeof(float), &p)); w->pse[i] = (float*)p; }
    w->pse_set = (long)B * 2 * d;
    w->pb3.assign(c->audiodec.size(), nullptr);
    for (size_t i = 0; i < w->pb3.size(); ++i) {
      if (!c->audiodec[i].wpp) continue;
      const size_t n = (size_t)B * c->cone_len[i] * 2 * d;          // cone_len rows: (cone_len - 1) bulk rows + the presum row
      CHK(ws_get(c, "dec.pb3_" + std::to_string(i), 2 * n * sizeof(float), &p)); w->pb3[i] = (float*)p; w->pb3_set[i] = (long)n;
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ decode v1: fused kernels, one stream
static int decode_step_launch(dctts_ctx* c, const DecodeWs& w, int B, int N, hipStream_t st) {
  const int d = c->cfg.d;
  // AudioEnc: one new row per utterance, taps read the per-layer history
  const RowMap r1{B, 1, nullptr, w.step};
  View cur = w.ypad;
  for (size_t i = 0; i < c->audioenc.size(); ++i) { CHK(run_conv(c, c->audioenc[i], cur, nullptr, w.ae[i], r1, st)); cur = w.ae[i]; }
  // windowed attention for the rows AudioDec C_1 must emit, with the CURRENT window
  AttnWinParams a;
  a.Qh = w.ae.back().p; a.q_bstride = w.ae.back().bstride; a.q_row0 = w.ae.back().row0; a.q_stride = d;
  a.K = w.kv.p; a.V = w.kv.p + d; a.kv_stride = 2 * d; a.kv_bstride = N;
  a.N = N; a.d = d; a.win = c->cfg.attention_win_size;
  a.step = w.step; a.step_val = 0; a.offs = c->cone_dev[0]; a.R = c->cone_len[0];
  a.pm_all = w.pm_all; a.B = B;
  a.rbuf = w.rbuf.p; a.r_bstride = w.rbuf.bstride; a.r_row0 = w.rbuf.row0; a.r_set = 0;
  hipLaunchKernelGGL(attention_window_kernel, dim3((a.R + 3) / 4, B), dim3(256), 0, st, a);
  HIPCHK(hipGetLastError());
  // AudioDec dependency cone
  cur = w.rbuf;
  const size_t nl = c->audiodec.size();
  for (size_t i = 0; i < nl; ++i) {
    const RowMap rm{B, c->cone_len[i], c->cone_dev[i], w.step};
    if (i + 1 == nl) {
      // sigmoid(logits) of frame j becomes S[j+1]: write at ypad row (PAD + 1 + j); raw logits kept per frame
      View yo = w.ypad; yo.row0 = w.ypad.row0 + 1;
      CHK(run_conv(c, c->audiodec[i], cur, nullptr, yo, rm, st, 0, &w.logits));
    } else {
      CHK(run_conv(c, c->audiodec[i], cur, nullptr, w.ad[i], rm, st));
      cur = w.ad[i];
    }
  }
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(64), 0, st, w.step);
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ decode v2: split kernels, two streams
static size_t hsplit_smem(int MF) {
  return (size_t)8 * 2 * (MF == 32 ? 16 : 4) * 64 * sizeof(float);      // split-K reduction buffer
}

static RowNorm make_norm(const DevLayer& prod, const float* P, const View* res) {
  RowNorm n; memset(&n, 0, sizeof(n));
  n.P = P; n.np = prod.hc ? 2 * prod.cout : prod.cout;
  n.g1 = prod.g1; n.b1 = prod.b1; n.g2 = prod.g2; n.b2 = prod.b2; n.act = prod.act;
  n.ngroups = (prod.cout + 15) / 16;
  if (res) { n.res = res->p; n.res_bstride = res->bstride; n.res_row0 = res->row0; n.res_stride = res->stride; n.res_set = res->set; }
  return n;
}

struct SplitExtra {                     // decode v3 additions to a split launch
  const float* presum = nullptr; int presum_rstride = 0;   // chain: per-row presum replaces the bias
  const View* raw = nullptr;                                // chain: bare contraction -> raw[b][frame]
  int mask_last = 0;                                        // bulk: the last row of every utterance is a presum row
  int np_out = 0;                                           // bulk: floats per output row when the output is a slice of wider rows
};

// One split GEMM launch for frame `frame`.  MF = 16: chain (newest frame, R = 1, offs = null); MF = 32: bulk (cone rows at offsets < 0).
static int run_split(dctts_ctx* c, int MF, const DevLayer& L, int B, int R, const int* offs, int frame, int pro, const RowNorm* nrm,
                     const View* xmat, const View& xsrc, float* pout, hipStream_t st,
                     const float* stats_in = nullptr, float* stats_out = nullptr, int tile_rows16 = 0, const View* xmat2 = nullptr,
                     int xm2_toff = 0, const SplitExtra* ex = nullptr) {
  SplitParams p; memset(&p, 0, sizeof(p));
  if (ex) {
    p.presum = ex->presum; p.presum_rstride = ex->presum_rstride; p.mask_last = ex->mask_last;
    if (ex->np_out && MF != 32) return fail(DCTTS_ERR_STATE, "split kernel: output slices belong to the 32-row form");
    if (ex->raw) { p.raw_out = ex->raw->p; p.raw_bstride = ex->raw->bstride; p.raw_row0 = ex->raw->row0; p.raw_stride = ex->raw->stride; }
    if ((ex->presum || ex->raw) && (MF != 16 || R != 1)) return fail(DCTTS_ERR_STATE, "split kernel: presum / raw output belong to the chain (16-row form, one row per utterance)");
    if (ex->mask_last && (L.ntaps != 3 || L.cin_p != 256 || L.tap_off[2] != 0 || pro != PRO_RAW)) return fail(DCTTS_ERR_STATE, "split kernel: presum rows belong to a causal 3-tap layer over 256 channels");
  }
  if (xmat2) { p.xmat2 = xmat2->p; p.xm2_bstride = xmat2->bstride; p.xm2_stride = xmat2->stride; p.xm2_toff = xm2_toff; }
  p.M = B * R; p.R = R; p.b0 = 0; p.offs = offs; p.step = nullptr; p.step_val = frame;
  p.pro = pro; if (nrm) p.nrm = *nrm;
  if (xmat) { p.xmat = xmat->p; p.xm_bstride = xmat->bstride; p.xm_row0 = xmat->row0; p.xm_stride = xmat->stride; p.xm_set = xmat->set; }
  p.xsrc = xsrc.p; p.xs_bstride = xsrc.bstride; p.xs_row0 = xsrc.row0; p.xs_stride = xsrc.stride; p.xs_set = xsrc.set;
  p.ntaps = L.ntaps; for (int j = 0; j < 3; ++j) p.tap_off[j] = L.tap_off[j];
  p.cin = L.cin; p.cin_p = L.cin_p;
  p.wp = (MF == 16) ? L.wp16 : L.wp; p.bias = L.bias; p.cout = L.cout; p.hc = L.hc ? 1 : 0;
  p.np_out = (ex && ex->np_out) ? ex->np_out : (L.hc ? 2 * L.cout : L.cout); p.pout = pout; p.stats_in = stats_in; p.stats_out = stats_out;
  if (pro == PRO_MEL && (MF != 16 || L.ntaps != 1 || !stats_in || !nrm || nrm->np != L.cin))
    return fail(DCTTS_ERR_STATE, "split kernel: the mel prologue feeds a k = 1 layer whose input width is the mel row");
  if (pro != PRO_RAW && pro != PRO_MEL && (MF != 16 || L.cin_p != 256 || !stats_in))
    return fail(DCTTS_ERR_STATE, "split kernel: LN prologue needs the 16-row form, 256 input channels and producer statistics");
  if (L.ntaps > 1 && L.cin_p != 256) return fail(DCTTS_ERR_STATE, "split kernel: multi-tap layers must have 256 input channels");
  const int groups = L.hc ? L.cout / MF : (L.cout + 2 * MF - 1) / (2 * MF);
  p.ngroups = groups;
  p.tile_rows = (MF == 16) ? (tile_rows16 ? tile_rows16 : 8) : MF;     // chain: 8 rows per workgroup (half of an MFMA tile: half the activation bytes per CU)
  int nblk = ((p.M + p.tile_rows - 1) / p.tile_rows) * groups;
  if (MF == 32 && nblk > c->bulk_cap) nblk = c->bulk_cap;
  const size_t sm = hsplit_smem(MF);
  // specialised 16-row forms: causal k = 3 over 256 channels (NT = 3), k = 1 over 256 channels (NT = 1); anything else is generic
  int nt = 0;
  if (MF == 16 && L.cin == L.cin_p && L.cin_p == 256) nt = (L.ntaps == 3 && L.tap_off[2] == 0) ? 3 : (L.ntaps == 1 ? 1 : 0);
  else if (MF == 16 && L.ntaps == 1 && L.cin == L.cin_p && L.cin_p == 512) nt = 2;
  else if (MF == 16 && L.ntaps == 1 && L.cin_p <= 128) nt = 4;
  if (ex && (ex->presum || ex->raw) && nt != 1) return fail(DCTTS_ERR_STATE, "split kernel: presum / raw output need the k = 1 x 256-channel chain form");
  // the chain-only forms (NT = 1, 2, 4) take a (column groups, row tiles) grid and assume one row per utterance
  const bool chainrow = (MF == 16) && (nt == 1 || nt == 2 || nt == 4);
  if (chainrow && (R != 1 || offs)) return fail(DCTTS_ERR_STATE, "split kernel: the chain forms take one row per utterance and no offset table");
  const dim3 grid16 = chainrow ? dim3(groups, (p.M + p.tile_rows - 1) / p.tile_rows) : dim3(nblk);
#define DCTTS_LAUNCH16(NTV) hipLaunchKernelGGL((hsplit_kernel<16, false, 0, NTV, false>), grid16, dim3(512), sm, st, p)
  if (MF == 16) {
    if (nt == 3) DCTTS_LAUNCH16(3); else if (nt == 1) DCTTS_LAUNCH16(1); else if (nt == 2) DCTTS_LAUNCH16(2);
    else if (nt == 4) DCTTS_LAUNCH16(4); else DCTTS_LAUNCH16(0);
  }
#undef DCTTS_LAUNCH16
  else {
    const int kg = L.ntaps * L.cin_p / 8;                     // k-groups of 8; the 32-row form is instantiated per K (straight-line K loop)
    const bool plain = (pro == PRO_RAW) && L.cin == L.cin_p;   // hbulk_kernel: software-pipelined across items
    if (ex && ex->mask_last && !(kg == 96 && plain)) return fail(DCTTS_ERR_STATE, "split kernel: presum rows need hbulk_kernel<12>");
    const bool profb = c->prof_id == DCTTS_PROF_BULK_GEMM && kg == 96 && plain;
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    if (profb) { HIPCHK(hipEventCreate(&pe0)); HIPCHK(hipEventCreate(&pe1)); HIPCHK(hipEventRecord(pe0, st)); }
    if (kg == 96 && plain)      hipLaunchKernelGGL((hbulk_kernel<12>), dim3(nblk), dim3(512), sm, st, p);
    else if (kg == 64 && plain) hipLaunchKernelGGL((hbulk_kernel<8>), dim3(nblk), dim3(512), sm, st, p);
    else if (kg == 32 && plain) hipLaunchKernelGGL((hbulk_kernel<4>), dim3(nblk), dim3(512), sm, st, p);
    else return fail(DCTTS_ERR_STATE, "split kernel (32-row form): raw input rows, K = 256, 512 or 768");
    if (profb) { HIPCHK(hipEventRecord(pe1, st)); c->prof_ev.emplace_back(pe0, pe1); c->prof_cnt.push_back(1); c->prof_rows += p.M; }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- Do two streams really run at the same time?  The decode's two streams wait for EACH OTHER inside kernels (and, in the fallback forms, through stream memory
// operations), so they must sit on different hardware queues.  HIP maps the streams of one priority onto a small pool of queues (four by default) and lets a new
// stream SHARE the queue of an existing one once the pool is full; two streams that share a queue run their commands one after the other.  Round 6 met exactly that:
// a caller that had created a high-priority stream of its own (torch.cuda.Stream(priority=-1)) beside one engine's pair left the NEXT engine's pair on one queue --
// every decode of that engine then ran into its bounded waits (error word 36; with stream-operation meetings it would have hung).  So the pair is TESTED when it is
// created: a kernel on one stream waits (bounded, ~0.5 s: a device that is busy with other work -- another context's SSRN, another process -- must not look like a shared
// queue) for a flag a kernel on the other stream sets; if it times out the second stream is replaced by a new one (the rejected ones are held until the end, so that
// the pool's next pick is a different queue).
__global__ void conc_wait_kernel(const unsigned* __restrict__ flag, unsigned* __restrict__ out) {
  unsigned ok = 0;
  for (int i = 0; i < (1 << 19) && !ok; ++i) {
    ok = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
    if (!ok) __builtin_amdgcn_s_sleep(32);
  }
  __hip_atomic_store(out, ok ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void conc_set_kernel(unsigned* __restrict__ flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
static int streams_run_concurrently(dctts_ctx* c, hipStream_t a, hipStream_t b, bool* ok) {
  if (!c->conc_flags) { HIPCHK(hipMalloc((void**)&c->conc_flags, 64 * sizeof(unsigned))); }
  HIPCHK(dev_zero_now(c->conc_flags, 64 * sizeof(unsigned)));
  hipLaunchKernelGGL(conc_wait_kernel, dim3(1), dim3(1), 0, a, (const unsigned*)c->conc_flags, c->conc_flags + 32);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(conc_set_kernel, dim3(1), dim3(1), 0, b, c->conc_flags);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(a)); HIPCHK(hipStreamSynchronize(b));
  unsigned r = 0;
  HIPCHK(hipMemcpy(&r, c->conc_flags + 32, sizeof(unsigned), hipMemcpyDeviceToHost));
  *ok = (r == 1u);
  return 0;
}

static int decode_streams_init(dctts_ctx* c) {
  if (c->s_bulk) return 0;
  {
    // Rounds 2-3: the side ("bulk") stream was throughput work that only had to finish within a frame period -- lowest priority, so that the dispatcher preferred the
    // chain's launches.  Since round 4 the side stream is as long as the chain, and its team kernel (xcone_kernel) needs all of its 128 workgroups resident to get
    // through its barriers: with another stream's big kernels on the device (SSRN of the previous batch: tools/soak.py, phase C) a low-priority xcone workgroup could
    // wait for a CU longer than its team-mates' bounded spins (20 ms; 76 of 1500 decodes reported it).  Highest priority: a freed CU goes to the decode first.
    int lo = 0, hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(hipStreamCreateWithPriority(&c->s_bulk, hipStreamNonBlocking, hi));
    // ... and so do the chain's launches: they run on a high-priority stream of the context, between two events on the caller's stream (decode_impl).  On the caller's
    // own (default-priority) stream the chain's team kernels met the same fate as xcone_kernel once that one had priority: 9 of 1500 decodes beside SSRN + vocoder
    // reported a team hand-off time-out of the chain.
    if (!c->s_chain) HIPCHK(hipStreamCreateWithPriority(&c->s_chain, hipStreamNonBlocking, hi));      // (a call that failed further down may come again)
    if (!c->ev_in) HIPCHK(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
    if (!c->ev_out) HIPCHK(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
    // the pair must run concurrently (above): replace the side stream until it does
    std::vector<hipStream_t> rejected;
    bool ok = false;
    int rc = 0;
    // (not when the streams meet through EVENTS only -- DCTTS_SYNC_VALUES=0, what rocprofv3 --pmc selects: counter collection runs dispatches one at a time across
    //  all queues, the test could only fail there, and event meetings need no concurrency)
    if (!c->sync_values) ok = true;
    for (int attempt = 0; attempt < 8 && !ok; ++attempt) {
      rc = streams_run_concurrently(c, c->s_chain, c->s_bulk, &ok);
      if (rc != 0 || ok) break;
      rejected.push_back(c->s_bulk); c->s_bulk = nullptr;
      if (hipStreamCreateWithPriority(&c->s_bulk, hipStreamNonBlocking, hi) != hipSuccess) { rc = fail(DCTTS_ERR_HIP, "hipStreamCreateWithPriority (decode side stream)"); break; }
    }
    c->stream_retries = (int)rejected.size();
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    if (rc != 0) return rc;
    if (!ok) {
      if (c->s_bulk) { (void)hipStreamDestroy(c->s_bulk); c->s_bulk = nullptr; }
      return fail(DCTTS_ERR_STATE, "decode: could not obtain two high-priority streams that run concurrently (the process holds too many high-priority streams: HIP lets "
                                   "new streams share a hardware queue); the decode's two streams wait for each other and must not share one");
    }
  }
  // the two streams hand data to each other through device memory only: device-scope release on the event markers (no system-scope flush)
  const unsigned evf = (unsigned)hipEventReleaseToDevice | hipEventDisableTiming;
  HIPCHK(hipEventCreateWithFlags(&c->ev_fork, evf));
  for (int i = 0; i < 4; ++i) { HIPCHK(hipEventCreateWithFlags(&c->ev_chain[i], evf)); HIPCHK(hipEventCreateWithFlags(&c->ev_bulk[i], evf)); }
  HIPCHK(hipFuncSetAttribute((const void*)hbulk_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)hbulk_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)hbulk_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)hbulk_group_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)xgroup_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));      // (its passengers: hbulk_body items)
  HIPCHK(hipFuncSetAttribute((const void*)xgroup_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  return 0;
}

static void destroy_graphs(dctts_ctx* c) {
  for (hipGraphExec_t g : c->bulk3_g) if (g) (void)hipGraphExecDestroy(g);
  c->bulk3_g.clear(); c->graphs3_geom.clear();
}

template <typename F>
static int capture_piece(hipStream_t cs, hipGraphExec_t* out, F&& body) {
  hipGraph_t gr = nullptr;
  HIPCHK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
  const int rc = body();
  const hipError_t e = hipStreamEndCapture(cs, &gr);
  if (rc != 0) { if (gr) (void)hipGraphDestroy(gr); return rc; }
  HIPCHK(e);
  HIPCHK(hipGraphInstantiate(out, gr, nullptr, nullptr, 0));
  (void)hipGraphDestroy(gr);
  return 0;
}

// ---- the decode's device tables are cached per (kind, key): a key spells everything a table depends on (geometry, buffer addresses, flags), the buffers of a
// geometry are stable (dctts_ctx::ws), so a geometry that comes back finds its tables again -- nothing is freed on a shape change (round 5; ws_trim drops all)
static bool tab_lookup(dctts_ctx* c, const char* kind, const std::string& key, dctts_ctx::TabSlot* out) {
  auto it = c->tabcache.find(std::string(kind) + "|" + key);
  if (it == c->tabcache.end()) return false;
  *out = it->second; return true;
}
static void tab_store(dctts_ctx* c, const char* kind, const std::string& key, void* tab, size_t bytes, void* mem = nullptr, int n0 = 0) {
  dctts_ctx::TabSlot s; s.tab = tab; s.mem = mem; s.n0 = n0; s.bytes = bytes;
  c->tabcache[std::string(kind) + "|" + key] = s;
}
static void drop_decode_tables(dctts_ctx* c) {      // (the caller has synchronised the device)
  for (auto& kv : c->tabcache) { if (kv.second.tab) (void)hipFree(kv.second.tab); if (kv.second.mem) (void)hipFree(kv.second.mem); }
  c->tabcache.clear();
  c->aepre_tab = c->mlp_tab = c->xmlp_tab = c->xtail_tab = c->xg_tab = c->xc_tab = nullptr; c->xg_mem = nullptr;
  c->aepre_geom.clear(); c->mlp_geom.clear(); c->xmlp_geom.clear(); c->xtail_geom.clear(); c->xg_geom.clear(); c->xc_geom.clear();
  if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
  if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
  c->graph_geom.clear();
  destroy_graphs(c);
}

// ------------------------------------------------------------------------------------------------ decode v3: hoisted taps (decode3_kernels.h)
// chain piece j (j = -1 .. T-1), caller's stream:
//     AudioDec HC_2 .. C_11 of frame j  [j >= 0; needs bulk piece j]
//     AudioEnc C_1 .. HC_13 of frame j+1 (C_1's prologue finalises mel frame j) [needs the AudioEnc presums of bulk piece j+1]
//     attnq(j+1): Q[j+1], window of frame j+2, C_1's presum;  AudioDec C_1 of frame j+1 (K = 256 on Q[j+1]; also C1Q[j+1])
// bulk piece f (f = 0 .. T-1), side stream, after chain piece f-2, overlapping chain piece f-1:
//     AudioEnc presums for row f (one grouped launch) | C_1 cone rows (rowc1_kernel) | per k=3 AudioDec layer: cone rows + the
//     presum row of frame f in one GEMM, then LN / gate of the cone rows.
static inline int aepre_stride(const dctts_ctx* c) { return c->aepre_layers + 3; }      // descriptors per parity copy of the table (v3_aepre_table)
static int v3_aepre_table(dctts_ctx* c, const DecodeWs& w, int B, bool c1qw_ahead) {
  const std::string g = std::to_string(B) + ":" + std::to_string((int)c1qw_ahead) + ":" + std::to_string((size_t)w.ae[0].p) + ":" + std::to_string((size_t)w.pse.back()) + ":" + std::to_string((size_t)w.c1qw.p) + ":" + std::to_string((int)c->chain_one);
  if (c->aepre_tab && c->aepre_geom == g) return 0;
  { dctts_ctx::TabSlot ts; if (tab_lookup(c, "aepre", g, &ts)) { c->aepre_tab = ts.tab; c->aepre_layers = ts.n0; c->aepre_geom = g; return 0; } }
  std::vector<SplitParams> tab;
  const std::vector<DevLayer>& AP = c->ae_p;
  for (int par = 0; par < 2; ++par)
    for (size_t i = 0; i < AP.size(); ++i) {
      if (!AP[i].wpp) continue;
      const DevLayer& L = AP[i];
      if (i == 0 || L.cin_p != 256 || L.cout != 256) return fail(DCTTS_ERR_STATE, "v3: AudioEnc k=3 layers must be 256 -> 2 x 256");
      SplitParams p; memset(&p, 0, sizeof(p));
      p.M = B; p.R = 1; p.ngroups = L.cout / 32; p.tile_rows = 32; p.pro = PRO_RAW;
      const View& x = w.ae[i - 1];
      p.xsrc = x.p; p.xs_bstride = x.bstride; p.xs_row0 = x.row0; p.xs_stride = x.stride; p.xs_set = 0;
      p.ntaps = 2; p.tap_off[0] = L.tap_off[0]; p.tap_off[1] = L.tap_off[1]; p.cin = L.cin; p.cin_p = L.cin_p;
      p.wp = L.wpp; p.bias = L.bias; p.cout = L.cout; p.hc = 1; p.np_out = 2 * L.cout; p.pout = w.pse[i] + par * w.pse_set;
      tab.push_back(p);
      if (i + 1 == AP.size()) {
        // riding in the same launch: the newest row of the C1Q . diag(gamma1) W2[q] cache (row f-1 when the launch's step is f+1)
        for (int q = 0; q < 3; ++q) {
          const DevLayer& H = c->hc2_wt2[q];
          SplitParams h; memset(&h, 0, sizeof(h));
          h.M = B; h.R = 1; h.ngroups = H.cout / 32; h.tile_rows = 32; h.pro = PRO_RAW;
          h.xsrc = w.c1q.p; h.xs_bstride = w.c1q.bstride; h.xs_row0 = w.c1q.row0; h.xs_stride = w.c1q.stride; h.xs_set = 0;
          h.ntaps = 2; h.tap_off[0] = h.tap_off[1] = -2; h.cin = H.cin; h.cin_p = H.cin_p;
          h.wp = H.wp; h.bias = H.bias; h.cout = H.cout; h.hc = 1; h.np_out = 6 * H.cout; h.pout = w.c1qw.p + (long)q * 2 * H.cout;
          h.abs_bstride = w.c1qw.bstride; h.abs_row0 = w.c1qw.row0; h.abs_toff = -2;
          h.step_val = (c1qw_ahead && !c->chain_one) ? 1 : 0;       // launched from the chain's stream, one piece earlier (decode_v3): the row is the chain's newest C1Q row
                                                                    // (chain_one: the passengers' step is already one further: they compute AudioEnc's presums of row j + 2)
          tab.push_back(h);
        }
        // ... and the same three once more without the zero half of K (K = 256, hbulk_body<4>): xtail_kernel's passengers run these -- side-stream piece j + 1
        // waits for them (the fold's second phase polls their counter), and at K = 512 they took 9 - 11 us of which half was zeros (round 5)
        for (int q = 0; q < 3; ++q) {
          SplitParams h = tab[tab.size() - 3];
          const DevLayer& H = c->hc2_wt[q];
          h.ntaps = 1; h.tap_off[0] = h.tap_off[1] = -2; h.cin = H.cin; h.cin_p = H.cin_p; h.wp = H.wp;
          tab.push_back(h);
        }
      }
    }
  if (tab.empty()) return fail(DCTTS_ERR_STATE, "v3: no causal k=3 AudioEnc layers");
  c->aepre_tab = nullptr;
  HIPALLOC(hipMalloc(&c->aepre_tab, tab.size() * sizeof(SplitParams)));
  HIPCHK(hipMemcpy(c->aepre_tab, tab.data(), tab.size() * sizeof(SplitParams), hipMemcpyHostToDevice));
  c->aepre_layers = (int)tab.size() / 2 - 3; c->aepre_geom = g;      // (a parity copy = aepre_layers descriptors + the three K = 256 twins: aepre_stride)
  tab_store(c, "aepre", g, c->aepre_tab, tab.size() * sizeof(SplitParams), nullptr, c->aepre_layers);
  return 0;
}

// AudioEnc presums for row f (into parity copy f & 1): bias + the taps that are final a whole chain piece before row f is computed
// part 0: everything; 1: AudioEnc's presums only (the first aepre_layers - 3 descriptors); 2: the three C1QW descriptors only
static int v3_aepre(dctts_ctx* c, int B, int f, hipStream_t st, int part = 0, unsigned wait_val = 0) {
  const int ipl = ((B + 31) / 32) * (c->cfg.d / 32);
  const SplitParams* tab = (const SplitParams*)c->aepre_tab + (size_t)(f & 1) * aepre_stride(c);
  int n = c->aepre_layers;
  if (part == 1) n -= 3;
  if (part == 2) { tab += c->aepre_layers - 3; n = 3; }
  hipLaunchKernelGGL((hbulk_group_kernel<8>), dim3(n * ipl), dim3(512), hsplit_smem(32), st, tab, ipl, f,
                     wait_val ? (const unsigned*)c->wait_ctr : nullptr, wait_val, (int*)(c->wait_ctr + 64));
  HIPCHK(hipGetLastError());
  return 0;
}

static int v3_vw(dctts_ctx* c, const DecodeWs& w, int B, int N, hipStream_t st) {
  if (N > c->iota_n) return fail(DCTTS_ERR_ARG, "decode: N too large");
  const int d = c->cfg.d;
  const View v{w.kv.p + d, (long)N, 0, 2 * d, 0};                    // V = channels d..2d of TextEnc's output rows
  CHK(run_split(c, 32, c->ad_vw, B, N, c->iota_dev, 0, PRO_RAW, nullptr, nullptr, v, w.vw, st));
  const View vw{w.vw, (long)N, 0, d, 0};
  SplitExtra ex; ex.np_out = 6 * d;
  for (int q = 0; q < 3; ++q)                                           // VWW[n][q] = VW[n] . diag(gamma1) W2[q]
    CHK(run_split(c, 32, c->hc2_wt[q], B, N, c->iota_dev, 0, PRO_RAW, nullptr, nullptr, vw, w.vww + (long)q * 2 * d, st, nullptr, nullptr, 0, nullptr, 0, &ex));
  return 0;
}

// The two row operations of a side-stream piece (decode3_kernels.h): AudioDec C_1 and HC_2 over their cone rows for frame f.  Launches of their own, or -- round 5,
// the folded form -- the first two phases of xcone_kernel's launch, whose per-frame table then carries these parameters.
static RowC1Params fill_rowc1(dctts_ctx* c, const DecodeWs& w, int B, int N, int f) {
  const std::vector<DevLayer>& AD = c->audiodec;
  const int d = c->cfg.d;
  RowC1Params q; memset(&q, 0, sizeof(q));
  q.B = B; q.R = c->cone_len[0] - 1; q.offs = c->cone3_dev[0]; q.frame = f;
  q.Qh = w.ae.back().p; q.q_bstride = w.ae.back().bstride; q.q_row0 = w.ae.back().row0; q.q_stride = d;
  q.K = w.kv.p; q.k_stride = 2 * d; q.VW = w.vw; q.vw_stride = d; q.kv_bstride = N;
  q.C1Q = w.c1q.p; q.c_bstride = w.c1q.bstride; q.c_row0 = w.c1q.row0; q.c_stride = w.c1q.stride;
  q.bias = AD[0].bias; q.g = AD[0].g1; q.be = AD[0].b1;
  q.N = N; q.d = d; q.win = c->cfg.attention_win_size; q.pm_all = w.pm_all;
  q.x = w.ad[0].p; q.x_bstride = w.ad[0].bstride; q.x_row0 = w.ad[0].row0; q.x_stride = w.ad[0].stride; q.x_set = w.ad[0].set;
  q.scal = w.scal.p; q.s_bstride = w.scal.bstride; q.s_row0 = w.scal.row0;
  if (c->ae_pass && f > 0) { q.wait = c->wait_ctr + 16; q.wait_val = (unsigned)f; q.wait_err = (int*)(c->wait_ctr + 64); }      // row f - 1 of the C1Q . W2 cache: passengers of chain piece f - 1
  q.contig = c->cone_contig[0];
  return q;
}
static RowHc2Params fill_rowhc2(dctts_ctx* c, const DecodeWs& w, int B, int N, int f) {
  const std::vector<DevLayer>& AD = c->audiodec;
  const int par = f & 1;
  RowHc2Params q; memset(&q, 0, sizeof(q));
  const int R = c->cone_len[1];
  q.B = B; q.R = R; q.offs = c->cone3_dev[1]; q.frame = f;
  for (int t3 = 0; t3 < 3; ++t3) q.tap_off[t3] = AD[1].tap_off[t3];
  q.scal = w.scal.p; q.s_bstride = w.scal.bstride; q.s_row0 = w.scal.row0;
  q.VWW = w.vww; q.kv_bstride = N;
  q.C1QW = w.c1qw.p; q.c_bstride = w.c1qw.bstride; q.c_row0 = w.c1qw.row0;
  q.consts = c->hc2_consts; q.bias = AD[1].bias; q.g1 = AD[1].g1; q.b1 = AD[1].b1; q.g2 = AD[1].g2; q.b2 = AD[1].b2;
  q.x1 = w.ad[0].p; q.x1_bstride = w.ad[0].bstride; q.x1_row0 = w.ad[0].row0; q.x1_stride = w.ad[0].stride; q.x1_set = w.ad[0].set;
  q.x2 = w.ad[1].p; q.x2_bstride = w.ad[1].bstride; q.x2_row0 = w.ad[1].row0; q.x2_stride = w.ad[1].stride; q.x2_set = w.ad[1].set;
  q.presum = w.pb3[1] + (long)par * w.pb3_set[1] + (long)(R - 1) * 2 * AD[1].cout; q.presum_rstride = (long)R * 2 * AD[1].cout;
  q.N = N; q.win = c->cfg.attention_win_size; q.pm_all = w.pm_all;
  q.contig = c->cone_contig[1];
  return q;
}

static int v3_bulk_rest(dctts_ctx* c, const DecodeWs& w, int B, int N, int T, int f, hipStream_t sb, unsigned wait_val) {
  const int d = c->cfg.d;
  const std::vector<DevLayer>& AD = c->audiodec;
  const int par = f & 1;
  // one grouped launch: AudioEnc presums of row f+1 (consumed by chain piece f; inputs are rows <= f-1) and the newest row (f-1) of
  // the C1Q . diag(gamma1) W2 cache that rowhc2_kernel reads below
  // (with the team kernels the side stream is the longer one: AudioEnc's presums then run on the chain's stream, in front of the piece that uses them)
  // (wait_val != 0: the launch first polls the chain's counter for that value -- the piece's input row comes from the chain's stream)
  if (!c->c1qw_chain) CHK(v3_aepre(c, B, f + 1, sb, c->xc_on ? 2 : 0, wait_val));
  const bool fold = c->xc_on && c->side_fold;            // round 5: both row operations are the first phases of xcone_kernel's launch (its per-frame table carries their parameters)
  if (c->cone_len[0] > 1 && !fold) {
    const RowC1Params q = fill_rowc1(c, w, B, N, f);
    hipLaunchKernelGGL(rowc1_kernel, dim3((q.R + ROWC1_NW - 1) / ROWC1_NW, B), dim3(ROWC1_NW * 64), 0, sb, q);
    HIPCHK(hipGetLastError());
  }
  size_t first_gemm = 1;
  if (AD.size() > 1 && AD[1].wpp && AD[1].tap_off[1] == -1) {
    // HC_2 over its cone rows + its presum row: a row operation on the cached V.W / Q.W products (no GEMM, no separate LN pass)
    if (!fold) {
      const RowHc2Params q = fill_rowhc2(c, w, B, N, f);
      hipLaunchKernelGGL(rowhc2_kernel, dim3((q.R + ROWHC2_NW - 1) / ROWHC2_NW, B), dim3(ROWHC2_NW * 64), 0, sb, q);
      HIPCHK(hipGetLastError());
    }
    first_gemm = 2;
  }
  if (c->side_pre && f + 1 < T) CHK(v3_aepre(c, B, f + 1, sb, 1));      // AudioEnc's presums of row f + 1 (inputs: rows <= f - 1): consumed by the AudioEnc run of chain piece f, which starts behind this piece
  if (c->xc_on) {                                        // HC_3 .. HC_7 and their row passes: one launch, teams inside one XCD (xcone_kernel.h)
    const XConeParams* xp = (const XConeParams*)c->xc_tab + f;
    const bool prof = c->prof_id == DCTTS_PROF_XCONE && f >= 100 && (f & 15) == 8;          // full-size cones only; eager decode only (graph mode 0)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, sb)); }
    hipLaunchKernelGGL(xcone_kernel, dim3(128), dim3(512), 0, sb, xp, (const int*)(w.pm_all + (long)f * B));
    HIPCHK(hipGetLastError());
    if (prof) { HIPCHK(hipEventRecord(e1, sb)); c->prof_ev.emplace_back(e0, e1); c->prof_cnt.push_back(1); long rows = 0; for (size_t i = 2; i < (c->tail_on ? (size_t)(c->np_eff + 1) : AD.size()); ++i) if (AD[i].hc) rows += (long)B * c->cone_len[i]; c->prof_rows += rows; }
    return 0;
  }
  for (size_t i = first_gemm; i < AD.size(); ++i) {
    if (!AD[i].wpp) continue;
    const int R = c->cone_len[i], Rb = R - 1;                          // Rb cone rows at offsets < 0, then the presum row (offset 0)
    float* pout = w.pb3[i] + (long)par * w.pb3_set[i];
    SplitExtra ex; ex.mask_last = 1;
    // few rows (the last cone layers): 16-row x 16-channel-group items spread them over (rows / 16) x 16 short workgroups instead
    // of one latency-bound round of a handful of 32-row items
    const int mf = (R <= 16) ? 16 : 32;          // by rows per utterance, not by B: results stay bitwise shard-invariant
    CHK(run_split(c, mf, AD[i], B, R, c->cone3_dev[i], f, PRO_RAW, nullptr, nullptr, w.ad[i - 1], pout, sb, nullptr, nullptr, 16, nullptr, 0, &ex));
    if (Rb <= 0) continue;
    LnRowsParams q; memset(&q, 0, sizeof(q));
    q.M = B * Rb; q.R = Rb; q.Rp = R; q.b0 = 0; q.offs = c->cone3_dev[i]; q.step = nullptr; q.step_val = f; q.hc = 1;
    q.nrm = make_norm(AD[i], pout, &w.ad[i - 1]);
    q.x = w.ad[i].p; q.x_bstride = w.ad[i].bstride; q.x_row0 = w.ad[i].row0; q.x_stride = w.ad[i].stride; q.x_set = w.ad[i].set;
    hipLaunchKernelGGL(ln_rows_kernel, dim3((q.M + 3) / 4), dim3(256), 0, sb, q);
    HIPCHK(hipGetLastError());
  }
  return 0;
}


static int prof_close_run(dctts_ctx* c, hipStream_t st) {
  if (!c->prof_run_e0) return 0;
  hipEvent_t e1 = nullptr;
  HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e1, st));
  c->prof_ev.emplace_back(c->prof_run_e0, e1); c->prof_cnt.push_back(c->prof_run_n);
  c->prof_run_e0 = nullptr; c->prof_run_n = 0;
  return 0;
}

// One v3 chain layer on chain3_kernel (256 input channels).  `prod` = the layer whose pre-norm rows `P` (+ partial statistics
// `stats_in`) are this layer's input (nullptr: PRO_RAW, `xin` is the input row view); `res` = highway residual of `prod`;
// `xmat` = where the rebuilt input row is kept.  The frame offset is folded into every base pointer here.
static int run_chain3(dctts_ctx* c, const DevLayer& L, int B, int j, const DevLayer* prod, const float* P, const float* stats_in,
                      const View* res, const View* xmat, const View* xin, float* pout, float* stats_out, const SplitExtra* ex, hipStream_t st) {
  if (L.cin != 256 || L.cin_p != 256 || L.ntaps != (L.tap2 ? 2 : 1) || !L.wp16) return fail(DCTTS_ERR_STATE, "chain3: k = 1 over 256 channels");
  if (L.tap2 && (!prod || !xmat || !L.hc)) return fail(DCTTS_ERR_STATE, "chain3: the tap -1 form is a highway layer fed by a rebuilt row");
  Chain3Params p; memset(&p, 0, sizeof(p));
  const long par = j & 1;
  auto row = [&](const View& v) { return v.p + par * v.set + (v.row0 + j) * (long)v.stride; };
  p.B = B;
  int pro = PRO_RAW;
  if (prod) {
    pro = prod->hc ? PRO_LN_HC : PRO_LN_C;
    p.P = P; p.p_bs = prod->hc ? 2 * prod->cout : prod->cout; p.stats = stats_in;
    p.g1 = prod->g1; p.b1 = prod->b1; p.g2 = prod->g2; p.b2 = prod->b2; p.relu = (prod->act == ACT_RELU) ? 1 : 0;
    if (prod->cout != 256 || !stats_in) return fail(DCTTS_ERR_STATE, "chain3: producer must emit 256 channels with statistics");
    if (prod->hc) { if (!res) return fail(DCTTS_ERR_STATE, "chain3: highway producer needs its residual"); p.res = row(*res); p.res_bs = (int)(res->bstride * res->stride); }
    if (xmat) { p.xm = row(*xmat); p.xm_bs = (int)(xmat->bstride * xmat->stride); }
    if (L.tap2) { p.xt = row(*xmat) - xmat->stride; p.xt_bs = p.xm_bs; }          // the same history buffer, one time step back
  } else {
    if (!xin) return fail(DCTTS_ERR_STATE, "chain3: raw input row missing");
    p.P = row(*xin); p.p_bs = (int)(xin->bstride * xin->stride);
  }
  p.wp = L.wp16; p.add = L.bias; p.add_bs = 0;
  if (ex && ex->presum) { p.add = ex->presum; p.add_bs = ex->presum_rstride; }
  if (ex && ex->raw) { p.raw = row(*ex->raw); p.raw_bs = (int)(ex->raw->bstride * ex->raw->stride); }
  p.pout = pout; p.np_out = L.hc ? 2 * L.cout : L.cout; p.stats_out = stats_out; p.cout = L.cout;
  if (c->sig_next) { p.sig = c->sig_ptr; p.sig_val = c->sig_next; c->sig_next = 0; }
  if (c->wait2_next) {
    if (!(ex && ex->presum)) return fail(DCTTS_ERR_STATE, "chain3: the in-kernel wait guards a presum addend");
    p.wait2 = c->wait_ctr + 32; p.wait_val = c->wait2_next; p.gate_err = (int*)(c->wait_ctr + 64); c->wait2_next = 0;
  }
  const dim3 grid(L.hc ? L.cout / 16 : (L.cout + 31) / 32, (B + 7) / 8);
  // measurement (dctts_hip_debug.h): HIP events on the launch stream around sampled launches of the time-dominant decode kernel
  // Consecutive launches of the kernel share ONE event pair (a pair around every 5 us launch measures its own marker packets:
  // 8.5 us instead of 5.3): the first opens the run, the next launch of anything else -- or the end of the piece -- closes it.
  const bool prof = c->prof_id == DCTTS_PROF_CHAIN_HC && c->prof_frame && pro == PRO_LN_HC && L.hc && !L.tap2;
  if (prof && !c->prof_run_e0) { HIPCHK(hipEventCreate(&c->prof_run_e0)); HIPCHK(hipEventRecord(c->prof_run_e0, st)); c->prof_run_n = 0; }
  if (!prof) CHK(prof_close_run(c, st));
#define C3(PRO_, HC_) hipLaunchKernelGGL((chain3_kernel<PRO_, HC_>), grid, dim3(512), 0, st, p)
  if (pro == PRO_LN_HC && L.hc && !L.tap2 && c->trace_on && c->trace_n < 64) {   // DCTTS_TRACE: stamped instantiation
    p.ts = c->trace_buf + 32 * 64 * c->trace_n++;
    hipLaunchKernelGGL((chain3_kernel<PRO_LN_HC, true, false, true>), grid, dim3(512), 0, st, p);
  } else
  if (L.tap2 && pro == PRO_LN_C) hipLaunchKernelGGL((chain3_kernel<PRO_LN_C, true, true>), grid, dim3(512), 0, st, p);
  else if (L.tap2 && pro == PRO_LN_HC) hipLaunchKernelGGL((chain3_kernel<PRO_LN_HC, true, true>), grid, dim3(512), 0, st, p);
  else if (pro == PRO_RAW && !L.hc) C3(PRO_RAW, false);
  else if (pro == PRO_LN_C && L.hc) C3(PRO_LN_C, true);
  else if (pro == PRO_LN_C) C3(PRO_LN_C, false);
  else if (pro == PRO_LN_HC && L.hc) C3(PRO_LN_HC, true);
  else if (pro == PRO_LN_HC) C3(PRO_LN_HC, false);
  else return fail(DCTTS_ERR_STATE, "chain3: unsupported layer form");
#undef C3
  HIPCHK(hipGetLastError());
  if (prof) { ++c->prof_run_n; c->prof_rows += B; }
  return 0;
}



// Utterances per team and round (round 6).  Rounds 3-5: always four -- a batch of 8 then ran on two of the eight teams (XCDs) while six idled, and the frame cost what
// it costs at B = 32.  Now a batch of at most 8 / 16 is dealt one / two utterances to a team, so that all eight teams work (the side stream's cone GEMMs and row phases
// shrink with the rows a team owns; the chain's latency-bound layers do not).  The arithmetic of an utterance does not depend on the slot it sits in: results are bitwise
// those of the four-utterance form (DCTTS_XGROUP=2 forces it: tests).  Only the forms whose three team kernels all know about it (the merged chain forms).
static inline int team_u_for(const dctts_ctx* c, int B) {
  if (!c->dec_merge || c->xgroup == 2) return 4;
  return B > 16 ? 4 : (B > 8 ? 2 : 1);
}
static inline int team_rounds(int B, int U) { return ((B + U - 1) / U + 7) / 8; }

// ---- xgroup_kernel plumbing: one XGroupParams per (chain piece, network) in device memory; exchange buffers + team barriers + error word
struct XgMem { float* xch[2]; float* sch[2]; float* xch_m; float* sch_m; float* xch_h; float* sch_h; unsigned* bar; unsigned* bar_cone; unsigned* bar_mlp; int* err; int bpad; size_t bar_words; };
static inline int xg_hgroups(int B) { const int bpad = (B + 3) / 4 * 4; return B <= 16 ? bpad : bpad / 4; }      // utterance groups of xtail_kernel's highway exchange: one per utterance when a small batch is dealt singly (team_u_for)
static size_t xg_mem_floats(int B) { const int bpad = (B + 3) / 4 * 4; return (size_t)2 * (2 * bpad * 512 + 2 * bpad * 64) + (size_t)(2 * bpad * 512 + 2 * bpad * 64) + (size_t)2 * xg_hgroups(B) * XT_MAXM * (512 + 64) + (size_t)3 * ((bpad / 4 + 7) / 8 * 8) * 32 + 64; }
static XgMem xg_mem(dctts_ctx* c, int B) {
  XgMem m; m.bpad = (B + 3) / 4 * 4;
  float* q = c->xg_mem;
  for (int n = 0; n < 2; ++n) { m.xch[n] = q; q += (size_t)2 * m.bpad * 512; m.sch[n] = q; q += (size_t)2 * m.bpad * 64; }
  m.xch_m = q; q += (size_t)2 * m.bpad * 512; m.sch_m = q; q += (size_t)2 * m.bpad * 64;      // the k = 1 layers' exchange: xmlp_kernel [2][bpad][256] rows + [2][bpad][16][2] statistics; xtail_kernel the same with a tag beside every value: [2][bpad][256][2], [2][bpad][16][4]
  m.xch_h = q; q += (size_t)2 * xg_hgroups(B) * XT_MAXM * 512; m.sch_h = q; q += (size_t)2 * xg_hgroups(B) * XT_MAXM * 64;   // xtail_kernel's highway layers: [2][groups][20][512], [2][groups][20][16][4]
  m.bar_words = (size_t)((m.bpad / 4 + 7) / 8 * 8) * 32;
  m.bar = (unsigned*)q; q += m.bar_words;                     // the chain's teams (xgroup_kernel)
  m.bar_cone = (unsigned*)q; q += m.bar_words;                // the side stream's teams (xcone_kernel): the two run concurrently
  m.bar_mlp = (unsigned*)q; q += m.bar_words;                 // the chain's teams between the two xgroup runs (xmlp_kernel)
  m.err = (int*)q;
  return m;
}

// Chain piece j (j = -1 .. T-1) launches, in this order: the AudioDec run of frame j (j >= 0; since the merged form of round 4 the front of xtail_kernel's
// launch, v3_xtail_table, and not launched from this table), ..., the AudioEnc run of frame j + 1 (j + 1 < T).
// The team barriers count arrivals monotonically over the whole decode, so every launch is told the count it starts from.
static int v3_xgroup_table(dctts_ctx* c, const DecodeWs& w, int B, int T, bool insig, bool cwait) {
  const std::string g = geom("xg", B, T) + ":" + std::to_string((size_t)w.pe[0]) + ":" + std::to_string((size_t)w.ae[0].p) + ":" + std::to_string((size_t)w.pb3[1]) + ":" + std::to_string((size_t)w.ad[0].p) + ":" +
                        std::to_string((int)insig) + ":" + std::to_string((int)cwait) + ":" + std::to_string((size_t)c->sig_ptr) + ":" + std::to_string((size_t)c->wait_ctr) + ":" +
                        std::to_string((size_t)c->aepre_tab) + ":" + std::to_string((int)c->ae_pass) + ":" + std::to_string((int)c->ae_pass_split) + ":" + std::to_string(c->trace_frame) + ":" + std::to_string((int)c->tail_on) + ":" + std::to_string((int)c->dec_merge) + ":" + std::to_string((int)c->attn_fold) + ":" +
                        std::to_string((size_t)w.kv.p) + ":" + std::to_string((size_t)w.vw) + ":" + std::to_string((size_t)w.c1q.p) + ":" + std::to_string((size_t)w.pm_all);
  if (c->xg_tab && c->xg_geom == g) return 0;
  { dctts_ctx::TabSlot ts; if (tab_lookup(c, "xg", g, &ts)) { c->xg_tab = ts.tab; c->xg_mem = (float*)ts.mem; c->xg_T = ts.n0; c->xg_geom = g; return 0; } }
  c->xg_tab = nullptr; c->xg_mem = nullptr;
  HIPALLOC(hipMalloc((void**)&c->xg_mem, xg_mem_floats(B) * sizeof(float)));
  HIPCHK(dev_zero_now(c->xg_mem, xg_mem_floats(B) * sizeof(float)));
  const XgMem m = xg_mem(c, B);
  const std::vector<DevLayer>& AE = c->ae_c; const std::vector<DevLayer>& AD = c->ad_c;
  auto rowp = [](const View& v, long par, int j) { return v.p + par * v.set + (v.row0 + j) * (long)v.stride; };
  std::vector<XGroupParams> tab((size_t)2 * (T + 1));
  unsigned arrivals = 0;
  for (int piece = -1; piece < T; ++piece) {
    for (int net = 0; net < 2; ++net) {                         // 0 = AudioDec highway layers of frame `piece`, 1 = AudioEnc highway layers of frame `piece + 1`
      const int j = net ? piece + 1 : piece;
      XGroupParams p; memset(&p, 0, sizeof(p));
      if (j < 0 || j >= T) { tab[(size_t)2 * (piece + 1) + net] = p; continue; }
      const long par = j & 1;
      const std::vector<DevLayer>& Lr = net ? AE : AD;
      size_t i0 = 0; while (i0 < Lr.size() && !Lr[i0].hc) ++i0;
      size_t i1 = i0; while (i1 < Lr.size() && Lr[i1].hc) ++i1;
      if (net == 0 && c->tail_on) i1 = i0 + 3;                 // AudioDec HC_2 .. HC_4 only: HC_5 .. HC_7 run in xtail_kernel, the launch behind this one
      const int L = (int)(i1 - i0);
      if (i0 == 0 || L < 2 || L > 10 || Lr[i0 - 1].cout != 256 || Lr[i0 - 1].act != ACT_NONE) return fail(DCTTS_ERR_STATE, "xgroup: a run of 2..10 highway layers after a linear 256-channel layer");
      p.B = B; p.L = L; p.U = c->team_u;
      const std::vector<float*>& P = net ? w.pe : w.pd; const std::vector<float*>& S = net ? w.se : w.sd;
      const std::vector<View>& H = net ? w.ae : w.ad;
      p.P0 = P[i0 - 1]; p.p0_bs = 256; p.stats0 = S[i0 - 1]; p.pg1 = Lr[i0 - 1].g1; p.pb1 = Lr[i0 - 1].b1;
      for (int k = 0; k < L; ++k) {
        const size_t i = i0 + k; const DevLayer& Ly = Lr[i];
        if (Ly.cout != 256 || Ly.cin != 256 || !Ly.wp16 || !Ly.wp16c) return fail(DCTTS_ERR_STATE, "xgroup: 256-channel causal k=3 highway layers only");
        XGroupLayer& q = p.lay[k];
        q.wp = Ly.wp16; q.g1 = Ly.g1; q.b1 = Ly.b1; q.g2 = Ly.g2; q.b2 = Ly.b2; q.tap2 = Ly.tap2 ? 1 : 0;
        if (net) { q.presum = w.pse[i] + par * w.pse_set; q.presum_bs = 512; }
        else { q.presum = w.pb3[i] + par * w.pb3_set[i] + (long)(c->cone_len[i] - 1) * 512; q.presum_bs = c->cone_len[i] * 512; }
        const View& hin = H[i - 1];                             // this layer's input rows
        const bool keep = net ? true : (k + 1 == L);            // AudioEnc: every row is history; AudioDec: only the residual the next launch (C_8) needs
        if (keep) { q.xm = rowp(hin, net ? 0 : par, j); q.xm_bs = (int)(hin.bstride * hin.stride); }
        if (Ly.tap2) { q.xt = rowp(hin, 0, j) - hin.stride; q.xt_bs = (int)(hin.bstride * hin.stride); }
        else { q.xt = q.wp; q.xt_bs = 0; }
      }
      p.pout = P[i1 - 1]; p.stats_out = S[i1 - 1];
      p.xch = m.xch[net]; p.sch = m.sch[net]; p.xch_set = m.bpad * 512; p.sch_set = m.bpad * 64;
      p.bar = m.bar; p.bar_base = arrivals; p.err = m.err;
      if (net == 1 && c->attn_fold) {
        // the attention row of frame j and AudioDec C_1 of frame j behind the run's last layer (what attnq_kernel + chain3_kernel<RAW> compute, v3_chain_enc)
        const size_t la = AE.size() - 1;
        const int d = c->cfg.d;
        p.attn = 1; p.N = c->cfg.max_N; p.win = c->cfg.attention_win_size;
        p.pm = w.pm_all + (long)j * B; p.pm_next = w.pm_all + (long)(j + 1) * B;
        p.K = w.kv.p; p.k_stride = 2 * d; p.VW = w.vw; p.vw_stride = d; p.kv_bs = c->cfg.max_N;
        p.qhist = w.ae[la].p + (w.ae[la].row0 + j) * (long)w.ae[la].stride; p.q_bs = (int)(w.ae[la].bstride * w.ae[la].stride);
        p.c1_wp = c->ad_c1q.wp16; p.c1_bias = c->audiodec[0].bias;
        p.c1_raw = w.c1q.p + (w.c1q.row0 + j) * (long)w.c1q.stride; p.raw_bs = (int)(w.c1q.bstride * w.c1q.stride);
        p.c1_pout = w.pd[0]; p.c1_stats = w.sd[0];
      }
      if (!(net == 0 && c->dec_merge))                          // (merged form: the AudioDec run is part of xtail_kernel's launch, which has barrier words of its own)
        arrivals += (unsigned)team_rounds(B, c->team_u) * (unsigned)(L - 1 + p.attn) * 16u;      // (a team with more than one utterance group runs them in turn: xgroup_kernel.h)
      if (net == 0) {                                           // the first launch of chain piece j: publishes the chain's counter and waits for bulk piece j
        if (insig) { p.sig = c->sig_ptr; p.sig_val = (unsigned)(j + 1); }
        if (cwait) { p.wait2 = c->wait_ctr + 32; p.wait_val = (unsigned)(j + 1); }
        if (c->ae_pass && j + 1 < T) {
          // passengers: AudioEnc's presums of row j + 1 (inputs: rows <= j - 1; the AudioEnc run of frame j + 1 follows on this stream) and row j of
          // the C1Q . W2 cache (the table's last three descriptors), which side-stream piece j + 1 needs behind its first launch.
          // Round 4 (ae_pass_split): only those three descriptors ride here -- 24 workgroups find CUs between the side stream's row kernels at once, 104
          // found them only when this launch's teams had finished (~4 us on the chain); AudioEnc's presums ride in the PREVIOUS piece's AudioEnc launch.
          const int ipl = ((B + 31) / 32) * (c->cfg.d / 32);
          const int first = c->ae_pass_split ? c->aepre_layers - 3 : 0;
          p.ptab = (const SplitParams*)c->aepre_tab + (size_t)((j + 1) & 1) * aepre_stride(c) + first;
          p.p_ipl = ipl; p.p_blocks = (c->aepre_layers - first) * ipl; p.p_step = j + 1; p.p_count_from = c->aepre_layers - 3 - first;
          p.pdone = (unsigned*)m.err + 2; p.pdone_target = (unsigned)(j + 1) * (unsigned)(3 * ipl); p.psig = c->wait_ctr + 16; p.psig_val = (unsigned)(j + 1);
        }
      } else if (c->ae_pass && c->ae_pass_split && j + 1 < T) {
        // the AudioEnc run of frame j (launched in chain piece j - 1) carries AudioEnc's presums of row j + 1: their inputs are rows <= j - 1, final since chain
        // piece j - 2, and their consumer is the AudioEnc run of frame j + 1, a later launch on this stream.  They start when the side stream's xcone_kernel
        // lets go of its CUs (~15 us before this launch ends) and are done before it.
        const int ipl = ((B + 31) / 32) * (c->cfg.d / 32);
        p.ptab = (const SplitParams*)c->aepre_tab + (size_t)((j + 1) & 1) * aepre_stride(c);
        p.p_ipl = ipl; p.p_blocks = (c->aepre_layers - 3) * ipl; p.p_step = j + 1; p.p_count_from = c->aepre_layers - 3;      // (none of them is counted)
      }
      if (c->trace_frame >= 0 && piece == c->trace_frame) {     // DCTTS_TRACE: this piece's two launches record their phase boundaries (unset = -1, which is also the first piece's index)
        if (!c->trace_buf) { HIPCHK(hipMalloc((void**)&c->trace_buf, 64 * 64 * 32 * sizeof(long long))); HIPCHK(dev_zero_now(c->trace_buf, 64 * 64 * 32 * sizeof(long long))); }
        p.ts = c->trace_buf + 64 * 64 * 32 - 192 - 256 * (2 - net);
      }
      tab[(size_t)2 * (piece + 1) + net] = p;
    }
  }
  { const hipError_t e_ = hipMalloc(&c->xg_tab, tab.size() * sizeof(XGroupParams)); if (e_ != hipSuccess) { (void)hipFree(c->xg_mem); c->xg_mem = nullptr; c->xg_tab = nullptr; HIPALLOC(e_); } }
  HIPCHK(hipMemcpy(c->xg_tab, tab.data(), tab.size() * sizeof(XGroupParams), hipMemcpyHostToDevice));
  c->xg_geom = g; c->xg_T = T;
  tab_store(c, "xg", g, c->xg_tab, tab.size() * sizeof(XGroupParams) + xg_mem_floats(B) * sizeof(float), c->xg_mem, T);
  return 0;
}

static int v3_xgroup_launch(dctts_ctx* c, int B, int piece, int net, hipStream_t st) {
  const XGroupParams* p = (const XGroupParams*)c->xg_tab + (size_t)2 * (piece + 1) + net;
  const int teams = (B + 3) / 4;
  // measurement (dctts_hip_debug.h): HIP events on the launch stream around the launches of every 16th frame
  // (the AudioEnc runs only: ten layers and nothing else in the launch)
  const bool prof = c->prof_id == DCTTS_PROF_XGROUP && c->prof_frame && net == 1;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, st)); }
  const int ipl_ = ((B + 31) / 32) * (c->cfg.d / 32);
  int pass = 0;                                                                                                     // passenger workgroups, as in the table
  if (net == 0 && c->ae_pass && piece >= 0 && piece + 1 < c->xg_T) pass = (c->ae_pass_split ? 3 : c->aepre_layers) * ipl_;
  if (net == 1 && c->ae_pass && c->ae_pass_split && piece + 1 >= 0 && piece + 2 < c->xg_T) pass = (c->aepre_layers - 3) * ipl_;
  // always 128 team workgroups (8 teams of 16, one team per XCD): more could starve the other stream of CUs while they poll for it
  if (c->trace_frame >= 0 && piece == c->trace_frame) hipLaunchKernelGGL(xgroup_kernel<true>, dim3(128 + pass), dim3(512), pass ? hsplit_smem(32) : 0, st, p);      // DCTTS_TRACE: stamped
  else hipLaunchKernelGGL(xgroup_kernel<false>, dim3(128 + pass), dim3(512), pass ? hsplit_smem(32) : 0, st, p);
  HIPCHK(hipGetLastError());
  if (prof) { HIPCHK(hipEventRecord(e1, st)); c->prof_ev.emplace_back(e0, e1); c->prof_cnt.push_back(1); c->prof_rows += 10; }   // prof_rows counts LAYERS here
  return 0;
}

// ---- xcone_kernel plumbing: one XConeParams per frame in device memory (layers HC_3 .. HC_7 of AudioDec's cone, parity copies folded in)
static int v3_xcone_table(dctts_ctx* c, const DecodeWs& w, int B, int T, bool insig) {
  const bool fold = c->side_fold && c->cone_len[0] - 1 <= 96 && c->cone_len[1] <= 96 && c->audiodec.size() > 1 && c->audiodec[1].wpp && c->audiodec[1].tap_off[1] == -1;
  c->side_fold = fold;
  const std::string g = geom("xc", B, T) + ":" + std::to_string((size_t)w.pb3[2]) + ":" + std::to_string((size_t)w.ad[1].p) + ":" + std::to_string((size_t)c->xg_mem) + ":" + std::to_string((int)insig) + ":" + std::to_string((size_t)c->wait_ctr) + ":" + std::to_string((int)c->tail_on) + ":" + std::to_string(c->np_eff) + ":" + std::to_string(c->trace_frame) +
                        ":" + std::to_string((int)fold) + ":" + std::to_string((int)c->ae_pass) + ":" + std::to_string((size_t)w.kv.p) + ":" + std::to_string((size_t)w.vw) + ":" + std::to_string((size_t)w.pm_all) + ":" + std::to_string((size_t)w.scal.p);
  if (c->xc_tab && c->xc_geom == g) return 0;
  { dctts_ctx::TabSlot ts; if (tab_lookup(c, "xc", g, &ts)) { c->xc_tab = ts.tab; c->xc_geom = g; return 0; } }
  c->xc_tab = nullptr;
  const std::vector<DevLayer>& AD = c->audiodec;
  const XgMem m = xg_mem(c, B);
  size_t i0 = 2, i1 = i0; while (i1 < AD.size() && AD[i1].hc) ++i1;     // HC_3 .. the last highway layer
  const int tail = c->tail_on ? 1 : 0;
  if (tail) i1 = i0 + 2 + (size_t)(c->np_eff - 3);                       // HC_3, HC_4 (round 6: + HC_5) and the last one's cone rows: the chain's xtail_kernel takes it from there
  const int L = (int)(i1 - i0);
  if (L < 1 || L > 5) return fail(DCTTS_ERR_STATE, "xcone: 1..5 highway layers behind HC_2");
  std::vector<XConeParams> tab((size_t)T);
  for (int f = 0; f < T; ++f) {
    const long par = f & 1;
    XConeParams p; memset(&p, 0, sizeof(p));
    p.B = B; p.L = L; p.frame = f; p.tail_rows = tail; p.U = c->team_u;
    for (int k = 0; k < L; ++k) {
      const size_t i = i0 + k; const DevLayer& Ly = AD[i];
      if (Ly.cout != 256 || Ly.cin != 256 || Ly.cin_p != 256 || Ly.ntaps != 3 || Ly.tap_off[2] != 0 || !Ly.wp16 || c->cone_len[i] > 64) return fail(DCTTS_ERR_STATE, "xcone: causal k=3 highway layers over 256 channels, <= 64 cone rows");
      XConeLayer& q = p.lay[k];
      q.wp = Ly.wp16; q.bias = Ly.bias; q.g1 = Ly.g1; q.b1 = Ly.b1; q.g2 = Ly.g2; q.b2 = Ly.b2;
      const View& in = w.ad[i - 1]; const View& out = w.ad[i];
      q.xin = in.p + par * in.set; q.xin_bstride = in.bstride; q.xin_row0 = in.row0; q.xin_stride = in.stride;
      q.xout = out.p + par * out.set; q.xout_bstride = out.bstride; q.xout_row0 = out.row0; q.xout_stride = out.stride;
      q.pout = w.pb3[i] + par * w.pb3_set[i];
      q.offs = c->cone3_dev[i]; q.R = c->cone_len[i];
      for (int t3 = 0; t3 < 3; ++t3) q.tap_off[t3] = Ly.tap_off[t3];
    }
    p.bar = m.bar_cone; p.bar_base = (unsigned)f * (unsigned)team_rounds(B, c->team_u) * (unsigned)(2 * L - 1 + tail + (fold ? 2 : 0)) * 16u; p.err = m.err;      // (utterance groups of a team in turn)
    if (fold) { p.fold = 1; p.rc1 = fill_rowc1(c, w, B, c->cfg.max_N, f); p.rhc2 = fill_rowhc2(c, w, B, c->cfg.max_N, f); }
    if (insig) {                                                // the launch's last team publishes "side-stream piece f complete" itself
      p.done = (unsigned*)m.err + 1; p.done_target = (unsigned)(f + 1) * (unsigned)(((B + c->team_u - 1) / c->team_u < 8) ? (B + c->team_u - 1) / c->team_u : 8);      // one count per team (at most 8)
      p.sig = c->wait_ctr + 32; p.sig_val = (unsigned)(f + 1);
      if (f + 1 < T) { p.wait = c->wait_ctr; p.wait_val = (unsigned)(f + 1); p.wait_err = (int*)(c->wait_ctr + 64); }      // what side-stream piece f + 1 starts from
    }
    if (f == c->trace_frame) {                                  // DCTTS_TRACE: this frame's launch records its phase boundaries
      if (!c->trace_buf) { HIPCHK(hipMalloc((void**)&c->trace_buf, 64 * 64 * 32 * sizeof(long long))); HIPCHK(dev_zero_now(c->trace_buf, 64 * 64 * 32 * sizeof(long long))); }
      p.ts = c->trace_buf + 64 * 64 * 32 - 192;
    }
    tab[f] = p;
  }
  HIPALLOC(hipMalloc(&c->xc_tab, tab.size() * sizeof(XConeParams)));
  HIPCHK(hipMemcpy(c->xc_tab, tab.data(), tab.size() * sizeof(XConeParams), hipMemcpyHostToDevice));
  c->xc_geom = g;
  tab_store(c, "xc", g, c->xc_tab, tab.size() * sizeof(XConeParams));
  return 0;
}

// ---- mlp_rows_kernel plumbing: one MlpRowsParams per frame in device memory
static int v3_mlp_table(dctts_ctx* c, const DecodeWs& w, int B, int T) {
  const std::string g = geom("mlp", B, T) + ":" + std::to_string((size_t)w.pd[0]) + ":" + std::to_string((size_t)w.ypad.p) + ":" + std::to_string((size_t)w.ad[0].p) + ":" + std::to_string((size_t)w.pe[0]);
  if (c->mlp_tab && c->mlp_geom == g) return 0;
  { dctts_ctx::TabSlot ts; if (tab_lookup(c, "mlp", g, &ts)) { c->mlp_tab = ts.tab; c->mlp_geom = g; return 0; } }
  c->mlp_tab = nullptr;
  const std::vector<DevLayer>& AE = c->ae_c; const std::vector<DevLayer>& AD = c->ad_c;
  size_t lh = 0; for (size_t i = 0; i < AD.size(); ++i) if (AD[i].hc) lh = i;          // last highway layer of AudioDec (HC_7)
  size_t nh = 0; while (nh < AE.size() && !AE[nh].hc) ++nh;                             // AudioEnc k=1 head (C_1..C_3)
  const int ntail = (int)(AD.size() - lh - 1), nhead = (int)nh;
  if (lh < 1 || ntail < 1 || ntail + nhead > 7 || nhead < 1) return fail(DCTTS_ERR_STATE, "v3 mlp: unexpected layer structure");
  auto fill = [&](const DevLayer& L, MlpLayer* m) -> int {
    if (!L.wraw || L.hc || L.ntaps != 1 || (L.cin_real & 7) || (L.cout & 3) || L.cin_real > 256 || L.cout > 256) return fail(DCTTS_ERR_STATE, "v3 mlp: unsupported layer shape");
    m->w = L.wraw; m->bias = L.bias; m->g = L.g1; m->be = L.b1; m->cin = L.cin_real; m->cout = L.cout; m->relu = (L.act == ACT_RELU) ? 1 : 0;
    return 0;
  };
  std::vector<MlpRowsParams> tab((size_t)T);
  for (int j = 0; j < T; ++j) {
    MlpRowsParams p; memset(&p, 0, sizeof(p));
    p.B = B; p.frame = j; p.par = j & 1;
    p.nrm = make_norm(AD[lh], w.pd[lh], &w.ad[lh - 1]);
    int n = 0;
    for (size_t i = lh + 1; i < AD.size(); ++i) CHK(fill(AD[i], &p.L[n++]));
    p.mel_layer = n - 1;
    if (j + 1 < T) for (size_t i = 0; i < nh; ++i) CHK(fill(AE[i], &p.L[n++]));     // the last frame has no next frame to encode
    p.nlayers = n;
    p.ypad = w.ypad.p; p.y_bstride = w.ypad.bstride; p.y_row = w.ypad.row0 + 1 + j; p.y_stride = w.ypad.stride;
    p.logits = w.logits.p; p.l_bstride = w.logits.bstride; p.l_row = j; p.l_stride = w.logits.stride;
    p.pout = w.pe[nh - 1]; p.stats_out = w.se[nh - 1];
    tab[j] = p;
  }
  HIPALLOC(hipMalloc(&c->mlp_tab, tab.size() * sizeof(MlpRowsParams)));
  HIPCHK(hipMemcpy(c->mlp_tab, tab.data(), tab.size() * sizeof(MlpRowsParams), hipMemcpyHostToDevice));
  c->mlp_geom = g;
  tab_store(c, "mlp", g, c->mlp_tab, tab.size() * sizeof(MlpRowsParams));
  return 0;
}

// The k = 1 layers around the mel frame for frame j as XMlpParams (shared by xmlp_kernel and xtail_kernel); `prod` = index of the AudioDec highway layer whose
// pre-norm rows / statistics / input rows the launch starts from.
static int fill_xmlp(dctts_ctx* c, const DecodeWs& w, int B, int T, int j, size_t prod, const XgMem& m, int slots, XMlpParams* out) {
  const std::vector<DevLayer>& AE = c->ae_c; const std::vector<DevLayer>& AD = c->ad_c;
  size_t lh = 0; for (size_t i = 0; i < AD.size(); ++i) if (AD[i].hc) lh = i;          // last highway layer of AudioDec (HC_7)
  size_t nh = 0; while (nh < AE.size() && !AE[nh].hc) ++nh;                             // AudioEnc k=1 head (C_1..C_3)
  const int ntail = (int)(AD.size() - lh - 1), nhead = (int)nh;
  if (lh < 1 || ntail < 1 || ntail + nhead > 7 || nhead < 1 || AD[prod].cout != 256 || !AD[prod].hc || prod < 1) return fail(DCTTS_ERR_STATE, "xmlp: unexpected layer structure");
  auto fill = [&](const DevLayer& L, XMlpLayer* x, bool mel) -> int {
    if (!L.wp16 || L.hc || L.ntaps != 1 || (L.cin_p & 15) || (L.cout & 15) || L.cin_p > 256 || L.cout > 256) return fail(DCTTS_ERR_STATE, "xmlp: unsupported layer shape");
    x->wp = L.wp16; x->bias = L.bias; x->g = L.g1; x->be = L.b1; x->nkg = L.cin_p / 16; x->cout = L.cout; x->act = mel ? ACT_SIGMOID : L.act; x->pad_ = 0;
    return 0;
  };
  const int rounds = team_rounds(B, c->team_u);
  XMlpParams p; memset(&p, 0, sizeof(p));
  const long par = j & 1;
  p.B = B;
  p.P0 = w.pd[prod]; p.p0_bs = 2 * AD[prod].cout; p.stats0 = w.sd[prod];
  p.g1 = AD[prod].g1; p.b1 = AD[prod].b1; p.g2 = AD[prod].g2; p.b2 = AD[prod].b2;
  const View& rv = w.ad[prod - 1];
  p.res = rv.p + par * rv.set + (rv.row0 + j) * (long)rv.stride; p.res_bs = (int)(rv.bstride * rv.stride);
  int n = 0;
  for (size_t i = lh + 1; i < AD.size(); ++i) { CHK(fill(AD[i], &p.lay[n], i + 1 == AD.size())); ++n; }
  p.mel_layer = n - 1;
  if (j + 1 < T) for (size_t i = 0; i < nh; ++i) { CHK(fill(AE[i], &p.lay[n], false)); ++n; }     // the last frame has no next frame to encode
  p.nl = n;
  p.ymel = w.ypad.p + (w.ypad.row0 + 1 + j) * (long)w.ypad.stride; p.y_bs = (int)(w.ypad.bstride * w.ypad.stride);      // the +1 shift of train.py:51
  p.logits = w.logits.p + (w.logits.row0 + j) * (long)w.logits.stride; p.l_bs = (int)(w.logits.bstride * w.logits.stride);
  p.pout = w.pe[nh - 1]; p.stats_out = w.se[nh - 1];
  p.xch = m.xch_m; p.sch = m.sch_m; p.xch_set = m.bpad * 256; p.sch_set = m.bpad * 32;
  p.bar = m.bar_mlp; p.bar_base = (unsigned)j * (unsigned)rounds * (unsigned)slots * 16u; p.err = m.err;
  *out = p;
  return 0;
}

// xmlp_kernel's per-frame parameters: the same seven layers as v3_mlp_table, column-split over the teams (needs xg_mem: call after v3_xgroup_table)
static int v3_xmlp_table(dctts_ctx* c, const DecodeWs& w, int B, int T) {
  const std::string g = geom("xmlp", B, T) + ":" + std::to_string((size_t)w.pd[0]) + ":" + std::to_string((size_t)w.ypad.p) + ":" + std::to_string((size_t)w.ad[0].p) + ":" + std::to_string((size_t)w.pe[0]) + ":" + std::to_string((size_t)c->xg_mem);
  if (c->xmlp_tab && c->xmlp_geom == g) return 0;
  { dctts_ctx::TabSlot ts; if (tab_lookup(c, "xmlp", g, &ts)) { c->xmlp_tab = ts.tab; c->xmlp_geom = g; return 0; } }
  c->xmlp_tab = nullptr;
  const std::vector<DevLayer>& AD = c->ad_c;
  size_t lh = 0; for (size_t i = 0; i < AD.size(); ++i) if (AD[i].hc) lh = i;
  const XgMem m = xg_mem(c, B);
  std::vector<XMlpParams> tab((size_t)T);
  for (int j = 0; j < T; ++j) CHK(fill_xmlp(c, w, B, T, j, lh, m, 7, &tab[j]));
  HIPALLOC(hipMalloc(&c->xmlp_tab, tab.size() * sizeof(XMlpParams)));
  HIPCHK(hipMemcpy(c->xmlp_tab, tab.data(), tab.size() * sizeof(XMlpParams), hipMemcpyHostToDevice));
  c->xmlp_geom = g;
  tab_store(c, "xmlp", g, c->xmlp_tab, tab.size() * sizeof(XMlpParams));
  return 0;
}

// xtail_kernel's per-frame parameters: AudioDec's last three highway layers (HC_5 .. HC_7 over 5 / 3 / 1 rows per utterance) + the k = 1 layers
static int v3_xtail_table(dctts_ctx* c, const DecodeWs& w, int B, int T, bool insig, bool cwait) {
  const std::string g = geom("xtail", B, T) + ":" + std::to_string((size_t)w.pd[0]) + ":" + std::to_string((size_t)w.ypad.p) + ":" + std::to_string((size_t)w.ad[0].p) + ":" + std::to_string((size_t)w.pe[0]) + ":" + std::to_string((size_t)c->xg_mem) + ":" + std::to_string(c->trace_frame) + ":" +
                        std::to_string((int)c->dec_merge) + ":" + std::to_string(c->np_eff) + ":" + std::to_string((int)insig) + ":" + std::to_string((int)cwait) + ":" + std::to_string((size_t)c->sig_ptr) + ":" + std::to_string((size_t)c->wait_ctr) + ":" + std::to_string((size_t)c->aepre_tab) + ":" + std::to_string((int)c->ae_pass) + ":" + std::to_string((size_t)w.pb3[1]) + ":" + std::to_string((int)c->chain_one);
  if (c->xtail_tab && c->xtail_geom == g) return 0;
  { dctts_ctx::TabSlot ts; if (tab_lookup(c, "xtail", g, &ts)) { c->xtail_tab = ts.tab; c->xtail_geom = g; return 0; } }
  c->xtail_tab = nullptr;
  const std::vector<DevLayer>& AD = c->audiodec;                         // (the full layers: all three taps in wp16)
  const int NPv = c->dec_merge ? c->np_eff : 3, NHv = 6 - NPv;           // newest-row layers in front / cone layers behind them (xtail_kernel.h: NP)
  const size_t h0 = (size_t)(1 + NPv);                                   // NP = 3: C_1, HC_2, HC_3, HC_4 | HC_5, HC_6, HC_7 | C_8 ..;  NP = 4: ... HC_5 | HC_6, HC_7 | C_8 ..
  for (size_t k = 0; k < (size_t)NHv; ++k) {
    const DevLayer& Ly = AD[h0 + k];
    if (!Ly.hc || Ly.cout != 256 || Ly.cin != 256 || Ly.cin_p != 256 || Ly.ntaps != 3 || Ly.tap_off[2] != 0 || !Ly.wp16) return fail(DCTTS_ERR_STATE, "xtail: causal k=3 highway layers over 256 channels");
    if ((k > 0 || NPv == 4) && (Ly.tap_off[0] != -2 || Ly.tap_off[1] != -1)) return fail(DCTTS_ERR_STATE, "xtail: the layers behind the first one have dilation 1");
  }
  int nout[3] = {0, 0, 0};
  for (int k = 0; k < NHv; ++k) nout[k] = c->cone_len[h0 + k];
  if (NPv == 3 && (nout[0] != 5 || nout[1] != 3 || nout[2] != 1 || 3 * nout[0] != c->cone_len[h0 - 1] || 4 * nout[0] > XT_MAXM)) return fail(DCTTS_ERR_STATE, "xtail: cone 15 / 5 / 3 / 1");
  if (NPv == 4 && (nout[0] != 3 || nout[1] != 1 || c->cone_len[h0 - 1] != nout[0] + 2)) return fail(DCTTS_ERR_STATE, "xtail: cone 5 / 3 / 1");
  const int nin0 = c->cone_len[h0 - 1];                                  // input rows per utterance of the first cone layer (its producer's cone)
  const XgMem m = xg_mem(c, B);
  const int groups = xg_hgroups(B);
  const View& xv = w.ad[h0 - 1];                                         // HC_4's output rows: the side stream's xcone_kernel writes its cone rows (parity copy of the frame)
  std::vector<XTailParams> tab((size_t)T);
  for (int j = 0; j < T; ++j) {
    XTailParams p; memset(&p, 0, sizeof(p));
    CHK(fill_xmlp(c, w, B, T, j, h0 - 1, m, c->dec_merge ? (c->chain_one ? 14 : 13) : 10, &p.m));      // (chain_one: + the meeting in front of the AudioEnc run, xchain_kernel)
    p.m.xch_set = m.bpad * 512; p.m.sch_set = m.bpad * 64;      // (value, tag) pairs
    const long par = j & 1;
    if (c->dec_merge) {
      // merged form: AudioDec HC_2 .. HC_4 at the newest row run in front (what the AudioDec run of xgroup_kernel did, v3_xgroup_table), and this launch is the
      // first one of chain piece j: it publishes the chain's counter, waits for side-stream piece j, and carries the passengers
      const std::vector<DevLayer>& ADc = c->ad_c;                         // (centre tap in wp16)
      p.np = NPv;
      p.pP0 = w.pd[0]; p.pstats0 = w.sd[0]; p.pg1 = ADc[0].g1; p.pb1 = ADc[0].b1;
      for (int k = 0; k < NPv; ++k) {
        const size_t i = 1 + (size_t)k; const DevLayer& Ly = ADc[i];
        if (!Ly.hc || Ly.cout != 256 || Ly.cin != 256 || !Ly.wp16 || !Ly.wp16c || Ly.tap2) return fail(DCTTS_ERR_STATE, "xtail: the newest-row layers are 256-channel causal k=3 highway layers");
        XTailP& q = p.pl[k];
        q.wp = Ly.wp16; q.g1 = Ly.g1; q.b1 = Ly.b1; q.g2 = Ly.g2; q.b2 = Ly.b2;
        q.presum = w.pb3[i] + par * w.pb3_set[i] + (long)(c->cone_len[i] - 1) * 512; q.presum_bs = c->cone_len[i] * 512;
      }
      p.xchp = m.xch[0]; p.schp = m.sch[0]; p.xchp_set = m.bpad * 512; p.schp_set = m.bpad * 64;
      if (insig) { p.sig = c->sig_ptr; p.sig_val = (unsigned)(j + 1); }
      if (cwait) { p.wait2 = c->wait_ctr + 32; p.wait_val = (unsigned)(j + 1); }
      if (c->ae_pass && j + 1 < T) {                                      // passengers: AudioEnc's presums of row j + 1 and row j of the C1Q . W2 cache (counted: rowc1_kernel polls psig)
        const int ipl = ((B + 31) / 32) * (c->cfg.d / 32);
        // chain_one: the AudioEnc run that reads the presums is part of THIS launch, and passengers run on other XCDs: they compute row j + 2's (inputs: rows <= j,
        // complete since the previous piece) for the next launch; the C1Q . W2 descriptors carry step_val 0 then, so their row stays j
        const int ps = c->chain_one ? j + 2 : j + 1;
        p.ptab = (const SplitParams*)c->aepre_tab + (size_t)(ps & 1) * aepre_stride(c);
        p.p_ipl = ipl; p.p_blocks = c->aepre_layers * ipl; p.p_step = ps; p.p_count_from = c->aepre_layers - 3;
        p.pdone = (unsigned*)m.err + 2; p.pdone_target = (unsigned)(j + 1) * (unsigned)(3 * ipl); p.psig = c->wait_ctr + 16; p.psig_val = (unsigned)(j + 1);
      }
    }
    p.nh = NHv; p.nin0 = nin0; p.frame = j; p.U = c->team_u;
    for (int k = 0; k < NHv; ++k) {
      const DevLayer& Ly = AD[h0 + k];
      XTailHc& q = p.hc[k];
      q.wp = Ly.wp16; q.bias = Ly.bias; q.g1 = Ly.g1; q.b1 = Ly.b1; q.g2 = Ly.g2; q.b2 = Ly.b2;
      q.nout = nout[k]; q.nin = k == 0 ? nin0 : nout[k - 1]; q.ts = (k == 0 && NPv == 3) ? nout[0] : 1;      // (the input row of (output row r, tap) is r + (2 - tap) * ts)
    }
    p.xin = xv.p + par * xv.set + (xv.row0 + j) * (long)xv.stride; p.xin_bs = xv.bstride * (long)xv.stride; p.xin_stride = xv.stride;
    if (NPv == 3) {
      for (int kk = 0; kk < 3; ++kk)
        for (int r = 0; r < nout[0]; ++r) p.in_off[kk * nout[0] + r] = AD[h0].tap_off[2 - kk] - r;      // input row q = kk * 5 + r: time t - r + tap offset
    } else {
      for (int q = 0; q < nin0; ++q) p.in_off[q] = -q;                      // dilation 1: input row q is time t - q
    }
    p.xch = m.xch_h; p.sch = m.sch_h; p.xch_set = groups * XT_MAXM * 512; p.sch_set = groups * XT_MAXM * 64;
    if (j == c->trace_frame) {
      if (!c->trace_buf) { HIPCHK(hipMalloc((void**)&c->trace_buf, 64 * 64 * 32 * sizeof(long long))); HIPCHK(dev_zero_now(c->trace_buf, 64 * 64 * 32 * sizeof(long long))); }
      p.ts = c->trace_buf + 64 * 64 * 32 - 64;      // (the slot mlp_rows_kernel's stamps use in the row-split form)
    }
    tab[j] = p;
  }
  // every input row the first layer stages (but the newest) must be a cone row the side stream produces
  for (int q = 1; q < nin0; ++q) {
    bool found = false;
    std::vector<int> offs((size_t)c->cone_len[h0 - 1]);
    HIPCHK(hipMemcpy(offs.data(), c->cone3_dev[h0 - 1], offs.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int o : offs) if (o == tab[0].in_off[q]) found = true;
    if (!found) return fail(DCTTS_ERR_STATE, "xtail: an input row of the first cone layer is not in its producer's cone");
  }
  HIPALLOC(hipMalloc(&c->xtail_tab, tab.size() * sizeof(XTailParams)));
  HIPCHK(hipMemcpy(c->xtail_tab, tab.data(), tab.size() * sizeof(XTailParams), hipMemcpyHostToDevice));
  c->xtail_geom = g;
  tab_store(c, "xtail", g, c->xtail_tab, tab.size() * sizeof(XTailParams));
  return 0;
}

static int v3_mlp_launch(dctts_ctx* c, int B, int j, hipStream_t st) {
  CHK(prof_close_run(c, st));
  if (c->tail_on && c->xtail_tab) {
    const bool prof = c->prof_id == DCTTS_PROF_XTAIL && c->prof_frame;      // measurement (dctts_hip_debug.h): HIP events around the launch of every 16th frame
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, st)); }
    int pass = 0;                                                 // passenger workgroups, as in the table
    if (c->dec_merge && c->ae_pass && j + 1 < c->xg_T) pass = c->aepre_layers * (((B + 31) / 32) * (c->cfg.d / 32));
    if (c->chain_one && j + 1 < c->xg_T) {
      // round 5: the whole chain piece as ONE launch -- these layers, a team barrier, the AudioEnc run of frame j + 1 with the attention row and AudioDec C_1
      const XGroupParams* pg = (const XGroupParams*)c->xg_tab + (size_t)2 * (j + 1) + 1;
      const XTailParams* pt = (const XTailParams*)c->xtail_tab + j;
      if (c->np_eff == 4) {
        if (c->trace_on) hipLaunchKernelGGL((xchain_kernel<true, 4>), dim3(128 + pass), dim3(512), 0, st, pt, pg);
        else hipLaunchKernelGGL((xchain_kernel<false, 4>), dim3(128 + pass), dim3(512), 0, st, pt, pg);
      } else if (c->trace_on) hipLaunchKernelGGL((xchain_kernel<true, 3>), dim3(128 + pass), dim3(512), 0, st, pt, pg);
      else hipLaunchKernelGGL((xchain_kernel<false, 3>), dim3(128 + pass), dim3(512), 0, st, pt, pg);
    } else if (c->dec_merge && c->np_eff == 4) {
      if (c->trace_on) hipLaunchKernelGGL((xtail_kernel<true, 4>), dim3(128 + pass), dim3(512), 0, st, (const XTailParams*)c->xtail_tab + j);
      else hipLaunchKernelGGL((xtail_kernel<false, 4>), dim3(128 + pass), dim3(512), 0, st, (const XTailParams*)c->xtail_tab + j);
    } else if (c->trace_on) hipLaunchKernelGGL((xtail_kernel<true, 3>), dim3(128 + pass), dim3(512), 0, st, (const XTailParams*)c->xtail_tab + j);      // DCTTS_TRACE: stamped instantiation
    else hipLaunchKernelGGL((xtail_kernel<false, 3>), dim3(128 + pass), dim3(512), 0, st, (const XTailParams*)c->xtail_tab + j);
    HIPCHK(hipGetLastError());
    if (prof) { HIPCHK(hipEventRecord(e1, st)); c->prof_ev.emplace_back(e0, e1); c->prof_cnt.push_back(1); c->prof_rows += 10; }
    return 0;
  }
  if (c->xmlp_on && c->xmlp_tab) {
    hipLaunchKernelGGL(xmlp_kernel, dim3(128), dim3(512), 0, st, (const XMlpParams*)c->xmlp_tab + j);
    HIPCHK(hipGetLastError());
    return 0;
  }
  const MlpRowsParams* pm = (const MlpRowsParams*)c->mlp_tab + j;
  if (c->trace_on) {                                              // DCTTS_TRACE: stamped instantiation, stamps at the end of the trace buffer
    hipLaunchKernelGGL((mlp_rows_kernel<2, true>), dim3((B + 1) / 2), dim3(512), 0, st, pm, c->trace_buf + 64 * 64 * 32 - 64);
  } else {
    hipLaunchKernelGGL((mlp_rows_kernel<2, false>), dim3((B + 1) / 2), dim3(512), 0, st, pm, (long long*)nullptr);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// AudioDec HC_2 .. C_11 for frame j (its C_1 ran at the end of the previous chain piece)
static int v3_chain_dec(dctts_ctx* c, const DecodeWs& w, int B, int j, hipStream_t sm) {
  const std::vector<DevLayer>& AD = c->ad_c;
  const int par = j & 1;
  if (c->xg_on) {                                               // the newest-row layers as one launch (its table entry carries the piece's signal / wait) ...
    c->sig_next = 0; c->wait2_next = 0;
    if (c->dec_merge) return 0;                                 // ... or as the front of xtail_kernel's launch (v3_mlp_launch, called next)
    return v3_xgroup_launch(c, B, j, 0, sm);
  }
  for (size_t i = 1; i < AD.size(); ++i) {
    if (!AD[i].hc) break;                                       // C_8 .. C_11 run inside mlp_rows_kernel (launched by the caller)
    SplitExtra ex;
    if (AD[i].wp16c) { ex.presum = w.pb3[i] + (long)par * w.pb3_set[i] + (long)(c->cone_len[i] - 1) * 2 * AD[i].cout; ex.presum_rstride = c->cone_len[i] * 2 * AD[i].cout; }
    CHK(run_chain3(c, AD[i], B, j, &AD[i - 1], w.pd[i - 1], w.sd[i - 1], (AD[i - 1].hc && i >= 2) ? &w.ad[i - 2] : nullptr, &w.ad[i - 1], nullptr,
                   w.pd[i], w.sd[i], &ex, sm));
  }
  return 0;
}

// AudioEnc for frame j (C_1's prologue finalises mel frame j-1 when j > 0), attention row j, AudioDec C_1 of frame j.
static int v3_chain_enc(dctts_ctx* c, const DecodeWs& w, int B, int N, int j, hipStream_t sm) {
  const int d = c->cfg.d;
  const std::vector<DevLayer>& AE = c->ae_c;
  const std::vector<DevLayer>& AD = c->ad_c;
  for (size_t i = 0; i < AE.size(); ++i) {
    if (j > 0 && !AE[i].hc) continue;                           // C_1 .. C_3 of frame j ran inside frame j-1's mlp_rows_kernel
    if (i == 0) {                                               // frame 0 only: S[0] is the zero row
      CHK(run_split(c, 16, AE[0], B, 1, nullptr, j, PRO_RAW, nullptr, nullptr, w.ypad, w.pe[0], sm, nullptr, w.se[0]));
    } else if (c->xg_on && AE[i].hc) {
      if (!AE[i - 1].hc && !(c->chain_one && j >= 1)) CHK(v3_xgroup_launch(c, B, j - 1, 1, sm));    // the whole run of highway layers (HC_4 .. HC_13) of frame j, launched from chain piece j - 1
                                                                                                     // (chain_one: it ran as the back of that piece's one launch, v3_mlp_launch)
    } else {
      SplitExtra ex;
      if (AE[i].wp16c) { ex.presum = w.pse[i] + (long)(j & 1) * w.pse_set; ex.presum_rstride = 2 * AE[i].cout; }
      CHK(run_chain3(c, AE[i], B, j, &AE[i - 1], w.pe[i - 1], w.se[i - 1], (AE[i - 1].hc && i >= 2) ? &w.ae[i - 2] : nullptr, &w.ae[i - 1], nullptr,
                     w.pe[i], w.se[i], &ex, sm));
    }
  }
  if (c->attn_fold) return 0;                                   // the AudioEnc run's launch carried the attention row and AudioDec C_1 (xgroup_kernel.h)
  const size_t la = AE.size() - 1;
  AttnQParams a; memset(&a, 0, sizeof(a));
  a.B = B; a.frame = j;
  a.nrm = make_norm(AE[la], w.pe[la], &w.ae[la - 1]);
  a.qhist = w.ae[la].p; a.q_bstride = w.ae[la].bstride; a.q_row0 = w.ae[la].row0; a.q_stride = d;
  a.K = w.kv.p; a.k_stride = 2 * d; a.VW = w.vw; a.vw_stride = d; a.kv_bstride = N;
  a.bias = c->audiodec[0].bias; a.N = N; a.d = d; a.win = c->cfg.attention_win_size; a.pm_all = w.pm_all; a.presum = w.ps0;
  CHK(prof_close_run(c, sm));
  hipLaunchKernelGGL(attnq_kernel, dim3((B + 3) / 4), dim3(256), 0, sm, a);
  HIPCHK(hipGetLastError());
  SplitExtra ex; ex.presum = w.ps0; ex.presum_rstride = d; ex.raw = &w.c1q;
  return run_chain3(c, c->ad_c1q, B, j, nullptr, nullptr, nullptr, nullptr, nullptr, &w.ae[la], w.pd[0], w.sd[0], &ex, sm);
}

static int write_trace3(dctts_ctx* c, int j) {
  std::vector<long long> h(64 * 64 * 32);
  HIPCHK(hipMemcpy(h.data(), c->trace_buf, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  if (c->trace_file.empty()) return 0;
  FILE* f = fopen(c->trace_file.c_str(), "w");
  if (!f) return 0;
  long long t0 = 0;
  for (int k = 0; k < c->trace_n; ++k) for (int wg = 0; wg < 64; ++wg) { const long long e = h[(k * 64 + wg) * 32]; if (e && (!t0 || e < t0)) t0 = e; }
  fprintf(f, "# chain piece %d, in-kernel stamps (100 MHz wall clock), microseconds\n", j);
  if (c->trace_n > 0) fprintf(f, "# chain3_kernel<LN_HC, HC> launches (DCTTS_XGROUP=0): since the first entry of the piece\n# idx wgs | first_entry last_entry | median over workgroups, wave 0: entry->loads_issued ->landed ->mfma_done ->barrier ->end | last_end | median over workgroups of (last wave - first wave): entry, landed, mfma_done\n");
  for (int k = 0; k < c->trace_n; ++k) {
    std::vector<double> ph[8]; long long e0 = 0, e1 = 0, x1 = 0; int nw = 0;
    for (int wg = 0; wg < 64; ++wg) {
      const long long* o = &h[(k * 64 + wg) * 32];
      if (!o[0] || !o[5]) continue;
      if (!nw || o[0] < e0) e0 = o[0]; if (!nw || o[0] > e1) e1 = o[0]; if (!nw || o[5] > x1) x1 = o[5];
      for (int q = 0; q < 5; ++q) ph[q].push_back((o[q + 1] - o[q]) / 100.0);
      for (int q = 0; q < 3; ++q) {
        long long mn = o[8 + 8 * q], mx = mn;
        for (int w = 1; w < 8; ++w) { mn = std::min(mn, o[8 + 8 * q + w]); mx = std::max(mx, o[8 + 8 * q + w]); }
        ph[5 + q].push_back((mx - mn) / 100.0);
      }
      ++nw;
    }
    if (!nw) continue;
    fprintf(f, "%2d %3d | %8.2f %8.2f |", k, nw, (e0 - t0) / 100.0, (e1 - t0) / 100.0);
    for (int q = 0; q < 5; ++q) { std::sort(ph[q].begin(), ph[q].end()); fprintf(f, " %6.2f", ph[q][ph[q].size() / 2]); }
    fprintf(f, " | %8.2f |", (x1 - t0) / 100.0);
    for (int q = 5; q < 8; ++q) { std::sort(ph[q].begin(), ph[q].end()); fprintf(f, " %6.2f", ph[q][ph[q].size() / 2]); }
    fprintf(f, "\n");
  }
  for (int net = 0; net < 2; ++net) {
    const long long* o = &h[64 * 64 * 32 - 192 - 256 * (2 - net)];
    if (!o[0]) continue;
    fprintf(f, "# xgroup_kernel, %s run (workgroup 0, thread 0), microseconds since its entry: first row built | per layer: contraction + partial sums written, slice reduced, published, barrier passed, exchanged rows landed; behind the last layer, the attention + C_1 tail: output row rebuilt, operands landed + contraction written, finished\n ", net ? "AudioEnc" : "AudioDec");
    fprintf(f, " %6.2f |", (o[1] - o[0]) / 100.0);
    for (int i = 2; i < 120 && o[i]; ++i) fprintf(f, " %6.2f%s", (o[i] - o[0]) / 100.0, ((i - 2) % 5 == 4) ? " |" : "");
    fprintf(f, "\n");
    const long long* ot = &h[64 * 64 * 32 - 64];
    if (net && c->chain_one && ot[0]) fprintf(f, "  (one launch per chain piece: this run was entered %.2f us after xtail_kernel's part of the launch below was)\n", (o[0] - ot[0]) / 100.0);
  }
  {
    const long long* o = &h[64 * 64 * 32 - 192];
    if (o[0]) {
      fprintf(f, "# xcone_kernel (workgroup 0, thread 0), microseconds since its entry%s; per layer: row tables | contraction done | barrier passed | row pass done | barrier passed\n ",
              c->side_fold ? "; folded form: C_1's cone rows done | barrier passed | HC_2's cone rows done | barrier passed, then" : "");
      int last = 0;
      for (int i = 1; i < 60 && o[i]; ++i) { fprintf(f, " %6.2f", (o[i] - o[0]) / 100.0); last = i; }
      if (last > 0 && o[101] > o[100]) fprintf(f, "   (shader clock over the launch: %.0f MHz)", (double)(o[101] - o[100]) / ((o[last] - o[0]) / 100.0));
      fprintf(f, "\n");
      if (o[60]) {
        fprintf(f, "  first GEMM layer, per wave: first requests out, then the end of each (row tile, K half) unit:");
        for (int wv = 0; wv < 8; ++wv) { fprintf(f, "%s w%d", wv ? " |" : "", wv); for (int k = 0; k < 5 && o[60 + wv * 5 + k]; ++k) fprintf(f, " %.2f", (o[60 + wv * 5 + k] - o[0]) / 100.0); }
        fprintf(f, "\n");
      }
      const long long* ot = &h[64 * 64 * 32 - 64];          // the chain launch of the same frame (the stamps are one wall clock): when did this side piece run relative to it?
      if (ot[0] && c->tail_on && last > 0) fprintf(f, "  (relative to the entry of the chain's launch of this frame: entered %.2f, last stamp %.2f us)\n", (o[0] - ot[0]) / 100.0, (o[last] - ot[0]) / 100.0);
    }
  }
  {
    const long long* o = &h[64 * 64 * 32 - 64];
    if (o[0] && c->tail_on) {
      int i0 = 2;
      if (c->dec_merge) {
        const int npv = c->np_eff;
        fprintf(f, "# xtail_kernel, merged form (workgroup 0, thread 0), microseconds since its entry: first row built (incl. the wait for the side stream) | end of each newest-row layer (HC_2 .. HC_%d) | cone rows in LDS | per cone layer (HC_%d .. HC_7): MFMAs issued + partial sums written, reduced + published, barrier passed, exchanged rows landed, rows rebuilt | then the end of every k = 1 layer\n ", npv + 1, npv + 2);
        fprintf(f, " %6.2f |", (o[1] - o[0]) / 100.0);
        for (int i = 2; i < 2 + npv && o[i]; ++i) fprintf(f, " %6.2f", (o[i] - o[0]) / 100.0);
        fprintf(f, " | %6.2f |", (o[2 + npv] - o[0]) / 100.0);
        i0 = 3 + npv;
      } else {
        fprintf(f, "# xtail_kernel (workgroup 0, thread 0), microseconds since its entry: rows staged + newest row rebuilt | per highway layer: MFMAs issued + partial sums written, reduced + published, barrier passed, exchanged rows landed, rows rebuilt | then the end of every k = 1 layer\n ");
        fprintf(f, " %6.2f |", (o[1] - o[0]) / 100.0);
      }
      const int ncone = c->dec_merge ? 5 * (6 - c->np_eff) : 15;
      for (int i = i0; i < 56 && o[i]; ++i) fprintf(f, " %6.2f%s", (o[i] - o[0]) / 100.0, (i < i0 + ncone && (i - i0) % 5 == 4) ? " |" : "");
      if (o[56] && o[58]) fprintf(f, "\n  passengers: the first one ran %.2f .. %.2f, the last one %.2f .. %.2f", (o[56] - o[0]) / 100.0, (o[57] - o[0]) / 100.0, (o[58] - o[0]) / 100.0, (o[59] - o[0]) / 100.0);
      fprintf(f, "\n");
    } else if (o[0]) {
      fprintf(f, "# mlp_rows_kernel (workgroup 0, thread 0), microseconds since its entry: rows rebuilt | per layer: loads landed, FMAs done, partial sums exchanged, row finished\n");
      fprintf(f, "  %6.2f |", (o[1] - o[0]) / 100.0);
      for (int i = 2; i + 3 < 32 && o[i + 3]; i += 4) fprintf(f, "  %6.2f %6.2f %6.2f %6.2f |", (o[i] - o[0]) / 100.0, (o[i + 1] - o[0]) / 100.0, (o[i + 2] - o[0]) / 100.0, (o[i + 3] - o[0]) / 100.0);
      fprintf(f, "\n");
    }
  }
  fclose(f);
  return 0;
}

static int decode_v3(dctts_ctx* c, const DecodeWs& w, int B, int N, int T, hipStream_t st) {
  CHK(decode_streams_init(c));
  // How the two streams meet (DESIGN.md sections 2b / 2c).  Default (chain_wait_inkernel): two counters in device memory that kernels poll and
  // write themselves -- the chain's is written by the first launch of the NEXT chain piece and polled by xcone_kernel's tail, the side stream's
  // is written by xcone_kernel's last team and polled by the chain piece's first launch; no stream operation per frame.  DCTTS_CHAIN_WAIT=0:
  // the same counters as stream memory operations (hipStreamWriteValue32 / hipStreamWaitValue32) around the pieces.
  // Fallback (no stream memory operations on the device, or DCTTS_SYNC_VALUES=0, which rocprofv3 --pmc needs): events.
  if (c->sync_values && !c->ctr_chain) {
    int can = 0; (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device);
    if (!can) c->sync_values = 0;
  }
  const bool vs = c->sync_values != 0;
  if (vs && !c->ctr_chain) {
    // Stream memory operations: a write packet after a piece, a compare-and-wait packet before the piece that needs it.  The
    // command processor polls the counter itself: no signal objects, no interrupt, and (measured) ~10 us less per frame on the
    // chain's stream than hipEventRecord + hipStreamWaitEvent.
    HIPCHK(hipExtMallocWithFlags((void**)&c->ctr_chain, 8, hipMallocSignalMemory));
    HIPCHK(hipExtMallocWithFlags((void**)&c->ctr_bulk, 8, hipMallocSignalMemory));
  }
  const bool insig = vs;                                    // the chain's counter is written by the first launch of the next piece
  const bool cwait = vs && c->chain_wait_inkernel;          // the chain's wait for the bulk's counter happens inside that launch (chain3_kernel: wait2)
  if (cwait && !c->wait_ctr) {
    HIPCHK(hipMalloc((void**)&c->wait_ctr, 128 * sizeof(unsigned)));
    HIPCHK(dev_zero_now(c->wait_ctr, 128 * sizeof(unsigned)));
  }
  CHK(v3_mlp_table(c, w, B, T));
  c->xg_on = c->xgroup != 0 && c->xgroup_ok && c->prof_id != DCTTS_PROF_CHAIN_HC;      // (that timing id looks at chain3_kernel launches; so does DCTTS_TRACE with DCTTS_XGROUP=0)
  c->xc_on = c->xcone != 0 && c->xgroup_ok && c->prof_id != DCTTS_PROF_BULK_GEMM &&
             c->audiodec.size() > 1 && c->audiodec[1].wpp && c->audiodec[1].tap_off[1] == -1;      // (behind rowhc2_kernel)
  c->tail_on = c->xg_on && c->xc_on && c->chain_tail >= 2 && c->audiodec.size() == 11 && c->cone_len.size() > 6 && c->cone_len[4] == 5 && c->cone_len[5] == 3 && c->cone_len[6] == 1;
  c->xmlp_on = c->xg_on && c->chain_tail >= 1 && !c->tail_on;
  c->dec_merge = c->tail_on && (c->chain_tail == 2 || c->chain_tail == 6);          // (5: xtail_kernel behind an AudioDec run of xgroup_kernel, the first round-4 form -- A/B)
  c->np_eff = (c->dec_merge && c->tail_np == 4) ? 4 : 3;
  c->team_u = team_u_for(c, B);
  c->attn_fold = c->xg_on && c->chain_tail >= 1 && c->chain_tail != 3 && c->cfg.d == 256 && c->ad_c1q.wp16 != nullptr;      // (3: xtail_kernel without the fold -- A/B)
  c->chain_one = c->dec_merge && c->chain_tail == 2 && c->attn_fold;      // (6: the chain piece as two launches, round 4's form -- A/B)
  // with in-kernel waits both stream meetings of a frame leave the command processor: the side stream's first launch polls the chain's
  // counter, and xcone_kernel's last team writes the side stream's
  const bool bsig = cwait && c->xc_on;
  // ... and the one small GEMM in front of the cone work (the newest row of the C1Q . W2 cache) rides in the chain's AudioEnc presum launch one piece
  // earlier: its input is the chain's own newest row, and the counter the side stream waits for is written by the launch behind it
  // (round 4, measured and switched off: the presum GEMMs as launches of their own on the SIDE stream -- the newest C1Q . W2 row in front of the row kernels,
  //  AudioEnc's presums between rowhc2_kernel and xcone_kernel -- instead of as passenger workgroups of the chain's AudioDec launch, where they only find CUs
  //  once that launch's teams have finished (~4 us): the two hbulk_group_kernel launches take 11 + 10.7 us on the side stream, which then bounds the frame at
  //  99.5 us against 89.3)
  c->side_pre = false;
  c->side_fold = c->xcone == 1 && c->xc_on && bsig;      // (needs the team kernels and their in-kernel stream meetings; v3_xcone_table checks the cone geometry)
  c->c1qw_chain = vs && c->xc_on && !c->side_pre;
  // ... or, with both team kernels, in no launch of its own at all: the AudioEnc presums and that row are passengers of the chain's AudioDec launch
  // (xgroup_kernel.h); the side stream's first launch (rowc1_kernel) polls the row's own counter before it ends
  c->ae_pass = bsig && c->xg_on && c->cone_len[0] > 1 && !c->side_pre;
  c->chain_one = c->chain_one && c->ae_pass;
  c->ae_pass_split = c->ae_pass && c->chain_tail == 4;      // (4: A/B -- measured: the AudioDec launch gets 3.4 us shorter, the AudioEnc launch 6 us longer: its passengers only find CUs when xcone_kernel's work ends, ~4 us before the launch's own teams do: 88.9 against 86.2 us per frame)
  CHK(v3_aepre_table(c, w, B, c->c1qw_chain));
  c->sig_ptr = cwait ? c->wait_ctr : (unsigned*)c->ctr_chain;
  if (c->xg_on || c->xc_on) {
    CHK(v3_xgroup_table(c, w, B, T, insig, cwait));            // (also allocates the memory both kernels meet through)
    if (c->xc_on) CHK(v3_xcone_table(c, w, B, T, bsig));
    const XgMem m = xg_mem(c, B);
    HIPCHK(hipMemsetAsync(m.bar, 0, (3 * m.bar_words + 64) * sizeof(unsigned), st));      // the three sets of team barriers and the error word
    if (c->xmlp_on) CHK(v3_xmlp_table(c, w, B, T));
    if (c->tail_on) {
      CHK(v3_xtail_table(c, w, B, T, insig, cwait));
      HIPCHK(hipMemsetAsync(m.xch_m, 0, (size_t)(2 * m.bpad * 512 + 2 * m.bpad * 64) * sizeof(float), st));      // the tagged exchange: a tag of the previous decode must not look current
    }
  }
  hipStream_t sb = c->s_bulk;
  // use_graph: 0 = every launch eager; 1 = the bulk piece of each frame is one hipGraph launch (the chain launches stay eager: a graph
  // launch per chain piece costs ~10 us of start-up on the critical path)
  const bool gr = c->use_graph != 0;
  auto chain_piece = [&](int j, hipStream_t s) -> int {      // j = -1: AudioEnc / attention / AudioDec C_1 of frame 0 only
    c->sig_next = (insig && j >= 0) ? (unsigned)(j + 1) : 0u;          // written by the piece's first launch (AudioDec HC_2)
    c->wait2_next = (cwait && j >= 0) ? (unsigned)(j + 1) : 0u;        // ... which also waits for bulk piece j
    // AudioEnc's presums of row j+1 (inputs: rows <= j-1, final since piece j-2): when the side stream is the longer one they run here, while this
    // piece would otherwise wait for it, instead of in front of the cone work
    if (j >= 0 && c->xc_on && !c->ae_pass && !c->side_pre && j + 1 < T) CHK(v3_aepre(c, B, j + 1, s, c->c1qw_chain ? 0 : 1));
    if (j >= 0) { CHK(v3_chain_dec(c, w, B, j, s)); CHK(v3_mlp_launch(c, B, j, s)); }   // AudioDec HC_2 .. HC_7; C_8 .. C_11, mel frame j, AudioEnc C_1 .. C_3 of frame j+1
    if (j + 1 < T) return v3_chain_enc(c, w, B, N, j + 1, s);
    return 0;
  };
  if (gr) {
    const std::string g = geom("graph3", B, T, N) + ":" + std::to_string(c->bulk_cap) + ":" + std::to_string((size_t)c->mlp_tab) + ":" + std::to_string((int)cwait) + ":" + std::to_string((int)vs) + ":" + std::to_string((int)c->xg_on) + ":" + std::to_string((int)c->xc_on) + ":" + std::to_string((int)c->ae_pass) + ":" + std::to_string((int)c->side_pre) + ":" + std::to_string((size_t)c->xc_tab) + ":" + std::to_string((size_t)c->wait_ctr) + ":" +
                          std::to_string((size_t)w.kv.p) + ":" + std::to_string((size_t)w.vw);
    if (c->bulk3_g.empty() || c->graphs3_geom != g) {
      if (!c->bulk3_g.empty()) HIPCHK(hipDeviceSynchronize());      // (an earlier decode's graphs may still be running on another stream)
      destroy_graphs(c);
      hipStream_t cs;
      HIPCHK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
      const int prof_keep = c->prof_id; c->prof_id = -1;
      c->bulk3_g.assign(T, nullptr);
      int rc = 0;
      for (int f = 0; f < T && rc == 0; ++f) rc = capture_piece(cs, &c->bulk3_g[f], [&]() { return v3_bulk_rest(c, w, B, N, T, f, cs, (cwait && f > 0) ? (unsigned)f : 0u); });
      c->prof_id = prof_keep;
      HIPCHK(hipStreamDestroy(cs));
      if (rc != 0) { destroy_graphs(c); return rc; }
      c->graphs3_geom = g;
    }
  }
  if (cwait) HIPCHK(hipMemsetAsync(c->wait_ctr, 0, 65 * sizeof(unsigned), st));      // the counters and THIS decode's wait error word ([64]; the sticky copy lives in dstat); st is ordered after the previous decode's last piece (and its decode_finish), and that piece after all bulk work
  if (vs) {
    HIPCHK(hipStreamWriteValue32(st, c->ctr_chain, 0u, 0));
    HIPCHK(hipStreamWriteValue32(st, c->ctr_bulk, 0u, 0));
  }
  CHK(v3_vw(c, w, B, N, st));                                              // V . W_top, once per batch
  CHK(v3_aepre(c, B, 0, st));                                              // row 0's AudioEnc presums (= the biases: every tap reads padding)
  if (c->chain_one && T > 1) CHK(v3_aepre(c, B, 1, st, 1));                // ... and row 1's (the same): piece j's passengers compute row j + 2's in this form
  HIPCHK(hipEventRecord(c->ev_fork, st));
  HIPCHK(hipStreamWaitEvent(sb, c->ev_fork, 0));
  const int tstep = c->trace_frame;
  auto bulk_piece = [&](int f) -> int {
    if (gr) HIPCHK(hipGraphLaunch(c->bulk3_g[f], sb)); else CHK(v3_bulk_rest(c, w, B, N, T, f, sb, (cwait && f > 0) ? (unsigned)f : 0u));
    if (bsig) return 0;
    if (vs) HIPCHK(hipStreamWriteValue32(sb, cwait ? (void*)(c->wait_ctr + 32) : (void*)c->ctr_bulk, (uint32_t)(f + 1), 0)); else HIPCHK(hipEventRecord(c->ev_bulk[f & 3], sb));
    return 0;
  };
  // DCTTS_PIECETIME=<frame>: timing events around 8 consecutive chain / bulk pieces starting there (measurement only)
  const int pt0 = c->piecetime;
  hipEvent_t pe_c[9][2], pe_b[9][2];
  if (pt0 >= 0) for (int i = 0; i < 9; ++i) for (int k = 0; k < 2; ++k) { HIPCHK(hipEventCreate(&pe_c[i][k])); HIPCHK(hipEventCreate(&pe_b[i][k])); }
  auto ptime = [&](int j) { return pt0 >= 0 && j >= pt0 && j < pt0 + 8; };
  CHK(bulk_piece(0));
  CHK(chain_piece(-1, st));
  if (!vs) HIPCHK(hipEventRecord(c->ev_chain[3], st));
  for (int j = 0; j < T; ++j) {
    if (j + 1 < T) {
      // bulk piece j+1 needs attnq(j) / C1Q[j]: end of chain piece j-1
      if (cwait) {} else if (vs) HIPCHK(hipStreamWaitValue32(sb, c->ctr_chain, (uint32_t)(j + 1), hipStreamWaitValueGte, 0xFFFFFFFFu)); else HIPCHK(hipStreamWaitEvent(sb, c->ev_chain[(j - 1) & 3], 0));
      if (ptime(j + 1)) HIPCHK(hipEventRecord(pe_b[j + 1 - pt0][0], sb));
      CHK(bulk_piece(j + 1));
      if (ptime(j + 1)) HIPCHK(hipEventRecord(pe_b[j + 1 - pt0][1], sb));
    }
    if (!cwait) { if (vs) HIPCHK(hipStreamWaitValue32(st, c->ctr_bulk, (uint32_t)(j + 1), hipStreamWaitValueGte, 0xFFFFFFFFu)); else HIPCHK(hipStreamWaitEvent(st, c->ev_bulk[j & 3], 0)); }
    if (ptime(j)) HIPCHK(hipEventRecord(pe_c[j - pt0][0], st));
    if (j == tstep) {
      if (!c->trace_buf) { HIPCHK(hipMalloc((void**)&c->trace_buf, 64 * 64 * 32 * sizeof(long long))); }
      HIPCHK(hipMemsetAsync(c->trace_buf, 0, (64 * 64 * 32 - 192) * sizeof(long long), st));      // (the last 192 words hold xcone's / mlp_rows' stamps)
      HIPCHK(hipMemsetAsync(c->trace_buf + 64 * 64 * 32 - 64, 0, 64 * sizeof(long long), st));
      c->trace_on = true; c->trace_n = 0;
    }
    c->prof_frame = (j & 15) == 8;
    CHK(chain_piece(j, st));
    CHK(prof_close_run(c, st));
    c->prof_frame = false;
    if (ptime(j)) HIPCHK(hipEventRecord(pe_c[j - pt0][1], st));
    if (!vs) HIPCHK(hipEventRecord(c->ev_chain[j & 3], st));
    if (c->trace_on) {
      c->trace_on = false;
      HIPCHK(hipStreamSynchronize(st));
      CHK(write_trace3(c, j));
    }
  }
  // the chain's last piece is on `st`; the bulk stream's last piece was consumed by it, so `st` is ordered after all decode work.
  c->fin_xerr = (c->xg_on || c->xc_on) ? xg_mem(c, B).err : nullptr;      // this decode's error words, folded into the sticky status by decode_finish
  c->fin_werr = cwait ? (const int*)(c->wait_ctr + 64) : nullptr;
  if (pt0 >= 0 && pt0 + 8 < T) {
    HIPCHK(hipStreamSynchronize(st)); HIPCHK(hipStreamSynchronize(sb));
    for (int i = 0; i < 8; ++i) {
      float dc = 0, db = 0, c2c = 0, b2c = 0, c2b = 0;
      (void)hipEventElapsedTime(&dc, pe_c[i][0], pe_c[i][1]);
      if (i > 0) { (void)hipEventElapsedTime(&db, pe_b[i][0], pe_b[i][1]); (void)hipEventElapsedTime(&c2c, pe_c[i - 1][1], pe_c[i][0]);
                   (void)hipEventElapsedTime(&b2c, pe_b[i][1], pe_c[i][0]); (void)hipEventElapsedTime(&c2b, pe_c[i - 1][0], pe_b[i][0]); }
      fprintf(stderr, "[dctts] frame %d: chain piece %.1f us, bulk piece %.1f us, prev chain end -> chain start %.1f us, bulk end -> chain start %.1f us, prev chain start -> bulk start %.1f us\n",
              pt0 + i, dc * 1e3, db * 1e3, c2c * 1e3, b2c * 1e3, c2b * 1e3);
    }
    for (int i = 0; i < 9; ++i) for (int k = 0; k < 2; ++k) { (void)hipEventDestroy(pe_c[i][k]); (void)hipEventDestroy(pe_b[i][k]); }
  }
  return 0;
}

// Decodes of DIFFERENT contexts on one device must not overlap: each runs two polling team kernels sized to half the CUs, and two pairs of them starve each
// other until the bounded waits give up (a resource deadlock).  Inside a process they are serialised here: a decode waits for the
// completion event of the last decode another context enqueued on the device.  (Another PROCESS on the same GPU cannot be seen from here: there the bounded
// waits, the poisoned outputs and dctts_decode_status are the protection, and dc_tts_amd.Engine repeats the decode with one launch per layer.)
struct DeviceLease { std::mutex mu; hipEvent_t done = nullptr; const dctts_ctx* owner = nullptr; };
static std::mutex g_lease_mu;                    // guards the map only; a device's lease has its own mutex, held across the whole ENQUEUE of a decode (decode_impl)
static std::map<int, DeviceLease> g_lease;
static DeviceLease& lease_of(int device) { std::lock_guard<std::mutex> lk(g_lease_mu); return g_lease[device]; }
static void lease_forget(const dctts_ctx* c) {   // dctts_destroy: the next context allocated at this address must not be taken for the lease's owner
  DeviceLease& ls = lease_of(c->device);
  std::lock_guard<std::mutex> lk(ls.mu);
  if (ls.owner == c) ls.owner = nullptr;
}

static int decode_status_init(dctts_ctx* c) {
  if (c->dstat) return 0;
  HIPCHK(hipMalloc((void**)&c->dstat, 4 * sizeof(int)));
  HIPCHK(dev_zero_now(c->dstat, 4 * sizeof(int)));
  HIPCHK(hipHostMalloc((void**)&c->dstat_host, 4 * sizeof(int), 0));
  for (int i = 0; i < 4; ++i) c->dstat_host[i] = 0;
  return 0;
}

static int decode_impl(dctts_ctx* c, const int32_t* L, int B, int N, int T, float* Y, int64_t* maxatt, float* alignments, hipStream_t st) {
  if (N != c->cfg.max_N) return fail(DCTTS_ERR_ARG, "decode: N must equal hp.max_N (mask built from it, networks.py:142)");
  CHK(decode_status_init(c));
  if (c->dstat_host[0]) {
    // an earlier decode failed on the device and nobody asked (dctts_decode_status reports and clears): refuse ONCE, so that the failure cannot go
    // unnoticed (its outputs were poisoned as well), then start clean
    const int ew = c->dstat_host[0];
    c->dstat_host[0] = 0; c->dstat_host[1] = 0;
    HIPCHK(hipMemsetAsync(c->dstat, 0, 2 * sizeof(int), st));
    if (ew & (2 | 8)) c->xgroup_ok = false;
    return fail(DCTTS_ERR_STATE, "decode: an EARLIER decode on this context failed on the device (error word " + std::to_string(ew) + ") and dctts_decode_status was not consulted; its outputs were poisoned (NaN / -1)");
  }
  // Two host threads decoding on two contexts of one device must not both see the old lease and enqueue overlapping decodes: the device's lease is held from here to the
  // event record behind this decode's last launch (enqueue only: ~10 ms of host time, nothing waits for the GPU under it).
  DeviceLease& ls = lease_of(c->device);
  std::lock_guard<std::mutex> lease_lk(ls.mu);
  if (ls.done && ls.owner != c) HIPCHK(hipStreamWaitEvent(st, ls.done, 0));
  // ... and decodes / TextEnc calls of THIS context from other streams (use groups; K / V of this decode live in TextEnc's output buffer until it ends)
  // (both groups' completion events -- and the device lease's -- are recorded on every way out of this function once something may have been enqueued)
  GroupGuard gte(c, dctts_ctx::GRP_TE, st), gdec(c, dctts_ctx::GRP_DEC, st);
  CHK(gte.acquire());
  CHK(gdec.acquire());
  struct LeaseDone {
    DeviceLease& ls; dctts_ctx* c; hipStream_t st;
    ~LeaseDone() {
      if (!ls.done && hipEventCreateWithFlags(&ls.done, hipEventDisableTiming) != hipSuccess) { ls.done = nullptr; return; }
      if (hipEventRecord(ls.done, st) == hipSuccess) ls.owner = c;
    }
  } lease_done{ls, c, st};
  c->fin_xerr = nullptr; c->fin_werr = nullptr;
  // dctts_decode_safe_once: THIS decode runs one launch per layer and meets the side stream through stream operations (no team kernel, no bounded in-kernel wait:
  // nothing in it can time out when the GPU is shared) -- the persistent settings (dctts_set_team_kernels, DCTTS_XGROUP / DCTTS_XCONE / DCTTS_CHAIN_WAIT,
  // a switch-off by dctts_decode_status) are left exactly as they were
  struct SafeOnce {
    dctts_ctx* c; bool on; int xg, xc, cw;
    explicit SafeOnce(dctts_ctx* c_) : c(c_), on(c_->safe_once && c_->safe_once_tid == std::this_thread::get_id()), xg(c_->xgroup), xc(c_->xcone), cw(c_->chain_wait_inkernel) { if (on) { c->safe_once = false; c->xgroup = 0; c->xcone = 0; c->chain_wait_inkernel = 0; } }
    ~SafeOnce() { if (on) { c->xgroup = xg; c->xcone = xc; c->chain_wait_inkernel = cw; } }
  } safe_once(c);
  DecodeWs w;
  CHK(decode_ws(c, B, N, T, &w, st));
  const bool v3 = (c->decode_mode == 3);
  if (!v3) { w.rbuf.set = 0; for (auto& v : w.ad) v.set = 0; }            // the simple form uses one copy of every buffer
  CHK(textenc_into(c, L, B, N, &w.kv, st));
  HIPCHK(hipMemsetAsync(w.step, 0, 256, st));                              // the simple form's device-side frame counter
  HIPCHK(hipMemsetAsync(w.pm_all, 0, (size_t)B * sizeof(int), st));      // prev_max_attentions = zeros (synthesize.py:46)
  if (!c->init_pm.empty()) {                                               // test hook (dctts_hip_debug.h): a seeded start state for this one decode
    if ((int)c->init_pm.size() != B) { c->init_pm.clear(); return fail(DCTTS_ERR_ARG, "decode: the seeded prev_max_attentions must have B entries"); }
    HIPCHK(hipMemcpyAsync(w.pm_all, c->init_pm.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    c->init_pm.clear();
  }
  if (v3) {
    CHK(decode_streams_init(c));
    // The chain's launches need the highest stream priority as well (decode_streams_init).  A caller's stream that has it is used as it is; otherwise the decode proper
    // runs on the context's own high-priority chain stream between two events on the caller's stream: everything that stream has done so far in front of the decode,
    // everything it does later behind it (the two hand-overs cost ~0.1 ms per decode: INTEGRATION.md recommends a high-priority stream to callers who care).
    int lo = 0, hi = 0, pr = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    bool own = st && hipStreamGetPriority(st, &pr) == hipSuccess && pr == hi && hi != lo;
    if (!own) (void)hipGetLastError();
    if (own && !c->sync_values) { c->own_tested = st; c->own_ok = true; }
    if (own && st != c->own_tested) {        // ... and only if it does not share a hardware queue with the side stream (decode_streams_init; tested once per stream handle)
      bool ok = false;
      CHK(streams_run_concurrently(c, st, c->s_bulk, &ok));
      c->own_tested = st; c->own_ok = ok;
    }
    if (own && !c->own_ok) own = false;
    if (own) {
      CHK(decode_v3(c, w, B, N, T, st));
    } else {
      HIPCHK(hipEventRecord(c->ev_in, st));
      HIPCHK(hipStreamWaitEvent(c->s_chain, c->ev_in, 0));
      const int rc = decode_v3(c, w, B, N, T, c->s_chain);
      HIPCHK(hipEventRecord(c->ev_out, c->s_chain));
      HIPCHK(hipStreamWaitEvent(st, c->ev_out, 0));
      CHK(rc);
    }
  } else if (c->use_graph) {
    const std::string g = geom("graph1", B, T, N) + ":" + std::to_string((size_t)w.kv.p);   // the captured launches bake in the TextEnc output pointer
    if (!c->graph_exec || c->graph_geom != g) {
      if (c->graph_exec) { HIPCHK(hipDeviceSynchronize()); (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
      if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
      hipStream_t cs;
      HIPCHK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
      HIPCHK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
      const int prof_keep = c->prof_id; c->prof_id = -1;
      int rc = decode_step_launch(c, w, B, N, cs);
      c->prof_id = prof_keep;
      hipError_t e = hipStreamEndCapture(cs, &c->graph);
      if (rc != 0) { (void)hipStreamDestroy(cs); return rc; }
      HIPCHK(e);
      HIPCHK(hipGraphInstantiate(&c->graph_exec, c->graph, nullptr, nullptr, 0));
      HIPCHK(hipStreamDestroy(cs));
      c->graph_geom = g;
    }
    for (int j = 0; j < T; ++j) HIPCHK(hipGraphLaunch(c->graph_exec, st));
  } else {
    for (int j = 0; j < T; ++j) CHK(decode_step_launch(c, w, B, N, st));
  }
  const int nm = c->cfg.n_mels;
  HIPCHK(hipMemcpy2DAsync(Y, (size_t)T * nm * sizeof(float), w.ypad.p + (w.ypad.row0 + 1) * nm,
                          (size_t)w.ypad.bstride * nm * sizeof(float), (size_t)T * nm * sizeof(float), B,
                          hipMemcpyDeviceToDevice, st));
  if (maxatt) {
    hipLaunchKernelGGL(traj_to_i64_kernel, dim3((B * T + 255) / 256), dim3(256), 0, st, w.pm_all, (long long*)maxatt, B, T);
    HIPCHK(hipGetLastError());
  }
  if (alignments) {
    // `g.alignments` as the LAST sess.run of the loop fetches it (synthesize.py:48, networks.py:153): every time row t against the window of step
    // T - 1, (B, N, T).  Q[t] for all t is AudioEnc's last history buffer; the window of step T - 1 is row T - 1 of the trajectory table.
    const View& q = w.ae.back();
    const int d = c->cfg.d;
    AttnFullParams p;
    p.Q = q.p + q.row0 * (long)q.stride; p.q_stride = q.stride; p.q_bstride = q.bstride;
    p.K = w.kv.p; p.k_stride = w.kv.stride; p.k_bstride = w.kv.bstride; p.V = w.kv.p + d; p.v_stride = w.kv.stride; p.v_bstride = w.kv.bstride;
    p.T = T; p.N = N; p.d = d; p.monotonic = 1; p.prev_max = w.pm_all + (long)(T - 1) * B; p.win = c->cfg.attention_win_size;
    p.R = nullptr; p.align = alignments; p.maxatt = nullptr;
    hipLaunchKernelGGL(attention_full_kernel, dim3(T, B), dim3(256), (d + N) * sizeof(float), st, p);
    HIPCHK(hipGetLastError());
  }
  // fold this decode's error words into the sticky status, poison the outputs if it failed, publish the status to the host
  const int inject = c->inject_err; c->inject_err = 0;
  hipLaunchKernelGGL(decode_finish_kernel, dim3(128), dim3(256), 0, st, c->fin_xerr, c->fin_werr, inject, c->dstat, Y, (long)B * T * nm,
                     (long long*)maxatt, (long)B * T, alignments, (long)B * N * T);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->dstat_host, c->dstat, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  CHK(gdec.release());
  return gte.release();
}

extern "C" int dctts_text2mel_decode(dctts_ctx* c, const int32_t* L, int B, int N, int T, float* Y, int64_t* maxatt, float* alignments, void* stream) {
  DevGuard dev_guard(c);
  CHK(check_ready(c, dev_guard));
  if (!L || !Y || B <= 0 || T <= 0) return fail(DCTTS_ERR_ARG, "decode: bad argument");
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  return oom_retry(c, [&]() -> int { CHK(ws_trim(c)); return decode_impl(c, L, B, N, T, Y, maxatt, alignments, (hipStream_t)stream); });
}

extern "C" int dctts_synthesize(dctts_ctx* c, const int32_t* L, int B, int N, int T, float* Y, float* Z, int64_t* maxatt, float* alignments, void* stream) {
  DevGuard dev_guard(c);
  CHK(check_ready(c, dev_guard));
  if (!L || !Y || !Z || B <= 0 || T <= 0) return fail(DCTTS_ERR_ARG, "synthesize: bad argument");
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  hipStream_t st = (hipStream_t)stream;
  return oom_retry(c, [&]() -> int {
    CHK(ws_trim(c));
    CHK(decode_impl(c, L, B, N, T, Y, maxatt, alignments, st));
    CHK(ssrn_impl(c, Y, B, T, nullptr, Z, st));                             // synthesize.py:57
    // SSRN's ReLU layers turn a poisoned (NaN) mel back into finite numbers: Z of a failed decode is poisoned explicitly.  The kernel reads the context's status
    // block, which the next decode (possibly from another stream) writes: it runs under the decode's use group, whose completion event is recorded behind it.
    GroupGuard gdec(c, dctts_ctx::GRP_DEC, st);
    CHK(gdec.acquire());
    hipLaunchKernelGGL(poison_if_failed_kernel, dim3(256), dim3(256), 0, st, c->dstat, Z, (long)B * 4 * T * c->cfg.n_linear);
    HIPCHK(hipGetLastError());
    return gdec.release();
  });
}

extern "C" int dctts_decode_status(dctts_ctx* c) {
  if (!c) return fail(DCTTS_ERR_ARG, "null ctx");
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  if (c->dstat_host && c->dstat_host[0]) {                  // reported once, then cleared on both sides so that the next decode starts clean
    DevGuard dev_guard(c);
    const int ew = c->dstat_host[0], nfail = c->dstat_host[1];
    c->dstat_host[0] = 0; c->dstat_host[1] = 0;
    (void)dev_zero_now(c->dstat, 2 * sizeof(int));
    // only a team that is not on one XCD (bits 2 / 8) says something permanent about this device; time-outs (bits 1 / 4 / 16 / 32: CUs or the side stream held
    // up by somebody else's work) are transient -- the team kernels stay on unless three reports in a row fail
    if (ew & (2 | 8)) c->xgroup_ok = false;
    else if (++c->team_fail_streak >= 3) c->xgroup_ok = false;
    return fail(DCTTS_ERR_STATE, "decode: " + std::to_string(nfail) + " decode(s) failed on the device since the last report (error word " + std::to_string(ew) +
                ": 1 / 2 = xgroup_kernel barrier time-out / a team not on one XCD, 4 / 8 = the same in xcone_kernel, 16 = the side stream never arrived, 32 = an in-kernel stream wait timed out, "
                "64 = injected by the debug hook): their outputs were overwritten with NaN / -1" + (c->xgroup_ok ? "" : "; further decodes run one launch per layer"));
  }
  c->team_fail_streak = 0;
  return 0;
}

extern "C" int dctts_set_team_kernels(dctts_ctx* c, int enable) {
  if (!c || enable < 0 || enable > 1) return fail(DCTTS_ERR_ARG, "team kernels: 0 or 1");
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  c->xgroup = enable; c->xcone = enable;
  if (enable) { c->xgroup_ok = true; c->team_fail_streak = 0; }
  return 0;
}

extern "C" int dctts_debug_team_kernels_state(dctts_ctx* c) {
  if (!c) return fail(DCTTS_ERR_ARG, "null ctx");
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  return (c->xgroup ? 1 : 0) | (c->xcone ? 2 : 0) | (c->xgroup_ok ? 4 : 0);
}

extern "C" int dctts_decode_safe_once(dctts_ctx* c) {
  if (!c) return fail(DCTTS_ERR_ARG, "null ctx");
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  c->safe_once = true; c->safe_once_tid = std::this_thread::get_id();
  return 0;
}

extern "C" int dctts_debug_inject_decode_error(dctts_ctx* c, int bits) {
  if (!c) return fail(DCTTS_ERR_ARG, "null ctx");
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  c->inject_err = bits ? (bits | 64) : 0;
  return 0;
}

extern "C" int dctts_set_decode_graph(dctts_ctx* c, int enable) {
  if (!c || enable < 0 || enable > 1) return fail(DCTTS_ERR_ARG, "decode graph mode must be 0 or 1");
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  c->use_graph = enable;
  return 0;
}

extern "C" int dctts_set_decode_mode(dctts_ctx* c, int mode) {
  if (!c || (mode != 0 && mode != 3)) return fail(DCTTS_ERR_ARG, "decode mode must be 3 (two-stream incremental form, the default) or 0 (simple one-stream form, cross-check)");
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  c->decode_mode = mode;
  return 0;
}

