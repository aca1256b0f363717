// dctts_api.hip -- host side of libdctts_hip.so: context, weight packing, network drivers, decode loop.
// Implements include/dctts_hip.h.  Reference call sites: networks.py:14-292, train.py:48-77,
// synthesize.py:45-57.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/dctts_hip.h"
#include "../../include/dctts_hip_debug.h"
#include "api_common.h"
#include "attn_kernels.h"
#include "decode_kernels.h"
#include "decode_fallback_kernels.h"
#include "decode3_kernels.h"
#include "hconv_kernel.h"
#include "hconv16_kernel.h"
#include "xgroup_kernel.h"
#include "xcone_kernel.h"
#include "xmlp_kernel.h"
#include "xtail_kernel.h"

using namespace dctts;

namespace dctts {

// (weight prefetch depth BD, scalar tile bases SB) per shape: what measured best on this MI355X (hconv_kernel.h; profiles/r03_hconv_lab.txt)
#define HCONV_CASE(E, NT_, NW_, BD_, SB_)                                                        \
  if (s.epi == E && s.nt == NT_ && s.nw == NW_) {                                                \
    hipLaunchKernelGGL((hconv_kernel<E, NT_, NW_, BD_, SB_>), grid, dim3(NW_ * 64), 0, stream, p); \
    return hipGetLastError();                                                                    \
  }

hipError_t launch_hconv(const ConvShape& s, const ConvParams& p, hipStream_t stream, int tiles) {
  const dim3 grid(tiles >= 0 ? tiles : (p.M + 31) / 32);
  if (p.M <= 0 || grid.x == 0) return hipSuccess;
  if (p.wx) {                                // a k = 1 layer of 32 x 32 + 1 columns: 8 waves x 4 tiles + the last column on the vector ALU (hconv_kernel.h: XC)
    if (s.epi != EPI_C || p.cout != 1025 || p.ntaps != 1 || p.cin_p > 1120) return hipErrorInvalidValue;
    // ring depth 1 and registers capped at 128 (two workgroups per CU): 10.45 ms per SSRN pass at B = 32 against 10.50 (depth 2, one workgroup per CU) and 10.71 (11 waves x 3 tiles)
    // (round 6: + the k-group's requests spread behind single MFMAs, SG = 1: 459 -> 447 us for 768 items, tools/micro/hconv_lab)
    // (with the registers capped at 128 -- XC = 2, round 5's form -- the spread costs 60 bytes of scratch; uncapped at ring depth 2 it is 451 us without any)
    hipLaunchKernelGGL((hconv_kernel<EPI_C, 4, 8, 2, 1, 0, 1, 0, 1>), grid, dim3(512), 0, stream, p);
    return hipGetLastError();
  }
  HCONV_CASE(EPI_HC, 2, 8, 2, 0)
  HCONV_CASE(EPI_HC, 4, 8, 2, 0)
  if (s.epi == EPI_HC && s.nt == 8 && s.nw == 8) {      // SSRN HC_11 / HC_12: SG = 1 (round 6): 2398 -> 2249 us for 768 items = 0.820 -> 0.874 of the fp32 MFMA peak (hconv_lab, same run)
    hipLaunchKernelGGL((hconv_kernel<EPI_HC, 8, 8, 1, 1, 0, 0, 0, 1>), grid, dim3(512), 0, stream, p);
    return hipGetLastError();
  }
  HCONV_CASE(EPI_C, 1, 4, 1, 0)
  HCONV_CASE(EPI_C, 1, 8, 1, 0)
  HCONV_CASE(EPI_C, 2, 8, 2, 0)
  HCONV_CASE(EPI_C, 4, 8, 1, 1)
  HCONV_CASE(EPI_C, 3, 11, 2, 1)
  return hipErrorInvalidConfiguration;
}

// The opt-in split-bf16 form (hconv_kernel.h: BF): whole 32-row items, no row tail.  parts = 2: a 64-tile highway layer as two column halves + the finishing pass.
hipError_t launch_hconv_bf16(const ConvShape& s, const ConvParams& p, hipStream_t stream) {
  if (p.M <= 0) return hipSuccess;
  const int items = (p.M + 31) / 32;
  if (p.wx) {
    if (s.epi != EPI_C || p.cout != 1025 || p.ntaps != 1 || p.cin_p > 1120) return hipErrorInvalidValue;
    hipLaunchKernelGGL((hconv_kernel<EPI_C, 4, 8, 2, 1, 0, 1, 1>), dim3(items), dim3(512), 0, stream, p);
    return hipGetLastError();
  }
  if (s.epi == EPI_HC && s.nt == 8 && s.nw == 8) {
    if (!p.raw_out || p.m_base != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL((hconv_kernel<EPI_HC, 4, 8, 2, 1, 2, 0, 1>), dim3(items, 2), dim3(512), 0, stream, p);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) return e_;
    hipLaunchKernelGGL(hc_tail_finish_kernel, dim3((p.M + 3) / 4), dim3(256), 0, stream, p, (const float*)p.raw_out, 1);
    return hipGetLastError();
  }
  if (s.epi == EPI_HC && s.nt == 4 && s.nw == 8) { hipLaunchKernelGGL((hconv_kernel<EPI_HC, 4, 8, 1, 1, 0, 0, 1>), dim3(items), dim3(512), 0, stream, p); return hipGetLastError(); }      // (ring depth 1: 274 against 298 us for HC_8's 768 items, tools/micro/hconv_lab)
  if (s.epi == EPI_C && s.nt == 4 && s.nw == 8) { hipLaunchKernelGGL((hconv_kernel<EPI_C, 4, 8, 2, 1, 0, 0, 1>), dim3(items), dim3(512), 0, stream, p); return hipGetLastError(); }
  if (s.epi == EPI_C && s.nt == 2 && s.nw == 8) { hipLaunchKernelGGL((hconv_kernel<EPI_C, 2, 8, 2, 1, 0, 0, 1>), dim3(items), dim3(512), 0, stream, p); return hipGetLastError(); }
  return hipErrorInvalidConfiguration;
}

#define HCONV_TAIL_CASE(NT_, NW_, BD_, SB_)                                                                       \
  if (s.epi == EPI_HC && s.nt == NT_ && s.nw == NW_) {                                                            \
    hipLaunchKernelGGL((hconv_kernel<EPI_HC, NT_, NW_, BD_, SB_, 1>), grid, dim3(NW_ * 64), 0, stream, p);         \
    hipError_t e_ = hipGetLastError();                                                                            \
    if (e_ != hipSuccess) return e_;                                                                              \
    hipLaunchKernelGGL(hc_tail_finish_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, p, (const float*)p.raw_out, 3); \
    return hipGetLastError();                                                                                     \
  }

#define HCONV_CTAIL_CASE(NT_, NW_, BD_, SB_)                                                                      \
  if (s.epi == EPI_C && s.nt == NT_ && s.nw == NW_) {                                                             \
    hipLaunchKernelGGL((hconv_kernel<EPI_C, NT_, NW_, BD_, SB_, 1>), grid, dim3(NW_ * 64), 0, stream, p);          \
    hipError_t e_ = hipGetLastError();                                                                            \
    if (e_ != hipSuccess) return e_;                                                                              \
    hipLaunchKernelGGL(c_tail_finish_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, p, (const float*)p.raw_out, 3); \
    return hipGetLastError();                                                                                     \
  }

hipError_t launch_hconv_tail(const ConvShape& s, const ConvParams& p, hipStream_t stream) {
  const int rows = p.M - p.m_base;
  if (rows <= 0) return hipSuccess;
  if (!p.raw_out) return hipErrorInvalidValue;
  const dim3 grid((rows + 31) / 32, 3);
  if (s.epi == EPI_HC) {
    if (p.ntaps != 3) return hipErrorInvalidValue;
    HCONV_TAIL_CASE(2, 8, 2, 0)
    HCONV_TAIL_CASE(4, 8, 2, 0)
    if (s.nt == 8 && s.nw == 8) {          // HC_11 / HC_12's row tail: the same spread of the k-group's requests as their main launch (SG = 1)
      hipLaunchKernelGGL((hconv_kernel<EPI_HC, 8, 8, 1, 1, 1, 0, 0, 1>), grid, dim3(512), 0, stream, p);
      hipError_t e_ = hipGetLastError();
      if (e_ != hipSuccess) return e_;
      hipLaunchKernelGGL(hc_tail_finish_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, p, (const float*)p.raw_out, 3);
      return hipGetLastError();
    }
  } else {
    if (p.ntaps != 1 || p.cout > 1280) return hipErrorInvalidValue;
    HCONV_CTAIL_CASE(4, 8, 1, 1)
    HCONV_CTAIL_CASE(3, 11, 2, 1)
  }
  return hipErrorInvalidConfiguration;
}

// a highway layer of 512 channels as quarter-column items (hconv_kernel.h: RAW = 2): rows p.m_base .. p.M - 1 (ConvParams::raw_out / raw_ld set), then the finishing pass
hipError_t launch_hconv_cols(const ConvShape& s, const ConvParams& p, hipStream_t stream) {
  const int rows = p.M - p.m_base;
  if (rows <= 0) return hipSuccess;
  if (!p.raw_out || s.epi != EPI_HC || s.nt != 4 || s.nw != 8 || p.cout != 512) return hipErrorInvalidValue;
  const dim3 grid((rows + 31) / 32, 4);
  hipLaunchKernelGGL((hconv_kernel<EPI_HC, 2, 4, 4, 1, 2>), grid, dim3(256), 0, stream, p);      // (weight requests four k-groups ahead, wave-uniform tile bases: 2.04 -> 2.01 ms for TextEnc; round 6: with the requests spread behind single MFMAs, SG = 1, 1.96 -> 2.01 ms: not here)
  hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) return e_;
  hipLaunchKernelGGL(hc_tail_finish_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, p, (const float*)p.raw_out, 1);
  return hipGetLastError();
}

#define HCONV16_CASE(E, NT_, NW_, BD_, SB_)                                                                   \
  if (s.epi == E && s.nt == NT_ && s.nw == NW_) {                                                             \
    hipLaunchKernelGGL((hconv16_kernel<E, NT_, NW_, BD_, SB_>), grid, dim3(NW_ * 64), 0, stream, p, m_start); \
    return hipGetLastError();                                                                                 \
  }

hipError_t launch_hconv16(const ConvShape& s, const ConvParams& p, int m_start, hipStream_t stream) {
  if (p.M <= m_start) return hipSuccess;
  const dim3 grid((p.M - m_start + 15) / 16);
  HCONV16_CASE(EPI_HC, 4, 8, 2, 0)
  HCONV16_CASE(EPI_HC, 8, 8, 2, 0)
  HCONV16_CASE(EPI_HC, 16, 8, 1, 1)
  HCONV16_CASE(EPI_C, 8, 8, 2, 0)
  HCONV16_CASE(EPI_C, 6, 11, 2, 1)
  return hipErrorInvalidConfiguration;
}

}  // namespace dctts

// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
int dctts_set_error(int code, const std::string& msg) { return fail(code, msg); }
#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e__ = (x);                                                                           \
    if (e__ != hipSuccess)                                                                          \
      return fail(DCTTS_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e__) + " @" + std::to_string(__LINE__)); \
  } while (0)
#define CHK(x) do { int r__ = (x); if (r__ != 0) return r__; } while (0)
// Device allocations of the workspace / table caches: an out-of-memory failure is remembered (per thread), so that the entry point can drop the caches
// (ws_trim_now: one device synchronisation) and run the call once more instead of failing a serving loop whose shapes vary (oom_retry).
static thread_local bool g_oom = false;
#define HIPALLOC(x)                                                                                 \
  do {                                                                                              \
    hipError_t e__ = (x);                                                                           \
    if (e__ == hipErrorOutOfMemory) { (void)hipGetLastError(); g_oom = true; }                      \
    if (e__ != hipSuccess)                                                                          \
      return fail(DCTTS_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e__) + " @" + std::to_string(__LINE__)); \
  } while (0)

static const int PAD = 64;   // zero rows in front of / behind every tap-read activation buffer (>= 2*27)

struct HostTensor { std::vector<float> v; std::vector<int64_t> shape; };

struct DevLayer {
  ConvShape shape{0, 0, 0};
  int ntaps = 1, cin = 0, cin_p = 0, tap_off[3] = {0, 0, 0}, cout = 0, act = ACT_NONE;
  float *wp = nullptr, *bias = nullptr, *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
  bool deconv_phase = false; int phase = 0;
  float* wp16r = nullptr;         // SSRN 4T-resolution layers: packing for hconv16_kernel (row-tail launches)
  float* wpb = nullptr;           // opt-in split-bf16 form (dctts_set_split_bf16): the kernel as bf16 (hi, mid) fragments for v_mfma_f32_32x32x16_bf16; the 1025-column layers: their first 1024 columns
  float* wpx = nullptr; float* wxcol = nullptr;   // SSRN's 1025-column layers: the first 1024 columns as 32 tiles + the last column's weights (hconv_kernel<..., XC = 1>); the row tail keeps wp
  bool col_split = false;         // TextEnc's three-tap highway layers: when the 32-row items fill less than 3/4 of the CUs the layer runs as quarter-COLUMN items + a finishing pass (run_conv)
  bool tap_tail = false;          // SSRN highway layers: the rows left over after exact rounds may run as 32-row items x taps (run_conv).  SSRN only: which rows
                                  //   take that form depends on the batch, and Text2Mel's outputs stay bitwise equal across batch compositions
  ConvShape shape16{0, 0, 0};
  float* wp16 = nullptr;          // decode layers: second packing for 16x16x4 tiles (hsplit_kernel<16>)
  float* wraw = nullptr;          // decode k=1 layers: the kernel in TF layout (Cin, Cout) for mlp_rows_kernel
  float* wp16c = nullptr;         // decode causal k=3 layers (v3): centre tap only, 16x16x4 tiles (the chain contracts K = 256)
  float* wpp = nullptr;           // decode causal k=3 layers (v3): the two older taps, 32x32x2 tiles (presum GEMM, K = 512)
  bool tap2 = false;              // v3 chain view of a dilation-1 AudioEnc layer: the chain contracts taps -1 and 0 (K = 512), the presum holds tap -2 only
  int cin_real = 0;
  bool hc = false;
};

struct View {            // a (B, rows, stride) activation buffer; row0 = index of t = 0 inside a batch item
  float* p; long bstride; long row0; int stride;
  long set = 0;          // floats between the two frame-parity copies of the buffer (0 = single copy)
};

struct Buf { void* p = nullptr; size_t bytes = 0; };
struct Arena { void* base = nullptr; size_t size = 0, used = 0; };

struct dctts_ctx {
  dctts_config cfg;
  int device = 0;
  std::map<std::string, HostTensor> hw;
  bool finalized = false;
  std::vector<Arena> warena;           // weights
  std::map<std::string, std::vector<Arena>> wsarena;   // workspaces, one pool per geometry prefix ("dec.", "ssrn.", ...)
  size_t wbytes = 0;
  std::vector<DevLayer> textenc, audioenc, audiodec, ssrn;
  float* embed = nullptr;
  std::vector<int*> cone_dev; std::vector<int> cone_len;
  std::vector<int> cone_contig;        // per AudioDec layer: its cone offsets are 0, -1, -2, ... without a gap (C_1, HC_2: the row phases compute t instead of loading the table)
  std::map<std::string, Buf> ws;       // named workspaces; the key includes the geometry its prefix is selected for (ws_select)
  // Workspaces are cached PER GEOMETRY and only grow (round 5): "te." / "t2m." / "dec." / "ssrn." each select the geometry string of the running call, a
  // buffer "dec.ypad" lives under the key "dec@<geometry>.ypad" in the arena pool "dec@<geometry>.", and a geometry that comes back finds its buffers (and the
  // decode its device tables: tabcache) as it left them -- no hipDeviceSynchronize, no hipFree when a serving loop alternates between batch shapes.  Only when
  // the cached workspaces exceed ws_limit bytes does the next call drop ALL of them behind one device synchronisation (ws_trim).
  std::map<std::string, std::string> ws_sel;
  hipStream_t ws_stream = nullptr;     // the running call's stream: a NEW arena is zero-filled on it (ordered before the call's kernels)
  size_t ws_limit = (size_t)96 << 30;  // dctts_create lowers it to 60 % of the device memory that is free then (dctts_set_workspace_limit overrides)
  std::vector<void*> graveyard; size_t graveyard_bytes = 0;      // outgrown tail_ws / cols_ws buffers: freed by ws_trim / dctts_destroy, never while a stream may still use them
  struct TabSlot { void* tab = nullptr; void* mem = nullptr; int n0 = 0; size_t bytes = 0; };
  std::map<std::string, TabSlot> tabcache;                       // the decode's device tables per (kind, geometry key)
  // Calls that share scratch are ordered against each other whatever streams they come from (round 5): every entry point waits for the completion event of the
  // last call of its group that came from ANOTHER stream and records its own behind its last launch.  Groups: TextEnc (te.*, cols_ws; the decode holds it too --
  // K / V live in te.kv until the decode ends), the full-sequence AudioEnc / AudioDec functions (t2m.*), the decode (dec.*, the team kernels' exchange memory,
  // both decode streams), SSRN (ssrn.*, tail_ws).  Calls of DIFFERENT groups on different streams overlap (SSRN of batch n beside the decode of batch n + 1).
  struct UseGroup { hipEvent_t done = nullptr; hipStream_t last = nullptr; bool used = false; };
  enum { GRP_TE = 0, GRP_T2M = 1, GRP_DEC = 2, GRP_SSRN = 3, GRP_N = 4 };
  UseGroup grp[GRP_N];
  std::recursive_mutex mu;             // the host side of a context is not re-entrant: entry points of one context are serialised (enqueue only; nothing waits for the GPU under it)
  int use_graph = 0;                   // decode: 0 = every launch eager, 1 = the side stream's work as one hipGraph per frame
  int decode_mode = 3;                 // 3 = two-stream incremental form (the default), 0 = simple form: fused kernels, one stream, a device-side
                                       //     frame counter (cross-check: a different implementation of the same arithmetic)
  // ---- decode mode 3
  std::vector<DevLayer> ae_c, ad_c;    // chain view of AudioEnc / AudioDec: k=3 layers reduced to their centre tap (k=1 layers unchanged)
  std::vector<DevLayer> ae_p;          // presum view of AudioEnc's k=3 layers (taps -2d, -d; K = 512); entries of k=1 layers are unused
  DevLayer ad_c1q, ad_vw;              // AudioDec C_1 split by input rows: Q half (chain, 16-row tiles), A.V half (V . W_top precompute)
  DevLayer hc2_wt[3], hc2_wt2[3];      // AudioDec HC_2 as a row operation (rowhc2_kernel): diag(gamma_C1) W2[q], K = 256 / zero-padded to K = 512
  float* hc2_consts = nullptr;         // [3 taps][beta1 . W2[q], b1 . Wt_q, 1^T Wt_q][512]
  float* zeros512 = nullptr;
  std::vector<int*> cone3_dev;         // per AudioDec layer: cone offsets < 0 (descending) followed by 0 (the presum row)
  int* iota_dev = nullptr; int iota_n = 0;
  std::vector<hipGraphExec_t> bulk3_g; std::string graphs3_geom;   // one small linear graph per frame for the side stream, frame index baked into every launch
  void* aepre_tab = nullptr; std::string aepre_geom; int aepre_layers = 0;
  void* mlp_tab = nullptr; std::string mlp_geom;
  // How the chain runs AudioDec behind C_1 (DCTTS_CHAIN_TAIL): 2 (default) = xtail_kernel, merged form: the newest-row layers HC_2 .. HC_4, HC_5 .. HC_7 over the
  // few cone rows they need and the seven k = 1 layers around the mel frame, ONE launch in team form (the side stream stops behind HC_4); 5 = the same with
  // HC_2 .. HC_4 as an xgroup_kernel launch in front; 1 = xmlp_kernel: the seven k = 1 layers in team form (HC_5 .. HC_7 stay split between the chain's run
  // and the side stream); 0 = mlp_rows_kernel (round 2: split by rows); 3 / 4: A/B forms (tools/README.md)
  int chain_tail = 2; bool tail_on = false, xmlp_on = false;
  int tail_np = 4, np_eff = 3;         // round 6 (DCTTS_CHAIN_TAIL=7 selects 3): newest-row layers in front of the chain launch's cone layers: 4 = HC_2 .. HC_5 (HC_5's older cone rows on the side stream: xcone_kernel runs HC_3 .. HC_5),
                                       //   3 = rounds 4-5 (HC_2 .. HC_4; the side stream stops behind HC_4); np_eff: what this decode uses (the merged forms only)
  int team_u = 4;                      // utterances per team and round of THIS decode (decode_host.h: team_u_for); DCTTS_XGROUP=2 = the team kernels with four whatever the batch (A/B, tests)
  bool chain_one = false;              // round 5 (chain_tail == 2): a chain piece is ONE launch -- xtail_kernel's layers, a team barrier, the AudioEnc run + attention + C_1 (xchain_kernel); 6: two launches (round 4)
  bool dec_merge = false;              // chain_tail == 2: AudioDec's newest-row layers HC_2 .. HC_4 run in FRONT of xtail_kernel's cone layers in the same launch (chain_tail 2 and 6)
  bool ae_pass_split = false;          // round 4: AudioEnc's presums ride in the PREVIOUS piece's AudioEnc launch (a row ahead), only the C1Q . W2 row stays in the AudioDec launch
  bool side_fold = true;               // round 5: rowc1 / rowhc2 as the first phases of xcone_kernel's launch (DCTTS_XCONE=2: three launches per side-stream piece, rounds 3-4)
  bool side_pre = false;               // the small presum GEMMs (AudioEnc's presums of the next row, the newest C1Q . W2 row) run on the SIDE stream, which has the slack since xtail_kernel (round 4), instead of as passenger workgroups of the chain's AudioDec launch
  bool attn_fold = false;              // the newest row's attention + AudioDec C_1 run behind the AudioEnc run's last layer inside xgroup_kernel (chain_tail >= 1) instead of as two more launches
  void* xmlp_tab = nullptr; std::string xmlp_geom;
  void* xtail_tab = nullptr; std::string xtail_geom;
  // runs of chain highway layers as one launch whose workgroups meet inside one XCD (xgroup_kernel.h); DCTTS_XGROUP=0: one launch per layer
  int xgroup = 1;
  bool ae_pass = false; int xg_T = 0;  // AudioEnc's presums and the C1Q . W2 row ride in the AudioDec run's xgroup_kernel launch (passengers); frames of the xgroup table
  bool c1qw_chain = false;             // the C1Q . W2 row rides in the chain's AudioEnc presum launch (decode_v3)
  bool xg_on = false, xc_on = false;   // this decode uses them
  bool xgroup_ok = true;               // cleared for good when a decode reports that the placement assumption (block b on XCD b % 8) does not hold here
  void* xg_tab = nullptr; std::string xg_geom;   // per chain piece: [T + 1][2] XGroupParams (AudioDec run of frame j, AudioEnc run of frame j + 1)
  float* xg_mem = nullptr;             // exchange buffers (2 networks x 2 parity copies of rows + statistics), team barriers, error word: one allocation
  // Decode status (decode_host.h: decode_finish): the LAST kernel of every decode folds that decode's error words (team kernels, in-kernel waits, the
  // debug injection) into a STICKY device block that no decode ever clears -- [0] OR of the error bits, [1] failed decodes, [2] "the decode that just
  // ended failed" -- and poisons the decode's outputs (NaN / -1) when it failed; the block is copied to pinned host memory behind every decode and
  // cleared only by dctts_decode_status (or by the one refusal of the next decode call).
  int* dstat = nullptr; int* dstat_host = nullptr;
  const int* fin_xerr = nullptr; const int* fin_werr = nullptr;   // the running decode's own error words (decode_v3 sets them, decode_finish reads them)
  int inject_err = 0;                  // debug hook: error bits OR-ed into the NEXT decode's status (dctts_debug_inject_decode_error)
  int team_fail_streak = 0;            // consecutive failed status reports without a split-team bit: the team kernels are switched off at 3
  std::thread::id safe_once_tid;       // ... requested by this host thread: only a decode call of the same thread consumes it (another thread's decode in between keeps the team kernels)
  bool safe_once = false;              // dctts_decode_safe_once: the NEXT decode runs one launch per layer with stream-operation meetings, then the settings are as before
  // the tail of AudioDec's cone (HC_3 .. HC_7 and their row passes) as one launch per frame on the side stream (xcone_kernel.h); DCTTS_XCONE=0: nine launches
  int xcone = 1;
  void* xc_tab = nullptr; std::string xc_geom;   // per frame: XConeParams
  hipStream_t s_bulk = nullptr; hipEvent_t ev_fork = nullptr;
  unsigned* conc_flags = nullptr; int stream_retries = 0;       // decode_streams_init: the two decode streams are TESTED for running concurrently (decode_host.h); side streams rejected on the way
  hipStream_t own_tested = nullptr; bool own_ok = false;        // the last caller's high-priority stream tested against the side stream, and the result
  hipStream_t s_chain = nullptr; hipEvent_t ev_in = nullptr, ev_out = nullptr;      // decode mode 3: the chain's launches run on a high-priority stream of the context between two events on the caller's stream (decode_host.h)
  hipEvent_t ev_chain[4] = {nullptr, nullptr, nullptr, nullptr}, ev_bulk[4] = {nullptr, nullptr, nullptr, nullptr};
  int sync_values = 1;                 // the two streams meet through stream memory operations (hipStreamWriteValue32 / WaitValue32 on two counters) instead of events
                                       // (DCTTS_SYNC_VALUES=0; rocprofv3 --pmc needs events: read_env)
  uint32_t* ctr_chain = nullptr; uint32_t* ctr_bulk = nullptr;   // signal memory: chain pieces done + 1, bulk pieces done
  unsigned* sig_ptr = nullptr;         // ... the counter itself: signal memory (stream wait operations on the side stream) or wait_ctr[0] (the side stream polls in-kernel)
  unsigned sig_next = 0;               // value the next run_chain3 launch writes to the chain's counter (0 = none)
  int chain_wait_inkernel = 1;         // the chain's wait for the bulk's counter happens inside the piece's first launch (sc1 read of the one operand the bulk
                                       // produced for it) instead of a wait-value operation in front of it (DCTTS_CHAIN_WAIT=0)
  unsigned wait2_next = 0;             // counter value the next run_chain3 launch waits for (0 = none)
  unsigned* wait_ctr = nullptr;        // device memory, polled in-kernel: [0] chain pieces done + 1, [32] bulk pieces complete (written by xcone_kernel's last team, or a write-value operation), [64] error word
  hipGraph_t graph = nullptr; hipGraphExec_t graph_exec = nullptr; std::string graph_geom;   // decode mode 0: one step, replayed T times
  long long prof_rows = 0;             // output rows covered by the profiled launches since prof_enable
  int n_cu = 256;                      // CUs of the device (hipDeviceProp_t::multiProcessorCount)
  int bf16_packed = 0;                 // dctts_set_split_bf16 before finalize: 1 = SSRN's layers carry the bf16 packing, 2 = TextEnc's as well
  int bf16_mode = 0;                   // ... and what runs: 0 = exact fp32 (default), 1 = SSRN on the split-bf16 form, 2 = SSRN + TextEnc
  int pack_bf16_now = 0;               // (set while finalize builds a network whose layers get the packing)
  bool bf_te = false;                  // ... and the network is TextEnc (scratch of the TextEnc group: run_conv)
  bool bf_now = false;                 // set by the TextEnc / SSRN drivers around their run_conv calls when the split-bf16 form is selected for that network
  static constexpr int ssrn_xc = 1;    // SSRN's 1025-column layers: 8 waves x 4 tiles + one vector-ALU column in the main launch (rounds 1-4: 11 waves x 3 tiles, what their row tail still runs on)
  float* tail_ws = nullptr; size_t tail_ws_floats = 0;   // run_conv: the tap-split row tail's partial sums [3][tail rows][2C] (SSRN layers only)
  float* cols_ws = nullptr; size_t cols_ws_floats = 0;   // run_conv: the column-split layers' pre-norm rows [rows][2C] (TextEnc only).  NOT tail_ws: SSRN of the previous batch may run on
                                                         // another stream beside the next batch's TextEnc (tools/soak.py, phase C: with one buffer 536 of 3000 decodes differed, unreported)
  static constexpr int bulk_cap = 176;   // workgroups of a bulk (cone) launch of hbulk_kernel: fewer than CUs so the chain stream finds free ones
  // in-kernel trace (DCTTS_TRACE=<frame>, eager mode): wall-clock stamps of every chain launch of one frame
  long long* trace_buf = nullptr; int trace_n = 0; bool trace_on = false;
  // profiling
  // measurement knobs, read ONCE from the environment in dctts_create (tools/README.md); never consulted per call
  int trace_frame = -1, piecetime = -1;
  std::vector<int> init_pm;            // test hook: prev_max_attentions the NEXT decode starts from (consumed by it)
  std::string trace_file;
  int prof_id = -1;
  bool prof_frame = false;             // decode: the current frame is one of the sampled ones (every 16th) for DCTTS_PROF_CHAIN_HC
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
  std::vector<int> prof_cnt;           // launches bracketed by each event pair (runs of consecutive chain launches share one pair)
  hipEvent_t prof_run_e0 = nullptr; int prof_run_n = 0;
};

// Every entry point runs on the context's device and leaves the caller's current device as it found it.
struct DevGuard {
  int prev = -1, dev = -1; bool ok = true;
  explicit DevGuard(const dctts_ctx* c);
  ~DevGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};
static void destroy_graphs(dctts_ctx* c);
static void lease_forget(const dctts_ctx* c);
DevGuard::DevGuard(const dctts_ctx* c) {
  if (!c) return;
  dev = c->device;
  if (hipGetDevice(&prev) != hipSuccess) prev = -1;
  if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
}
static int round_up(int x, int m) { return (x + m - 1) / m * m; }
// hipMemset on device memory may return before the fill has run (it is queued on the null stream, which a non-blocking stream does not wait for): a buffer that kernels on
// ANOTHER stream are about to poll or count in must be zero before the call returns.  (Rounds 3-4 got that for free from the hipDeviceSynchronize of every table rebuild;
// without those, bench.py's first decode at a new batch size lost its team barriers to the late fill -- error word 1.)
static hipError_t dev_zero_now(void* p, size_t bytes) {
  hipError_t e = hipMemsetAsync(p, 0, bytes, nullptr);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(nullptr);
}

// ------------------------------------------------------------------------------------------------ weights
static int get_w(dctts_ctx* c, const std::string& name, const std::vector<int64_t>& shape, const HostTensor** out) {
  auto it = c->hw.find(name);
  if (it == c->hw.end()) return fail(DCTTS_ERR_WEIGHTS, "missing variable " + name);
  if (it->second.shape != shape) return fail(DCTTS_ERR_WEIGHTS, "bad shape for " + name);
  *out = &it->second;
  return 0;
}

// Device memory comes from a few large arenas (bump allocation, 256-byte aligned) instead of hundreds of small hipMallocs:
// every decode launch touches a different layer's weights and buffers, and with one allocation per tensor each launch
// started with address-translation misses (large contiguous arenas map with big pages and stay within TLB reach).
static const size_t ARENA_CHUNK = (size_t)512 << 20;      // weights: one context-lifetime pool
#ifndef WS_CHUNK_MB
#define WS_CHUNK_MB 64
#endif
static const size_t WS_CHUNK = (size_t)WS_CHUNK_MB << 20;          // workspaces: one pool per (prefix, geometry) -- a small geometry (B = 1) must not cost 3 x 512 MiB; bigger buffers get an arena of their own size
// `zero_on`: a workspace arena is zero-filled on the stream of the call that creates it (ordered in front of that call's kernels; calls from other streams are ordered
// behind it by the use groups); nullptr (weights, at dctts_weights_finalize): a synchronous fill.
static int arena_alloc(dctts_ctx* c, std::vector<Arena>& pool, size_t bytes, void** out, const hipStream_t* zero_on = nullptr) {
  bytes = (bytes + 255) & ~(size_t)255;
  for (Arena& a : pool) if (a.used + bytes <= a.size) { *out = (char*)a.base + a.used; a.used += bytes; return 0; }
  const size_t chunk = zero_on ? WS_CHUNK : ARENA_CHUNK;
  Arena a; a.size = bytes > chunk ? bytes : chunk; a.used = bytes;
  HIPALLOC(hipMalloc(&a.base, a.size));
  if (zero_on) {
    HIPCHK(hipMemsetAsync(a.base, 0, a.size, *zero_on));
#ifndef WS_ZERO_NOSYNC
    // ... and the fill has COMPLETED when the first call at a new geometry goes on (round 6): with 64 MiB arenas a decode variant's first run at a new batch size
    // came out wrong (tools/flaky_probe.py: not with 512 MiB arenas, not on any later run) -- the fill and the decode's two own streams, although ordered through
    // events, see dev_zero_now.  One stream synchronisation per NEW arena, i.e. on the first call of a shape; nothing on the calls that find their buffers.
    HIPCHK(hipStreamSynchronize(*zero_on));
#endif
  } else HIPCHK(hipMemset(a.base, 0, a.size));
  pool.push_back(a);
  *out = a.base;
  return 0;
}

static int upload(dctts_ctx* c, const std::vector<float>& h, float** d) {
  void* p = nullptr;
  CHK(arena_alloc(c, c->warena, h.size() * sizeof(float), &p));
  HIPCHK(hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  c->wbytes += h.size() * sizeof(float);
  *d = (float*)p;
  return 0;
}

// Pack W(tap, c, col) into MFMA fragment order: [tile][kgroup][lane][4].
//   MF = 32 (v_mfma_f32_32x32x2_f32): lane l, element i <- k = 8*kg + 4*(l>>5) + i, column l&31 of the tile
//   MF = 16 (v_mfma_f32_16x16x4_f32): lane l, element i <- k = 16*kg + 4*(l>>4) + i, column l&15 of the tile
// Tiles of a highway layer alternate gate (H1) / info (H2) blocks of the same MF channels.
static std::vector<float> pack_bw(const std::function<float(int, int, int)>& W, int ntaps, int cin_real, int cin_p,
                                  int tiles, int cout, bool hc, int MF) {
  const int KGS = (MF == 32) ? 8 : 16, KG = ntaps * cin_p / KGS, sh = (MF == 32) ? 5 : 4;
  std::vector<float> out((size_t)tiles * KG * 256, 0.f);
  for (int gt = 0; gt < tiles; ++gt)
    for (int kg = 0; kg < KG; ++kg)
      for (int l = 0; l < 64; ++l) {
        int col;
        if (hc) { const int ch = (gt / 2) * MF + (l & (MF - 1)); col = ch < cout ? (gt & 1) * cout + ch : -1; }
        else    { const int ch = gt * MF + (l & (MF - 1));       col = ch < cout ? ch : -1; }
        if (col < 0) continue;
        for (int i = 0; i < 4; ++i) {
          const int k = kg * KGS + 4 * (l >> sh) + i, tap = k / cin_p, cc = k % cin_p;
          if (cc < cin_real) out[(((size_t)gt * KG + kg) * 64 + l) * 4 + i] = W(tap, cc, col);
        }
      }
  return out;
}
// The split-bf16 packing (hconv_kernel.h: BF): [tile][16-k group][hi | mid][lane][8 bf16]; lane l, element e <- k = 16 g + 8 (l >> 5) + e, column l & 31 of the tile.
// hi = bf16(w), mid = bf16(w - hi), both round-to-nearest-even.  Returned as floats (two bf16 per float) so that it travels through upload().
static uint16_t f2bf16(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf16_2f(uint16_t h) { const uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static std::vector<float> pack_bw_bf16(const std::function<float(int, int, int)>& W, int ntaps, int cin_real, int cin_p, int tiles, int cout, bool hc) {
  const int KG = ntaps * cin_p / 16;
  std::vector<uint16_t> out((size_t)tiles * KG * 2 * 64 * 8, 0);
  for (int gt = 0; gt < tiles; ++gt)
    for (int kg = 0; kg < KG; ++kg)
      for (int l = 0; l < 64; ++l) {
        int col;
        if (hc) { const int ch = (gt / 2) * 32 + (l & 31); col = ch < cout ? (gt & 1) * cout + ch : -1; }
        else    { const int ch = gt * 32 + (l & 31);       col = ch < cout ? ch : -1; }
        if (col < 0) continue;
        for (int e = 0; e < 8; ++e) {
          const int k = kg * 16 + 8 * (l >> 5) + e, tap = k / cin_p, cc = k % cin_p;
          if (cc >= cin_real) continue;
          const float w = W(tap, cc, col);
          const uint16_t hi = f2bf16(w), mid = f2bf16(w - bf16_2f(hi));
          const size_t base = (((size_t)gt * KG + kg) * 2) * 64 * 8;
          out[base + (size_t)l * 8 + e] = hi;
          out[base + 64 * 8 + (size_t)l * 8 + e] = mid;
        }
      }
  std::vector<float> f(out.size() / 2);
  memcpy(f.data(), out.data(), out.size() * 2);
  return f;
}

static std::vector<float> pack_b(const std::function<float(int, int, int)>& W, int ntaps, int cin_real, int cin_p,
                                 const ConvShape& s, int cout, bool hc) {
  return pack_bw(W, ntaps, cin_real, cin_p, s.nt * s.nw, cout, hc, 32);
}

static int make_C(dctts_ctx* c, const std::string& scope, int cin_real, int cin_read, int cout, int act, DevLayer* L, bool dec = false, bool tail = false) {
  const HostTensor *k, *b, *be, *ga;
  CHK(get_w(c, scope + "/conv1d/kernel", {1, cin_real, cout}, &k));
  CHK(get_w(c, scope + "/conv1d/bias", {cout}, &b));
  CHK(get_w(c, scope + "/normalize/beta", {cout}, &be));
  CHK(get_w(c, scope + "/normalize/gamma", {cout}, &ga));
  L->shape = pick_shape(EPI_C, cout);
  L->ntaps = 1; L->cin = cin_read; L->cin_p = round_up(cin_real, 32); L->cout = cout; L->act = act;
  L->tap_off[0] = L->tap_off[1] = L->tap_off[2] = 0;
  const float* kv = k->v.data();
  auto W = [=](int, int cc, int col) { return kv[(size_t)cc * cout + col]; };
  CHK(upload(c, pack_b(W, 1, cin_real, L->cin_p, L->shape, cout, false), &L->wp));
  if (dec) CHK(upload(c, pack_bw(W, 1, cin_real, L->cin_p, 2 * ((cout + 31) / 32), cout, false, 16), &L->wp16));
  if (dec) CHK(upload(c, k->v, &L->wraw));
  if (tail) {
    L->shape16 = pick_shape16(EPI_C, cout);
    CHK(upload(c, pack_bw(W, 1, cin_real, L->cin_p, L->shape16.nt * L->shape16.nw, cout, false, 16), &L->wp16r));
  }
  if (c->pack_bf16_now) {
    if (cout == 1025) CHK(upload(c, pack_bw_bf16(W, 1, cin_real, L->cin_p, 32, 1024, false), &L->wpb));      // + wxcol below: the last column stays an fp32 dot product
    else CHK(upload(c, pack_bw_bf16(W, 1, cin_real, L->cin_p, L->shape.nt * L->shape.nw, cout, false), &L->wpb));
  }
  if (((tail && c->ssrn_xc) || c->pack_bf16_now) && cout == 1025 && L->cin_p <= 1120) {
    if (tail && c->ssrn_xc) CHK(upload(c, pack_bw(W, 1, cin_real, L->cin_p, 32, 1024, false, 32), &L->wpx));
    std::vector<float> col((size_t)L->cin_p, 0.f);
    for (int cc = 0; cc < cin_real; ++cc) col[cc] = kv[(size_t)cc * cout + 1024];
    CHK(upload(c, col, &L->wxcol));
  }
  L->cin_real = cin_real;
  CHK(upload(c, b->v, &L->bias)); CHK(upload(c, ga->v, &L->g1)); CHK(upload(c, be->v, &L->b1));
  return 0;
}

static int make_HC(dctts_ctx* c, const std::string& scope, int C, int k, int rate, bool causal, DevLayer* L, bool dec = false, bool tail = false) {
  const HostTensor *kw, *b, *b1, *g1, *b2, *g2;
  CHK(get_w(c, scope + "/conv1d/kernel", {k, C, 2 * C}, &kw));
  CHK(get_w(c, scope + "/conv1d/bias", {2 * C}, &b));
  CHK(get_w(c, scope + "/H1/beta", {C}, &b1)); CHK(get_w(c, scope + "/H1/gamma", {C}, &g1));
  CHK(get_w(c, scope + "/H2/beta", {C}, &b2)); CHK(get_w(c, scope + "/H2/gamma", {C}, &g2));
  L->shape = pick_shape(EPI_HC, C);
  L->ntaps = k; L->cin = C; L->cin_p = round_up(C, 32); L->cout = C; L->act = ACT_NONE;
  for (int j = 0; j < 3; ++j) L->tap_off[j] = 0;
  if (k == 3) {
    // tf.layers.conv1d tap j reads x[t + j*rate - pad_left]: CAUSAL pad_left = 2*rate, SAME pad_left = rate
    const int pl = causal ? 2 * rate : rate;
    for (int j = 0; j < 3; ++j) L->tap_off[j] = j * rate - pl;
  }
  const float* kv = kw->v.data();
  auto W = [=](int tap, int cc, int col) { return kv[((size_t)tap * C + cc) * (2 * C) + col]; };
  CHK(upload(c, pack_b(W, k, C, L->cin_p, L->shape, C, true), &L->wp));
  if (dec) CHK(upload(c, pack_bw(W, k, C, L->cin_p, 2 * (C / 16), C, true, 16), &L->wp16));
  if (dec && k == 3 && causal) {
    auto Wc = [=](int, int cc, int col) { return kv[((size_t)2 * C + cc) * (2 * C) + col]; };        // tap 2 sees x[t]
    CHK(upload(c, pack_bw(Wc, 1, C, L->cin_p, 2 * (C / 16), C, true, 16), &L->wp16c));
    CHK(upload(c, pack_bw(W, 2, C, L->cin_p, L->shape.nt * L->shape.nw, C, true, 32), &L->wpp));   // taps 0, 1 see x[t-2d], x[t-d]
  }
  if (tail) {
    L->shape16 = pick_shape16(EPI_HC, C);
    CHK(upload(c, pack_bw(W, k, C, L->cin_p, L->shape16.nt * L->shape16.nw, C, true, 16), &L->wp16r));
  }
  if (c->pack_bf16_now) CHK(upload(c, pack_bw_bf16(W, k, C, L->cin_p, L->shape.nt * L->shape.nw, C, true), &L->wpb));
  L->hc = true;
  CHK(upload(c, b->v, &L->bias));
  CHK(upload(c, g1->v, &L->g1)); CHK(upload(c, b1->v, &L->b1));
  CHK(upload(c, g2->v, &L->g2)); CHK(upload(c, b2->v, &L->b2));
  return 0;
}

// conv2d_transpose (1,3,Cout,Cin), stride 2, 'same' (modules.py:232-239):
//   out[2t]   = b + x[t] W0 + x[t-1] W2      (phase 0: taps {0,-1})
//   out[2t+1] = b + x[t] W1                  (phase 1: tap  {0})
static int make_D(dctts_ctx* c, const std::string& scope, int C, DevLayer* even, DevLayer* odd) {
  const HostTensor *kw, *b, *be, *ga;
  CHK(get_w(c, scope + "/conv2d_transpose/kernel", {1, 3, C, C}, &kw));
  CHK(get_w(c, scope + "/conv2d_transpose/bias", {C}, &b));
  CHK(get_w(c, scope + "/normalize/beta", {C}, &be));
  CHK(get_w(c, scope + "/normalize/gamma", {C}, &ga));
  const float* kv = kw->v.data();
  float *bias, *g, *bb;
  CHK(upload(c, b->v, &bias)); CHK(upload(c, ga->v, &g)); CHK(upload(c, be->v, &bb));
  for (int ph = 0; ph < 2; ++ph) {
    DevLayer* L = ph ? odd : even;
    L->shape = pick_shape(EPI_C, C);
    L->ntaps = ph ? 1 : 2; L->cin = C; L->cin_p = round_up(C, 32); L->cout = C; L->act = ACT_NONE;
    L->tap_off[0] = 0; L->tap_off[1] = -1; L->tap_off[2] = 0;
    L->deconv_phase = true; L->phase = ph;
    auto W = [=](int tap, int cc, int col) {
      const int j = ph ? 1 : (tap == 0 ? 0 : 2);
      return kv[((size_t)j * C + col) * C + cc];          // kernel[0][j][out][in]
    };
    CHK(upload(c, pack_b(W, L->ntaps, C, L->cin_p, L->shape, C, false), &L->wp));
    if (c->pack_bf16_now) CHK(upload(c, pack_bw_bf16(W, L->ntaps, C, L->cin_p, L->shape.nt * L->shape.nw, C, false), &L->wpb));
    L->bias = bias; L->g1 = g; L->b1 = bb;
  }
  return 0;
}

static std::vector<std::vector<int>> audiodec_cone(const std::vector<DevLayer>& ad) {
  // output-row offsets (relative to the newest frame) each AudioDec layer must emit
  std::vector<std::vector<int>> out(ad.size());
  std::vector<int> need{0};
  for (int i = (int)ad.size() - 1; i >= 0; --i) {
    out[i] = need;
    if (ad[i].ntaps > 1) {
      std::vector<char> mark(4096, 0);
      for (int o : need) for (int j = 0; j < ad[i].ntaps; ++j) mark[-(o + ad[i].tap_off[j])] = 1;
      need.clear();
      for (int o = 0; o < 4096; ++o) if (mark[o]) need.push_back(-o);
    }
  }
  return out;
}


// Measurement / A-B knobs (tools/README.md).  Read once per context: the decode path itself never calls getenv.
static void read_env(dctts_ctx* c) {
  auto geti = [](const char* n, int* v) { if (const char* e = getenv(n)) *v = atoi(e); };
  geti("DCTTS_SYNC_VALUES", &c->sync_values); geti("DCTTS_CHAIN_WAIT", &c->chain_wait_inkernel); geti("DCTTS_XGROUP", &c->xgroup); geti("DCTTS_XCONE", &c->xcone); geti("DCTTS_CHAIN_TAIL", &c->chain_tail);
  geti("DCTTS_TRACE", &c->trace_frame);
  if (c->chain_tail == 7) { c->chain_tail = 2; c->tail_np = 3; }      // 7 = the default form with round 5's split of the cone (A/B) geti("DCTTS_PIECETIME", &c->piecetime);
  if (const char* e = getenv("DCTTS_TRACE_FILE")) c->trace_file = e;
  // rocprofv3 --pmc serialises dispatches ACROSS queues: a launch that polls the other stream's counter would never see it move
  // (it only times out, with wrong results).  Under counter collection the two decode streams meet through events instead.
  if (const char* e = getenv("ROCPROF_COUNTER_COLLECTION")) { if (atoi(e) != 0 && !getenv("DCTTS_SYNC_VALUES")) c->sync_values = 0; }
}

// ------------------------------------------------------------------------------------------------ C ABI: lifetime
extern "C" const char* dctts_last_error(void) { return g_err.c_str(); }

extern "C" int dctts_create(dctts_ctx** out, int device, const dctts_config* cfg) {
  if (!out || !cfg) return fail(DCTTS_ERR_ARG, "null argument");
  if (cfg->d != 256) return fail(DCTTS_ERR_ARG, "attention kernels are specialised for d == 256");
  if (cfg->attention_win_size < 1 || cfg->attention_win_size > MAXWIN)
    return fail(DCTTS_ERR_ARG, "attention_win_size must be 1..3: the decode attention kernels are unrolled for a 3-key window (hyperparams.py:32)");
  if (cfg->max_N < 1 || cfg->n_mels < 4 || cfg->n_mels > 128 || (cfg->n_mels & 3)) return fail(DCTTS_ERR_ARG, "unsupported max_N / n_mels");
  dctts_ctx* c = new dctts_ctx();
  c->cfg = *cfg; c->device = device;
  DevGuard dev_guard(c);
  if (!dev_guard.ok) { delete c; return fail(DCTTS_ERR_HIP, "hipSetDevice: no such device"); }
  int ncu = 0;
  if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->n_cu = ncu;
  read_env(c);
  {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr > 0) { const size_t lim = fr / 10 * 6; if (lim < c->ws_limit) c->ws_limit = lim; }
    else (void)hipGetLastError();
  }
  *out = c;
  return 0;
}

static void free_ws(dctts_ctx* c) {
  for (auto& kv : c->wsarena) for (Arena& a : kv.second) (void)hipFree(a.base);
  c->wsarena.clear();
  c->ws.clear();
}

extern "C" int dctts_destroy(dctts_ctx* c) {
  if (!c) return 0;
  DevGuard dev_guard(c);
  (void)hipDeviceSynchronize();        // nothing of this context may still be running when its memory goes
  if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
  if (c->graph) (void)hipGraphDestroy(c->graph);
  destroy_graphs(c);
  for (int i = 0; i < 4; ++i) { if (c->ev_chain[i]) (void)hipEventDestroy(c->ev_chain[i]); if (c->ev_bulk[i]) (void)hipEventDestroy(c->ev_bulk[i]); }
  if (c->s_bulk) (void)hipStreamDestroy(c->s_bulk);
  if (c->s_chain) (void)hipStreamDestroy(c->s_chain);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_out) (void)hipEventDestroy(c->ev_out);
  if (c->ctr_chain) (void)hipFree(c->ctr_chain);
  if (c->ctr_bulk) (void)hipFree(c->ctr_bulk);
  if (c->wait_ctr) (void)hipFree(c->wait_ctr);
  if (c->conc_flags) (void)hipFree(c->conc_flags);
  if (c->dstat) (void)hipFree(c->dstat);
  if (c->dstat_host) (void)hipHostFree(c->dstat_host);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->trace_buf) (void)hipFree(c->trace_buf);
  if (c->tail_ws) (void)hipFree(c->tail_ws);
  if (c->cols_ws) (void)hipFree(c->cols_ws);
  for (void* p : c->graveyard) (void)hipFree(p);
  for (auto& kv : c->tabcache) { if (kv.second.tab) (void)hipFree(kv.second.tab); if (kv.second.mem) (void)hipFree(kv.second.mem); }
  for (auto& u : c->grp) if (u.done) (void)hipEventDestroy(u.done);
  lease_forget(c);
  free_ws(c);
  for (Arena& a : c->warena) (void)hipFree(a.base);
  for (int* p : c->cone_dev) (void)hipFree(p);
  for (int* p : c->cone3_dev) (void)hipFree(p);
  if (c->iota_dev) (void)hipFree(c->iota_dev);
  for (auto& e : c->prof_ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  delete c;
  return 0;
}

extern "C" int dctts_weights_set(dctts_ctx* c, const char* name, const float* data, const int64_t* shape, int ndim) {
  if (!c || !name || !data || !shape || ndim < 1 || ndim > 4) return fail(DCTTS_ERR_ARG, "bad argument");
  if (c->finalized) return fail(DCTTS_ERR_STATE, "weights already finalized");
  HostTensor t; size_t n = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
  t.v.assign(data, data + n);
  c->hw[name] = std::move(t);
  return 0;
}

extern "C" int dctts_weights_finalize(dctts_ctx* c) {
  if (!c) return fail(DCTTS_ERR_ARG, "null ctx");
  if (c->finalized) return 0;
  DevGuard dev_guard(c);
  if (!dev_guard.ok) return fail(DCTTS_ERR_HIP, "hipSetDevice");
  const dctts_config& g = c->cfg;
  const int d = g.d, cc = g.c, F = g.n_linear, Fp = round_up(F, 32);
  char nm[64];
  // ---- TextEnc (networks.py:14-71)
  c->pack_bf16_now = c->bf16_packed >= 2;
  {
    const std::string s = "Text2Mel/TextEnc/";
    const HostTensor* tab;
    CHK(get_w(c, s + "embed_1/lookup_table", {g.vocab_size, g.e}, &tab));
    std::vector<float> t = tab->v;
    for (int i = 0; i < g.e; ++i) t[i] = 0.f;                     // modules.py:36-38: row 0 := zeros
    CHK(upload(c, t, &c->embed));
    int i = 2; DevLayer L;
    snprintf(nm, 64, "C_%d", i++); L = DevLayer(); CHK(make_C(c, s + nm, g.e, g.e, 2 * d, ACT_RELU, &L)); c->textenc.push_back(L);
    snprintf(nm, 64, "C_%d", i++); L = DevLayer(); CHK(make_C(c, s + nm, 2 * d, 2 * d, 2 * d, ACT_NONE, &L)); c->textenc.push_back(L);
    for (int rep = 0; rep < 2; ++rep) for (int j = 0, r = 1; j < 4; ++j, r *= 3) {
      snprintf(nm, 64, "HC_%d", i++); L = DevLayer(); CHK(make_HC(c, s + nm, 2 * d, 3, r, false, &L)); L.col_split = true; c->textenc.push_back(L); }
    for (int rep = 0; rep < 2; ++rep) { snprintf(nm, 64, "HC_%d", i++); L = DevLayer(); CHK(make_HC(c, s + nm, 2 * d, 3, 1, false, &L)); L.col_split = true; c->textenc.push_back(L); }
    for (int rep = 0; rep < 2; ++rep) { snprintf(nm, 64, "HC_%d", i++); L = DevLayer(); CHK(make_HC(c, s + nm, 2 * d, 1, 1, false, &L)); L.col_split = true; c->textenc.push_back(L); }
  }
  c->pack_bf16_now = 0;                 // Text2Mel's decode networks are never run in reduced form: the attention trajectory is fed back
  // ---- AudioEnc (networks.py:73-124), causal
  {
    const std::string s = "Text2Mel/AudioEnc/";
    int i = 1; DevLayer L;
    snprintf(nm, 64, "C_%d", i++); L = DevLayer(); CHK(make_C(c, s + nm, g.n_mels, g.n_mels, d, ACT_RELU, &L, true)); c->audioenc.push_back(L);
    snprintf(nm, 64, "C_%d", i++); L = DevLayer(); CHK(make_C(c, s + nm, d, d, d, ACT_RELU, &L, true)); c->audioenc.push_back(L);
    snprintf(nm, 64, "C_%d", i++); L = DevLayer(); CHK(make_C(c, s + nm, d, d, d, ACT_NONE, &L, true)); c->audioenc.push_back(L);
    for (int rep = 0; rep < 2; ++rep) for (int j = 0, r = 1; j < 4; ++j, r *= 3) {
      snprintf(nm, 64, "HC_%d", i++); L = DevLayer(); CHK(make_HC(c, s + nm, d, 3, r, true, &L, true)); c->audioenc.push_back(L); }
    for (int rep = 0; rep < 2; ++rep) { snprintf(nm, 64, "HC_%d", i++); L = DevLayer(); CHK(make_HC(c, s + nm, d, 3, 3, true, &L, true)); c->audioenc.push_back(L); }
  }
  // ---- AudioDec (networks.py:157-212), causal
  {
    const std::string s = "Text2Mel/AudioDec/";
    int i = 1; DevLayer L;
    snprintf(nm, 64, "C_%d", i++); L = DevLayer(); CHK(make_C(c, s + nm, 2 * d, 2 * d, d, ACT_NONE, &L, true)); c->audiodec.push_back(L);
    for (int j = 0, r = 1; j < 4; ++j, r *= 3) { snprintf(nm, 64, "HC_%d", i++); L = DevLayer(); CHK(make_HC(c, s + nm, d, 3, r, true, &L, true)); c->audiodec.push_back(L); }
    for (int rep = 0; rep < 2; ++rep) { snprintf(nm, 64, "HC_%d", i++); L = DevLayer(); CHK(make_HC(c, s + nm, d, 3, 1, true, &L, true)); c->audiodec.push_back(L); }
    for (int rep = 0; rep < 3; ++rep) { snprintf(nm, 64, "C_%d", i++); L = DevLayer(); CHK(make_C(c, s + nm, d, d, d, ACT_RELU, &L, true)); c->audiodec.push_back(L); }
    snprintf(nm, 64, "C_%d", i++); L = DevLayer(); CHK(make_C(c, s + nm, d, d, g.n_mels, ACT_SIGMOID, &L, true)); c->audiodec.push_back(L);
    auto cone = audiodec_cone(c->audiodec);
    for (auto& v : cone) {
      int* dp = nullptr;
      HIPCHK(hipMalloc((void**)&dp, v.size() * sizeof(int)));
      HIPCHK(hipMemcpy(dp, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
      c->cone_dev.push_back(dp); c->cone_len.push_back((int)v.size());
      { bool cg = true; for (size_t q = 0; q < v.size(); ++q) cg = cg && v[q] == -(int)q; c->cone_contig.push_back(cg ? 1 : 0); }
      std::vector<int> v3(v.begin() + 1, v.end()); v3.push_back(0);       // v[0] == 0: the newest row goes last (presum row)
      HIPCHK(hipMalloc((void**)&dp, v3.size() * sizeof(int)));
      HIPCHK(hipMemcpy(dp, v3.data(), v3.size() * sizeof(int), hipMemcpyHostToDevice));
      c->cone3_dev.push_back(dp);
    }
    // ---- decode v3 views
    auto centre = [](const DevLayer& L) { DevLayer x = L; if (L.wp16c) { x.ntaps = 1; x.tap_off[0] = x.tap_off[1] = x.tap_off[2] = 0; x.wp16 = L.wp16c; } return x; };
    auto older = [](const DevLayer& L) { DevLayer x = L; if (L.wpp) { x.ntaps = 2; x.tap_off[2] = 0; x.wp = L.wpp; } return x; };
    for (const DevLayer& L : c->audioenc) { c->ae_c.push_back(centre(L)); c->ae_p.push_back(older(L)); }
    // AudioEnc layers of dilation 1: tap -1 reads the row the chain produced one frame earlier, so it stays in the chain (K = 512)
    // and the presum -- computed a whole chain piece ahead, inside the bulk graph -- carries bias + tap -2 only
    for (size_t i = 0; i < c->audioenc.size(); ++i) {
      const DevLayer& L = c->audioenc[i];
      if (!L.wpp || L.tap_off[1] != -1) continue;
      char sc[96]; snprintf(sc, 96, "Text2Mel/AudioEnc/HC_%d/conv1d/kernel", (int)i + 1);
      const HostTensor* kw; const int C = L.cout;
      CHK(get_w(c, sc, {3, C, 2 * C}, &kw));
      const float* kv2 = kw->v.data();
      auto W12 = [=](int tap, int cc, int col) { return kv2[((size_t)(tap + 1) * C + cc) * (2 * C) + col]; };
      auto W0z = [=](int tap, int cc, int col) { return tap == 0 ? kv2[((size_t)cc) * (2 * C) + col] : 0.f; };
      DevLayer& X = c->ae_c[i]; X.ntaps = 2; X.tap2 = true;
      CHK(upload(c, pack_bw(W12, 2, C, L.cin_p, 2 * (C / 16), C, true, 16), &X.wp16));
      DevLayer& Pz = c->ae_p[i]; Pz.tap_off[0] = Pz.tap_off[1] = -2;
      CHK(upload(c, pack_bw(W0z, 2, C, L.cin_p, L.shape.nt * L.shape.nw, C, true, 32), &Pz.wpp));
    }
    for (const DevLayer& L : c->audiodec) c->ad_c.push_back(centre(L));
    const HostTensor* k1;
    CHK(get_w(c, s + "C_1/conv1d/kernel", {1, 2 * d, d}, &k1));
    const float* kv = k1->v.data();
    auto Wtop = [=](int, int cc, int col) { return kv[(size_t)cc * d + col]; };              // rows that multiply A.V (networks.py:150-151)
    auto Wbot = [=](int, int cc, int col) { return kv[(size_t)(d + cc) * d + col]; };        // rows that multiply Q
    c->ad_c1q = c->audiodec[0]; c->ad_c1q.cin = c->ad_c1q.cin_p = c->ad_c1q.cin_real = d; c->ad_c1q.wp = nullptr; c->ad_c1q.wraw = nullptr;
    CHK(upload(c, pack_bw(Wbot, 1, d, d, 2 * ((d + 31) / 32), d, false, 16), &c->ad_c1q.wp16));
    c->ad_vw = c->ad_c1q; c->ad_vw.wp16 = nullptr;
    CHK(upload(c, pack_bw(Wtop, 1, d, d, (d + 31) / 32, d, false, 32), &c->ad_vw.wp));
    CHK(upload(c, std::vector<float>(d, 0.f), &c->ad_vw.bias));
    {
      // HC_2 as a row operation (decode3_kernels.h: rowhc2_kernel)
      const HostTensor *k2, *ga1, *be1, *bi1;
      CHK(get_w(c, s + "HC_2/conv1d/kernel", {3, d, 2 * d}, &k2));
      CHK(get_w(c, s + "C_1/normalize/gamma", {d}, &ga1)); CHK(get_w(c, s + "C_1/normalize/beta", {d}, &be1)); CHK(get_w(c, s + "C_1/conv1d/bias", {d}, &bi1));
      const float* w2 = k2->v.data(); const float* gam = ga1->v.data();
      std::vector<float> consts((size_t)3 * 3 * 2 * d, 0.f);
      for (int q = 0; q < 3; ++q)
        for (int cc = 0; cc < d; ++cc)
          for (int col = 0; col < 2 * d; ++col) {
            const double wv = w2[((size_t)q * d + cc) * (2 * d) + col], wt = (double)gam[cc] * wv;
            consts[((size_t)q * 3 + 0) * 2 * d + col] += (float)((double)be1->v[cc] * wv);
            consts[((size_t)q * 3 + 1) * 2 * d + col] += (float)((double)bi1->v[cc] * wt);
            consts[((size_t)q * 3 + 2) * 2 * d + col] += (float)wt;
          }
      CHK(upload(c, consts, &c->hc2_consts));
      CHK(upload(c, std::vector<float>(2 * d, 0.f), &c->zeros512));
      for (int q = 0; q < 3; ++q) {
        auto Wt = [=](int tap, int cc, int col) { return tap == 0 ? gam[cc] * w2[((size_t)q * d + cc) * (2 * d) + col] : 0.f; };
        DevLayer L = c->audiodec[1];
        L.ntaps = 1; L.tap_off[0] = L.tap_off[1] = L.tap_off[2] = 0; L.bias = c->zeros512; L.wp16 = L.wp16c = L.wpp = nullptr;
        CHK(upload(c, pack_bw(Wt, 1, d, d, L.shape.nt * L.shape.nw, d, true, 32), &L.wp));
        c->hc2_wt[q] = L;
        L.ntaps = 2;
        CHK(upload(c, pack_bw(Wt, 2, d, d, L.shape.nt * L.shape.nw, d, true, 32), &L.wp));
        c->hc2_wt2[q] = L;
      }
    }
    c->iota_n = g.max_N > 1024 ? g.max_N : 1024;
    std::vector<int> io(c->iota_n); for (int q = 0; q < c->iota_n; ++q) io[q] = q;
    HIPCHK(hipMalloc((void**)&c->iota_dev, io.size() * sizeof(int)));
    HIPCHK(hipMemcpy(c->iota_dev, io.data(), io.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  // ---- SSRN (networks.py:214-292), SAME
  c->pack_bf16_now = c->bf16_packed >= 1;
  {
    const std::string s = "SSRN/";
    int i = 1; DevLayer L, L2;
    snprintf(nm, 64, "C_%d", i++); L = DevLayer(); CHK(make_C(c, s + nm, g.n_mels, g.n_mels, cc, ACT_NONE, &L)); c->ssrn.push_back(L);
    for (int j = 0, r = 1; j < 2; ++j, r *= 3) { snprintf(nm, 64, "HC_%d", i++); L = DevLayer(); CHK(make_HC(c, s + nm, cc, 3, r, false, &L)); L.tap_tail = true; c->ssrn.push_back(L); }
    for (int rep = 0; rep < 2; ++rep) {
      snprintf(nm, 64, "D_%d", i++); L = DevLayer(); L2 = DevLayer(); CHK(make_D(c, s + nm, cc, &L, &L2)); c->ssrn.push_back(L); c->ssrn.push_back(L2);
      for (int j = 0, r = 1; j < 2; ++j, r *= 3) { snprintf(nm, 64, "HC_%d", i++); L = DevLayer(); CHK(make_HC(c, s + nm, cc, 3, r, false, &L, false, rep == 1)); L.tap_tail = true; c->ssrn.push_back(L); }
    }
    // everything from HC_8 on runs at 4T rows: those layers also get the 16-row packing for the row-tail launch
    snprintf(nm, 64, "C_%d", i++); L = DevLayer(); CHK(make_C(c, s + nm, cc, cc, 2 * cc, ACT_NONE, &L, false, true)); c->ssrn.push_back(L);
    for (int rep = 0; rep < 2; ++rep) { snprintf(nm, 64, "HC_%d", i++); L = DevLayer(); CHK(make_HC(c, s + nm, 2 * cc, 3, 1, false, &L, false, true)); L.tap_tail = true; c->ssrn.push_back(L); }
    snprintf(nm, 64, "C_%d", i++); L = DevLayer(); CHK(make_C(c, s + nm, 2 * cc, 2 * cc, F, ACT_NONE, &L, false, true)); c->ssrn.push_back(L);
    for (int rep = 0; rep < 2; ++rep) { snprintf(nm, 64, "C_%d", i++); L = DevLayer(); CHK(make_C(c, s + nm, F, Fp, F, ACT_RELU, &L, false, true)); c->ssrn.push_back(L); }
    snprintf(nm, 64, "C_%d", i++); L = DevLayer(); CHK(make_C(c, s + nm, F, Fp, F, ACT_SIGMOID, &L, false, true)); c->ssrn.push_back(L);
  }
  c->pack_bf16_now = 0;
  HIPCHK(hipDeviceSynchronize());
  c->hw.clear();
  c->finalized = true;
  return 0;
}

// ------------------------------------------------------------------------------------------------ workspaces
// Select the geometry the buffers of `prefix` ("dec.", "ssrn.", ...) belong to for the running call, and the stream a new arena is zero-filled on.
static void ws_select(dctts_ctx* c, const std::string& prefix, const std::string& geometry, hipStream_t st) {
  c->ws_sel[prefix] = geometry; c->ws_stream = st;
}

static int ws_get(dctts_ctx* c, const std::string& name, size_t bytes, void** out) {
  const size_t dot = name.find('.');
  const std::string prefix = name.substr(0, dot + 1);
  auto sel = c->ws_sel.find(prefix);
  const std::string pool = name.substr(0, dot) + "@" + (sel == c->ws_sel.end() ? std::string() : sel->second) + ".";
  Buf& b = c->ws[pool + name.substr(dot + 1)];
  if (b.bytes != bytes) {
    // a new buffer (or, within one geometry, a changed size: the old slice stays in its pool).  Arenas are zero-filled when created: pad rows / columns stay
    // zero because kernels never write them.
    CHK(arena_alloc(c, c->wsarena[pool], bytes, &b.p, &c->ws_stream));
    b.bytes = bytes;
  }
  *out = b.p;
  return 0;
}

static int ws_view(dctts_ctx* c, const std::string& name, int B, long rows, long row0, int stride, View* v) {
  void* p;
  CHK(ws_get(c, name, (size_t)B * rows * stride * sizeof(float), &p));
  *v = View{(float*)p, rows, row0, stride, 0};
  return 0;
}

static int ws_view2(dctts_ctx* c, const std::string& name, int B, long rows, long row0, int stride, View* v) {   // two parity copies
  void* p;
  CHK(ws_get(c, name, (size_t)2 * B * rows * stride * sizeof(float), &p));
  *v = View{(float*)p, rows, row0, stride, (long)B * rows * stride};
  return 0;
}

extern "C" size_t dctts_device_bytes(const dctts_ctx* c) {
  size_t n = 0;
  for (const Arena& a : c->warena) n += a.size;
  for (auto& kv : c->wsarena) for (const Arena& a : kv.second) n += a.size;
  n += (c->tail_ws_floats + c->cols_ws_floats) * sizeof(float) + c->graveyard_bytes;
  for (auto& kv : c->tabcache) n += kv.second.bytes;      // the decode's device tables and team exchange memory
  return n;
}

// tail_ws / cols_ws only grow: a bigger buffer is allocated, the outgrown one is kept (another stream's earlier call may still be using it) until ws_trim / dctts_destroy.
// (Every launch that reads or writes the scratch buffer is fully written before it is read within the same call: no zero fill.)
static int grow_scratch(dctts_ctx* c, float** buf, size_t* floats, size_t need) {
  if (*buf) { c->graveyard.push_back(*buf); c->graveyard_bytes += *floats * sizeof(float); *buf = nullptr; *floats = 0; }
  HIPALLOC(hipMalloc((void**)buf, need * sizeof(float)));
  *floats = need;
  return 0;
}

// ------------------------------------------------------------------------------------------------ one conv launch
struct RowMap { int B; int R; const int* offs; const int* step; };

static int run_conv(dctts_ctx* c, const DevLayer& L, const View& in, const int* gather, const View& out,
                    const RowMap& rm, hipStream_t st, int out_zero_to = 0, const View* out2 = nullptr,
                    int out_tmul = 1, int out_tadd = 0) {
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.in = in.p; p.gather = gather; p.gather_n = c->cfg.vocab_size; p.in_bstride = in.bstride; p.in_row0 = in.row0; p.in_stride = in.stride;
  p.cin = L.cin; p.cin_p = L.cin_p; p.ntaps = L.ntaps;
  for (int j = 0; j < 3; ++j) p.tap_off[j] = L.tap_off[j];
  p.M = rm.B * rm.R; p.R = rm.R; p.offs = rm.offs; p.step = rm.step;
  p.wp = L.wp; p.bias = L.bias; p.g1 = L.g1; p.b1 = L.b1; p.g2 = L.g2; p.b2 = L.b2; p.cout = L.cout;
  p.out = out.p; p.out_bstride = out.bstride; p.out_row0 = out.row0; p.out_stride = out.stride;
  p.out_tmul = out_tmul; p.out_tadd = out_tadd; p.out_zero_to = out_zero_to;
  if (out2) { p.out2 = out2->p; p.out2_bstride = out2->bstride; p.out2_row0 = out2->row0; p.out2_stride = out2->stride; }
  p.act = L.act;
  const int kid = L.shape.epi * 10000 + L.shape.nt * 100 + L.shape.nw;
  const bool prof = (c->prof_id == kid);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, st)); }
  // Row split: exact rounds of 32-row items on hconv_kernel; what is left over
  //   * of a highway layer with three taps, when three times as many workgroups still fit in two rounds: 32-row items x TAPS + a finishing pass
  //     (hconv_kernel.h: RAW) -- a third of an item's time per round instead of a whole one;
  //   * of the 4T-resolution k = 1 layers, when it is at most 0.6 of a round: 16-row items (hconv16_kernel.h).
  int tiles32 = (p.M + 31) / 32, m_tail = p.M;
  // The opt-in split-bf16 form (dctts_set_split_bf16): whole 32-row items for every row, contraction on the bf16 matrix pipe.  `bf` is set by the network drivers
  // for TextEnc / SSRN only; the layer must carry the packing.
  if (c->bf_now && L.wpb && !rm.step && ((L.cout == 1025 && L.wxcol) || (L.shape.nw == 8 && (L.shape.nt == 2 || L.shape.nt == 4 || L.shape.nt == 8)))) {
    p.wp = L.wpb; p.wx = (L.cout == 1025) ? L.wxcol : nullptr;
    if (L.shape.epi == EPI_HC && L.shape.nt == 8) {
      // the two column halves' raw output: SSRN-group scratch (tail_ws) -- unless the caller is TextEnc (a config whose 2d-wide highway layers have 64 tiles), whose
      // group owns cols_ws: TextEnc of the next batch may run beside SSRN of the previous one on another stream
      float** buf = c->bf_te ? &c->cols_ws : &c->tail_ws; size_t* fl = c->bf_te ? &c->cols_ws_floats : &c->tail_ws_floats;
      const size_t need = (size_t)p.M * 2 * L.cout;
      if (need > *fl) CHK(grow_scratch(c, buf, fl, need));
      p.m_base = 0; p.raw_out = *buf; p.raw_ld = 2 * L.cout;
    }
    HIPCHK(launch_hconv_bf16(L.shape, p, st));
    if (prof) { HIPCHK(hipEventRecord(e1, st)); c->prof_ev.emplace_back(e0, e1); c->prof_cnt.push_back(1); c->prof_rows += p.M; }
    return 0;
  }
  // Column split (round 4): TextEnc's 512-channel highway layers run as quarter-column items, three or four of them to a CU at once, + the finishing
  // pass.  At B = 32 their 180 32-row items fill 70 % of the CUs and a layer takes a whole item's time; as quarters it takes 3/4 of it.  ALWAYS, whatever the
  // batch: which form a row takes must not depend on the batch it is decoded in (K and V feed the attention; Text2Mel's outputs are bitwise equal across batch
  // compositions: tests/test_gpu_parity.py), and at sizes where the items fill the rounds the split costs only the finishing pass (~5 %).
  if (L.col_split && L.shape.epi == EPI_HC && !gather && !rm.step && L.shape.nt == 4 && L.shape.nw == 8 && L.cout == 512) {
    const int raw_ld = 2 * L.cout;
    const size_t need = (size_t)p.M * raw_ld;
    if (need > c->cols_ws_floats) CHK(grow_scratch(c, &c->cols_ws, &c->cols_ws_floats, need));
    p.m_base = 0; p.raw_out = c->cols_ws; p.raw_ld = raw_ld;
    HIPCHK(launch_hconv_cols(L.shape, p, st));
    if (prof) { HIPCHK(hipEventRecord(e1, st)); c->prof_ev.emplace_back(e0, e1); c->prof_cnt.push_back(1); c->prof_rows += p.M; }
    return 0;
  }
  const int full = (tiles32 / c->n_cu) * c->n_cu, left = tiles32 - full;
  const bool hc3 = L.tap_tail && L.shape.epi == EPI_HC && L.ntaps == 3 && !gather && (L.shape.nt == 2 || L.shape.nt == 4 || L.shape.nt == 8) && (L.cout % 256) == 0 && L.cout <= 1024;
  // ... and of the 4T-resolution k = 1 layers of SSRN (the ones that carry the 16-row packing): thirds of K instead of taps, same finishing idea
  const bool c3 = L.wp16r && L.shape.epi == EPI_C && L.ntaps == 1 && !gather && !L.deconv_phase && L.cout <= 1280 &&
                  ((L.shape.nt == 4 && L.shape.nw == 8) || (L.shape.nt == 3 && L.shape.nw == 11));
  const bool tap_tail = (hc3 || c3) && left > 0 && 3 * left <= 2 * c->n_cu && !rm.step;      // (not in decode mode 0's captured launches: the partial-sum buffer is allocated on demand)
  if (tap_tail) { tiles32 = full; m_tail = full * 32; }
  else if (L.wp16r && left * 10 <= c->n_cu * 6) { tiles32 = full; m_tail = full * 32; }
  if (L.wpx) { p.wp = L.wpx; p.wx = L.wxcol; }      // (the main launch only: the row tail below keeps the 33-tile packing)
  HIPCHK(launch_hconv(L.shape, p, st, tiles32));
  p.wp = L.wp; p.wx = nullptr;
  if (prof) { HIPCHK(hipEventRecord(e1, st)); c->prof_ev.emplace_back(e0, e1); c->prof_cnt.push_back(1); c->prof_rows += m_tail < p.M ? m_tail : p.M; }
  if (m_tail < p.M && tap_tail) {
    // the row tail of a big highway layer: 32-row items x taps + a finishing pass (hconv_kernel.h: RAW) instead of 16-row items
    const int raw_ld = (L.shape.epi == EPI_HC) ? 2 * L.cout : round_up(L.cout, 32);
    const size_t need = (size_t)3 * (p.M - m_tail) * raw_ld;
    if (need > c->tail_ws_floats) CHK(grow_scratch(c, &c->tail_ws, &c->tail_ws_floats, need));
    p.m_base = m_tail; p.raw_out = c->tail_ws; p.raw_ld = raw_ld;
    const bool prof16 = (c->prof_id == 50000 + L.shape16.epi * 10000 + L.shape16.nt * 100 + L.shape16.nw);      // (the tail of the same layer, whatever its form)
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (prof16) { HIPCHK(hipEventCreate(&t0)); HIPCHK(hipEventCreate(&t1)); HIPCHK(hipEventRecord(t0, st)); }
    HIPCHK(launch_hconv_tail(L.shape, p, st));
    if (prof16) { HIPCHK(hipEventRecord(t1, st)); c->prof_ev.emplace_back(t0, t1); c->prof_cnt.push_back(1); c->prof_rows += p.M - m_tail; }
  } else if (m_tail < p.M) {
    p.wp = L.wp16r;
    const bool prof16 = (c->prof_id == 50000 + L.shape16.epi * 10000 + L.shape16.nt * 100 + L.shape16.nw);      // the 16-row tail launch of the same layer
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (prof16) { HIPCHK(hipEventCreate(&t0)); HIPCHK(hipEventCreate(&t1)); HIPCHK(hipEventRecord(t0, st)); }
    HIPCHK(launch_hconv16(L.shape16, p, m_tail, st));
    if (prof16) { HIPCHK(hipEventRecord(t1, st)); c->prof_ev.emplace_back(t0, t1); c->prof_cnt.push_back(1); c->prof_rows += p.M - m_tail; }
  }
  return 0;
}

static int check_ready(dctts_ctx* c, const DevGuard& g) {
  if (!c) return fail(DCTTS_ERR_ARG, "null ctx");
  if (!c->finalized) return fail(DCTTS_ERR_WEIGHTS, "weights not finalized");
  if (!g.ok) return fail(DCTTS_ERR_HIP, "hipSetDevice");
  return 0;
}

static std::string geom(const char* tag, int a, int b, int cdim = 0) {
  char s[96]; snprintf(s, 96, "%s:%d:%d:%d", tag, a, b, cdim); return s;
}

static void drop_ws_prefix(dctts_ctx* c, const std::string& prefix) {      // every geometry of `prefix` ("dbg." -> the pools "dbg@...")
  const std::string pp = prefix.substr(0, prefix.size() - 1) + "@";
  for (auto it = c->ws.begin(); it != c->ws.end();) {
    if (it->first.compare(0, pp.size(), pp) == 0) it = c->ws.erase(it);
    else ++it;
  }
  for (auto pit = c->wsarena.begin(); pit != c->wsarena.end();) {
    if (pit->first.compare(0, pp.size(), pp) == 0) { for (Arena& a : pit->second) (void)hipFree(a.base); pit = c->wsarena.erase(pit); }
    else ++pit;
  }
}

// ---- use groups (dctts_ctx::grp): order this call behind the last call of the group that came from another stream; publish this call's end
static int grp_acquire(dctts_ctx* c, int g, hipStream_t st) {
  dctts_ctx::UseGroup& u = c->grp[g];
  // (always, also when the group's last call came from a stream with the same handle: a destroyed stream's handle can be given to a new stream, and a wait for an
  //  event of the waiting stream itself costs nothing on the device)
  if (u.used) HIPCHK(hipStreamWaitEvent(st, u.done, 0));
  return 0;
}
static int grp_release(dctts_ctx* c, int g, hipStream_t st) {
  dctts_ctx::UseGroup& u = c->grp[g];
  if (!u.done) HIPCHK(hipEventCreateWithFlags(&u.done, hipEventDisableTiming));
  HIPCHK(hipEventRecord(u.done, st));
  u.last = st; u.used = true;
  return 0;
}

static size_t ws_cached_bytes(const dctts_ctx* c) {
  size_t n = c->graveyard_bytes + (c->tail_ws_floats + c->cols_ws_floats) * sizeof(float);
  for (auto& kv : c->wsarena) for (const Arena& a : kv.second) n += a.size;
  for (auto& kv : c->tabcache) n += kv.second.bytes;
  return n;
}
static void drop_decode_tables(dctts_ctx* c);
// Called at the top of every entry point that uses workspaces: when the cached geometries have outgrown ws_limit, ALL of them are dropped behind one device
// synchronisation (the only place left where a call waits for the GPU because of a shape change; dctts_set_workspace_limit).
static int ws_trim(dctts_ctx* c, bool force = false) {
  if (!force && ws_cached_bytes(c) <= c->ws_limit) return 0;
#ifdef DCTTS_DEBUG_LOG
  fprintf(stderr, "[dctts] ws_trim: cached %.1f MB, limit %.1f MB, force %d\n", ws_cached_bytes(c) / 1e6, c->ws_limit / 1e6, (int)force);
#endif
  HIPCHK(hipDeviceSynchronize());
  drop_decode_tables(c);
  free_ws(c);
  for (void* p : c->graveyard) (void)hipFree(p);
  c->graveyard.clear(); c->graveyard_bytes = 0;
  if (c->tail_ws) { (void)hipFree(c->tail_ws); c->tail_ws = nullptr; c->tail_ws_floats = 0; }
  if (c->cols_ws) { (void)hipFree(c->cols_ws); c->cols_ws = nullptr; c->cols_ws_floats = 0; }
  return 0;
}

// An entry point's body, run once more behind a forced trim when a cache allocation ran out of device memory (the failed attempt's launches have completed by
// then: ws_trim synchronises the device before it frees anything; its use groups published their events on the way out: GroupGuard).
template <typename F>
static int oom_retry(dctts_ctx* c, F&& body) {
  g_oom = false;
  int rc = body();
#ifdef DCTTS_DEBUG_LOG
  if (rc != 0) fprintf(stderr, "[dctts] entry point failed rc=%d oom=%d: %s\n", rc, (int)g_oom, g_err.c_str());
#endif
  if (rc != 0 && g_oom) {
    g_oom = false;
    const std::string first = g_err;
    if (ws_trim(c, true) != 0) return fail(DCTTS_ERR_HIP, first);
    rc = body();
  }
  return rc;
}
// A call's hold on a use group: acquire() orders it behind the group's last call, and the group's completion event is recorded on EVERY way out once anything
// may have been enqueued -- an early error return must not leave launches on shared scratch that the next call of the group does not wait for.
struct GroupGuard {
  dctts_ctx* c; int g; hipStream_t st; bool held = false;
  GroupGuard(dctts_ctx* c_, int g_, hipStream_t st_) : c(c_), g(g_), st(st_) {}
  int acquire() { const int rc = grp_acquire(c, g, st); held = (rc == 0); return rc; }
  int release() { held = false; return grp_release(c, g, st); }
  ~GroupGuard() { if (held) (void)grp_release(c, g, st); }
};

extern "C" int dctts_set_split_bf16(dctts_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 2) return fail(DCTTS_ERR_ARG, "split-bf16 mode: 0 (exact fp32), 1 (SSRN), 2 (SSRN + TextEnc)");
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  if (!c->finalized) { c->bf16_packed = mode; c->bf16_mode = mode; return 0; }      // before dctts_weights_finalize: the level that gets packed (and runs)
  if (mode > c->bf16_packed) return fail(DCTTS_ERR_STATE, "split-bf16: this level was not requested before dctts_weights_finalize (the bf16 weight packing does not exist)");
  c->bf16_mode = mode;
  return 0;
}

extern "C" int dctts_set_workspace_limit(dctts_ctx* c, size_t bytes) {
  if (!c) return fail(DCTTS_ERR_ARG, "null ctx");
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->ws_limit = bytes;
  return 0;
}

// ------------------------------------------------------------------------------------------------ TextEnc
static int textenc_into(dctts_ctx* c, const int32_t* L, int B, int N, View* kv_out, hipStream_t st) {
  const int D2 = 2 * c->cfg.d;
  ws_select(c, "te.", geom("te", B, N), st);
  View a, b, kv;
  CHK(ws_view(c, "te.a", B, PAD + N + PAD, PAD, D2, &a));
  CHK(ws_view(c, "te.b", B, PAD + N + PAD, PAD, D2, &b));
  CHK(ws_view(c, "te.kv", B, N, 0, D2, &kv));
  const RowMap rm{B, N, nullptr, nullptr};
  const View tab{c->embed, 0, 0, c->cfg.e};
  const size_t nl = c->textenc.size();
  struct BfScope { dctts_ctx* c; BfScope(dctts_ctx* c_, bool on) : c(c_) { c->bf_now = on; c->bf_te = on; } ~BfScope() { c->bf_now = false; c->bf_te = false; } } bf_scope(c, c->bf16_mode >= 2);
  CHK(run_conv(c, c->textenc[0], tab, (const int*)L, a, rm, st));          // embed + C_2
  View cur = a, nxt = b;
  for (size_t i = 1; i < nl; ++i) {
    const View& o = (i + 1 == nl) ? kv : nxt;
    CHK(run_conv(c, c->textenc[i], cur, nullptr, o, rm, st));
    View t = cur; cur = nxt; nxt = t;
  }
  *kv_out = kv;
  return 0;
}

extern "C" int dctts_textenc_fwd(dctts_ctx* c, const int32_t* L, int B, int N, float* K, float* V, void* stream) {
  DevGuard dev_guard(c);
  CHK(check_ready(c, dev_guard));
  if (!L || !K || !V || B <= 0 || N <= 0) return fail(DCTTS_ERR_ARG, "textenc: bad argument");
  hipStream_t st = (hipStream_t)stream;
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  return oom_retry(c, [&]() -> int {
    CHK(ws_trim(c));
    GroupGuard gte(c, dctts_ctx::GRP_TE, st);
    CHK(gte.acquire());
    View kv;
    CHK(textenc_into(c, L, B, N, &kv, st));
    const int d = c->cfg.d;
    HIPCHK(hipMemcpy2DAsync(K, d * sizeof(float), kv.p, 2 * d * sizeof(float), d * sizeof(float), (size_t)B * N, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpy2DAsync(V, d * sizeof(float), kv.p + d, 2 * d * sizeof(float), d * sizeof(float), (size_t)B * N, hipMemcpyDeviceToDevice, st));
    return gte.release();
  });
}

// ------------------------------------------------------------------------------------------------ AudioEnc / AudioDec (full sequence)
static int t2m_ws(dctts_ctx* c, int B, int T, View* a, View* b, hipStream_t st) {
  ws_select(c, "t2m.", geom("t2m", B, T), st);
  CHK(ws_view(c, "t2m.a", B, PAD + T, PAD, c->cfg.d, a));
  CHK(ws_view(c, "t2m.b", B, PAD + T, PAD, c->cfg.d, b));
  return 0;
}

extern "C" int dctts_audioenc_fwd(dctts_ctx* c, const float* S, int B, int T, float* Q, void* stream) {
  DevGuard dev_guard(c);
  CHK(check_ready(c, dev_guard));
  if (!S || !Q || B <= 0 || T <= 0) return fail(DCTTS_ERR_ARG, "audioenc: bad argument");
  hipStream_t st = (hipStream_t)stream;
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  return oom_retry(c, [&]() -> int {
    CHK(ws_trim(c));
    GroupGuard gt(c, dctts_ctx::GRP_T2M, st);
    CHK(gt.acquire());
    View a, b; CHK(t2m_ws(c, B, T, &a, &b, st));
    const RowMap rm{B, T, nullptr, nullptr};
    const View vin{const_cast<float*>(S), T, 0, c->cfg.n_mels}, vout{Q, T, 0, c->cfg.d};
    const size_t nl = c->audioenc.size();
    View cur = vin, nxt = a, other = b;
    for (size_t i = 0; i < nl; ++i) {
      const View& o = (i + 1 == nl) ? vout : nxt;
      CHK(run_conv(c, c->audioenc[i], cur, nullptr, o, rm, st));
      cur = nxt; View t = nxt; nxt = other; other = t;
    }
    return gt.release();
  });
}

extern "C" int dctts_audiodec_fwd(dctts_ctx* c, const float* R, int B, int T, float* logits, float* Y, void* stream) {
  DevGuard dev_guard(c);
  CHK(check_ready(c, dev_guard));
  if (!R || !Y || B <= 0 || T <= 0) return fail(DCTTS_ERR_ARG, "audiodec: bad argument");
  hipStream_t st = (hipStream_t)stream;
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  return oom_retry(c, [&]() -> int {
    CHK(ws_trim(c));
    GroupGuard gt(c, dctts_ctx::GRP_T2M, st);
    CHK(gt.acquire());
    View a, b; CHK(t2m_ws(c, B, T, &a, &b, st));
    const RowMap rm{B, T, nullptr, nullptr};
    const View vin{const_cast<float*>(R), T, 0, 2 * c->cfg.d}, vout{Y, T, 0, c->cfg.n_mels}, vlog{logits, T, 0, c->cfg.n_mels};
    const size_t nl = c->audiodec.size();
    View cur = vin, nxt = a, other = b;
    for (size_t i = 0; i < nl; ++i) {
      const bool last = (i + 1 == nl);
      CHK(run_conv(c, c->audiodec[i], cur, nullptr, last ? vout : nxt, rm, st, 0, (last && logits) ? &vlog : nullptr));
      cur = nxt; View t = nxt; nxt = other; other = t;
    }
    return gt.release();
  });
}

// ------------------------------------------------------------------------------------------------ Attention (full)
extern "C" int dctts_attention_fwd(dctts_ctx* c, const float* Q, const float* K, const float* V, int B, int T, int N,
                                   int monotonic, const int32_t* prev_max, float* R, float* alignments,
                                   int64_t* max_attentions, void* stream) {
  DevGuard dev_guard(c);
  CHK(check_ready(c, dev_guard));
  if (!Q || !K || !V || !R || B <= 0 || T <= 0 || N <= 0) return fail(DCTTS_ERR_ARG, "attention: bad argument");
  if (monotonic && !prev_max) return fail(DCTTS_ERR_ARG, "attention: monotonic mode needs prev_max_attentions");
  if (monotonic && N != c->cfg.max_N) return fail(DCTTS_ERR_ARG, "attention: monotonic mask is built from hp.max_N (networks.py:142); N must equal it");
  const int d = c->cfg.d;
  AttnFullParams p;
  p.Q = Q; p.q_stride = d; p.q_bstride = T; p.K = K; p.k_stride = d; p.k_bstride = N; p.V = V; p.v_stride = d; p.v_bstride = N;
  p.T = T; p.N = N; p.d = d; p.monotonic = monotonic; p.prev_max = prev_max; p.win = c->cfg.attention_win_size;
  p.R = R; p.align = alignments; p.maxatt = (long long*)max_attentions;
  hipLaunchKernelGGL(attention_full_kernel, dim3(T, B), dim3(256), (d + N) * sizeof(float), (hipStream_t)stream, p);
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ SSRN
static View sub_batch(const View& v, int b0) { View r = v; r.p += (long)b0 * v.bstride * v.stride; return r; }

// The layers of networks.py:214-292 over utterances [b0, b0 + Bs) of a batch whose buffers are `ws` (utterance-major: a sub-batch is a pointer offset).
static int ssrn_layers(dctts_ctx* c, const View* ws, const View& vin0, const View& vz0, const View* vlog0, int b0, int Bs, int T, hipStream_t st) {
  const int Fp = round_up(c->cfg.n_linear, 32);
  View v[10];
  for (int k = 0; k < 10; ++k) v[k] = sub_batch(ws[k], b0);
  const View &s1a = v[0], &s1b = v[1], &s2a = v[2], &s2b = v[3], &s4a = v[4], &s4b = v[5], &w4a = v[6], &w4b = v[7], &z4a = v[8], &z4b = v[9];
  const View vin = sub_batch(vin0, b0), vz = sub_batch(vz0, b0);
  View vlog; if (vlog0) vlog = sub_batch(*vlog0, b0);
  const RowMap r1{Bs, T, nullptr, nullptr}, r2{Bs, 2 * T, nullptr, nullptr}, r4{Bs, 4 * T, nullptr, nullptr};
  const std::vector<DevLayer>& S = c->ssrn;
  size_t i = 0;
  struct BfScope { dctts_ctx* c; BfScope(dctts_ctx* c_, bool on) : c(c_) { c->bf_now = on; } ~BfScope() { c->bf_now = false; } } bf_scope(c, c->bf16_mode >= 1);
  CHK(run_conv(c, S[i++], vin, nullptr, s1a, r1, st));            // C_1
  CHK(run_conv(c, S[i++], s1a, nullptr, s1b, r1, st));            // HC_2
  CHK(run_conv(c, S[i++], s1b, nullptr, s1a, r1, st));            // HC_3
  CHK(run_conv(c, S[i++], s1a, nullptr, s2a, r1, st, 0, nullptr, 2, 0));   // D_4 even rows
  CHK(run_conv(c, S[i++], s1a, nullptr, s2a, r1, st, 0, nullptr, 2, 1));   // D_4 odd rows
  CHK(run_conv(c, S[i++], s2a, nullptr, s2b, r2, st));            // HC_5
  CHK(run_conv(c, S[i++], s2b, nullptr, s2a, r2, st));            // HC_6
  CHK(run_conv(c, S[i++], s2a, nullptr, s4a, r2, st, 0, nullptr, 2, 0));   // D_7
  CHK(run_conv(c, S[i++], s2a, nullptr, s4a, r2, st, 0, nullptr, 2, 1));
  CHK(run_conv(c, S[i++], s4a, nullptr, s4b, r4, st));            // HC_8
  CHK(run_conv(c, S[i++], s4b, nullptr, s4a, r4, st));            // HC_9
  CHK(run_conv(c, S[i++], s4a, nullptr, w4a, r4, st));            // C_10
  CHK(run_conv(c, S[i++], w4a, nullptr, w4b, r4, st));            // HC_11
  CHK(run_conv(c, S[i++], w4b, nullptr, w4a, r4, st));            // HC_12
  CHK(run_conv(c, S[i++], w4a, nullptr, z4a, r4, st, Fp));        // C_13 (pad columns written as 0)
  CHK(run_conv(c, S[i++], z4a, nullptr, z4b, r4, st, Fp));        // C_14
  CHK(run_conv(c, S[i++], z4b, nullptr, z4a, r4, st, Fp));        // C_15
  CHK(run_conv(c, S[i++], z4a, nullptr, vz, r4, st, 0, vlog0 ? &vlog : nullptr));   // C_16 + sigmoid
  return 0;
}

static int ssrn_impl(dctts_ctx* c, const float* Y, int B, int T, float* logits, float* Z, hipStream_t st) {
  GroupGuard gs(c, dctts_ctx::GRP_SSRN, st);
  CHK(gs.acquire());
  const int cc = c->cfg.c, F = c->cfg.n_linear, Fp = round_up(F, 32);
  ws_select(c, "ssrn.", geom("ssrn", B, T), st);
  View ws[10];
  CHK(ws_view(c, "ssrn.s1a", B, PAD + T + PAD, PAD, cc, &ws[0])); CHK(ws_view(c, "ssrn.s1b", B, PAD + T + PAD, PAD, cc, &ws[1]));
  CHK(ws_view(c, "ssrn.s2a", B, PAD + 2 * T + PAD, PAD, cc, &ws[2])); CHK(ws_view(c, "ssrn.s2b", B, PAD + 2 * T + PAD, PAD, cc, &ws[3]));
  CHK(ws_view(c, "ssrn.s4a", B, PAD + 4 * T + PAD, PAD, cc, &ws[4])); CHK(ws_view(c, "ssrn.s4b", B, PAD + 4 * T + PAD, PAD, cc, &ws[5]));
  CHK(ws_view(c, "ssrn.w4a", B, PAD + 4 * T + PAD, PAD, 2 * cc, &ws[6])); CHK(ws_view(c, "ssrn.w4b", B, PAD + 4 * T + PAD, PAD, 2 * cc, &ws[7]));
  CHK(ws_view(c, "ssrn.z4a", B, 4 * T, 0, Fp, &ws[8])); CHK(ws_view(c, "ssrn.z4b", B, 4 * T, 0, Fp, &ws[9]));
  const View vin{const_cast<float*>(Y), T, 0, c->cfg.n_mels}, vz{Z, 4L * T, 0, F}, vlog{logits, 4L * T, 0, F};
  CHK(ssrn_layers(c, ws, vin, vz, logits ? &vlog : nullptr, 0, B, T, st));
  return gs.release();
}

extern "C" int dctts_ssrn_fwd(dctts_ctx* c, const float* Y, int B, int T, float* logits, float* Z, void* stream) {
  DevGuard dev_guard(c);
  CHK(check_ready(c, dev_guard));
  if (!Y || !Z || B <= 0 || T <= 0) return fail(DCTTS_ERR_ARG, "ssrn: bad argument");
  hipStream_t st = (hipStream_t)stream;
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  return oom_retry(c, [&]() -> int { CHK(ws_trim(c)); return ssrn_impl(c, Y, B, T, logits, Z, st); });
}

// ------------------------------------------------------------------------------------------------ decode (synthesize.py:45-54)
#include "decode_host.h"

// ------------------------------------------------------------------------------------------------ per-layer test hook
// Runs ONE device layer of a network on a caller tensor X (B,T,Cin) -> out (B,T',Cout); T' = 2T for a
// transposed conv (both phases run).  Used by tests/ to compare every kernel shape class with the oracle.
extern "C" int dctts_debug_layer(dctts_ctx* c, const char* net, int index, const float* X, int B, int T, float* out, void* stream) {
  DevGuard dev_guard(c);
  CHK(check_ready(c, dev_guard));
  if (!net || !X || !out || B <= 0 || T <= 0) return fail(DCTTS_ERR_ARG, "debug_layer: bad argument");
  hipStream_t st = (hipStream_t)stream;
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  HIPCHK(hipDeviceSynchronize());           // a test hook: its scratch ("dbg.") is re-created per call, and run_conv may touch tail_ws / cols_ws of any group
  const std::string n(net);
  const std::vector<DevLayer>* V = n == "textenc" ? &c->textenc : n == "audioenc" ? &c->audioenc : n == "audiodec" ? &c->audiodec : n == "ssrn" ? &c->ssrn : nullptr;
  if (!V || index < 0 || index >= (int)V->size()) return fail(DCTTS_ERR_ARG, "debug_layer: unknown net / index");
  const DevLayer& L = (*V)[index];
  const RowMap rm{B, T, nullptr, nullptr};
  struct BfScope { dctts_ctx* c; BfScope(dctts_ctx* c_, bool on) : c(c_) { c->bf_now = on; } ~BfScope() { c->bf_now = false; } } bf_scope(c, (n == "ssrn" && c->bf16_mode >= 1) || (n == "textenc" && c->bf16_mode >= 2));
  struct TeScope { dctts_ctx* c; TeScope(dctts_ctx* c_, bool on) : c(c_) { c->bf_te = on; } ~TeScope() { c->bf_te = false; } } te_scope(c, n == "textenc" && c->bf16_mode >= 2);
  if (n == "textenc" && index == 0) {      // embed + C_2: X is really int32 ids (B,T)
    const View tab{c->embed, 0, 0, c->cfg.e}, vo{out, T, 0, L.cout};
    return run_conv(c, L, tab, (const int*)X, vo, rm, st);
  }
  const int cin_real = (L.cin == round_up(c->cfg.n_linear, 32)) ? c->cfg.n_linear : L.cin;
  drop_ws_prefix(c, "dbg.");
  ws_select(c, "dbg.", "", st);
  View vi;
  CHK(ws_view(c, "dbg.in", B, PAD + T + PAD, PAD, L.cin, &vi));
  for (int b = 0; b < B; ++b)
    HIPCHK(hipMemcpy2DAsync(vi.p + ((long)b * vi.bstride + PAD) * L.cin, (size_t)L.cin * sizeof(float),
                            X + (long)b * T * cin_real, (size_t)cin_real * sizeof(float), (size_t)cin_real * sizeof(float), T,
                            hipMemcpyDeviceToDevice, st));
  if (L.deconv_phase) {
    if (L.phase != 0 || index + 1 >= (int)V->size()) return fail(DCTTS_ERR_ARG, "debug_layer: give the even phase of a D layer");
    const View vo{out, 2L * T, 0, L.cout};
    CHK(run_conv(c, L, vi, nullptr, vo, rm, st, 0, nullptr, 2, 0));
    return run_conv(c, (*V)[index + 1], vi, nullptr, vo, rm, st, 0, nullptr, 2, 1);
  }
  const View vo{out, T, 0, L.cout};
  return run_conv(c, L, vi, nullptr, vo, rm, st);
}


// Calibration aid for the HBM counters (FETCH_SIZE / WRITE_SIZE): a float4 grid-stride copy of known size.
__global__ void __launch_bounds__(256) calib_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int dctts_debug_seed_prev_max(dctts_ctx* c, const int32_t* prev_max, int B) {
  if (!c || !prev_max || B <= 0) return fail(DCTTS_ERR_ARG, "seed_prev_max: bad argument");
  for (int b = 0; b < B; ++b) if (prev_max[b] < 0 || prev_max[b] >= c->cfg.max_N) return fail(DCTTS_ERR_ARG, "seed_prev_max: values must lie in [0, max_N)");
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  c->init_pm.assign(prev_max, prev_max + B);
  return 0;
}

extern "C" int dctts_debug_set_trace(dctts_ctx* c, int frame, const char* file) {
  if (!c) return fail(DCTTS_ERR_ARG, "null ctx");
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  c->trace_frame = frame < 0 ? -1 : frame;                  // (the decode's tables are keyed by it: the next decode builds the ones that carry the stamp buffer)
  c->trace_file = (frame >= 0 && file) ? file : "";
  return 0;
}

__global__ void xcd_census_kernel(int* __restrict__ out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (int)xg_xcc_id();
}
extern "C" int dctts_debug_xcd_census(dctts_ctx* c, int32_t* xcc128, int32_t* n_cu, void* stream) {
  if (!c || !xcc128) return fail(DCTTS_ERR_ARG, "xcd_census: null argument");
  DevGuard dev_guard(c);
  if (!dev_guard.ok) return fail(DCTTS_ERR_HIP, "hipSetDevice");
  int* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, 128 * sizeof(int)));
  hipLaunchKernelGGL(xcd_census_kernel, dim3(128), dim3(512), 0, (hipStream_t)stream, d);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(xcc128, d, 128 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(d);
  HIPCHK(e);
  if (n_cu) *n_cu = c->n_cu;
  return 0;
}

extern "C" int dctts_debug_copy(const float* src, float* dst, size_t nfloats, void* stream) {
  hipLaunchKernelGGL(calib_copy_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, nfloats / 4);
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ profiling aid
extern "C" int dctts_prof_enable(dctts_ctx* c, int kernel_id) {
  if (!c) return fail(DCTTS_ERR_ARG, "null ctx");
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  if (kernel_id >= 0) c->prof_rows = 0;
  c->prof_id = kernel_id;
  return 0;
}

extern "C" int dctts_prof_collect(dctts_ctx* c, int* launches, double* total_ms) {
  if (!c || !launches || !total_ms) return fail(DCTTS_ERR_ARG, "null argument");
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  double tot = 0; int n = 0;
  for (size_t i = 0; i < c->prof_ev.size(); ++i) {
    auto& e = c->prof_ev[i];
    HIPCHK(hipEventSynchronize(e.second));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e.first, e.second));
    tot += ms; n += (i < c->prof_cnt.size()) ? c->prof_cnt[i] : 1;
    (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
  }
  c->prof_ev.clear(); c->prof_cnt.clear();
  *launches = n; *total_ms = tot;
  return 0;
}

extern "C" int dctts_prof_rows(dctts_ctx* c, long long* rows) {
  if (!c || !rows) return fail(DCTTS_ERR_ARG, "null argument");
  std::lock_guard<std::recursive_mutex> lk_(c->mu);
  *rows = c->prof_rows;
  return 0;
}
