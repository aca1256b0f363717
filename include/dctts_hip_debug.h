/* dctts_hip_debug.h -- test and measurement hooks of libdctts_hip.so.  NOT part of the drop-in surface (include/dctts_hip.h):
 * nothing a consumer of the synthesis path needs lives here.  Used by tests/ (per-layer parity), bench.py (kernel timing for the
 * roofline objects) and tools/ (PMC calibration).
 * Environment variables the library reads ONCE, in dctts_create (measurement / A-B only; see tools/README.md): DCTTS_*.
 * DCTTS_TRACE=<frame> + DCTTS_TRACE_FILE=<path> make a decode write in-kernel wall-clock stamps of that frame's chain launches. */
#ifndef DCTTS_HIP_DEBUG_H
#define DCTTS_HIP_DEBUG_H
#include "dctts_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: run ONE device layer of a network ("textenc" | "audioenc" | "audiodec" | "ssrn") on a caller
 * tensor X (B,T,Cin) -> out (B,T',Cout) (T' = 2T for a transposed conv, index = its even phase;
 * "textenc" index 0 = embed + C_2 and takes int32 ids).  Layer order = networks.py source order, with
 * each D layer occupying two consecutive indices.  Synchronises (allocates a scratch copy of X). */
int dctts_debug_layer(dctts_ctx* ctx, const char* net, int index, const float* X, int B, int T, float* out, void* stream);

/* Test hook: the NEXT decode on this context (dctts_text2mel_decode / dctts_synthesize) starts from these prev_max_attentions
 * (host array, B values in [0, max_N)) instead of the reference's zeros (synthesize.py:46), then the seed is dropped.  Random weights
 * never walk the attention to the end of a 180-character text, so this is how tests put the window on keys max_N-3 .. max_N-1 at the
 * production geometry (networks.py:142-147: the window clipped to 2, then 1 keys; the cached V.W / V.W.W tables read at their last rows). */
int dctts_debug_seed_prev_max(dctts_ctx* ctx, const int32_t* prev_max, int B);

/* Test hook: the NEXT decode on this context behaves as if a bounded in-kernel wait had given up (`bits` of the error word, | 64): its outputs are
 * poisoned and the sticky status is raised exactly as for a real time-out; the decode itself runs normally.  bits = 0 withdraws the injection. */
int dctts_debug_inject_decode_error(dctts_ctx* ctx, int bits);

/* Test hook: the context's PERSISTENT team-kernel settings as bits: 1 = xgroup requested, 2 = xcone requested, 4 = still allowed (not switched off by
 * dctts_decode_status).  A checked decode's safe retry (dctts_decode_safe_once) must leave this word exactly as it found it. */
int dctts_debug_team_kernels_state(dctts_ctx* ctx);

/* Measurement hook (bench.py: roofline.frac_gemm_phase): what DCTTS_TRACE / DCTTS_TRACE_FILE do, at run time -- decodes that follow write the in-kernel
 * wall-clock stamps of frame `frame`'s launches (xchain_kernel's two parts, xcone_kernel: phase boundaries, and every wave's unit ends in its first GEMM layer)
 * to `file`; frame < 0 switches it off.  The traced frame's launches are the stamped instantiations; every other frame runs the production kernels. */
int dctts_debug_set_trace(dctts_ctx* ctx, int frame, const char* file);

/* Measurement hook (bench.py: `placement`): where the device puts the workgroups of a 128-block launch -- xcc[b] = the XCD (HW_REG_XCC_ID) block b ran on -- and the
 * compute units the device reports.  The decode's team kernels are fast when blocks b, b + 8, b + 16, ... share an XCD and 128 + 128 workgroups of 512 threads are
 * co-resident (one per CU); they are CORRECT either way (a split team is detected and that decode repeated one launch per layer). */
int dctts_debug_xcd_census(dctts_ctx* ctx, int32_t* xcc128, int32_t* n_cu, void* stream);

/* Calibration aid for the HBM PMC counters: float4 copy of nfloats floats (nfloats % 4 == 0) on `stream`. */
int dctts_debug_copy(const float* src, float* dst, size_t nfloats, void* stream);

/* Measurement aid for bench.py's roofline objects: HIP events are recorded on the launch stream around launches of one
 * kernel while enabled; collect() synchronises those events, returns the number of launches and their summed duration,
 * and clears the list.  kernel_id:
 *   epi*10000 + NT*100 + NW   every launch of hconv_kernel<epi, NT, NW> (epi 0 = C, 1 = HC), e.g. 10808 = SSRN HC_11 / HC_12;
 *   50000 + epi*10000 + NT*100 + NW   the 16-row tail launches hconv16_kernel<epi, NT, NW> of the layers that are split by rows, e.g. 61608 = HC_11 / HC_12's tail;
 *   DCTTS_PROF_XGROUP         xgroup_kernel (a run of newest-row highway layers of the decode chain as one launch), on every 16th frame only;
 *                             prof_rows then counts LAYERS (10 for the AudioEnc run, the only run timed; the AudioDec layers are part of xtail_kernel's launch since round 4);
 *   DCTTS_PROF_XCONE          xcone_kernel (the tail of AudioDec's cone on the side stream), frames >= 100, every 16th; eager decode only (graph mode 0);
 *   DCTTS_PROF_XTAIL          xtail_kernel (merged form: AudioDec's newest-row layers HC_2 .. HC_4, HC_5 .. HC_7 over their cone rows and the seven k = 1 layers around the
 *                             mel frame, the first launch of a chain piece), every 16th frame; prof_rows is not meaningful for it;
 *   DCTTS_PROF_CHAIN_HC       chain3_kernel<LN_HC, HC> (one newest-row highway layer per launch: the form used when DCTTS_XGROUP=0), every 16th frame;
 *   DCTTS_PROF_BULK_GEMM      hbulk_kernel<12> (the cone GEMM of HC_3 when DCTTS_XCONE=0); eager decode only.  Enabling either of the last two
 *                             switches the corresponding team kernel off for the decodes that follow. */
#define DCTTS_PROF_CHAIN_HC 30000
#define DCTTS_PROF_BULK_GEMM 30001
#define DCTTS_PROF_XGROUP 30002
#define DCTTS_PROF_XCONE 30003
#define DCTTS_PROF_XTAIL 30004
int dctts_prof_enable(dctts_ctx* ctx, int kernel_id);
int dctts_prof_collect(dctts_ctx* ctx, int* launches, double* total_ms);
/* Output rows those launches covered, summed since the last prof_enable (a layer's rows may be split between the
 * 32-row kernel -- the one timed here -- and a 16-row tail launch, see csrc/hconv16_kernel.h; decode kernels: rows = utterances
 * (chain) or cone rows (bulk) per launch). */
int dctts_prof_rows(dctts_ctx* ctx, long long* rows);

/* Measurement aid (bench.py roofline of the vocoder): while enabled, HIP events are recorded on the launch stream around
 * every launch of the Griffin-Lim iteration kernel (gl_iter_wave_kernel); collect() synchronises them, returns the number
 * of launches and their summed duration, and clears the list. */
int dctts_vocoder_prof_enable(dctts_vocoder* v, int enable);
int dctts_vocoder_prof_collect(dctts_vocoder* v, int* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* DCTTS_HIP_DEBUG_H */
