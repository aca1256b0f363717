"""bench.py -- mel frames/s + RTF of the DC-TTS synthesis path on MI355X (BASELINE.json metric).

One "step" = one full synthesis of a batch of B=32 utterances on each GPU: TextEnc once, the 210-step
autoregressive Text2Mel decode (exact-parity incremental algorithm), one SSRN pass -> (32, 840, 1025)
linear spectrogram; inputs (character ids) and weights are resident in HBM before the timed region.
Multi-GPU: utterances shard embarrassingly, 32 per GPU, no collective on the data path ("weak").

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

What the JSON line holds besides the contract's fields (everything is measured in this run unless it says otherwise):
  roofline            the kernel that dominates the step by TIME (rocprofv3 --stats: profiles/r06_kernel_stats.md): xcone_kernel, the re-evaluation of
                      AudioDec's dependency cone on the decode's side stream; HIP-event timed on ITS stream inside the timed region (every 16th
                      frame from frame 100 on), fp32-MFMA bound; `frac` = the whole launch, `frac_gemm_phase` = its GEMM layers' own phases (in-kernel stamps);
                      `traffic` = PMC bytes per launch (profiles/r06_pmc_decode.json, separate passes)
  kernels             the same figures for xchain_kernel (the chain's launch; DCTTS_CHAIN_TAIL=6: xtail_kernel and xgroup_kernel), the FLOP-dominant kernel (SSRN HC_11/12 + its tail
                      launch) and SSRN's 1025-column layers (event-timed in untimed extra passes)
  phases / phase_rooflines   TextEnc / decode / SSRN times and their fractions of both roofs (SURVEY 8d algorithmic work)
  placement           per rank: where a 128-block launch lands (XCD census), compute units, whether the rank's timed decodes ran on the team kernels
  fallback            the step with the team kernels off (one launch per layer): what a rank falls back to when a team is not on one XCD / the waits give up
  other_configs       BASELINE configs[1] (decode only), [2] (SSRN only, B=128), [4] (max_T=1000, B=8 = one GPU's share); decode-only at B = 64 / 128;
                      pipelined_depth2[_with_vocoder]: two batches in flight on two streams of one engine (a SECOND line: the headline stays the serial batch)
  cpu_baseline        the reference's loop restated on torch-CPU fp32 (oracle/torch_ref.py) on the host cores, bounded sample; + the numpy
                      oracle and its incremental variant
  host_transfer / gather     PCIe-inclusive figures (never `value`)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense fp32 matrix peak
PEAK_HBM_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PROF_SSRN_HC = 1 * 10000 + 8 * 100 + 8    # hconv_kernel<EPI_HC, NT=8, NW=8>: SSRN HC_11 / HC_12 (C = 1024)
PROF_SSRN_HC_TAIL = 50000 + 1 * 10000 + 16 * 100 + 8     # the row-tail launches of the same layers (tap-split items + finishing pass; the id is the 16-row shape's)
PROF_SSRN_C1025 = 0 * 10000 + 3 * 100 + 11               # SSRN C_13 .. C_16 (1025 columns): the id is their row-tail shape <EPI_C, NT=3, NW=11>; the main launch is the XC form since round 5
PROF_XGROUP = 30002               # include/dctts_hip_debug.h: xgroup_kernel, sampled every 16th frame (prof_rows counts layers)
PROF_XCONE = 30003                # xcone_kernel (eager decode only)
PROF_XTAIL = 30004                # xtail_kernel, sampled every 16th frame


def both_roofs(flop, nbytes, ms):
    """achieved FLOP/s and B/s of `flop` / `nbytes` of ALGORITHMIC work done in `ms`, as fractions of both roofs."""
    tf = flop / (ms * 1e-3) / 1e12
    gb = nbytes / (ms * 1e-3) / 1e9
    return {"tflops": round(tf, 3), "frac_mfma": round(tf / PEAK_F32_MFMA_TFLOPS, 4), "algorithmic_GBps": round(gb, 1),
            "frac_hbm": round(gb / PEAK_HBM_GBPS, 4)}


def cpu_baseline(hp, W):
    """The reference's algorithm restated for the CPU (TensorFlow is not installable here), timed on the GPU box's host cores on a bounded
    sample, as BASELINE.md section 3 prescribes: torch-CPU fp32 with torch.set_num_threads(all cores) -- oracle/torch_ref.py, a few steps of the
    full-recompute loop (synthesize.py:47-54, TextEnc recomputed every step as the reference does) plus one SSRN pass, prorated per mel frame.
    Beside it: the numpy oracle on the same sample (what rounds 1-2 reported), and the oracle's INCREMENTAL variant (oracle/incremental_ref.py:
    the algorithm the HIP path runs), so that the algorithmic and the hardware speed-up can be told apart."""
    import torch as _torch
    from dc_tts_amd.weights import synthetic_text
    from oracle import dctts_ref as O
    from oracle import torch_ref as TR
    from oracle.incremental_ref import incremental_decode_v3
    ncpu = os.cpu_count() or 1
    Bt, STEPS_T = 8, 4                                        # 4 timed loop steps at B = 8 (after one untimed warm-up step) per thread count
    Lt = synthetic_text(hp, B=Bt, seed=99)
    Pt = TR.params(W)
    def torch_leg(threads, steps_t):
        """seconds per mel frame of the full-recompute loop + SSRN at `threads` torch threads (one untimed warm-up step first)"""
        _torch.set_num_threads(threads)
        TR.synthesize(Lt, W, hp, steps=1, run_ssrn=False)
        t0 = time.perf_counter()
        Yt, _, _ = TR.synthesize(Lt, W, hp, steps=steps_t, run_ssrn=False)
        t_step = (time.perf_counter() - t0) / steps_t            # seconds per loop step for Bt utterances
        t0 = time.perf_counter()
        with _torch.no_grad():
            TR.SSRN(_torch.from_numpy(Yt), Pt, hp)
        t_ssrn = (time.perf_counter() - t0) / Bt                 # seconds per utterance
        return t_step / Bt + t_ssrn / hp.max_T                   # one loop step yields one mel frame per utterance
    # BASELINE.md section 3 prescribes all host cores.  The GPU box shows 256 cores but its container is granted 16 (cgroup cpu.max): with
    # set_num_threads(256) the sample runs at 0.03 mel frames/s (profiles/r03_bench_allcores.json: 67 s per loop step, spinning threads), with 64
    # at 17.6, with 16 at 68.  So "all cores" = the cores this process may use; the leg climbs through thread counts up to that and stops as
    # soon as more threads are clearly slower; `value` is the BEST count tried, `cores` says which.
    try:
        navail = len(os.sched_getaffinity(0))
    except Exception:
        navail = ncpu
    quota = None
    try:                                                      # cgroup v2 CPU quota of this container ("max" = none): the GPU box shows 256 cores and grants 16
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(round(int(q) / int(per))))
            navail = min(navail, quota)
    except Exception:
        pass
    tried = {}
    t_leg0 = time.perf_counter()
    for th in sorted({min(16, navail), min(64, navail), navail}):
        if tried and (time.perf_counter() - t_leg0 > 30.0 or tried[max(tried)] > 1.5 * min(tried.values())):
            break
        tried[th] = torch_leg(th, STEPS_T)
    best_th = min(tried, key=tried.get)
    per_frame_t = tried[best_th]
    _torch.set_num_threads(best_th)
    steps_t = STEPS_T
    # ---- the numpy oracle on the same kind of sample
    Bs, steps = 2, 3
    L = synthetic_text(hp, B=Bs, seed=99)
    Y = np.zeros((Bs, hp.max_T, hp.n_mels), np.float32)
    prev = np.zeros((Bs,), np.int32)
    O.text2mel_graph(L, Y, prev, W, hp)                       # warm BLAS threads
    t0 = time.perf_counter()
    for j in range(steps):
        g = O.text2mel_graph(L, Y, prev, W, hp)               # full graph incl. TextEnc, like the reference
        Y[:, j, :] = g["Y"][:, j, :]
        prev = g["max_attentions"][:, j].astype(np.int32)
    t_step = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    O.SSRN(Y[:1], W, hp)
    t_ssrn = time.perf_counter() - t0
    per_frame = t_step / Bs + t_ssrn / hp.max_T
    Ti = 120                                                  # > 85: the full AudioDec cone is re-evaluated in the later steps
    t0 = time.perf_counter()
    incremental_decode_v3(L, W, hp.replace(max_T=Ti), np.float32)
    t_inc = (time.perf_counter() - t0) / (Bs * Ti)            # seconds per mel frame, decode incl. one TextEnc per utterance
    per_frame_inc = t_inc + t_ssrn / hp.max_T
    try:                                                      # threads numpy's BLAS actually runs on (the matmuls are the work)
        from threadpoolctl import threadpool_info
        cores_np = max([int(i.get("num_threads", 1)) for i in threadpool_info() if i.get("user_api") == "blas"] or [1])
    except Exception:
        cores_np = ncpu
    return {"value": 1.0 / per_frame_t, "unit": "mel frames/s", "cores": best_th, "host_cores": ncpu, "usable_cores": navail, "cgroup_cpu_quota_cores": quota, "kind": "port",
            "sample": f"{steps_t} timed steps of the restated synthesize.py loop (full Text2Mel graph incl. TextEnc per step, B={Bt}, N={hp.max_N}, "
                      f"T={hp.max_T}) + 1 SSRN pass (B={Bt}), torch-CPU fp32 (oracle/torch_ref.py) at torch.set_num_threads({best_th}) = the fastest of the "
                      f"thread counts tried, one untimed warm-up step each, prorated per mel frame",
            "rtf": per_frame_t / hp.seconds_per_mel_frame,
            "torch_threads_tried": {str(th): {"value": 1.0 / v, "unit": "mel frames/s", "rtf": v / hp.seconds_per_mel_frame} for th, v in sorted(tried.items())},
            "torch_more_threads_note": "the GPU box's container has a CPU quota of 16 cores beside 256 visible ones: set_num_threads(64) gives 17.6, "
                                       "set_num_threads(256) 0.03 mel frames/s on the same sample (profiles/r03_bench_allcores.json, gpurun_out of round 3)",
            "numpy_variant": {"value": 1.0 / per_frame, "unit": "mel frames/s", "cores": cores_np, "rtf": per_frame / hp.seconds_per_mel_frame,
                              "sample": f"the same loop in numpy fp32 (oracle/dctts_ref.py, B={Bs}, {steps} steps + 1 SSRN pass at B=1; BLAS threads = cores)"},
            "incremental_variant": {"value": 1.0 / per_frame_inc, "unit": "mel frames/s", "rtf": per_frame_inc / hp.seconds_per_mel_frame,
                                    "sample": f"oracle/incremental_ref.incremental_decode_v3 (the HIP path's algorithm in numpy: TextEnc once, "
                                              f"cached AudioEnc, cone re-evaluation), B={Bs}, T={Ti}, + the numpy SSRN pass, prorated per mel frame",
                                    "algorithmic_speedup_over_reference_loop": round(per_frame / per_frame_inc, 1)}}


def vocoder_cpu_baseline(hp, mag1):
    """oracle/vocoder_ref.spectrogram2wav (numpy restatement of utils.py:67-114; librosa is not installable) on ONE utterance."""
    from oracle import vocoder_ref as V
    t0 = time.perf_counter()
    V.spectrogram2wav(mag1, hp, np.float32)
    dt = time.perf_counter() - t0
    return {"value": (mag1.shape[0] / hp.r) / dt, "unit": "mel frames/s", "cores": 1, "kind": "port",
            "sample": f"1 utterance ({mag1.shape[0]} linear frames, n_iter={hp.n_iter}), numpy fp32 FFTs, single thread",
            "seconds_per_utterance": dt}


def vocoder_section(hp, Z, ms_synth, with_cpu):
    """Untimed extra (NOT part of `value`: BASELINE's metric excludes Griffin-Lim): the vocoder tail on the batch
    the timed loop just produced, with the roofline of its dominant kernel."""
    from dc_tts_amd.utils import Vocoder
    B, F, nb = Z.shape
    v = Vocoder(hp)
    v.spectrogram2wav_device(Z); torch.cuda.synchronize()
    reps = 3
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    v.prof_enable(True)
    e0.record()
    for _ in range(reps):
        wav, bounds = v.spectrogram2wav_device(Z)
    e1.record(); torch.cuda.synchronize()
    v.prof_enable(False)
    n_launch, it_ms = v.prof_collect()
    ms = e0.elapsed_time(e1) / reps
    L = hp.hop_length * (F - 1)
    # algorithmic bytes of one gl_iter_wave_kernel launch: the signal in, the magnitudes in, the windowed frames out (fp32)
    alg = 4.0 * B * (L + F * nb + F * hp.win_length)
    avg = it_ms / max(n_launch, 1)
    ach = alg / (avg * 1e-3) / 1e9 if n_launch else None
    audio_s = B * L / hp.sr
    out = {"ms_per_batch": round(ms, 3), "n_iter": hp.n_iter, "rtf": ms * 1e-3 / audio_s,
           "mel_frames_per_s": round(B * (F // hp.r) / (ms * 1e-3), 1),
           "end_to_end_ms_per_batch": round(ms_synth + ms, 3),
           "end_to_end_mel_frames_per_s": round(B * (F // hp.r) / ((ms_synth + ms) * 1e-3), 1),
           "roofline": {"bound": "hbm", "achieved": None if ach is None else round(ach, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": None if ach is None else round(ach / PEAK_HBM_GBPS, 4), "traffic": None,
                        "kernel": "gl_iter_wave_kernel (one Griffin-Lim iteration: window, rfft, phase projection, irfft, window; "
                                  "one wave per frame)", "launches": n_launch, "avg_launch_ms": round(avg, 4),
                        "algorithmic_bytes_per_launch": alg,
                        "note": "VALU-issue bound (~2.5 k wave instructions per frame), not HBM bound: see DESIGN.md"},
           "device_bytes": v.device_bytes()}
    tj = os.path.join(ROOT, "profiles", "r01_vocoder_pmc.json")
    if B == 32 and F == 840 and os.path.exists(tj):           # PMC FETCH_SIZE x2 + WRITE_SIZE per launch, separate rocprofv3 passes
        out["roofline"]["traffic"] = json.load(open(tj))["hbm_bytes_per_launch"]
        out["roofline"]["traffic_unit"] = "bytes/launch (PMC, separate pass of round 1: profiles/r01_vocoder.md; kernel unchanged since)"
    if with_cpu:
        out["cpu_baseline"] = vocoder_cpu_baseline(hp, Z[0].cpu().numpy())
    v.close()
    return out


def xcone_phase_stamps(eng, L, T):
    """In-kernel wall-clock stamps (100 MHz) of ONE side-stream launch of a steady-state frame (dctts_debug_set_trace): the two row phases and, per GEMM
    layer, contraction / barrier / row pass / barrier.  Returns microseconds: {"row_phases", "gemm" (sum of the layers' contraction phases incl. their row tables),
    "row_passes_and_barriers", "launch"} or None."""
    import tempfile
    frame = min(150, T - 2)
    if frame < 100:
        return None
    path = os.path.join(tempfile.gettempdir(), f"dctts_bench_trace_{os.getpid()}.txt")
    try:
        eng.debug_set_trace(frame, path)
        eng.text2mel(L); torch.cuda.synchronize()
        eng.debug_set_trace(-1)
        eng.text2mel(L); torch.cuda.synchronize()             # (back on the production instantiations, tables rebuilt)
        lines = open(path).read().splitlines()
        os.remove(path)
    except Exception as e:                                   # a measurement aid: never fatal
        print(f"[bench] in-kernel stamps unavailable: {e}", file=sys.stderr)
        return None
    for i, ln in enumerate(lines):
        if ln.startswith("# xcone_kernel") and i + 1 < len(lines):
            v = [float(x) for x in lines[i + 1].split("(")[0].split()]
            if len(v) < 8 or (len(v) - 4) % 4:
                return None
            gemm = sum(v[4 + 4 * k] - v[3 + 4 * k] for k in range((len(v) - 4) // 4))
            return {"row_phases": round(v[3], 2), "gemm": round(gemm, 2), "row_passes_and_barriers": round(v[-1] - v[3] - gemm, 2), "launch": round(v[-1], 2),
                    "gemm_layers": (len(v) - 4) // 4, "frame": frame}
    return None


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def cached_synthetic_weights(hp, seed, rank, world):
    """The seeded random-init weights (209.5 MB, ~3 s to draw): rank 0 draws them once per box and leaves them in /tmp, the other ranks (and the
    runs at the next N) load the file instead of drawing the same numbers again."""
    from dc_tts_amd.weights import synthetic_weights
    import torch.distributed as dist
    path = os.path.join("/tmp", f"dctts_synthetic_weights_seed{seed}_N{hp.max_N}.npz")
    def load():
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    if rank == 0 and not os.path.exists(path):
        W = synthetic_weights(hp, seed=seed, perturb=True)
        tmp = path + f".{os.getpid()}.tmp.npz"
        try:
            np.savez(tmp, **W); os.replace(tmp, path)
        except OSError:
            pass                                                   # a read-only /tmp costs nothing but the cache
    else:
        W = None
    if world > 1:
        dist.barrier()
    if W is None:
        try:
            W = load()
        except Exception:
            W = synthetic_weights(hp, seed=seed, perturb=True)
    return W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU (BASELINE: 32)")
    ap.add_argument("--max-T", type=int, default=210)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--graph-mode", type=int, default=0, help="0 eager (default: the side stream is 4 launches per frame since round 3), 1 the side stream's work of a frame as one hipGraph")
    ap.add_argument("--decode-mode", type=int, default=3, help="3 = default (two-stream incremental decode), 0 = fused full-row kernels on one stream (cross-check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vocoder", action="store_true", help="skip the untimed vocoder-tail section (SURVEY 8f-2)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed per-kernel passes and the other BASELINE configs")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL; the driver's multi-GPU runs) or gloo (plumbing test: ranks may share a GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch.distributed as dist
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    host_group = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            host_group = dist.new_group(backend="gloo")       # the host gather of SURVEY 8e (results meet on rank 0's host, not on a GPU)
        else:
            dist.init_process_group(args.dist_backend)
            host_group = dist.group.WORLD

    from dc_tts_amd.engine import Engine
    from dc_tts_amd.hyperparams import hp as hp0
    from dc_tts_amd.sharding import gather_to_host
    from dc_tts_amd.weights import synthetic_text, synthetic_weights

    hp = hp0.replace(max_T=args.max_T)
    B, T = args.batch, hp.max_T
    W = cached_synthetic_weights(hp, 1234, rank, world)
    gm = 0 if args.no_graph else args.graph_mode
    eng = Engine(W, hp, device=local, decode_graph=gm)
    eng.set_decode_mode(args.decode_mode)
    L = torch.from_numpy(synthetic_text(hp, B=B, seed=1234 + rank)).cuda()
    # The caller's stream: a HIGH-PRIORITY HIP stream.  The decode's team kernels need the highest stream priority to stay clear of other streams' kernels on the
    # device (tools/soak.py phase C); a caller's stream that has it carries the chain's launches itself, on a default-priority stream the library moves
    # them to a high-priority stream of its own between two events (+0.1 ms per decode).
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.Stream(priority=-1))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fallback = None
    if args.share_gpu and world > 1:
        # testing only: team kernels of TWO PROCESSES on one GPU each hold compute units the other one's teams are waiting for (inside a process the library's device
        # lease serialises decodes; there is none across processes) -- every bounded wait would give up.  The test of the launch line runs one launch per layer.
        eng.set_team_kernels(False)
        fallback = "--share-gpu: team kernels off (two processes on one GPU)"
    for _ in range(args.warmup):
        eng.synthesize(L)
        torch.cuda.synchronize()
        try:                                                         # a failed team hand-off (workgroups of a team not on one XCD) invalidates THAT decode and switches the
            eng.decode_status()                                      # library to one launch per layer: a warm-up step may absorb it, the timed region may not (checked below)
        except RuntimeError as e:
            fallback = str(e)
            print(f"[bench] warm-up decode reported: {e}", file=sys.stderr)
    barrier()
    chain_prof = args.decode_mode == 3

    def timed_region():
        eng.prof_enable(PROF_XCONE if chain_prof else -1)    # HIP events on the side stream around the xcone_kernel launch of every 16th frame (frames >= 100: full cones)
        t0 = time.perf_counter()
        out = None
        for _ in range(args.steps):
            out = eng.synthesize(L)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if world > 1:
            dist.barrier()
        eng.prof_enable(-1)
        n, ms = eng.prof_collect()
        return out, t1 - t0, n, ms, eng.prof_rows()

    # A decode whose bounded in-kernel wait gave up is invalid, and so is a timing that contains it: EVERY rank then repeats the region, and a rank whose own decode failed
    # runs one launch per layer from there on (the line says so: `placement`).  Three attempts: with two ranks on ONE GPU (the test's --share-gpu) the second rank's team
    # kernels can still time out beside the first rank's per-layer launches; a rank on per-layer launches has no bounded wait left, so the third attempt cannot fail.
    for attempt in range(3):
        (Y, Z, mx), elapsed, n_chain, chain_ms, chain_layers = timed_region()
        local_fail = 0
        try:
            eng.decode_status()
        except RuntimeError as e:
            local_fail = 1
            fallback = str(e)
            print(f"[bench] a decode of the timed region reported (attempt {attempt + 1}): {e}", file=sys.stderr)
        if world > 1:
            tf = torch.tensor([local_fail], dtype=torch.int32, device="cuda" if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(tf, op=dist.ReduceOp.MAX)
            any_fail = int(tf.item())
        else:
            any_fail = local_fail
        if not any_fail:
            break
        if local_fail:
            eng.set_team_kernels(False)
        barrier()
    else:
        raise RuntimeError("bench: the timed region contained a failed decode three times in a row: " + str(fallback))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                     # measurement only: no collective on the data path
        elapsed = float(t.item())

    # ---- the host gather of SURVEY 8e, timed separately (never part of `value`): every rank's Z -> its pinned host buffer ->
    #      rank 0's host (gloo); with one rank this is the plain D2H copy
    Zh = torch.empty(Z.shape, dtype=Z.dtype, pin_memory=True)
    barrier()
    tg0 = time.perf_counter()
    gathered = gather_to_host(Z, Zh, host_group, world, rank)
    if world > 1:
        dist.barrier()
    gather_s = time.perf_counter() - tg0
    if world > 1:
        t = torch.tensor([gather_s], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gather_s = float(t.item())
    if rank == 0 and world > 1:
        assert gathered is not None and tuple(gathered.shape) == (world * B, Z.shape[1], Z.shape[2])

    # ---- placement: where a 128-block launch lands (the team kernels' speed rests on blocks b, b + 8, ... sharing an XCD and on 256 co-resident workgroups;
    #      their correctness does not), and whether THIS rank's timed decodes ran on the team kernels -- from every rank, not only rank 0
    xcc, n_cu = eng.debug_xcd_census()
    k0 = int(xcc[0])
    nx = int(xcc.max()) + 1
    mine = {"rank": rank, "decode_team_kernels": fallback is None, "status_report": fallback, "compute_units": n_cu, "xcds_seen": int(len(set(xcc.tolist()))),
            "blocks_b_and_b_plus_8_share_an_xcd": bool(all(int(xcc[b]) == int(xcc[b % 8]) for b in range(128))),
            "round_robin_from_block_0": bool(nx > 1 and all(int(xcc[b]) == (k0 + b) % nx for b in range(128))),
            "team_kernels_state_bits": eng.debug_team_kernels_state()}
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine, group=host_group)
    else:
        ranks = [mine]
    if rank == 0:
        frames = world * B * T * args.steps
        value = frames / elapsed
        ms_step = elapsed / args.steps * 1e3
        rtf = elapsed / (world * B * args.steps * T * hp.seconds_per_mel_frame)
        d = hp.d
        # ---- roofline: the kernel that dominates the step by TIME (rocprofv3 --stats, profiles/r06_kernel_stats.md): xcone_kernel, AudioDec HC_3 and HC_4
        #      (round 6: and HC_5) over the rows of a frame's dependency cone (45 / 15 / 5 rows per utterance incl. the presum row) + their layer-norm / gate passes
        #      (HC_6, HC_7 run on the chain), one launch per frame on the decode's side stream.  Unit of work = one cone ROW of one
        #      layer: a (3 x 256) x 512 fp32 contraction = 2 * 768 * 512 FLOP; algorithmic bytes of a launch = the rows in and out (256 channels each) +
        #      the layers' weights once.
        row_flop = 2.0 * 3 * d * 2 * d
        roof = {"bound": "mfma", "achieved": None, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                "kernel": "xcone_kernel: AudioDec HC_3, HC_4 and (round 6) HC_5 (256 ch, k = 3, dilations 3 / 9 / 27) over the rows of a frame's dependency cone "
                          "(45 / 15 / 5 rows per utterance incl. the chain's presum row) + their layer-norm / gate row passes, ONE launch per frame on the decode's side stream; the launch "
                          "also carries AudioDec C_1 / HC_2 over their cone rows as its first two team phases (row operations; the FLOP count here is the GEMM layers' alone).  "
                          "16x16x4 fp32 MFMA, a 16-workgroup team per four utterances inside one XCD; round 6: a layer's 96 KB weight slice lives in LDS (global_load_lds) and a wave "
                          "contracts whole row tiles against it -- no split-K exchange inside a layer (rounds 3-5: slice in registers, K split over the waves, an LDS reduction per pass).  "
                          "The cone's last two layers (3 / 1 rows per utterance) run on the chain: kernels[] has xchain_kernel",
                "launches": n_chain, "sampled": "every 16th frame from frame 100 on (full-size cones) of the timed region, HIP events on the side stream",
                "avg_launch_ms": None, "rows_per_launch": None, "flop_per_row": row_flop,
                "note": "runs concurrently with the chain's kernels on the other half of the CUs (128 of 256: at most 0.5 of the roof); its launch ends with "
                        "the team leaders polling the chain's counter, so the event-timed duration includes that wait whenever the chain is the longer stream "
                        "(DESIGN.md section 2).  `frac` prices the WHOLE event-timed launch against the GEMM layers' FLOPs; `frac_gemm_phase` prices the GEMM layers' own phases "
                        "(in-kernel stamps of one launch, `phases_us`) -- the figure that says how well the contraction itself runs"}
        if n_chain > 0 and chain_layers > 0:
            avg = chain_ms / n_chain
            rpl = chain_layers / n_chain
            n_gemm_layers = 3 if rpl / B > 62 else 2                               # 45 + 15 (+ 5) rows per utterance
            alg_bytes = 4.0 * (rpl * (d + d) + n_gemm_layers * 3 * d * 2 * d)
            tf = row_flop * rpl / (avg * 1e-3) / 1e12
            roof.update(avg_launch_ms=round(avg, 5), rows_per_launch=round(rpl, 1), flop_per_launch=row_flop * rpl, algorithmic_bytes_per_launch=alg_bytes,
                        achieved=round(tf, 3), frac=round(tf / PEAK_F32_MFMA_TFLOPS, 4),
                        frac_hbm=round(alg_bytes / (avg * 1e-3) / 1e9 / PEAK_HBM_GBPS, 5))
        # (a) the launch also carries the two row phases (no MFMA work) and its barriers / row passes: the GEMM layers' own phases from in-kernel stamps of one
        #     steady-state launch (an untimed extra decode), so that `frac` (whole launch, event-timed) and `frac_gemm_phase` can be told apart
        if chain_prof and not args.no_extras and roof.get("flop_per_launch"):
            st = xcone_phase_stamps(eng, L, T)
            if st:
                tfg = roof["flop_per_launch"] / (st["gemm"] * 1e-6) / 1e12
                roof.update(phases_us=st, achieved_gemm_phase=round(tfg, 3), frac_gemm_phase=round(tfg / PEAK_F32_MFMA_TFLOPS, 4),
                            frac_gemm_phase_of_the_128_cus_the_kernel_owns=round(2 * tfg / PEAK_F32_MFMA_TFLOPS, 4))
        for tag in ("r06", "r05"):
            tj = os.path.join(ROOT, "profiles", f"{tag}_pmc_decode.json")
            if os.path.exists(tj):
                pj = json.load(open(tj))
                if "xcone_kernel" in pj:
                    roof["traffic"] = pj["xcone_kernel"]["hbm_bytes_per_launch"]
                    roof["traffic_unit"] = (f"bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes: profiles/{tag}_pmc_decode.json; includes Infinity-Cache hits).  "
                                            "Form: counter collection serialises queues, so under it the two decode streams meet through events (DCTTS_SYNC_VALUES=0) and a chain piece runs as "
                                            "the split launches; xcone_kernel itself is the shipping kernel" + ("" if tag == "r06" else " of ROUND 5 (register-resident weight slices, HC_3 / HC_4 only)"))
                break
        flop_frame = 2 * 3.0789e9 / T + 8.167e6 + 142.254e6 + 0.26e6 + 187.310e6      # SURVEY 8d, per mel frame and utterance
        out = {
            "metric": "mel frames/sec (Text2Mel->SSRN, LJ hyper-parameters)", "value": round(value, 1), "unit": "mel frames/s",
            "rtf": rtf, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (seeded character ids, seeded random-init weights)",
            "config": {"workload": f"full Text2Mel autoregressive decode + SSRN, batch={B}/GPU, max_N={hp.max_N}, "
                                   f"max_T={T} mel frames -> ({B},{4 * T},{hp.n_linear}) per GPU; exact-parity incremental decode",
                       "batch_per_gpu": B, "max_N": hp.max_N, "max_T": T, "decode_mode": args.decode_mode, "decode_graph_mode": gm,
                       "decode_team_kernels": all(r["decode_team_kernels"] for r in ranks),
                       "caller_stream": "a high-priority HIP stream (the decode's chain launches run on the caller's stream when it has the highest priority; from a "
                                        "default-priority stream the library moves them to its own high-priority stream between two events: +0.1 ms per decode)",
                       "sharding": f"{world} x {B} utterances, no collective"},
            "pipeline_tflops": round(value * flop_frame / 1e12, 2),
            "pipeline_frac_of_f32_mfma_peak": round(value * flop_frame / 1e12 / (PEAK_F32_MFMA_TFLOPS * world), 4),
            "roofline": roof, "device_bytes": eng.device_bytes(), "placement": ranks,
            "gather": {"what": "every rank's Z (B,4T,1025) -> its pinned host buffer (D2H) -> rank 0's host over gloo (SURVEY 8e); one rank: the D2H copy",
                       "bytes_per_rank": Z.numel() * 4, "seconds": round(gather_s, 5),
                       "value_incl_gather": round(world * B * T / (elapsed / args.steps + gather_s), 1),
                       "note": "serial upper bound on the cost: the gather of batch n can overlap the compute of batch n+1"},
        }
        if not args.no_extras:
            out.update(extras(eng, args, hp, W, L, Y, Z, B, T, gm, ms_step))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(hp, W)
        if world == 1 and not args.no_vocoder:
            out["vocoder"] = vocoder_section(hp, Z, ms_step, not args.no_cpu_baseline)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pipelined(eng, hp, L, Y_serial, Z_serial, B, T, ms_serial, with_vocoder, n=10):
    """Steady state of a two-deep pipeline on ONE GPU: the caller's (high-priority) stream runs TextEnc + decode of batch i + 1 while a second stream runs SSRN
    (and optionally the Griffin-Lim vocoder) of batch i.  Wall clock over n batches incl. the un-overlapped last SSRN, so the figure is slightly pessimistic."""
    from dc_tts_amd.utils import Vocoder
    main = torch.cuda.current_stream()
    s2 = torch.cuda.Stream()
    voc = Vocoder(hp) if with_vocoder else None
    res = {}

    def run(k):
        last = None
        for _ in range(k):
            Yi, mi = eng.text2mel(L)
            ev = torch.cuda.Event(); ev.record(main)
            with torch.cuda.stream(s2):
                s2.wait_event(ev)
                Yi.record_stream(s2)
                Zi = eng.ssrn(Yi, want_logits=False)[1]
                wav = voc.spectrogram2wav_device(Zi) if voc is not None else None
            last = (Yi, Zi, wav)
        return last
    run(2); torch.cuda.synchronize()
    t0 = time.perf_counter()
    Yl, Zl, _ = run(n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    eng.decode_status()
    same = bool(torch.equal(Yl, Y_serial)) and bool(torch.equal(Zl, Z_serial))
    res.update(workload=f"{n} batches of B={B}, max_T={T}: TextEnc + decode of batch i+1 on the caller's stream beside SSRN" + (" + the Griffin-Lim vocoder" if with_vocoder else "") +
                        " of batch i on a second stream of the same engine (wall clock incl. the last batch's un-overlapped tail)",
               ms_per_batch=round(dt * 1e3, 3), mel_frames_per_s=round(B * T / dt, 1), vs_serial=round(ms_serial / (dt * 1e3), 3) if not with_vocoder else None,
               outputs_bitwise_equal_to_serial=same)
    if voc is not None:
        voc.close()
    return res


def extras(eng, args, hp, W, L, Y, Z, B, T, gm, ms_step):
    """Untimed passes on rank 0 after the timed region: per-phase times, the two other kernels that matter, the other BASELINE
    configurations that fit one GPU."""
    from dc_tts_amd.engine import Engine
    from dc_tts_amd.weights import synthetic_text
    d, c = hp.d, hp.c
    res = {}
    # ---- phases
    # (order matters: this runs behind the host gather, i.e. after ~0.1 s of GPU idle, and a 2 ms kernel sequence timed first reads 10 % slow while the clocks come back --
    #  TextEnc 2.2 ms measured first against 1.95 ms behind any other work, tools/scratch/te_bench_dbg.py; in the pipeline it runs right behind the previous batch's SSRN)
    ms_t2m = timed(lambda: eng.text2mel(L))
    ms_ssrn = timed(lambda: eng.ssrn(Y, want_logits=False))
    ms_te = timed(lambda: eng.text_enc(L), reps=10)
    dec_ms = ms_t2m - ms_te
    res["phases"] = {"textenc_ms": round(ms_te, 3), "text2mel_total_ms": round(ms_t2m, 3),
                     "decode_us_per_step": round(dec_ms * 1e3 / T, 2), "ssrn_ms": round(ms_ssrn, 3)}
    # fractions of BOTH roofs from SURVEY 8d's algorithmic work (FLOP and fp32 bytes per utterance) and the phase times
    res["phase_rooflines"] = {
        "textenc": both_roofs(B * 2 * 3.0789e9, 68.6e6 + B * 0.37e6, ms_te),
        "decode": dict(both_roofs(B * T * (8.167e6 + 142.254e6 + 0.26e6), T * (27285440.0 + B * 125e3), dec_ms),
                       bound="latency: 16 dependent all-to-all highway layers + 7 k=1 layers + attention + AudioDec C_1 per frame in ONE launch on the chain's "
                             "stream (round 5; round 4: two), beside the cone re-evaluation on the side stream (ONE launch, round 4: three; the two streams are about equally long); the cone work of AudioDec C_1 / HC_2 runs as row operations on "
                             "cached products, so fewer FLOPs are EXECUTED than the algorithmic count used here (DESIGN.md section 2)"),
        "ssrn": both_roofs(B * T * 187.310e6, B * (67200 + 3444000) + 113641532.0, ms_ssrn),
    }
    res["roofline_frac_decode_phase_mfma"] = res["phase_rooflines"]["decode"]["frac_mfma"]
    # ---- (c) what the step costs when the team kernels are OFF (one launch per layer, what a decode falls back to when a team is not on one XCD or the bounded waits
    #      give up -- e.g. a partition mode with fewer CUs than 128 + 128 workgroups): one untimed extra pass, so that a fallback on another node is interpretable
    if args.decode_mode == 3:
        eng.set_team_kernels(False)
        try:
            ms_fb = timed(lambda: eng.synthesize(L), reps=2)
            eng.decode_status()
            Yf, Zf, mf = eng.synthesize(L); torch.cuda.synchronize()
            res["fallback"] = {"what": "dctts_set_team_kernels(0): every decode layer a launch of its own (chain3_kernel / hbulk_kernel / row kernels), the two streams as before",
                               "ms_per_step": round(ms_fb, 3), "mel_frames_per_s": round(B * T / (ms_fb * 1e-3), 1), "vs_team_kernels": round(ms_step / ms_fb, 3),
                               "max_abs_dY_vs_team_kernels": float((Yf - Y).abs().max())}
        finally:
            eng.set_team_kernels(True)
        eng.synthesize(L); torch.cuda.synchronize(); eng.decode_status()
    # ---- kernels: event-timed extra passes
    kern = []
    eng.prof_enable(PROF_SSRN_HC)
    for _ in range(2):
        eng.ssrn(Y, want_logits=False)
    torch.cuda.synchronize(); eng.prof_enable(-1)
    n, ms = eng.prof_collect(); rows = eng.prof_rows()
    if n:
        C = 2 * c
        rpl = rows / n
        kern.append(dict(kernel="hconv_kernel<EPI_HC,NT=8,NW=8> (SSRN HC_11 / HC_12: 1024 ch, k=3, fused LN + gate): the largest kernel by FLOPs",
                         bound="mfma", launches=n, avg_launch_ms=round(ms / n, 4), rows_per_launch=rpl, layer_rows=B * 4 * T,
                         **both_roofs(2.0 * rpl * 3 * C * 2 * C, 4.0 * (rpl * C * 2 + 3 * C * 2 * C), ms / n)))
    F = hp.n_linear
    for kid, what, K_, N_ in ((PROF_SSRN_HC_TAIL, "hconv_kernel<EPI_HC,NT=8,NW=8,RAW> + hc_tail_finish_kernel (SSRN HC_11 / HC_12: the rows left after three exact rounds, as 72 32-row items x 3 taps + a finishing pass)", 3 * 2 * c, 2 * 2 * c),
                              (PROF_SSRN_C1025, "hconv_kernel<EPI_C,NT=4,NW=8,XC> (SSRN C_14 / C_15 / C_16 and C_13: 1025 columns, k=1, fused LN + activation; round 5: 8 waves x 4 tiles + the 1025th column on the vector ALU, main launch)", None, F)):
        eng.prof_enable(kid)
        for _ in range(2):
            eng.ssrn(Y, want_logits=False)
        torch.cuda.synchronize(); eng.prof_enable(-1)
        n, ms = eng.prof_collect(); rows = eng.prof_rows()
        if n:
            rpl = rows / n
            Kk = K_ if K_ is not None else (3 * F + 2 * c) / 4.0          # C_13 reads 1024 channels, C_14..C_16 1025: mean over the four launches of a pass
            kern.append(dict(kernel=what, bound="mfma", launches=n, avg_launch_ms=round(ms / n, 4), rows_per_launch=rpl, layer_rows=B * 4 * T,
                             **both_roofs(2.0 * rpl * Kk * N_, 4.0 * (rpl * (Kk + N_ / (2 if K_ else 1)) + Kk * N_), ms / n)))
    if args.decode_mode == 3:
        eng.text2mel(L); torch.cuda.synchronize()
        eng.prof_enable(PROF_XGROUP); eng.text2mel(L); torch.cuda.synchronize(); eng.prof_enable(-1)
        n, ms = eng.prof_collect(); layers = eng.prof_rows()
        eng.prof_enable(PROF_XTAIL); eng.text2mel(L); torch.cuda.synchronize(); eng.prof_enable(-1)
        nt, mst = eng.prof_collect()
        one_launch = bool(nt) and not n          # round 5 default: a chain piece is ONE launch (xchain_kernel); DCTTS_CHAIN_TAIL=6: round 4's two launches
        if one_launch:
            # xtail_kernel's part per utterance and frame (round 6): the newest row of HC_2 .. HC_5 (K = 256 -> 512 columns each), HC_6 / HC_7 over 3 / 1 rows (K = 768),
            # C_8 .. C_10, C_11 (256 -> 80), AudioEnc C_1 (80 -> 256), C_2, C_3
            fl = 2.0 * (4 * d * 2 * d + 4 * 3 * d * 2 * d + 3 * d * d + d * hp.n_mels + hp.n_mels * d + 2 * d * d)
            by = 4.0 * (4 * d * 2 * d + 2 * 3 * d * 2 * d + 5 * d * d + 2 * d * hp.n_mels) + 4.0 * B * (4 * 2 * d + 4 * d + d + hp.n_mels)      # the 13 layers' weights once + per utterance four presum rows, the 4 staged rows, the mel frame, the output row
            lay_bytes = 4.0 * (d * 2 * d + 3 * B * 2 * d + 2 * B * 64 + B * d + 4 * d)      # one AudioEnc layer: weights 256 x 512, presum / rows out / rows in, statistics, kept row, LN parameters
            lay_flop = 2.0 * B * d * 2 * d
            fl_tail = 2.0 * B * (3 * d + d * d)                                               # attention logits over the 3-key window + C_1's Q half (256 x 256)
            by_tail = 4.0 * (d * d + B * (3 * 2 * d + 3 * d + 2 * d))
            e = dict(kernel="xchain_kernel (decode chain: a chain piece as ONE launch in team form -- xtail_kernel's part: AudioDec's newest-row layers HC_2 .. HC_5 (round 6: HC_5's "
                            "older cone rows moved to the side stream), HC_6 / HC_7 over the 3 / 1 cone rows they need, the seven k = 1 layers around the mel frame; a team barrier; "
                            "xgroup_kernel's part: the AudioEnc run HC_4 .. HC_13 of the next frame, its attention row and AudioDec C_1.  It also carries the presum GEMMs as passenger "
                            "workgroups, and its event-timed duration includes the wait for the side stream whenever that is the longer one; the two parts timed separately: "
                            "profiles/r06_chain_tail_split.txt, DCTTS_CHAIN_TAIL=6)",
                     bound="latency (25 dependent all-to-all layers)", launches=nt, avg_launch_ms=round(mst / nt, 5), layers_per_launch=25,
                     **both_roofs(fl * B + 10 * lay_flop + fl_tail, by + 10 * lay_bytes + by_tail, mst / nt))
            tj = os.path.join(ROOT, "profiles", "r06_pmc_decode.json")
            if os.path.exists(tj):
                pj = json.load(open(tj))
                if "xgroup_kernel" in pj and "xtail_kernel" in pj:
                    e["traffic"] = pj["xgroup_kernel"]["hbm_bytes_per_launch"] + pj["xtail_kernel"]["hbm_bytes_per_launch"]
                    e["traffic_note"] = ("PMC bytes per launch of the two parts, added (counter collection runs them as two launches: xtail_kernel + xgroup_kernel): eight "
                                         "XCD-local copies of every layer's weights, served by the Infinity Cache")
            kern.append(e)
        if nt and not one_launch:
            # per utterance and frame: the newest row of HC_2 .. HC_4 (K = 256: centre tap -> 512 columns), HC_5 / HC_6 / HC_7 over 5 / 3 / 1 rows (K = 768 -> 512 columns)
            # + C_8 .. C_10, C_11 (256 -> 80), AudioEnc C_1 (80 -> 256), C_2, C_3
            fl = 2.0 * (3 * d * 2 * d + 9 * 3 * d * 2 * d + 3 * d * d + d * hp.n_mels + hp.n_mels * d + 2 * d * d)
            by = 4.0 * (3 * d * 2 * d + 3 * 3 * d * 2 * d + 5 * d * d + 2 * d * hp.n_mels) + 4.0 * B * (3 * 2 * d + 14 * d + d + hp.n_mels)      # the 13 layers' weights once + per utterance three presum rows, the 14 staged rows, the mel frame, the output row
            kern.append(dict(kernel="xtail_kernel, merged form (decode chain, round 4: AudioDec's newest-row layers HC_2 .. HC_4, then HC_5 .. HC_7 over the 5 / 3 / 1 cone rows "
                                    "they need, then the seven k = 1 layers around the mel frame: ONE launch per frame in team form, the first of a chain piece's two launches; it also "
                                    "carries the presum GEMMs as passenger workgroups, and its event-timed duration includes the wait for the side stream whenever that is the longer one; "
                                    "the k = 1 layers hand their rows over without a barrier: every value travels with a sequence tag)",
                             bound="latency (13 dependent all-to-all layers)", launches=nt, avg_launch_ms=round(mst / nt, 5), layers_per_launch=13,
                             **both_roofs(fl * B, by, mst / nt)))
        if n:
            lpl = layers / n
            lay_bytes = 4.0 * (d * 2 * d + 3 * B * 2 * d + 2 * B * 64 + B * d + 4 * d)      # one layer: weights 256 x 512, presum / rows out / rows in, statistics, kept row, LN parameters
            lay_flop = 2.0 * B * d * 2 * d
            e = dict(kernel="xgroup_kernel (decode chain: a run of newest-row highway layers as ONE launch -- timed: the AudioEnc run, HC_4 .. HC_13 = ten layers + since round 4 "
                            "the attention row of the frame and AudioDec C_1 behind them; per layer a 32 x 256 x 512 contraction split over the 16 workgroups of a "
                            "4-utterance team, rows and statistics exchanged through the L2 of the team's XCD)",
                     bound="latency (16 dependent all-to-all layers per frame)", launches=n, avg_launch_ms=round(ms / n, 5), layers_per_launch=lpl,
                     algorithmic_bytes_per_layer=lay_bytes, flop_per_layer=lay_flop, **both_roofs(lay_flop * lpl, lay_bytes * lpl, ms / n))
            tj = os.path.join(ROOT, "profiles", "r06_pmc_decode.json")
            if os.path.exists(tj):
                pj = json.load(open(tj))
                if "xgroup_kernel" in pj:
                    e["traffic"] = pj["xgroup_kernel"]["hbm_bytes_per_launch"]
                    e["traffic_note"] = "PMC bytes per launch, mean over BOTH runs of a frame (6 and 10 layers): eight XCD-local copies of every layer's weights, served by the Infinity Cache"
            kern.append(e)
    res["kernels"] = kern
    # ---- host transfer (PCIe-inclusive figure, reported beside `value`, never as it)
    Lh = L.cpu().pin_memory()
    Zh = torch.empty(Z.shape, dtype=Z.dtype, pin_memory=True)
    def xfer():
        L.copy_(Lh, non_blocking=True); Zh.copy_(Z, non_blocking=True)
    ms_x = timed(xfer)
    res["host_transfer"] = {"h2d_bytes": Lh.numel() * 4, "d2h_bytes": Z.numel() * 4, "ms_per_batch": round(ms_x, 3),
                            "GBps": round(Z.numel() * 4 / (ms_x * 1e-3) / 1e9, 1),
                            "value_incl_transfer": round(B * T / ((ms_step + ms_x) * 1e-3), 1)}
    # ---- the other BASELINE configurations that fit one GPU (configs[3] is this run; [0] is the CPU plumbing case)
    oc = {}
    oc["config2_text2mel_decode_b32"] = {"workload": f"TextEnc + {T}-step decode, B={B}", "ms_per_batch": round(ms_t2m, 3),
                                         "mel_frames_per_s": round(B * T / (ms_t2m * 1e-3), 1), "rtf": ms_t2m * 1e-3 / (B * T * hp.seconds_per_mel_frame)}
    B3 = 128
    Y3 = torch.rand(B3, T, hp.n_mels, device=Y.device)
    ms3 = timed(lambda: eng.ssrn(Y3, want_logits=False), reps=2)
    oc["config3_ssrn_only_b128"] = dict(workload=f"SSRN only, B={B3}: ({B3},{T},80) -> ({B3},{4 * T},1025)", ms_per_batch=round(ms3, 3),
                                        mel_frames_per_s=round(B3 * T / (ms3 * 1e-3), 1), rtf=ms3 * 1e-3 / (B3 * T * hp.seconds_per_mel_frame),
                                        **both_roofs(B3 * T * 187.310e6, B3 * (67200 + 3444000) + 113641532.0, ms3))
    del Y3
    # ---- decode-only at larger batches (one pass of the decode serves the batch in rounds of 8 teams x 4 utterances: DESIGN.md section 9)
    for Bx in (64, 128):
        Lx = torch.from_numpy(synthetic_text(hp, B=Bx, seed=4321)).cuda()
        msx = timed(lambda: eng.text2mel(Lx), reps=2)
        oc[f"text2mel_decode_b{Bx}"] = {"workload": f"TextEnc + {T}-step decode, B={Bx} in ONE call", "ms_per_batch": round(msx, 3),
                                        "mel_frames_per_s": round(Bx * T / (msx * 1e-3), 1), "vs_b32": round((Bx * T / msx) / (B * T / ms_t2m), 3)}
        del Lx
    # ---- two batches in flight (NOT the headline: synthesize.py:45-57 is serial): SSRN of batch n on a second stream beside TextEnc + decode of batch n + 1, every
    #      batch still B utterances; then the same with the vocoder of batch n on that second stream as well.  One engine: calls of different kinds overlap, calls that
    #      share scratch are ordered by the library (include/dctts_hip.h, "streams and threads").  Outputs are compared bitwise with the serial run's.
    oc["pipelined_depth2"] = pipelined(eng, hp, L, Y, Z, B, T, ms_step, with_vocoder=False)
    if not args.no_vocoder:
        oc["pipelined_depth2_with_vocoder"] = pipelined(eng, hp, L, Y, Z, B, T, ms_step, with_vocoder=True)
    # ---- the OPT-IN split-bf16 contraction (dctts_set_split_bf16; NEVER the headline: `dtype` above is f32 and `value` is measured with exact fp32 everywhere).
    #      A second engine carries the bf16 weight packing; level 1 = SSRN, level 2 = SSRN + TextEnc; Text2Mel's decode stays fp32 in both.
    eb = Engine(W, hp, device=eng.device_index, decode_graph=gm, split_bf16=2)
    eb.set_decode_mode(args.decode_mode)
    sb = {}
    for level in (1, 2):
        eb.set_split_bf16(level)
        ms_b = timed(lambda: eb.synthesize(L))
        Yb, Zb, mb = eb.synthesize(L); torch.cuda.synchronize(); eb.decode_status()
        sb[f"level{level}"] = {"what": "SSRN on split-bf16 operands" if level == 1 else "SSRN + TextEnc on split-bf16 operands", "ms_per_batch": round(ms_b, 3),
                               "mel_frames_per_s": round(B * T / (ms_b * 1e-3), 1), "vs_fp32": round(ms_step / ms_b, 3),
                               "ssrn_ms": round(timed(lambda: eb.ssrn(Y, want_logits=False)), 3), "textenc_ms": round(timed(lambda: eb.text_enc(L)), 3),
                               "max_abs_vs_fp32_run": {"Y": float((Yb - Y).abs().max()), "Z": float((Zb - Z).abs().max())},
                               "attention_trajectory_equal_to_fp32_run": bool(torch.equal(mb, eng.text2mel(L)[1]))}
    sb["arithmetic"] = ("x = hi + mid (two bf16 terms, round-to-nearest), product = hi.hi + hi.mid + mid.hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate; dropped terms <= 2^-16 |x||w|; "
                        "bias / layer-norm / gate / activations / the 1025th column in fp32; measured against the float64 oracle: max|dZ| 3.3e-5 (fp32 form: 3.6e-6), "
                        "tests/test_gpu_parity.py::test_split_bf16_pipeline_error_and_untouched_decode")
    oc["split_bf16_opt_in"] = sb
    eb.close()
    T5, B5 = 1000, 8
    h5 = hp.replace(max_T=T5)
    e5 = Engine(W, h5, device=eng.device_index, decode_graph=gm)
    e5.set_decode_mode(args.decode_mode)
    L5 = torch.from_numpy(synthetic_text(h5, B=B5, seed=77)).cuda()
    ms5 = timed(lambda: e5.synthesize(L5), reps=2)
    oc["config5_long_form_t1000_b8"] = {"workload": f"full Text2Mel + SSRN, max_T={T5}, B={B5} (one GPU's share of 64)", "ms_per_batch": round(ms5, 3),
                                        "mel_frames_per_s": round(B5 * T5 / (ms5 * 1e-3), 1), "rtf": ms5 * 1e-3 / (B5 * T5 * h5.seconds_per_mel_frame)}
    e5.close()
    res["other_configs"] = oc
    return res


if __name__ == "__main__":
    main()
