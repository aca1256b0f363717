"""bench.py -- mel frames/s + RTF of the DC-TTS synthesis path on MI355X (BASELINE.json metric).

One "step" = one full synthesis of a batch of B=32 utterances on each GPU: TextEnc once, the 210-step
autoregressive Text2Mel decode (exact-parity incremental algorithm), one SSRN pass -> (32, 840, 1025)
linear spectrogram; inputs (character ids) and weights are resident in HBM before the timed region.
Multi-GPU: utterances shard embarrassingly, 32 per GPU, no collective on the data path ("weak").

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
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

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
PEAK_HBM_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E ~8 TB/s
DOMINANT_KERNEL_ID = 1 * 10000 + 8 * 100 + 8    # hconv_kernel<EPI_HC, NT=8, NW=8>: SSRN HC_11 / HC_12 (C = 1024)


def cpu_baseline(hp, W, seconds_budget=25.0):
    """The oracle (a port: numpy restatement of the reference, TF being uninstallable here) timed on the host
    cores, on a bounded sample: a few steps of the reference's full-recompute loop (synthesize.py:47-54, TextEnc
    recomputed every step as the reference does) for a small batch, plus one SSRN pass, prorated per mel frame."""
    from dc_tts_amd.weights import synthetic_text
    from oracle import dctts_ref as O
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
    t_step = (time.perf_counter() - t0) / steps               # seconds per loop step for Bs utterances
    t0 = time.perf_counter()
    O.SSRN(Y[:1], W, hp)
    t_ssrn = time.perf_counter() - t0                         # seconds per utterance
    per_frame = t_step / Bs + t_ssrn / hp.max_T               # one loop step yields one mel frame per utterance
    try:                                                      # threads numpy's BLAS actually runs on (the matmuls are the work)
        from threadpoolctl import threadpool_info
        cores = max([int(i.get("num_threads", 1)) for i in threadpool_info() if i.get("user_api") == "blas"] or [1])
    except Exception:
        cores = os.cpu_count()
    return {"value": 1.0 / per_frame, "unit": "mel frames/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"{steps} steps of the restated synthesize.py loop (full Text2Mel graph incl. TextEnc per step, "
                      f"B={Bs}, N={hp.max_N}, T={hp.max_T}) + 1 SSRN pass (B=1), numpy fp32 (BLAS threads = cores), "
                      f"prorated per mel frame",
            "rtf": per_frame / hp.seconds_per_mel_frame}


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
        out["roofline"]["traffic_unit"] = "bytes/launch (PMC, separate pass: profiles/r01_vocoder.md)"
    if with_cpu:
        out["cpu_baseline"] = vocoder_cpu_baseline(hp, Z[0].cpu().numpy())
    v.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU (BASELINE: 32)")
    ap.add_argument("--max-T", type=int, default=210)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--graph-mode", type=int, default=1, help="0 eager, 1 bulk pieces as hipGraphs (default), 2 chain pieces too")
    ap.add_argument("--decode-mode", type=int, default=1, help="1 = default (split kernels, two streams), 2 = + fused k=1 row MLP, 0 = fused full-row kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vocoder", action="store_true", help="skip the untimed vocoder-tail section (SURVEY 8f-2)")
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
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.dist_backend)

    from dc_tts_amd.engine import Engine
    from dc_tts_amd.hyperparams import hp as hp0
    from dc_tts_amd.layers import ssrn_layers
    from dc_tts_amd.weights import synthetic_text, synthetic_weights

    hp = hp0.replace(max_T=args.max_T)
    B, T = args.batch, hp.max_T
    W = synthetic_weights(hp, seed=1234, perturb=True)
    eng = Engine(W, hp, device=local, decode_graph=0 if args.no_graph else args.graph_mode)
    eng.set_decode_mode(args.decode_mode)
    L = torch.from_numpy(synthetic_text(hp, B=B, seed=1234 + rank)).cuda()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.synthesize(L)
    barrier()
    eng.prof_enable(DOMINANT_KERNEL_ID)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        Y, Z, mx = eng.synthesize(L)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if world > 1:
        dist.barrier()
    elapsed = t1 - t0
    eng.prof_enable(-1)
    n_launch, dom_ms = eng.prof_collect()
    dom_rows = eng.prof_rows()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                     # measurement only: no collective on the data path
        elapsed = float(t.item())

    # ---- untimed: per-phase breakdown (torch events on the launch stream)
    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    phases = None
    host_xfer = None
    if rank == 0:
        # PCIe-inclusive figure (reported beside `value`, never as it): ids host->device + Z device->pinned host per batch
        Lh = L.cpu().pin_memory()
        Zh = torch.empty(Z.shape, dtype=Z.dtype, pin_memory=True)
        def xfer():
            L.copy_(Lh, non_blocking=True); Zh.copy_(Z, non_blocking=True)
        ms_x = timed(xfer)
        host_xfer = {"h2d_bytes": Lh.numel() * 4, "d2h_bytes": Z.numel() * 4, "ms_per_batch": round(ms_x, 3),
                     "GBps": round(Z.numel() * 4 / (ms_x * 1e-3) / 1e9, 1)}
        ms_te = timed(lambda: eng.text_enc(L))
        ms_t2m = timed(lambda: eng.text2mel(L))
        ms_ssrn = timed(lambda: eng.ssrn(Y, want_logits=False))
        phases = {"textenc_ms": round(ms_te, 3), "text2mel_total_ms": round(ms_t2m, 3),
                  "decode_us_per_step": round((ms_t2m - ms_te) * 1e3 / T, 2), "ssrn_ms": round(ms_ssrn, 3)}

    if rank == 0:
        frames = world * B * T * args.steps
        value = frames / elapsed
        rtf = elapsed / (world * B * args.steps * T * hp.seconds_per_mel_frame)
        # roofline of the dominant kernel: algorithmic FLOPs of one launch = 2 * rows * K * N of the layer
        C = 2 * hp.c                                             # SSRN HC_11 / HC_12: 1024 -> 2048, k = 3
        # rows per launch as the library reports them: the layer's B*4T rows minus the row tail that goes to hconv16_kernel
        rows_per_launch = dom_rows / n_launch if n_launch else B * 4 * T
        flops_per_launch = 2.0 * rows_per_launch * (3 * C) * (2 * C)
        roof = {"bound": "mfma", "achieved": None, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": None,
                "traffic": None, "kernel": "hconv_kernel<EPI_HC,NT=8,NW=8> (SSRN HC_11/HC_12, 1024ch k=3, fused LN+gate): the largest kernel by FLOPs (27 % of the pipeline); by time the latency-bound decode chain dominates, see phase_rooflines.decode",
                "launches": n_launch, "avg_launch_ms": None, "flop_per_launch": flops_per_launch,
                "rows_per_launch": rows_per_launch, "layer_rows": B * 4 * T}
        tj = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if B == 32 and T == 210 and os.path.exists(tj):
            # HBM bytes per launch of this kernel from the PMC passes (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE),
            # collected separately with rocprofv3 --pmc (profiles/r01_pmc_traffic.md); bench.py cannot read counters itself
            roof["traffic"] = json.load(open(tj))["hbm_bytes_per_launch"]
            roof["traffic_unit"] = "bytes/launch (PMC, separate pass)"
        if n_launch > 0:
            avg_ms = dom_ms / n_launch
            ach = flops_per_launch / (avg_ms * 1e-3) / 1e12
            roof.update(achieved=round(ach, 2), frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4), avg_launch_ms=round(avg_ms, 4))
        # whole-pipeline algorithmic FLOPs per mel frame: TextEnc/T + AudioEnc + AudioDec cone + attention + SSRN
        flop_frame = 2 * 3.0789e9 / T + 8.167e6 + 142.254e6 + 0.26e6 + 187.310e6
        # per-phase fractions of both roofs from SURVEY 8d's algorithmic work (FLOP and fp32 bytes per utterance) and the phase times
        # above: by TIME the decode dominates and is latency-bound -- `roofline` below is the kernel that dominates by FLOPs
        def _pr(flop_utt, bytes_utt, ms):
            tf = B * flop_utt / (ms * 1e-3) / 1e12
            gb = B * bytes_utt / (ms * 1e-3) / 1e9
            return {"tflops": round(tf, 2), "frac_mfma": round(tf / PEAK_F32_MFMA_TFLOPS, 4), "algorithmic_GBps": round(gb, 1),
                    "frac_hbm": round(gb / PEAK_HBM_GBPS, 4)}
        dec_ms = phases["text2mel_total_ms"] - phases["textenc_ms"]
        phase_roof = {
            "textenc": _pr(2 * 3.0789e9, 68.6e6 / B + 0.37e6, phases["textenc_ms"]),
            "decode": dict(_pr(T * (8.167e6 + 142.254e6 + 0.26e6), T * (27285440.0 / B + 125e3), dec_ms),
                           bound="latency: 25 dependent launches per frame on the critical path (DESIGN.md section 4)"),
            "ssrn": _pr(T * 187.310e6, 67200 + 3444000 + 113641532.0 / B, phases["ssrn_ms"]),
        }
        out = {
            "metric": "mel frames/sec (Text2Mel->SSRN, LJ hyper-parameters)", "value": round(value, 1), "unit": "mel frames/s",
            "rtf": rtf, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (seeded character ids, seeded random-init weights)",
            "config": {"workload": f"full Text2Mel autoregressive decode + SSRN, batch={B}/GPU, max_N={hp.max_N}, "
                                   f"max_T={T} mel frames -> ({B},{4 * T},{hp.n_linear}) per GPU; exact-parity incremental decode",
                       "batch_per_gpu": B, "max_N": hp.max_N, "max_T": T, "decode_graph_mode": 0 if args.no_graph else args.graph_mode,
                       "sharding": f"{world} x {B} utterances, no collective"},
            "pipeline_tflops": round(value * flop_frame / 1e12, 2),
            "pipeline_frac_of_f32_mfma_peak": round(value * flop_frame / 1e12 / (PEAK_F32_MFMA_TFLOPS * world), 4),
            "phases": phases, "phase_rooflines": phase_roof, "roofline": roof, "device_bytes": eng.device_bytes(),
            "host_transfer": dict(host_xfer, value_incl_transfer=round(world * B * T / ((elapsed / args.steps) + host_xfer["ms_per_batch"] * 1e-3), 1),
                                  note="serial upper bound on the cost: the copy of batch n can overlap the compute of batch n+1"),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(hp, W)
        if world == 1 and not args.no_vocoder:
            out["vocoder"] = vocoder_section(hp, Z, elapsed / args.steps * 1e3, not args.no_cpu_baseline)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
