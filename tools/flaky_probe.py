"""First decodes at NEW geometries (new workspaces, new device tables, cold TLB) against the numpy statement of the incremental algorithm: the case in which a decode variant
came out wrong in round 6 (DCTTS_CHAIN_TAIL=0 / 1 with 64 MiB workspace arenas; never on a later run at the same geometry).  CT = DCTTS_CHAIN_TAIL of the engine under
test, BS = batch sizes, TS = frame counts (every (B, T) pair is a new geometry), NOTHER = other engines alive on the device (B = 32 syntheses before, SSRN calls between)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
os.environ["DCTTS_CHAIN_TAIL"] = os.environ.get("CT", "2")
import dc_tts_amd._lib as _L0
if os.environ.get('DCTTS_AB_LIB'): _L0.LIB_PATH = os.environ['DCTTS_AB_LIB']
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
from oracle.incremental_ref import incremental_decode_v3
W = synthetic_weights(hp, seed=1234, perturb=True)
others = [Engine(W, hp) for _ in range(int(os.environ.get("NOTHER", "1")))]
Lo = torch.from_numpy(synthetic_text(hp, B=32, seed=5)).cuda()
for e in others: e.synthesize(Lo)
torch.cuda.synchronize()
bad = runs = 0
for T in [int(x) for x in os.environ.get("TS", "100").split(",")]:
    h = hp.replace(max_T=T)
    eng = Engine(W, h)
    for B in [int(x) for x in os.environ.get("BS", "3,5,32").split(",")]:
        Lh = synthetic_text(h, B=B, seed=21 + B + T)
        Yr, trajr = incremental_decode_v3(Lh, W, h, np.float32)
        L = torch.from_numpy(Lh).cuda()
        for rep in range(int(os.environ.get("REPS", "2"))):
            for graph in (0, 1):
                eng.set_decode_graph(graph)
                if others and rep % 2: others[0].ssrn(torch.rand(8, 210, 80, device="cuda"))
                Y, mx = eng.text2mel(L); eng.synchronize()
                e = np.abs(Y.cpu().numpy() - Yr); ok = (mx.cpu().numpy() == trajr).all()
                runs += 1
                if e.max() > 1e-3 or not ok:
                    bad += 1
                    fr = np.argwhere(e.max(axis=2) > 1e-3)
                    print(f"T={T} B={B} rep={rep} graph={graph}: max err {e.max():.3e} traj_ok={ok}; first bad (utt, frame): {fr[:6].tolist()} n_bad_rows={len(fr)}")
    eng.close()
print(f"CHAIN_TAIL={os.environ['DCTTS_CHAIN_TAIL']}: bad runs: {bad} of {runs}")
