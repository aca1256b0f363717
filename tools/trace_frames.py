"""In-kernel stamps (tools/decode_trace.py) of SEVERAL frames of one process's decodes: is a slow layer of the traced frame slow in every frame, or a sample?
FRAMES=110,130,150,170,190 (default), TB = batch, HP=1 = called from a high-priority stream.  Uses the run-time hook dctts_debug_set_trace."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
eng = Engine(synthetic_weights(hp), hp, decode_graph=0)
L = torch.from_numpy(synthetic_text(hp, B=int(os.environ.get("TB", "32")))).cuda()
if os.environ.get("HP", "1") == "1": torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
eng.text2mel(L); torch.cuda.synchronize()
path = os.environ.get("OUT", "/tmp/trace_frames.txt")
for f in [int(x) for x in os.environ.get("FRAMES", "110,130,150,170,190").split(",")]:
    for rep in range(int(os.environ.get("REPS", "2"))):
        eng.debug_set_trace(f, path)
        eng.text2mel(L); torch.cuda.synchronize()
        eng.debug_set_trace(-1, path)
        print(f"=== frame {f} (decode {rep})")
        for line in open(path).read().splitlines():
            if not line.startswith("#"): print(line)
