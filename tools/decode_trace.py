"""In-kernel wall-clock stamps of one frame's chain3_kernel<LN_HC,HC> launches (decode mode 3) or rowchain_kernel's section timers
(DM=4).  DCTTS_TRACE / DCTTS_TRACE_FILE are read when the engine is created."""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
out = os.environ.setdefault('DCTTS_TRACE_FILE', 'gpurun_out/decode_trace.txt')
os.environ.setdefault('DCTTS_TRACE', '150')
import dc_tts_amd._lib as _L0
if os.environ.get("DCTTS_AB_LIB"): _L0.LIB_PATH = os.environ["DCTTS_AB_LIB"]
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
eng = Engine(synthetic_weights(hp), hp, decode_graph=int(os.environ.get("GM", "0")))
eng.set_decode_mode(int(os.environ.get("DM", "3")))
L = torch.from_numpy(synthetic_text(hp, B=int(os.environ.get("TB", "32")))).cuda()
if os.environ.get("HP"): torch.cuda.set_stream(torch.cuda.Stream(priority=-1))      # a high-priority caller's stream: the decode's chain runs on it directly
eng.text2mel(L); torch.cuda.synchronize()
eng.text2mel(L); torch.cuda.synchronize()
print(open(out).read())
