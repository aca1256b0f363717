set -u
R=$PWD
for rep in 1 2; do GM=0 HP=1 timeout 120 python tools/decode_time.py 2>&1 | grep -E "text2mel|rror"; done
timeout 200 python tools/scratch/te_reps.py 2>&1 | grep TextEnc
rocm-smi --showclocks 2>/dev/null | head -20
rocm-smi --showmemuse --showperflevel --showcomputepartition --showmemorypartition 2>/dev/null | head -30
