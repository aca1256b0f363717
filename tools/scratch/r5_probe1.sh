# round 5, first probe: how long are the chain / side pieces in the two chain forms (DCTTS_CHAIN_TAIL=2: xtail merged form, the default; 1: xgroup + xmlp + xcone over five layers)
set -u
R=$PWD; OUT=$R/gpurun_out/p1; mkdir -p $OUT
for tail in 2 1 5; do
  echo "== DCTTS_CHAIN_TAIL=$tail" | tee -a $OUT/probe1.txt
  for rep in 1 2; do DCTTS_CHAIN_TAIL=$tail GM=0 HP=1 timeout 120 python tools/decode_time.py 2>&1 | grep -E "text2mel|rror" | tee -a $OUT/probe1.txt; done
  DCTTS_CHAIN_TAIL=$tail DCTTS_PIECETIME=150 GM=0 HP=1 NREP=1 timeout 120 python tools/decode_time.py 2>&1 | grep -E "frame 15[0-7]" | tail -8 | tee -a $OUT/probe1.txt
done
