for rep in 1 2; do GM=0 HP=1 timeout 120 python tools/decode_time.py 2>&1 | grep -E "text2mel|rror"; done
echo "== rows skipped (timing only, wrong results)"
for rep in 1 2; do DCTTS_TIMING_SKIP_ROWS=1 GM=0 HP=1 timeout 120 python tools/decode_time.py 2>&1 | grep -E "text2mel|rror"; done
DCTTS_TIMING_SKIP_ROWS=1 DCTTS_PIECETIME=150 GM=0 HP=1 NREP=1 timeout 120 python tools/decode_time.py 2>&1 | grep -E "frame 15[0-7]" | tail -4
