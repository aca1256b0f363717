#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2l; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $R/tools/layer_trace.py > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_trace.csv" | head -2
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "dctts" in r["Kernel_Name"]]
half = len(rows) // 2
for r in rows[half:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f'{d:9.1f} us  grid {r["Grid_Size_X"]:>8}x{r["Grid_Size_Y"]}  wg {r["Workgroup_Size_X"]}  {r["Kernel_Name"][:90]}')
PY
