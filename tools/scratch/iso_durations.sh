# Durations of the decode kernels when nothing runs beside them (counter collection serialises the dispatches of both streams)
set -u
R=$PWD; OUT=$R/gpurun_out/iso; mkdir -p $OUT; rm -rf $OUT/t
cd /tmp; export TMPDIR=/tmp
DCTTS_SYNC_VALUES=0 DM=3 GM=0 timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv -d $OUT/t -- python $R/tools/decode_only.py 160 > $OUT/log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name'][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 100: print(f"{k:62s} n={len(v):5d}  mean of last quarter {sum(v[-len(v)//4:])/(len(v)//4):8.2f} us")
PY
