# Where a decode frame goes (round 5: two launches per frame, xchain_kernel on the chain and xcone_kernel on the side stream): per-kernel averages of a decode-only kernel trace (eager launches, T = ${PT:-160}: the printed frame is at 3/4 of the decode, i.e. with full cones) + the timeline of one steady-state frame
set -u
R=$PWD; OUT=$R/gpurun_out/dprobe; mkdir -p $OUT; rm -rf $OUT/trace
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/tools/decode_only.py ${PT:-160} > $OUT/trace.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(f'{int(r["Calls"]):6d} calls  avg {float(r["AverageNs"])/1e3:8.2f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms  {r["Percentage"]:>6}%  {r["Name"][:100]}')
PY
python $R/tools/trace_timeline.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) xcone_kernel 10
