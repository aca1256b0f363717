"""Print the per-dispatch timeline of one steady-state decode frame from a rocprofv3 kernel trace CSV (measurement helper)."""
import csv, sys
path = sys.argv[1]; anchor = sys.argv[2] if len(sys.argv) > 2 else "hbulk_group"; n = int(sys.argv[3]) if len(sys.argv) > 3 else 48
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Queue_Id']),
                     int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y'])))
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
i0 = idx[len(idx) * 3 // 4]
t0 = rows[i0][0]
for r in rows[i0 - 2:i0 + n]:
    nm = r[2].replace('void dctts::', '').replace('dctts::', '')[:42]
    print(f"{(r[0]-t0)/1000:9.2f} {(r[1]-t0)/1000:9.2f} dur {(r[1]-r[0])/1000:7.2f} q{r[3]} wg{r[4]:5d}x{r[5]} {nm}")
