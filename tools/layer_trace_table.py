"""Per-launch table of tools/layer_trace.py's second TextEnc + SSRN pass from a rocprofv3 --kernel-trace directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "dctts" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
print("# duration | workgroups x threads | kernel      (TextEnc: 16 launches, then SSRN; B = 32, N = 180, T = 210)")
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    wg = int(r["Workgroup_Size_X"])
    print(f'{d:9.1f} us  {int(r["Grid_Size_X"]) // wg:5d} x {wg:4d}  {r["Kernel_Name"].replace("void dctts::", "").replace("(dctts::ConvParams)", "").replace("(dctts::ConvParams, int)", "")}')
