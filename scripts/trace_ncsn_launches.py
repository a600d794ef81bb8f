"""Development: the kernel sequence of the LAST NCSN++ score call in a rocprofv3 --kernel-trace CSV (scripts/ncsn_trace_workload.py)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "ncsn_output_kernel" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
tot = 0
t_prev = None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = (e - s) / 1e3
    gap = 0 if t_prev is None else (s - t_prev) / 1e3
    t_prev = e
    tot += d
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{d:8.1f} us (+{gap:5.1f})  wg {int(r['Grid_Size_X'])//max(int(r['Workgroup_Size_X']),1):>6}  {n[:70]}")
print("sum of kernels", tot, "us; wall", (int(rows[b-1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
