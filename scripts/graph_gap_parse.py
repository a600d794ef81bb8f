"""Kernel-to-kernel gaps of the last captured score calls in a rocprofv3 kernel trace of scripts/graph_gap_workload.py."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-1500:]
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
gaps = collections.defaultdict(list)
durs = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(g)
    durs[short(a["Kernel_Name"])].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
print("kernel durations (us): median over the last 1500 launches")
for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"  {k:46s} n {len(v):4d}  median {v[len(v)//2]:7.2f}  sum {sum(v)/1e3:7.2f} ms")
print("gaps end -> next start (us): median")
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:14]:
    v.sort()
    print(f"  {k[0]:44s} -> {k[1]:44s} n {len(v):4d}  median {v[len(v)//2]:6.2f}  p90 {v[int(len(v)*0.9)]:6.2f}")
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"wall {1e-3*(t1-t0):.1f} us, kernels busy {1e-3*busy:.1f} us, idle share {(1 - busy/(t1-t0)):.3f}")
