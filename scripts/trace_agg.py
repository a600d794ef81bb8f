"""Aggregate a rocprofv3 kernel trace (csv) of the last score call by (kernel, grid) (development aid)."""
import csv, sys
def agg(path, marker="ncsn_pack_kernel"):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    sel = rows[idx[-1]:] if idx else rows
    tot = {}
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        g = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
        t = tot.setdefault((n, g), [0, 0.0]); t[0] += 1; t[1] += (e - s) / 1e3
    return tot
if __name__ == "__main__":
    ts = [agg(p) for p in sys.argv[1:]]
    keys = sorted(ts[0], key=lambda k: -ts[0][k][1])
    for k in keys[:int(__import__("os").environ.get("TOP", "18"))]:
        print(f"{k[0][:46]:46s} grid {k[1]:5d} x{ts[0][k][0]:3d} " + " ".join(f"{t.get(k,[1,0])[1]/max(t.get(k,[1,0])[0],1):8.1f}" for t in ts))
    print("total us:", [round(sum(v[1] for v in t.values())) for t in ts])
