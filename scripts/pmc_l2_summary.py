"""Per-kernel L2 (TCC) hit rate from the two passes of scripts/pmc_l2_collect.sh."""
import csv, glob, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_l2"
tot = {}
for c in ("TCC_HIT_sum", "TCC_MISS_sum"):
    f = glob.glob(f"{root}/{c}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:64]
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
    tot[c] = acc
for k in sorted(tot["TCC_HIT_sum"], key=lambda k: -tot["TCC_HIT_sum"][k][0] - tot["TCC_MISS_sum"][k][0]):
    h, n = tot["TCC_HIT_sum"][k]
    m = tot["TCC_MISS_sum"][k][0]
    if h + m < 1e5:
        continue
    print(f"{k:66s} launches {n:4d}  req/launch {(h+m)/n/1e3:9.1f} k  hit rate {h/(h+m):.3f}  miss bytes/launch (128 B) {m/n*128/1e6:8.2f} MB")
