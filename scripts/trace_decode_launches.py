import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last decode: find the last conv_out1_rows kernel, go back to previous one
idx = [i for i, r in enumerate(rows) if "conv_out1" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
tot = 0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{d:9.1f} us  grid {r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size','?'):>9}  {n[:90]}")
print("sum", tot)
