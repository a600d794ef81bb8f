"""Aggregate rocprofv3 --pmc counter CSVs (separate FETCH_SIZE / WRITE_SIZE passes over scripts/pmc_workload.py with
PART=score and PART=decode) into HBM bytes per launch of each kernel, per DiT score call and per decode, and write
profiles/<tag>.json.  gfx950 corrections per MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies 64 B per 128-B
request of wide coalesced reads -> doubled; WRITE_SIZE exact; both in KiB.
usage: pmc_summary.py <score_fetch_dir> <score_write_dir> <decode_fetch_dir|-> <decode_write_dir|-> <out.json> [commit] [note]"""
import collections, csv, glob, json, re, sys

SETUP = ("pack_weight", "wn_scale", "snake_params", "packed_row_sum", "bias_plus_wbeta", "pack_bias", "randn", "fill", "copyBuffer",
         "distribution", "elementwise", "rope_tables")
# bench.py call site -> substring of the rocprof kernel name at the C2 shape, headline (fp16) mode
SITES = {"dit.ff_in": "igemm_panel_kernel<1, 1, 4, 2, 64, 17, 2, 4>", "dit.ff_out": "igemm_panel_kernel<1, 1, 4, 3, 64, 9, 0, 4>",
         "dit.qkv_attention": "qkv_attention_kernel<1, 3, 3>", "dit.attn_out": "igemm_panel_kernel<1, 1, 2, 4, 64, 5, 1, 4>",
         "dit.residual_norm": "residual_norm_row_kernel<4>", "vae.residual_unit_fused": "ru_fused2_kernel"}
NCSN_SITES = {"ncsnpp.conv3x3_level0": "igemm_halo3x3_kernel<1, 256, 4>", "ncsnpp.conv3x3_level1": "igemm_halo3x3_kernel<1, 128, 1>",
              "ncsnpp.conv3x3_level2": "igemm2_kernel<1, 1, 128, 64, 3, 64, 0, 32>", "ncsnpp.gn_apply": "gn_apply_kernel"}

def load(dirname, counter):
    per = collections.defaultdict(list)
    for f in glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
            name = re.sub(r"\(.*", "", name).strip()
            if any(s in name for s in SETUP):
                continue
            per[name].append(float(r["Counter_Value"]))
    return per


def part(fetch_dir, write_dir, calls):
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    rows, total = {}, 0.0
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, []), write.get(name, [])
        n = max(len(f), len(w))
        hbm = (2 * sum(f) + sum(w)) * 1024
        rows[name] = {"launches": n // calls, "hbm_bytes_per_launch": hbm / max(n, 1),
                      "fetch_kib_raw_per_launch": sum(f) / max(len(f), 1), "write_kib_per_launch": sum(w) / max(len(w), 1)}
        total += hbm / calls
    return rows, total


score, score_total = part(sys.argv[1], sys.argv[2], 2)
decode, decode_total = part(sys.argv[3], sys.argv[4], 1) if sys.argv[3] != "-" else ({}, None)
sites = {}
is_ncsn = any("halo3x3" in n for n in score)
for site, pat in (NCSN_SITES if is_ncsn else SITES).items():
    for table in (score, decode):
        for name, r in table.items():
            if pat in name:
                sites[site] = {"kernel": name, **r}
    if site not in sites and not (site.startswith("vae.") and not decode):
        # a tile configuration changed and the table above was not updated: fail instead of silently dropping the site
        sys.exit(f"pmc_summary: no kernel matches call site {site} (pattern {pat!r}); kernels seen: {sorted(score)[:40]}")
out = {"commit": sys.argv[6] if len(sys.argv) > 6 else None,
       "note": (sys.argv[7] + "; " if len(sys.argv) > 7 else "") +
               "batch 64, T=32, fp16 headline mode; hbm = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md",
       "score_call_hbm_bytes": score_total, "decode_hbm_bytes": decode_total, "sites": sites,
       "score_kernels": score, "decode_kernels": decode}
json.dump(out, open(sys.argv[5], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("score_kernels", "decode_kernels")}, indent=1))
