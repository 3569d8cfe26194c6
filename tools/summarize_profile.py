#!/usr/bin/env python
"""Condense a tools/profile.sh output directory (gpurun_out/prof_<tag>) into the tracked files under profiles/:
  <round>_kernel_stats.csv   rocprofv3 --kernel-trace --stats rows of the ansfm kernels
  <round>_pmc_summary.json   per-kernel averages of every PMC counter (each counter group = its own run)
  pmc_traffic.json           HBM bytes per k_ck_overlap launch (bench.py's roofline.traffic)
usage: python tools/summarize_profile.py <tag> <round>        e.g.  r01d r01
The CSV holds FETCH_SIZE / WRITE_SIZE in KiB (x1024 here).  On gfx950 FETCH_SIZE reports half the bytes of
coalesced streaming reads (MI355X_MICROARCH.md, HBM section); tools/calib/fetch_calib.hip confirms the factor for
the 8-byte-per-lane loads this kernel issues (1 GiB read -> 0.500 GiB reported, plain and non-temporal alike), so
the doubled value is the one used; the raw value is kept beside it."""
import csv, glob, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    out = os.path.join(ROOT, "profiles")
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(stats)) if "ansfm" in r["Name"]]
    with open(os.path.join(out, rnd + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader(); w.writerows(rows)
    summ = defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            for r in csv.DictReader(open(f)):
                if "ansfm" not in r["Kernel_Name"]:
                    continue
                name = r["Kernel_Name"].split("(")[0]
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, cs in acc.items():
                for c, v in cs.items():
                    summ[k][c] = sum(v) / len(v)
    json.dump(summ, open(os.path.join(out, rnd + "_pmc_summary.json"), "w"), indent=1)
    ov = [k for k in summ if "k_ck_overlap<" in k]
    if ov:
        s = summ[ov[0]]
        fetch = s["FETCH_SIZE"] * 1024.0
        write = s["WRITE_SIZE"] * 1024.0
        traffic = {"W10000_G20_S8_L100": {
            "ck_overlap_hbm_bytes_per_launch": 2 * fetch + write,
            "fetch_bytes_raw": fetch, "fetch_bytes_x2_gfx950_correction": 2 * fetch, "write_bytes": write,
            "valu_insts_per_launch": s.get("SQ_INSTS_VALU"), "lds_insts_per_launch": s.get("SQ_INSTS_LDS"),
            "profile": tag,
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile.sh), average per "
                    "k_ck_overlap launch; FETCH doubled per MI355X_MICROARCH.md HBM section, factor confirmed for "
                    "8 B/lane loads with tools/calib/fetch_calib.hip (1 GiB read -> 0.500 GiB reported)"}}
        json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    for r in rows[:6]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"])
    if ov:
        print({k: round(v / 1e9, 3) for k, v in traffic["W10000_G20_S8_L100"].items() if isinstance(v, float)})


if __name__ == "__main__":
    main()
