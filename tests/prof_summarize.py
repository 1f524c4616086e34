"""Turns the rocprofv3 outputs of a profiled bench.py run (gpurun_out/<tag>/...) into the small,
committed summaries under profiles/.  Usage: python tests/prof_summarize.py gpurun_out/r01 profiles/r01
PMC units: FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B
(MI355X_MICROARCH.md "HBM"), so HBM read bytes = 2 * FETCH_SIZE * 1024 for wide coalesced reads."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)


def one(pattern):
    g = glob.glob(os.path.join(src, pattern))
    return g[0] if g else None


ks = one("stats/*/*kernel_stats.csv")
if ks:
    shutil.copy(ks, dst + "_kernel_stats.csv")
bj = os.path.join(src, "bench_stats.json")
if os.path.exists(bj):
    shutil.copy(bj, dst + "_bench_under_rocprof.json")


def pmc(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if "hifamd" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    napply = max(1, sum(1 for r in rows if "k_scatter_scale" in r["Kernel_Name"]) // 6)  # 6 sparse levels -> 6 S7 per apply
    by = collections.OrderedDict()
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void hifamd::", "").replace("hifamd::", "")
        by[k] = by.get(k, 0.0) + float(r["Counter_Value"])
    return napply, {k: v / napply for k, v in by.items()}


out = {}
f, w = one("pmc_fetch/*/*counter_collection.csv"), one("pmc_write/*/*counter_collection.csv")
if f and w:
    nf, bf = pmc(f, "FETCH_SIZE")
    nw, bw = pmc(w, "WRITE_SIZE")
    tf, tw = sum(bf.values()), sum(bw.values())
    out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --secondary 0",
           "applies_profiled": [nf, nw], "unit": "KiB per 64-RHS apply (default-parameter hierarchy)",
           "FETCH_SIZE_KiB": tf, "WRITE_SIZE_KiB": tw,
           "hbm_bytes_per_apply_corrected": (2.0 * tf + tw) * 1024.0,
           "correction": "read bytes = 2 * FETCH_SIZE * 1024 (gfx950 counts 128-B requests at 64 B); "
                         "WRITE_SIZE exact (k_gather_scale: 499,709 KiB reported vs 499,709 KiB written)",
           "per_kernel_KiB": {k: {"FETCH_SIZE": bf.get(k, 0.0), "WRITE_SIZE": bw.get(k, 0.0)} for k in bf}}
    json.dump(out, open(dst + "_pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
