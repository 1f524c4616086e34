"""Summarises the kernels of the LAST apply of a rocprofv3 --kernel-trace csv (dev helper)."""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "hifamd" in r["Kernel_Name"]]
napply = int(sys.argv[2])
verbose = len(sys.argv) > 3
n = len(rows) // napply
agg = collections.OrderedDict()
tot = 0.0
for r in rows[-n:]:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += dur
    name = r["Kernel_Name"].split("(")[0].replace("void hifamd::", "")
    blocks = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    key = name + (" [1 WG]" if blocks == 1 else "")
    a = agg.setdefault(key, [0, 0.0, 1e30, 0.0])
    a[0] += 1
    a[1] += dur
    a[2] = min(a[2], dur)
    a[3] = max(a[3], dur)
    if verbose:
        print(f"{name:40s} blocks={blocks:5d} {dur:10.1f} us")
for k, a in agg.items():
    print(f"{k:45s} calls={a[0]:5d} total={a[1] / 1e3:8.3f} ms avg={a[1] / a[0]:8.1f} us min={a[2]:7.1f} max={a[3]:8.1f}")
print("sum of kernel time per apply: %.3f ms" % (tot / 1e3))
