"""Attribute the FETCH_SIZE rows of `rocprofv3 --pmc FETCH_SIZE -- python3 tools/ab_lib_versions.py ... --pmc` to the libraries by
dispatch order (warm-up and every round launch library by library; each library launches two TN and two NN data passes).

    python tools/ab_lib_pmc.py <counter_collection.csv> tag1,tag2,...
"""
import csv
import sys
from collections import defaultdict

path, tags = sys.argv[1], sys.argv[2].split(",")
rows = defaultdict(list)   # symbol class -> [(dispatch id, value)]
with open(path) as f:
    for r in csv.DictReader(f):
        if r.get("Counter_Name") != "FETCH_SIZE":
            continue
        name = r["Kernel_Name"]
        if "gemm_kernel<1, 256, 0, 4" in name or "gemm_kernel<1,256,0,4" in name:
            cls = "TN"
        elif "gemm_kernel<0, 256, 0, 4" in name or "gemm_kernel<0,256,0,4" in name:
            cls = "NN"
        else:
            continue
        rows[cls].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for cls in ("TN", "NN"):
    seq = [v for _, v in sorted(rows[cls])]
    per = defaultdict(list)
    for i, v in enumerate(seq):
        per[tags[(i // 2) % len(tags)]].append(v)
    for t in tags:
        a = per[t]
        if a:
            gb = [x * 1024 * 2 / 1e9 for x in a]   # KiB -> bytes, x2: gfx950 correction of wide streaming reads (MI355X_MICROARCH.md, HBM)
            print("%s %-10s FETCH %.2f GB mean (min %.2f max %.2f) over %d launches" % (cls, t, sum(gb) / len(gb), min(gb), max(gb), len(gb)))
