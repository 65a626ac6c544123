"""Condense rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py` (C4) into the two
profiles/*_pmc_*.csv excerpts (cmfk kernels only) and profiles/traffic_c4.json.

usage: python tools/pmc_to_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [round-tag]

Corrections as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE is in KiB and tallies
the 128-byte requests of wide coalesced reads as 64 bytes (x2); WRITE_SIZE is in KiB and needs no factor."""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fetch_csv, write_csv = sys.argv[1], sys.argv[2]
tag = sys.argv[3] if len(sys.argv) > 3 else "r01"
m = d = p = 65536; k = 256
ALG = 4.0 * (m * d + m * k + d * k)  # one data matrix once + the factor operand + the output


def rows(path, counter):
    out = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "cmfk::" in r["Kernel_Name"]:
            out.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"]),
                        int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def excerpt(rs, counter, path):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value", "Duration_ns"])
        for disp, name, grid, val, dur in rs:
            w.writerow([disp, name, grid, counter, "%.6f" % val, dur])


fr, wr = rows(fetch_csv, "FETCH_SIZE"), rows(write_csv, "WRITE_SIZE")
excerpt(fr, "FETCH_SIZE", os.path.join(ROOT, "profiles", "%s_c4_n1_pmc_FETCH_SIZE.csv" % tag))
excerpt(wr, "WRITE_SIZE", os.path.join(ROOT, "profiles", "%s_c4_n1_pmc_WRITE_SIZE.csv" % tag))
out = {"_provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 2 "
                      "--warmup 1 --no-cpu-baseline` (C4, 1 GPU, %s), condensed by tools/pmc_to_traffic.py. Average over "
                      "every launch of the data-pass kernel symbol: FETCH_SIZE KiB x1024 x2 (gfx950 tallies the 128-B "
                      "requests of wide coalesced reads at 64 B: MI355X_MICROARCH.md, HBM section) + WRITE_SIZE KiB x1024." % tag,
       "unit": "bytes per launch"}
for key, sym in (("gemm_tn", "gemm_kernel<1, 256, 0,"), ("gemm_nn", "gemm_kernel<0, 256, 0,")):
    f = [v for _, n, _, v, _ in fr if sym in n]
    w = [v for _, n, _, v, _ in wr if sym in n]
    if not f or not w:
        continue
    fa, wa = sum(f) / len(f), sum(w) / len(w)
    out[key] = fa * 1024 * 2 + wa * 1024
    out[key + "_detail"] = {"fetch_raw_KiB": fa, "write_KiB": wa, "launches": len(f), "algorithmic_bytes": ALG}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_c4.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
