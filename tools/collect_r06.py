"""(Round 6: adapted from tools/collect_r05.py.)  Copy what tools/refresh_r06.sh left under gpurun_out/r06/ into profiles/ (tracked): bench lines, rocprofv3 kernel
stats, the condensed PMC collections, and the traffic / MFMA-utilisation summaries that bench.py and DESIGN.md cite.

Corrections as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE is in KiB and tallies the 128-byte
requests of wide (16 B per lane) reads as 64 bytes (x2); WRITE_SIZE is in KiB and needs no factor."""
import csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r06")
DST = os.path.join(ROOT, "profiles")
TAG = "r06"


def counters(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append((float(r["Counter_Value"]), int(r["Duration_ns"])))
    return out


LINES_ONLY = len(sys.argv) > 1 and sys.argv[1] == "lines"   # after `bash tools/refresh_r06.sh lines`: only the bench lines are new
C3_ONLY = len(sys.argv) > 1 and sys.argv[1] == "c3"         # after `bash tools/refresh_r06b.sh`: the C3 family (row kernel changed late in the round)
def copy(src, dst):
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, dst)
    else:
        print("missing:", src)


for w in (("c3", "c3r", "c3x") if C3_ONLY else ("c4", "c2", "c3", "c3r", "c3x", "c5", "c5z", "c5zs", "c5l", "c5l_l2x10")):
    copy(os.path.join(SRC, "%s.json" % w), os.path.join(DST, "%s_%s_n1_bench.json" % (TAG, w)))
for f_, t_ in (("c4_tol.json", "%s_c4_n1_tol_bench.json"), ("c4_notol40.json", "%s_c4_n1_bench_40_steps.json"),
               ("c3_norefine.json", "%s_c3_n1_bench_refine_rows_0.json"), ("c3_r05_clamp.json", "%s_c3_n1_bench_round5_clamp_path.json"),
               ("c3_no_early_exit.json", "%s_c3_n1_bench_no_early_exit.json"),
               ("c3_rowsym3.json", "%s_c3_n1_bench_row_symmetric_3.json"), ("c3_norank1.json", "%s_c3_n1_bench_rank1_clamp_0.json"), ("c3x_rowsym3.json", "%s_c3x_n1_bench_row_symmetric_3.json")):
    if C3_ONLY and not f_.startswith("c3"):
        continue
    copy(os.path.join(SRC, f_), os.path.join(DST, t_ % TAG))
if LINES_ONLY:
    sys.exit(0)
for w in (("c3", "c3x") if C3_ONLY else ("c4", "c2", "c3", "c3x", "c5")):
    copy(os.path.join(SRC, "prof_%s" % w, "%s_kernel_stats.csv" % w), os.path.join(DST, "%s_%s_n1_kernel_stats.csv" % (TAG, w)))
for w in (("c3", "c3x") if C3_ONLY else ("c4", "c3", "c3x", "c5", "c5z", "c5zs")):
    for cnt, d, f in (("FETCH_SIZE", "fetch", "f"), ("WRITE_SIZE", "write", "w")):
        shutil.copy(os.path.join(SRC, "pmc_%s_%s" % (w, d), "%s_cmfk_counters.csv" % f), os.path.join(DST, "%s_%s_n1_pmc_%s.csv" % (TAG, w, cnt)))
if not C3_ONLY:
    shutil.copy(os.path.join(SRC, "pmc_mfma_c4", "m_cmfk_counters.csv"), os.path.join(DST, "%s_c4_n1_pmc_MFMA.csv" % TAG))
for w in ("c3", "c3x"):   # (refresh_r06b.sh) matrix-pipe busy cycles of the row kernel; LDS counters of the C3X logit launches
    copy(os.path.join(SRC, "pmc_mfma_%s" % w, "m_cmfk_counters.csv"), os.path.join(DST, "%s_%s_n1_pmc_MFMA.csv" % (TAG, w)))
copy(os.path.join(SRC, "pmc_lds_c3x", "l_cmfk_counters.csv"), os.path.join(DST, "%s_c3x_n1_pmc_LDS.csv" % TAG))
copy(os.path.join(SRC, "pmc_lds_c3x_sym3", "l_cmfk_counters.csv"), os.path.join(DST, "%s_c3x_n1_pmc_LDS_row_symmetric_3.csv" % TAG))

# ---- C4: data-pass GEMMs
m = d = 65536; k = 256
# (in `c3` mode the blocks of the other workloads recompute their summaries from the PMC files already under profiles/: same numbers)
alg = 4.0 * (m * d + m * k + d * k)
f = counters(os.path.join(DST, "%s_c4_n1_pmc_FETCH_SIZE.csv" % TAG)); wr = counters(os.path.join(DST, "%s_c4_n1_pmc_WRITE_SIZE.csv" % TAG))
out = {"_provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/refresh_r06.sh) of `python3 bench.py --steps 2 "
                      "--warmup 1 --no-cpu-baseline` (C4, 1 GPU, %s), condensed by tools/collect_r06.py.  Average over every launch of the "
                      "data-pass kernel symbol: FETCH_SIZE KiB x1024 x2 (gfx950 correction) + WRITE_SIZE KiB x1024." % TAG, "unit": "bytes per launch"}
for key, sym in (("gemm_tn", "gemm_kernel<1, 256, 0,"), ("gemm_nn", "gemm_kernel<0, 256, 0,")):
    fv = [v for (n, c), vs in f.items() if sym in n and c == "FETCH_SIZE" for v, _ in vs]
    wv = [v for (n, c), vs in wr.items() if sym in n and c == "WRITE_SIZE" for v, _ in vs]
    if fv and wv:
        fa, wa = sum(fv) / len(fv), sum(wv) / len(wv)
        out[key] = fa * 1024 * 2 + wa * 1024
        out[key + "_detail"] = {"fetch_raw_KiB": fa, "write_KiB": wa, "launches": len(fv), "algorithmic_bytes": alg}
json.dump(out, open(os.path.join(DST, "traffic_c4.json"), "w"), indent=1)
print(json.dumps(out, indent=1))


# (C2: no PMC pass in round 6 -- profiles/traffic_c2.json stays the round-5 measurement; the data-pass kernel did not change)

# ---- C3 / C3Z / C3X: the fused row kernels (class launches: linear sampled sides with shared partial sums; logit launches: one row each)
for W3 in ("c3", "c3x"):
    f = counters(os.path.join(DST, "%s_%s_n1_pmc_FETCH_SIZE.csv" % (TAG, W3))); wr = counters(os.path.join(DST, "%s_%s_n1_pmc_WRITE_SIZE.csv" % (TAG, W3)))
    out = {"_provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/refresh_r06.sh) of `python3 bench.py --workload %s "
                          "--steps 1 --warmup 1 --no-cpu-baseline` (%s).  FETCH_SIZE KiB x1024 x2 (gfx950 correction) + WRITE_SIZE KiB x1024, averaged "
                          "over the launches of the row kernel.  Bytes past L2, Infinity-Cache hits included." % (W3, TAG), "unit": "bytes per launch"}
    allf, allw = [], []
    for key, pred in (("class_launches", lambda n: "row_hess_kernel" in n and ("3, 1>" in n or "4, 1>" in n)),
                      ("logit_launches", lambda n: "row_hess_kernel" in n and not ("3, 1>" in n or "4, 1>" in n))):
        fv = [v for (n, c), vs in f.items() if pred(n) and c == "FETCH_SIZE" for v, _ in vs]
        wv = [v for (n, c), vs in wr.items() if pred(n) and c == "WRITE_SIZE" for v, _ in vs]
        if fv and wv:
            out[key] = {"launches": len(fv), "fetch_bytes_per_launch": sum(fv) / len(fv) * 2048, "write_bytes_per_launch": sum(wv) / len(wv) * 1024}
            allf += fv; allw += wv
    if allf and allw:
        out["rowhess"] = sum(allf) / len(allf) * 2048 + sum(allw) / len(allw) * 1024
    json.dump(out, open(os.path.join(DST, "traffic_%s.json" % W3), "w"), indent=1)
    print(json.dumps(out, indent=1))

# ---- C5 / C5Z / C5ZS: blocked SpMM (two launches per iteration: X V gathers rows of V, X^T U rows of U)
for W5 in ("c5", "c5z", "c5zs"):
    pf, pw = os.path.join(DST, "%s_%s_n1_pmc_FETCH_SIZE.csv" % (TAG, W5)), os.path.join(DST, "%s_%s_n1_pmc_WRITE_SIZE.csv" % (TAG, W5))
    if not (os.path.exists(pf) and os.path.exists(pw)):
        print("missing PMC passes of", W5)
        continue
    f = counters(pf); wr = counters(pw)
    fv = [(v, dur) for (n, c), vs in f.items() if "spmm_blocked_kernel" in n and c == "FETCH_SIZE" for v, dur in vs]
    wv = [v for (n, c), vs in wr.items() if "spmm_blocked_kernel" in n and c == "WRITE_SIZE" for v, _ in vs]
    if not fv or not wv:
        print("no spmm_blocked_kernel launches in the PMC passes of", W5)
        continue
    nnz = 1e8; kp = 256
    gathered = nnz * (kp * 4 + 16)
    fa, wa = sum(v for v, _ in fv) / len(fv), sum(wv) / len(wv)
    dur = sum(dd for _, dd in fv) / len(fv)
    out = {"_provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/refresh_r06.sh) of `python3 bench.py --workload %s "
                          "--steps 2 --warmup 1 --no-cpu-baseline` (%s).  FETCH_SIZE KiB x1024 x2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE "
                          "KiB x1024, averaged over the launches of spmm_blocked_kernel<4> (X V and X^T U alternate).  FETCH_SIZE counts what the L2s "
                          "request from the fabric: Infinity-Cache hits included, L2 hits not." % (W5, TAG),
           "unit": "bytes per launch", "spmm": fa * 1024 * 2 + wa * 1024,
           "spmm_detail": {"fetch_raw_KiB": fa, "write_KiB": wa, "launches": len(fv), "avg_duration_ms_under_pmc": dur / 1e6,
                           "gathered_bytes_algorithmic": gathered, "l2_hit_fraction_of_gathers": 1.0 - (fa * 1024 * 2) / gathered,
                           "per_launch_fetch_raw_KiB": [v for v, _ in fv]}}
    json.dump(out, open(os.path.join(DST, "traffic_%s.json" % W5), "w"), indent=1)
    print(json.dumps(out, indent=1))

# ---- MFMA utilisation of the C4 data passes
mm = counters(os.path.join(DST, "%s_c4_n1_pmc_MFMA.csv" % TAG))
util = {"_provenance": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE of the C4 bench command (%s); utilisation = "
                       "MFMA busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); effective clock = GRBM_GUI_ACTIVE / 8 / duration" % TAG}
for key, sym in (("gemm_tn", "gemm_kernel<1, 256, 0,"), ("gemm_nn", "gemm_kernel<0, 256, 0,")):
    busy = [v for (n, c), vs in mm.items() if sym in n and c == "SQ_VALU_MFMA_BUSY_CYCLES" for v, _ in vs]
    act = [(v, dd) for (n, c), vs in mm.items() if sym in n and c == "GRBM_GUI_ACTIVE" for v, dd in vs]
    if busy and act:
        b = sum(busy) / len(busy); a = sum(v for v, _ in act) / len(act); dd = sum(x for _, x in act) / len(act)
        util[key] = {"mfma_busy_cycles": b, "grbm_gui_active": a, "utilisation": b / (1024.0 * a / 8.0), "effective_clock_GHz": a / 8.0 / dd, "launches": len(busy)}
# ---- ... and of the row kernel (tools/refresh_r06b.sh): class launches (two workgroups per CU) and logit launches (one) of C3, logit launches of C3X
for w in ("c3", "c3x"):
    pth = os.path.join(DST, "%s_%s_n1_pmc_MFMA.csv" % (TAG, w))
    if not os.path.exists(pth):
        continue
    mm = counters(pth)
    for key, pred in (("%s_rowhess_class_launches" % w, lambda n: "row_hess_kernel" in n and "4, 1>" in n),
                      ("%s_rowhess_logit_launches" % w, lambda n: "row_hess_kernel" in n and "4, 0>" in n)):
        busy = [v for (n, c), vs in mm.items() if pred(n) and c == "SQ_VALU_MFMA_BUSY_CYCLES" for v, _ in vs]
        act = [(v, dd) for (n, c), vs in mm.items() if pred(n) and c == "GRBM_GUI_ACTIVE" for v, dd in vs]
        if busy and act:
            b = sum(busy) / len(busy); a = sum(v for v, _ in act) / len(act); dd = sum(x for _, x in act) / len(act)
            util[key] = {"mfma_busy_cycles": b, "grbm_gui_active": a, "utilisation": b / (1024.0 * a / 8.0), "effective_clock_GHz": a / 8.0 / dd,
                         "launches": len(busy), "avg_duration_ms_under_pmc": dd / 1e6}
for tag_, fn in (("c3x_rowhess_lds", "%s_c3x_n1_pmc_LDS.csv" % TAG), ("c3x_rowhess_lds_row_symmetric_3", "%s_c3x_n1_pmc_LDS_row_symmetric_3.csv" % TAG)):
    pth = os.path.join(DST, fn)
    if os.path.exists(pth):
        ll = counters(pth)
        ent = {}
        for cn in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"):
            vals = [v for (n, c), vs in ll.items() if "row_hess_kernel" in n and c == cn for v, _ in vs]
            if vals:
                ent[cn] = sum(vals) / len(vals)
        util[tag_] = ent
util["_provenance_rowhess"] = ("rocprofv3 --pmc passes of `bench.py --workload c3|c3x --steps 1 --warmup 1` (tools/refresh_r06b.sh): the same quotients for "
                               "row_hess_kernel<256, 1, 0, 4, *>; LDS counters are sums over the SIMDs, averaged over the launches")
json.dump(util, open(os.path.join(DST, "mfma_util_%s.json" % TAG), "w"), indent=1)
print(json.dumps(util, indent=1))
