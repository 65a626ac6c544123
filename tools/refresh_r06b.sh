#!/bin/bash
# Round 6, second refresh: the k_pad = 256 row kernel changed (row_symmetric=4, 16-wide diagonal sub-blocks), so the C3-family bench lines,
# kernel stats and PMC passes are taken again; the other workloads keep the numbers of tools/refresh_r06.sh.  New here: the matrix-pipe
# busy cycles and the effective clock of the row kernel (c3, c3x), and its LDS bank-conflict cycles (c3x).
# Run on the GPU box from the repo root:  bash tools/refresh_r06b.sh   (results under gpurun_out/r06/; `python tools/collect_r06.py c3` files them)
set -ux
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06
mkdir -p "$O"
cd "$R"
B="timeout 600 python3 bench.py"
$B --workload c3 --steps 5 --warmup 3 > $O/c3.json 2> $O/c3.err < /dev/null
$B --workload c3 --steps 5 --warmup 3 --no-cpu-baseline --option refine_rows=0 > $O/c3_norefine.json 2> $O/c3_norefine.err < /dev/null
$B --workload c3 --steps 5 --warmup 3 --no-cpu-baseline --option eig_clamp=0 --option refine_rows_tol_ppm=0 > $O/c3_r05_clamp.json 2> $O/c3_r05_clamp.err < /dev/null
$B --workload c3 --steps 5 --warmup 3 --no-cpu-baseline --option eig_clamp=3 > $O/c3_no_early_exit.json 2> $O/c3_no_early_exit.err < /dev/null
$B --workload c3 --steps 5 --warmup 3 --no-cpu-baseline --option row_symmetric=3 > $O/c3_rowsym3.json 2> $O/c3_rowsym3.err < /dev/null
$B --workload c3 --steps 5 --warmup 3 --no-cpu-baseline --option rank1_clamp=0 > $O/c3_norank1.json 2> $O/c3_norank1.err < /dev/null
$B --workload c3r --steps 5 --warmup 3 --no-cpu-baseline > $O/c3r.json 2> $O/c3r.err < /dev/null
$B --workload c3x --steps 5 --warmup 3 > $O/c3x.json 2> $O/c3x.err < /dev/null
$B --workload c3x --steps 5 --warmup 3 --no-cpu-baseline --option row_symmetric=3 > $O/c3x_rowsym3.json 2> $O/c3x_rowsym3.err < /dev/null
cd /tmp && export TMPDIR=/tmp
for W in c3 c3x; do
  A="--workload $W --steps 3 --warmup 3"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -o $W -- python3 $R/bench.py $A --no-cpu-baseline > $O/prof_$W.log 2>&1 < /dev/null
done
for W in c3 c3x; do
  A="--workload $W --steps 1 --warmup 1"
  if [ $W = c3x ]; then A="$A --max-warmup 0"; fi
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${W}_fetch -o f -- python3 $R/bench.py $A --no-cpu-baseline > $O/pmc_${W}f.log 2>&1 < /dev/null
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${W}_write -o w -- python3 $R/bench.py $A --no-cpu-baseline > $O/pmc_${W}w.log 2>&1 < /dev/null
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_$W -o m -- python3 $R/bench.py $A --no-cpu-baseline > $O/pmc_m_$W.log 2>&1 < /dev/null
done
timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_lds_c3x -o l -- python3 $R/bench.py --workload c3x --steps 1 --warmup 1 --max-warmup 0 --no-cpu-baseline > $O/pmc_l_c3x.log 2>&1 < /dev/null
timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_lds_c3x_sym3 -o l -- python3 $R/bench.py --workload c3x --steps 1 --warmup 1 --max-warmup 0 --no-cpu-baseline --option row_symmetric=3 > $O/pmc_l_c3x_sym3.log 2>&1 < /dev/null
cd $R
find $O -name "*_kernel_trace.csv" -delete
for f in $(find $O -name "*counter_collection.csv"); do
  python3 - "$f" <<'PY'
import csv, sys
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
keep = [r for r in rows if "cmfk::" in r.get("Kernel_Name", "")]
with open(path.replace("counter_collection.csv", "cmfk_counters.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value", "Duration_ns"])
    for r in keep:
        w.writerow([r["Dispatch_Id"], r["Kernel_Name"], r["Grid_Size"], r["Counter_Name"], r["Counter_Value"],
                    int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
PY
  rm -f "$f"
done
find $O -size +20M -delete
du -sh $O
