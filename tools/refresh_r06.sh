#!/bin/bash
# Round-6 refresh of the numbers under profiles/: bench lines, rocprofv3 kernel stats, PMC traffic (C4, C3, C3X, C5, C5Z, C5ZS), MFMA busy (C4).
# Run on the GPU box from the repo root:  bash tools/refresh_r06.sh [lines]   (results under gpurun_out/r06/; tools/collect_r06.py files them)
# Not repeated from round 5 (the code behind them did not change): the null-collective per-rank clocks and the full-size N = 8 rehearsal.
set -ux
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06
mkdir -p "$O"
cd "$R"
MODE=${1:-all}
B="timeout 600 python3 bench.py"
$B --steps 20 --warmup 5 > $O/c4.json 2> $O/c4.err < /dev/null
$B --steps 40 --warmup 5 --tol 1e-4 --no-cpu-baseline > $O/c4_tol.json 2> $O/c4_tol.err < /dev/null
$B --steps 40 --warmup 5 --no-cpu-baseline > $O/c4_notol40.json 2> $O/c4_notol40.err < /dev/null
$B --workload c2 --steps 200 --warmup 20 > $O/c2.json 2> $O/c2.err < /dev/null
# per-row Newton lines: bench.py itself warms up until the clamp / refinement counts of consecutive iterations agree, and refuses the line
# (exit code 3) when the iterations behind the timed window cost more than 1.25 x the timed ones
$B --workload c3 --steps 5 --warmup 3 > $O/c3.json 2> $O/c3.err < /dev/null
$B --workload c3 --steps 5 --warmup 3 --no-cpu-baseline --option refine_rows=0 > $O/c3_norefine.json 2> $O/c3_norefine.err < /dev/null
$B --workload c3 --steps 5 --warmup 3 --no-cpu-baseline --option eig_clamp=0 --option refine_rows_tol_ppm=0 > $O/c3_r05_clamp.json 2> $O/c3_r05_clamp.err < /dev/null
$B --workload c3 --steps 5 --warmup 3 --no-cpu-baseline --option eig_clamp=3 > $O/c3_no_early_exit.json 2> $O/c3_no_early_exit.err < /dev/null
$B --workload c3r --steps 5 --warmup 3 --no-cpu-baseline > $O/c3r.json 2> $O/c3r.err < /dev/null
$B --workload c3x --steps 5 --warmup 3 > $O/c3x.json 2> $O/c3x.err < /dev/null
$B --workload c5 --steps 10 --warmup 3 > $O/c5.json 2> $O/c5.err < /dev/null
$B --workload c5z --steps 10 --warmup 3 --no-cpu-baseline > $O/c5z.json 2> $O/c5z.err < /dev/null
$B --workload c5zs --steps 10 --warmup 3 --no-cpu-baseline > $O/c5zs.json 2> $O/c5zs.err < /dev/null
$B --workload c5l --steps 10 --warmup 3 > $O/c5l.json 2> $O/c5l.err < /dev/null
$B --workload c5l_l2x10 --steps 10 --warmup 3 --no-cpu-baseline > $O/c5l_l2x10.json 2> $O/c5l_l2x10.err < /dev/null
if [ "$MODE" = "lines" ]; then exit 0; fi
cd /tmp && export TMPDIR=/tmp
for W in c4 c2 c3 c3x c5; do
  case $W in c4) A="--steps 10 --warmup 3";; c2) A="--workload c2 --steps 20 --warmup 5";; c3) A="--workload c3 --steps 3 --warmup 3";; c3x) A="--workload c3x --steps 3 --warmup 3";; c5) A="--workload c5 --steps 5 --warmup 2";; esac
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -o $W -- python3 $R/bench.py $A --no-cpu-baseline > $O/prof_$W.log 2>&1 < /dev/null
done
for W in c4 c3 c3x c5 c5z c5zs; do
  case $W in c4) A="--steps 2 --warmup 1";; c3) A="--workload c3 --steps 1 --warmup 1";; c3x) A="--workload c3x --steps 1 --warmup 1 --max-warmup 0";; c5) A="--workload c5 --steps 2 --warmup 1";; c5z) A="--workload c5z --steps 2 --warmup 1";; c5zs) A="--workload c5zs --steps 2 --warmup 1";; esac
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${W}_fetch -o f -- python3 $R/bench.py $A --no-cpu-baseline > $O/pmc_${W}f.log 2>&1 < /dev/null
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${W}_write -o w -- python3 $R/bench.py $A --no-cpu-baseline > $O/pmc_${W}w.log 2>&1 < /dev/null
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_c4 -o m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_m4.log 2>&1 < /dev/null
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
