#!/bin/bash
# Round-5 refresh of every number under profiles/: bench lines, rocprofv3 kernel stats, PMC traffic (C4, C2, C3, C5), MFMA busy (C4).
# Run on the GPU box from the repo root:  bash tools/refresh_r05.sh   (results under gpurun_out/r05/; tools/collect_r05.py files them)
set -ux
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05
mkdir -p "$O"
cd "$R"
# two stages: `bash tools/refresh_r05.sh` (everything), then -- after `python tools/collect_r05.py` has written profiles/traffic_*.json
# from this run's PMC passes -- `bash tools/refresh_r05.sh lines`: the bench lines again, so that the stored traffic figure each line
# quotes is the one measured on the same build
MODE=${1:-all}
python3 bench.py --steps 20 --warmup 5 > $O/c4.json 2> $O/c4.err
python3 bench.py --workload c2 --steps 200 --warmup 20 > $O/c2.json 2> $O/c2.err
python3 bench.py --workload c3 --steps 5 --warmup 2 > $O/c3.json 2> $O/c3.err
# c3z (the reference's default l2 = 0): the regime changes after ~7 iterations -- from then on the clamp of `_safe_invert` acts on every row
# of U and every one of them is redone in float64 -- so the line is taken BEHIND the transition; beside it the same run with the
# float64 refinement switched off (float32 spectral clamp only)
python3 bench.py --workload c3z --steps 3 --warmup 9 > $O/c3z.json 2> $O/c3z.err
python3 bench.py --workload c3z --steps 3 --warmup 9 --no-cpu-baseline --option refine_rows=0 > $O/c3z_norefine.json 2> $O/c3z_norefine.err
python3 bench.py --workload c3x --steps 3 --warmup 1 > $O/c3x.json 2> $O/c3x.err
# the reference's default stopping test (tol = 1e-4: the error metric every 10th iteration) INSIDE the timed region
python3 bench.py --steps 20 --warmup 5 --tol 1e-4 --no-cpu-baseline > $O/c4_tol.json 2> $O/c4_tol.err
python3 bench.py --workload c5 --steps 10 --warmup 3 > $O/c5.json 2> $O/c5.err
python3 bench.py --workload c5l --steps 10 --warmup 3 > $O/c5l.json 2> $O/c5l.err
if [ "$MODE" = "lines" ]; then exit 0; fi
# per-rank compute of the C4 shards (collectives stubbed out: CMF_COMM_BACKEND=null, sums = own partial x world: finite iterates), both
# MU protocols
for n in 2 4 8; do
  for mode in rsag allreduce; do
    RANK=0 LOCAL_RANK=0 WORLD_SIZE=$n MASTER_PORT=29999 CMF_COMM_BACKEND=null python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --mu-collective $mode > $O/c4_null${n}_$mode.json 2> $O/c4_null${n}_$mode.err
  done
done
# the N = 8 dress rehearsal of C4 at full size on this one GPU (host-staged collectives), both protocols, rows against N = 1
python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --dump-rows $O/rows_n1 > $O/c4_n1_2it.json 2> $O/c4_n1_2it.err
for mode in allreduce rsag; do
  CMF_BENCH_SAME_DEVICE=1 CMF_COMM_BACKEND=host CMF_COMM_TIMEOUT=900 timeout 1200 python3 bench.py --gpus 8 --steps 2 --warmup 0 --no-cpu-baseline \
      --mu-collective $mode --dump-rows $O/rows_n8_$mode > $O/c4_n8_same_device_$mode.json 2> $O/c4_n8_same_device_$mode.err
  python3 tools/compare_rows.py $O/rows_n1 $O/rows_n8_$mode --tol 1e-5 --out $O/c4_n8_same_device_${mode}_rows.json
done
CMF_BENCH_SAME_DEVICE=1 CMF_COMM_BACKEND=host CMF_COMM_TIMEOUT=900 timeout 900 python3 bench.py --gpus 8 --workload c4q --steps 2 --warmup 0 --no-cpu-baseline \
    > $O/c4q_n8_same_device_auto.json 2> $O/c4q_n8_same_device_auto.err
rm -f $O/rows_*.npz
python3 tools/refine_timing.py 1024,16384,512,256 0.5 > $O/refine_timing.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for W in c4 c2 c3 c5 c5l; do
  case $W in c4) A="--steps 10 --warmup 3";; c2) A="--workload c2 --steps 20 --warmup 5";; c3) A="--workload c3 --steps 3 --warmup 1";; c5) A="--workload c5 --steps 5 --warmup 2";; c5l) A="--workload c5l --steps 3 --warmup 1";; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -o $W -- python3 $R/bench.py $A --no-cpu-baseline > $O/prof_$W.log 2>&1
done
for W in c4 c2 c3 c3z c3x c5; do
  case $W in c4) A="--steps 2 --warmup 1";; c2) A="--workload c2 --steps 3 --warmup 1";; c3) A="--workload c3 --steps 1 --warmup 1";; c3z) A="--workload c3z --steps 1 --warmup 1";; c3x) A="--workload c3x --steps 1 --warmup 1";; c5) A="--workload c5 --steps 2 --warmup 1";; esac
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${W}_fetch -o f -- python3 $R/bench.py $A --no-cpu-baseline > $O/pmc_${W}f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${W}_write -o w -- python3 $R/bench.py $A --no-cpu-baseline > $O/pmc_${W}w.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_c4 -o m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_m4.log 2>&1
cd $R
find $O -name "*_kernel_trace.csv" -delete
# condense the PMC collections (cmfk kernels only) so that they fit the 64 MiB return channel
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
