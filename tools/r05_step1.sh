#!/bin/bash
# Round 5, step 1: the N = 8 dress rehearsal of the headline configuration at FULL size on one GPU (eight rank processes on the same
# device, host-staged collectives), both MU protocols, 16 rows of each factor against the N = 1 run; per-rank compute of the C4 shards
# with the collectives stubbed out ON FINITE ITERATES (CMF_COMM_BACKEND=null: sums = own partial x world); then the whole GPU suite.
set -ux
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05
mkdir -p "$O"
cd "$R"
python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --dump-rows $O/rows_n1 > $O/c4_n1_2it.json 2> $O/c4_n1_2it.err
for mode in allreduce rsag; do
  CMF_BENCH_SAME_DEVICE=1 CMF_COMM_BACKEND=host CMF_COMM_TIMEOUT=900 timeout 1200 python3 bench.py --gpus 8 --steps 2 --warmup 0 --no-cpu-baseline \
      --mu-collective $mode --dump-rows $O/rows_n8_$mode > $O/c4_n8_same_device_$mode.json 2> $O/c4_n8_same_device_$mode.err
  python3 tools/compare_rows.py $O/rows_n1 $O/rows_n8_$mode --tol 1e-5 --out $O/c4_n8_same_device_${mode}_rows.json
done
# the trial itself (auto) on eight host-staged ranks, at the test suite's shard size
CMF_BENCH_SAME_DEVICE=1 CMF_COMM_BACKEND=host CMF_COMM_TIMEOUT=900 timeout 900 python3 bench.py --gpus 8 --workload c4q --steps 2 --warmup 0 --no-cpu-baseline \
    > $O/c4q_n8_same_device_auto.json 2> $O/c4q_n8_same_device_auto.err
for n in 2 4 8; do
  for mode in rsag allreduce; do
    RANK=0 LOCAL_RANK=0 WORLD_SIZE=$n MASTER_PORT=29999 CMF_COMM_BACKEND=null python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --mu-collective $mode > $O/c4_null${n}_$mode.json 2> $O/c4_null${n}_$mode.err
  done
done
rm -f $O/rows_*.npz
tail -c 600 $O/*_rows.json
for f in $O/c4_null*.json $O/c4_n8*.json; do python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d.get("collective", {})
    print(sys.argv[1].split("/")[-1], "ms/step %.3f" % d["ms_per_step"], c.get("protocol"), c.get("launch_points_per_iteration"), c.get("replicas"), d["rel_residual"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
bash tools/r05_fullsuite.sh
