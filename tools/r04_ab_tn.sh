#!/bin/bash
# Round 4, VERDICT item 2: A/B of the TN data pass between library builds on ONE box in ONE process (timing, then FETCH_SIZE),
# plus this box's baseline bench lines and the per-rank compute of the N = 8 shard (null collectives).
# Run on the GPU box from the repository root: bash tools/r04_ab_tn.sh
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_ab
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 "$R/tools/ab_lib_versions.py" 65536,65536,65536,256 10 > "$O/ab_timing.txt" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O/pmc" -o ab --output-format csv -- python3 "$R/tools/ab_lib_versions.py" 65536,65536,65536,256 3 --pmc > "$O/ab_pmc_run.txt" 2>&1
f=$(find "$O/pmc" -name '*counter_collection.csv' | head -1)
python3 "$R/tools/ab_lib_pmc.py" "$f" HEAD,1077125,e0e8a47 > "$O/ab_fetch.txt" 2>&1
rm -rf "$O/pmc"
python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench_c4.json" 2> "$O/bench_c4.err"
python3 "$R/bench.py" --workload c2 --steps 200 --warmup 20 --no-cpu-baseline > "$O/bench_c2.json" 2> "$O/bench_c2.err"
python3 "$R/bench.py" --workload c3 --steps 5 --warmup 2 --no-cpu-baseline > "$O/bench_c3.json" 2> "$O/bench_c3.err"
for n in 2 4 8; do
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=$n MASTER_PORT=29999 CMF_COMM_BACKEND=null python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench_c4_null$n.json" 2> "$O/bench_c4_null$n.err"
done
tail -n 5 "$O/ab_timing.txt" "$O/ab_fetch.txt"
