#!/bin/bash
# Round 4: after the shift/mask fix of the GEMM staging indices and the row-blocked MU protocol:
# multi-process tests, A/B of the library builds again, per-rank compute of the N = 2/4/8 shards in both protocols.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_step2
mkdir -p "$O"
cd "$R"
python3 -m pytest tests/test_gpu_multiprocess.py tests/test_gpu_mu.py -x -q -m gpu > "$O/pytest_mp.txt" 2>&1
tail -n 5 "$O/pytest_mp.txt"
cd /tmp && export TMPDIR=/tmp
python3 "$R/tools/ab_lib_versions.py" 65536,65536,65536,256 10 > "$O/ab_timing.txt" 2>&1
cat "$O/ab_timing.txt"
python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench_c4.json" 2> "$O/bench_c4.err"
for n in 2 4 8; do
  for mode in rsag allreduce; do
    RANK=0 LOCAL_RANK=0 WORLD_SIZE=$n MASTER_PORT=29999 CMF_COMM_BACKEND=null python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --mu-collective $mode > "$O/bench_c4_null${n}_$mode.json" 2> "$O/bench_c4_null${n}_$mode.err"
  done
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_c4*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "ms %.3f"%d["ms_per_step"], {k:round(v,3) for k,v in d["roofline"]["per_class_ms_per_step"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
