#!/bin/bash
# Round 4: factor_times64_kernel software-pipelined -- parity tests, then C5 / C5L with the old and the new library alternating on one box
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_step15
mkdir -p "$O"
cd "$R"
timeout 900 python3 -m pytest tests/test_gpu_shared64.py tests/test_gpu_reassoc.py tests/test_gpu_sparse.py -x -q -m gpu > "$O/pytest.txt" 2>&1; tail -n 3 "$O/pytest.txt"
cp pycmf_amd/libcmfhip.so "$O/keep.so"
export PYCMF_AMD_SKIP_HASH_CHECK=1
for rep in 1 2; do
for v in old new; do
  cp tools/ab/lib${v}_ft64.so pycmf_amd/libcmfhip.so
  timeout 300 python3 bench.py --workload c5 --steps 10 --warmup 3 --no-cpu-baseline > "$O/bench_c5_${v}_$rep.json" 2> "$O/bench_c5_${v}_$rep.err"
  timeout 300 python3 bench.py --workload c5l --steps 3 --warmup 1 --no-cpu-baseline > "$O/bench_c5l_${v}_$rep.json" 2> "$O/bench_c5l_${v}_$rep.err"
done
done
cp "$O/keep.so" pycmf_amd/libcmfhip.so; rm -f "$O/keep.so"
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "it/s %.2f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in d["roofline"]["per_class_ms_per_step"].items()}, d["rel_residual"]["x"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace('.json', '.err')).read()[-400:])
PY
