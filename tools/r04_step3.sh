#!/bin/bash
# Round 4: cmf_run, mid-range parity, x-logit C3, estimator tests through the C loop
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_step3
mkdir -p "$O"
cd "$R"
python3 -m pytest tests/test_gpu_run_loop.py tests/test_gpu_midrange.py -x -q -m gpu -s > "$O/pytest_new.txt" 2>&1
tail -n 25 "$O/pytest_new.txt"
python3 -m pytest tests/test_gpu_estimator.py tests/test_gpu_conditioning.py "tests/test_gpu_fullsize.py::test_c3_full_size_newton_sampled_rows_vs_fp64" -x -q -m gpu --durations=8 > "$O/pytest_est.txt" 2>&1
tail -n 20 "$O/pytest_est.txt"
