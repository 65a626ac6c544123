#!/bin/bash
# the whole GPU suite with every test's duration, output kept under gpurun_out/r05_fullsuite
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05_fullsuite
mkdir -p "$O"
cd "$R"
python3 -m pytest tests -q -m gpu --durations=0 "$@" > "$O/pytest_gpu.txt" 2>&1
tail -n 12 "$O/pytest_gpu.txt"
