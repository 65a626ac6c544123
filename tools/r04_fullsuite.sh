#!/bin/bash
# the whole GPU suite, output kept under gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_fullsuite
mkdir -p "$O"
cd "$R"
python3 -m pytest tests -q -m gpu --durations=15 > "$O/pytest_gpu.txt" 2>&1
tail -n 30 "$O/pytest_gpu.txt"
