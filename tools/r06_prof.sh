#!/bin/bash
# r06: rocprofv3 kernel stats of one bench.py command on the GPU box.   tools/r06_prof.sh <name> <bench args...>
# Writes gpurun_out/r06/<name>_kernel_stats.csv (+ the bench line in <name>_prof.log).  Never blocks on stdin.
set -u
N=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$N -o $N -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/${N}_prof.log 2>&1 < /dev/null
f=$(find $O/prof_$N -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $O/${N}_kernel_stats.csv; head -16 $O/${N}_kernel_stats.csv | cut -c1-230; else echo "no kernel stats; log tail:"; tail -5 $O/${N}_prof.log; fi
rm -rf $O/prof_$N
