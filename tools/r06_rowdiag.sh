#!/bin/bash
# r06: timing-only diagnostic builds of the k_pad = 256 row kernel (python -m pycmf_amd.build --diag; wrong results) on the C3X logit launches.
# Prints the average launch duration of the row kernel per variant.   bash tools/r06_rowdiag.sh [workload]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06; mkdir -p $O; cd $R
W=${1:-c3x}
for D in ${DIAGS:-0 1 4 5 6 7 8 9 0}; do
  CMF_DIAG=1 timeout 300 python3 bench.py --workload $W --steps 2 --warmup 1 --max-warmup 0 --no-cpu-baseline --option row_diag=$D > $O/rowdiag_$D.json 2> $O/rowdiag_$D.err < /dev/null
  python3 - $D $O/rowdiag_$D.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = j["roofline"]
    print("row_diag=%s: row kernel %.3f ms per launch, %d launches; iteration %.1f ms" % (sys.argv[1], r["avg_launch_ms"], r["launches"], j["ms_per_step"]))
except Exception as e:
    print("row_diag=%s: no line (%s)" % (sys.argv[1], e))
PY
done
# A/B of the production variants on the same box (row_symmetric = 4: default; 5: the gather spread over five K-steps)
for S in ${SYMS:-}; do
  timeout 300 python3 bench.py --workload $W --steps 2 --warmup 1 --max-warmup 0 --no-cpu-baseline --option row_symmetric=$S > $O/rowsym_$S.json 2> $O/rowsym_$S.err < /dev/null
  python3 - $S $O/rowsym_$S.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = j["roofline"]
    print("row_symmetric=%s: row kernel %.3f ms per launch, %d launches; iteration %.1f ms" % (sys.argv[1], r["avg_launch_ms"], r["launches"], j["ms_per_step"]))
except Exception as e:
    print("row_symmetric=%s: no line (%s)" % (sys.argv[1], e))
PY
done
