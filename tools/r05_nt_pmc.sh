#!/bin/bash
# PMC look at the NT error pass at C4 (one call per option set): matrix-pipe busy, LDS bank conflicts, stalls
set -ux
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05_ntpmc
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1 || true
grep -i -E "LDS|MFMA|WAIT|STALL|BUSY" $O/avail.txt | head -80 > $O/avail_grep.txt
cat > /tmp/nt_once.py <<'PY'
import sys
sys.path.insert(0, sys.argv[1])
from pycmf_amd import _lib
ctx = _lib.Context(0)
ctx.set_problem(65536, 65536, 65536, 256)
ctx.fill_data_synthetic(0, 42, 0, 0); ctx.fill_data_synthetic(1, 43, 0, 0)
for w, s in ((0, 101), (1, 102), (2, 103)):
    ctx.fill_factor_synthetic(w, s, 0, (0.7979 / 256) ** 0.5)
for opt in (0, 1):
    ctx.set_option("nt_tile16", opt); ctx.set_option("nt_raster", 0)
    for _ in range(2):
        print(opt, ctx.residual_sq("linear", "linear"))
ctx.close()
PY
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM" ; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $O/p_$tag -o x -- python3 /tmp/nt_once.py $R > $O/log_$tag.txt 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05_ntpmc")
for f in sorted(glob.glob(O + "/p_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_kernel" not in k or "Li2ELi128" not in k.replace(" ", "") and "<2, 128" not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        agg[k]["_ns"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        n[k] += 1
    for k, v in agg.items():
        print(os.path.basename(os.path.dirname(os.path.dirname(f))), k[:80], {a: "%.4g" % b for a, b in v.items()}, n[k])
    os.remove(f)
PY
