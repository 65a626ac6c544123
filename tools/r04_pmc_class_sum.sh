#!/bin/bash
# Round 4: what class_sum_blocks_kernel waits for -- PMC passes restricted to that kernel (C3, one iteration)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_pmc_class_sum
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$O/counters_all.txt" 2>&1
grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|SQ_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TD_[A-Z0-9_]*" "$O/counters_all.txt" | sort -u > "$O/counter_names.txt"
wc -l "$O/counter_names.txt"
A="--workload c3 --steps 1 --warmup 1 --no-cpu-baseline"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU" \
           "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA_WRREQ_STALL_sum TCC_EA_RDREQ_STALL_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_GATE_EN1_sum" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex "class_sum_blocks" --output-format csv -d "$O/p$i" -o p -- python3 "$R/bench.py" $A > "$O/p$i.log" 2>&1
  f=$(find "$O/p$i" -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" "$set" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
dur = []
for r in rows:
    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("%-36s n=%d  mean %.4g  max %.4g" % (k, len(v), sum(v) / len(v), max(v)))
PY
  else
    echo "set $i failed:"; tail -n 5 "$O/p$i.log"
  fi
  find "$O/p$i" -name "*.csv" -size +5M -delete
done
