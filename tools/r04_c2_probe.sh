#!/bin/bash
# Round 4: where the C2 iteration's time is -- per-launch durations of the four data passes (rocprofv3 kernel trace), then the
# staging schedules / tile options as bench lines
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_c2_probe
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/trace" -o c2 -- python3 "$R/bench.py" --workload c2 --steps 20 --warmup 5 --no-cpu-baseline > "$O/trace_bench.json" 2> "$O/trace_bench.err"
cd "$R"
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/trace/**/c2_kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# the last 12 iterations: find the repeating pattern of gemm data passes
names = [s[0] for s in seq]
idx = [i for i, n in enumerate(names) if "gemm_kernel<1, 128" in n or "gemm_kernel<0, 128" in n]
tail = idx[-48:]
by = collections.defaultdict(list)
for j, i in enumerate(tail):
    by[(j % 4, "TN" if "<1," in names[i] else "NN")].append(seq[i][1] / 1e3)
for k in sorted(by): print("data pass", k, "n=%d avg %.1f us min %.1f max %.1f" % (len(by[k]), sum(by[k]) / len(by[k]), min(by[k]), max(by[k])))
# one iteration in order with gaps
i0 = tail[-8]
prev_end = None
for i in range(i0, len(seq)):
    n, d, s, e = seq[i]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-60s %8.1f us  gap %6.1f" % (n[:60], d / 1e3, gap))
    prev_end = e
    if i - i0 > 26: break
PY
python3 tools/ab_lib_versions.py 16384,8192,4096,128 60 > "$O/ab_pin.txt" 2>&1; tail -n 3 "$O/ab_pin.txt"
python3 tools/ab_lib_versions.py 16384,8192,4096,128 60 > "$O/ab_pin2.txt" 2>&1; tail -n 3 "$O/ab_pin2.txt"
for pipe in 0 1 2 3 4 5 10; do
  python3 bench.py --workload c2 --steps 300 --warmup 30 --no-cpu-baseline --option gemm_pipe=$pipe > "$O/bench_pipe$pipe.json" 2> "$O/bench_pipe$pipe.err"
done
python3 bench.py --workload c2 --steps 300 --warmup 30 --no-cpu-baseline --option gemm_tile512=1 > "$O/bench_tile512.json" 2> "$O/bench_tile512.err"
python3 bench.py --workload c2 --steps 300 --warmup 30 --no-cpu-baseline --option split_reduce_in_kernel=1 > "$O/bench_inred.json" 2> "$O/bench_inred.err"
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "it/s %.1f ms %.4f" % (d["value"], d["ms_per_step"]), {k: round(v, 4) for k, v in d["roofline"]["per_class_ms_per_step"].items()})
    except Exception as e:
        print(f, "ERR", e, open(f.replace('.json', '.err')).read()[-400:])
PY
