import json, sys, subprocess
for w, extra in (("c3", ["--steps","2","--warmup","1"]), ("c5l", ["--steps","2","--warmup","1"]), ("tiny3", ["--steps","3","--warmup","1"]), ("tiny5l", ["--steps","3","--warmup","1"])):
    out = subprocess.run([sys.executable, "bench.py", "--workload", w, "--no-cpu-baseline"] + extra, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print(w, round(d["ms_per_step"], 2), {k: v for k, v in d["conditioning"].items() if k != "note"}, flush=True)
