import runpy, sys, io, contextlib, json
sys.argv = ["bench.py"] + sys.argv[1:]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
j = json.loads(buf.getvalue().strip().splitlines()[-1])
print(sys.argv[1:], "torch imported:", "torch" in sys.modules, "value", round(j["value"], 3), "ms", round(j["ms_per_step"], 3))
