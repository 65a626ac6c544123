"""Host -> HBM upload rate of cmf_set_data_f64 / _f32 (pinned double-buffered staging, multi-threaded float64 -> float32 packing)
on a C4-sized row block: 16384 x 65536 float64 (8.6 GB of source, 4.3 GB on the wire); C4's X + Y are 8 such blocks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pycmf_amd import _lib
rows, cols = 16384, 65536
A = np.random.default_rng(0).random((rows, cols))
ctx = _lib.Context(0)
ctx.set_problem(rows, cols, 256, 256)
for rep in range(3):
    t = time.time(); ctx.set_data(0, A); ctx.sync(); dt = time.time() - t
    print("float64 source: %.3f s for %.2f GB of source = %.1f GB/s source, %.1f GB/s on the wire (float32)" % (dt, A.nbytes / 1e9, A.nbytes / 1e9 / dt, A.nbytes / 2e9 / dt), flush=True)
A32 = A.astype(np.float32)
for rep in range(2):
    t = time.time(); ctx.set_data(0, A32); ctx.sync(); dt = time.time() - t
    print("float32 source: %.3f s for %.2f GB = %.1f GB/s" % (dt, A32.nbytes / 1e9, A32.nbytes / 1e9 / dt), flush=True)
U = np.random.default_rng(1).random((1000000, 256))
ctx.set_problem(1000000, 256, 256, 256)
t = time.time(); ctx.set_factor(0, U); ctx.sync(); print("factor upload 1e6 x 256: %.3f s" % (time.time() - t))
t = time.time(); ctx.get_factor_into(0, U); print("factor download 1e6 x 256: %.3f s" % (time.time() - t))
ctx.close()
print("host threads used: %d of %d" % (min(16, os.cpu_count()), os.cpu_count()))
