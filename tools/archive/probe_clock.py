"""In-kernel shader clock of the headline GEMM (s_memtime / s_memrealtime stamps), after sustained load."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycmf_amd import _lib
m, d, p, k = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "65536,65536,4096,256").split(","))
ctx = _lib.Context(0)
ctx.set_problem(m, d, p, k)
ctx.fill_data_synthetic(0, 42); ctx.fill_data_synthetic(1, 43)
for w in range(3):
    ctx.fill_factor_synthetic(w, 100 + w, 0, (0.8 / k) ** 0.5)
t0 = time.time()
while time.time() - t0 < 2.0:
    ctx.mu_uz_update(0.0, 0.0, 1)
ctx.sync()
for _ in range(3):
    ghz, us = ctx.debug_clock()
    flops = 2.0 * 256 * 256 * d  # per workgroup tile (256 x 256 x K)
    cyc = ghz * 1e3 * us         # shader cycles of the main loop
    ideal = flops / 4096.0 * 64 / 4 / 2 * 2  # MFMAs per WG * 64 cycles / 4 SIMDs ... per SIMD: (flops/4096 MFMAs)/4 SIMDs * 64
    per_simd = flops / 4096.0 / 4.0 * 64.0
    print("in-kernel clock %.3f GHz, main loop %.1f us = %.3e cycles; MFMA-bound minimum %.3e cycles -> pipe busy %.1f%%; %.1f TF/s at this clock would be peak %.1f" %
          (ghz, us, cyc, per_simd, 100 * per_simd / cyc, 0, 157.3 * ghz / 2.4))
ctx.close()
