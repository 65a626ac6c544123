#!/usr/bin/env python3
"""A/B of the NT error pass (cmf_residual_sq) at C4 / C2: option nt_tile16 = 0 (256 x 128 x 32 tile, one workgroup per CU) against
1 (256 x 128 x 16, two per CU), each with and without the XCD-aware tile order (nt_raster).  Prints ms per call for both error terms together, TF/s and the values (they must be identical)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pycmf_amd import _lib  # noqa: E402


def run(m, d, p, k, reps=6, link="linear"):
    ctx = _lib.Context(0)
    ctx.set_problem(m, d, p, k)
    ctx.fill_data_synthetic(0, 42, 0, 0)
    ctx.fill_data_synthetic(1, 43, 0, 0)
    sc = (0.7979 / k) ** 0.5
    for w, seed in ((0, 101), (1, 102), (2, 103)):
        ctx.fill_factor_synthetic(w, seed, 0, sc)
    out = {}
    for rnd in range(2):
        for opt in (0, 1, 2, 4, 6):
            ctx.set_option("nt_tile16", opt & 1)
            ctx.set_option("nt_bn256", (opt >> 1) & 1)
            ctx.set_option("nt_debug", opt >> 2)
            val = ctx.residual_sq(link, link)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                val = ctx.residual_sq(link, link)
            ctx.sync()
            ms = (time.perf_counter() - t0) / reps * 1e3
            flops = 2.0 * k * d * (m + p)
            out.setdefault(opt, []).append(ms)
            print("m,d,p,k=%s link=%s nt_tile16 + 2 nt_bn256 + 4 no_targets = %d round %d: %.3f ms per metric, %.1f TF/s = %.3f of the fp32 MFMA peak; values %r"
                  % ((m, d, p, k), link, opt, rnd, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3, val), flush=True)
    ctx.close()
    return out


if __name__ == "__main__":
    run(65536, 65536, 65536, 256)
    run(32768, 16384, 8192, 256, reps=20)
