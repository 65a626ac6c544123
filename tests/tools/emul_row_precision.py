"""TEST INFRASTRUCTURE (CPU only): how much of the error of the cases in tests/tools/fuzz_flagged.jsonl is inherent to float32 per-row
Hessians?  Runs the float64 oracle three more times per case with (a) the inputs rounded to float32, (b) every per-row Hessian
rounded to float32 before its (float64) eigen-decomposition, (c) every per-row gradient rounded to float32 (+ accumulation-sized
noise), and prints the distance of each from the plain float64 result.  DESIGN.md section 7 quotes (b) and (c).

    OMP_NUM_THREADS=4 python tests/tools/emul_row_precision.py [max_k]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import cmf_oracle as O          # noqa: E402


def gen(c):
    rng = np.random.RandomState(c["seed"])
    m, d, p, k = c["m"], c["d"], c["p"], c["k"]
    xl, yl = c["x_link"], c["y_link"]
    X = rng.rand(m, d) if xl == "logit" else np.abs(rng.randn(m, d))
    Y = (rng.rand(d, p) < 0.3).astype(float) if (yl == "logit" and rng.rand() < 0.5) else (rng.rand(d, p) if yl == "logit" else np.abs(rng.randn(d, p)))
    if c["csr"]:
        X[rng.rand(m, d) < 0.9] = 0.0
    sc = 0.4 / np.sqrt(max(1.0, k / 8.0))
    U0, V0, Z0 = sc * rng.randn(m, k), sc * rng.randn(d, k), sc * rng.randn(p, k)
    if c["nn"] & 1: U0 = np.abs(U0)
    if c["nn"] & 2: V0 = np.abs(V0)
    if c["nn"] & 4: Z0 = np.abs(Z0)
    return X, Y, U0, V0, Z0


def step(c, X, Y, U, V, Z):
    np.random.seed(c["seed"] % (2 ** 31))
    nn, mask = c["nn"], c["mask"]
    U, V, Z = U.copy(), V.copy(), Z.copy()
    O.newton_update_step(X, Y, U, V, Z, c["alpha"], c["l1"], c["l2"], c["x_link"], c["y_link"], bool(nn & 1), bool(nn & 2), bool(nn & 4),
                         c["ratio"], c["pert"], update_U=bool(mask & 1), update_V=bool(mask & 2), update_Z=bool(mask & 4))
    return U, V, Z


def dist(ref, other):
    return ["%.1e" % (np.abs(a - b).max() / max(1e-3, np.abs(a).max())) for a, b in zip(ref, other)]


def main():
    max_k = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    here = os.path.dirname(os.path.abspath(__file__))
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    inv0, row0 = O.safe_invert, O._row_step
    for line in open(os.path.join(here, "fuzz_flagged.jsonl")):
        d = json.loads(line)
        c = d["case"]
        if c["k"] > max_k:
            continue
        X, Y, U0, V0, Z0 = gen(c)
        ref = step(c, X, Y, U0, V0, Z0)
        a = step(c, f32(X), f32(Y), f32(U0), f32(V0), f32(Z0))
        O.safe_invert = lambda H, pert: inv0(f32(H), pert)
        b = step(c, X, Y, U0, V0, Z0)
        O.safe_invert = inv0

        def row32(F, i, grad, Hinv, nn):
            g = np.asarray(grad, dtype=np.float64)
            noise = np.random.RandomState(i).randn(*g.shape) * 3e-7 * np.abs(g).max()
            return row0(F, i, f32(g) + noise, Hinv, nn)
        O._row_step = row32
        g = step(c, X, Y, U0, V0, Z0)
        O._row_step = row0
        print({k: c[k] for k in ("m", "d", "p", "k", "l2", "ratio", "pert")}, "device (float32 only)", ["%.1e" % v for v in d["err"]],
              "| inputs rounded", dist(ref, a), "| Hessians rounded", dist(ref, b), "| gradients rounded", dist(ref, g), flush=True)


if __name__ == "__main__":
    main()
