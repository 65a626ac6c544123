#!/usr/bin/env python3
"""r06 probe of the case campaign C flagged (alpha = 0, y logit, pert = 0.01; tests/tools/fuzz_r06c_replay.json): which rows of V are
off, what their float64 Hessians look like, and what the oracle's own step is sensitive to.   python tests/tools/r06_alpha0_probe.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
import fuzz_campaign as F          # noqa: E402
from oracle import cmf_oracle as O  # noqa: E402

c = json.load(open(os.path.join(HERE, "fuzz_r06c_replay.json")))[1][1]   # "no options"
info = {"want_rows": True}
errs = F.run_case(c, c["seed"], info)
print("err", errs, {k: v for k, v in info.items()})
# rebuild the oracle's state to look at the V rows
rng = np.random.RandomState(c["seed"])
m, d, p, k = c["m"], c["d"], c["p"], c["k"]
X = np.abs(rng.randn(m, d))
Y = (rng.rand(d, p) < 0.3).astype(float) if rng.rand() < 0.5 else rng.rand(d, p)
sc = 0.4 / np.sqrt(max(1.0, k / 8.0))
U0, V0, Z0 = sc * rng.randn(m, k), sc * rng.randn(d, k), sc * rng.randn(p, k)
U0 = np.abs(U0); Z0 = np.abs(Z0)
Ur, Vr, Zr = U0.copy(), V0.copy(), Z0.copy()
np.random.seed(c["seed"] % (2 ** 31))
O.newton_update_step(X, Y, Ur, Vr, Zr, c["alpha"], c["l1"], c["l2"], "linear", "logit", True, False, True, 1.0, c["pert"], update_U=True, update_V=False, update_Z=True)
print("after U, Z sweeps: |U| max %.3e  |Z| max %.3e  |V0| max %.3e" % (np.abs(Ur).max(), np.abs(Zr).max(), np.abs(V0).max()))
sig = lambda t: 1.0 / (1.0 + np.exp(-np.clip(t, -700, 700)))
for i, e in info["worst_v_rows"][:3]:
    s_ = sig(Zr @ V0[i])
    w = s_ * (1 - s_)
    H = (1 - c["alpha"]) * (Zr * w[:, None]).T @ Zr + c["l2"] * np.eye(k)
    ev = np.linalg.eigvalsh(H)
    g = (1 - c["alpha"]) * (s_ - Y[i]) @ Zr + c["l1"] * np.sign(V0[i]) + c["l2"] * V0[i]
    step = g @ O.safe_invert(H, c["pert"])
    H32 = H.astype(np.float32).astype(np.float64)
    step32 = g @ O.safe_invert(H32, c["pert"])
    print("row %d: dev err %.2e | eig max %.3e, #>= pert %d, #in [pert/2, 2 pert] %d, min %.3e | weights: max %.2e, #>1e-6: %d | |step| %.3e, float32-rounded H changes the step by %.2e of max|V|"
          % (i, e, ev.max(), int((ev >= c["pert"]).sum()), int(((ev > c["pert"] / 2) & (ev < 2 * c["pert"])).sum()), ev.min(), w.max(), int((w > 1e-6).sum()),
             np.abs(step).max(), np.abs(step - step32).max() / max(1e-3, np.abs(Vr).max())))
