"""The C ABI from a host that is not Python: tests/c/abi_consumer.c (plain C99, gcc) linked against libcmfhip.so runs the golden MU
steps of fixture g2 (minted from the reference's MUSolver.update_step, pycmf/cmf_solvers.py:248-263) and must reproduce the reference's
factors -- what a maintainer binding include/cmfhip.h from C, cgo or JNI would see."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu


def build_consumer(tmp_path):
    exe = str(tmp_path / "abi_consumer")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_consumer.c"),
           "-L", os.path.join(ROOT, "pycmf_amd"), "-lcmfhip", "-Wl,-rpath," + os.path.join(ROOT, "pycmf_amd"), "-o", exe]
    q = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert q.returncode == 0, q.stdout.decode()
    return exe


def write_problem(path, X, Y, U, V, Z, l1, l2, steps):
    m, d = X.shape
    p, k = Y.shape[1], U.shape[1]
    with open(path, "wb") as f:
        f.write(np.array([m, d, p, k, steps], dtype=np.int64).tobytes())
        f.write(np.array([l1, l2], dtype=np.float64).tobytes())
        for A in (X, Y, U, V, Z):
            f.write(np.ascontiguousarray(A, dtype=np.float64).tobytes())


@pytest.mark.parametrize("tag,l1,l2", [("plain", 0.0, 0.0), ("reg", 0.3, 0.7)])
@pytest.mark.parametrize("iters", [1, 10])
def test_c_host_reproduces_the_reference_mu_steps(tmp_path, tag, l1, l2, iters):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    g = load_golden("g2_mu_steps")
    X, Y, U0, V0, Z0 = g["X"], g["Y"], g["U0"], g["V0"], g["Z0"]
    exe = build_consumer(tmp_path)
    prob, res = str(tmp_path / "prob.bin"), str(tmp_path / "res.bin")
    write_problem(prob, X, Y, U0, V0, Z0, l1, l2, iters)
    q = subprocess.run([exe, prob, res], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert q.returncode == 0, q.stdout.decode()
    out = np.fromfile(res, dtype=np.float64)
    m, d = X.shape
    p, k = Y.shape[1], U0.shape[1]
    U, V, Z = out[:m * k].reshape(m, k), out[m * k:(m + d) * k].reshape(d, k), out[(m + d) * k:(m + d + p) * k].reshape(p, k)
    ex2, ey2 = out[-2:]
    rtol = 2e-5 if iters == 1 else 2e-4        # the tolerances of tests/test_gpu_mu.py for the same fixture through ctypes
    for name, got in (("U", U), ("V", V), ("Z", Z)):
        np.testing.assert_allclose(got, g["%s_dense_%s%d" % (tag, name, iters)], rtol=rtol, atol=1e-6)
    np.testing.assert_allclose([ex2, ey2], [np.linalg.norm(X - U @ V.T) ** 2, np.linalg.norm(Y - V @ Z.T) ** 2], rtol=1e-4)
