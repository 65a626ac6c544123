"""INTEGRATION.md section 1 is the binding a reference maintainer would add (pycmf/cmf.py:437-454 constructs the solver object
and calls its one method).  This test EXECUTES that code block as written -- only the library path is substituted -- so the
boundary documentation cannot rot silently."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 1."):text.index("## 2.")]
    blocks = re.findall(r"```python\n(.*?)```", sec, re.S)
    assert len(blocks) == 1, "section 1 holds exactly one python block: the stub"
    return blocks[0]


@pytest.mark.parametrize("tag,l1,l2", [("plain", 0.0, 0.0), ("reg", 0.3, 0.7)])
def test_integration_md_stub_reproduces_the_reference_steps(tag, l1, l2):
    from pycmf_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("no GPU visible: the gpu-marked tests need an MI355X")
    src = _stub_source()
    assert 'C.CDLL("libcmfhip.so")' in src
    ns = {}
    exec(compile(src.replace('"libcmfhip.so"', repr(_lib.LIB_PATH)), "INTEGRATION.md#1", "exec"), ns)
    g = load_golden("g2_mu_steps")
    # ten MUSolver.update_step calls of the reference on these inputs (tests/golden/make_golden.py), tol = 0: no early stop
    solver = ns["HipMUSolver"](max_iter=10, tol=0, l1_reg=l1, l2_reg=l2)
    U, V, Z = g["U0"].copy(), g["V0"].copy(), g["Z0"].copy()
    # the reference hands Z over as an F-ordered view (pycmf/cmf.py:202): the stub must cope with any strides
    Zf = np.asfortranarray(Z)
    Uo, Vo, Zo, n_iter = solver.fit_iterative_update(g["X"], g["Y"], U, V, Zf)
    assert n_iter == 10 and Uo is U and Zo is Zf                      # in place, like the reference (cmf_solvers.py:195)
    for got, name in ((U, "U"), (V, "V"), (Zf, "Z")):
        np.testing.assert_allclose(got, g["%s_dense_%s10" % (tag, name)], rtol=2e-4, atol=1e-6)
