"""Hardware unit test of the wave-level building blocks (compiled on the fly with hipcc)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_group_sum_dpp_permlane(tmp_path):
    """group_sum<8|16|32|64>: DPP adds + v_permlane16/32_swap (inline asm; the builtin form miscompiles when both
    operands carry the same value on ROCm 7.2) must equal a plain sum over every aligned lane group."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "groupsum_test")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-I", os.path.join(ROOT, "pycmf_amd", "csrc"), "-o", exe,
                    os.path.join(ROOT, "tests", "hip", "groupsum_test.hip")], check=True, capture_output=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "bad=0" in out, out
