"""Hardware check of the cross-lane helpers (csrc/ls_lanes.h): DPP row reductions, v_readlane wave reductions and the gfx950
v_permlane16/32_swap exchanges must agree exactly with the __shfl_xor butterflies they replace (tools/lane_check.hip)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_lane_helpers_match_shuffle_butterflies(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not present on this box")
    exe = str(tmp_path / "lane_check")
    src = os.path.join(ROOT, "tools", "lane_check.hip")
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-o", exe, src], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    for name in ("row16_sum", "wave_sum", "wave_max", "xor32_sum", "xor16_sum", "xor32_get", "xor16_get"):
        assert f"{name} max |diff| = 0" in run.stdout, run.stdout
