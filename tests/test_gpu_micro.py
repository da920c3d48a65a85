"""The layouts the matrix-core Hessian relies on (HessAcc / HessAcc32 in csrc/solve_g.h): operand and result layout of the 4-block 16x16x1 and the
2-block 32x32x1 fp32 MFMA, and the register exchanges (v_permlane32_swap / v_permlane16_swap) that bring a block into the lanes of its env.  The
micro programs print the layout they find and return non-zero when the exchange does not produce the row layout."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mfma_layout", "mfma_layout32"])
def test_mfma_layout_and_register_exchange(tmp_path, name):
    exe = tmp_path / name
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "-w", "-o", str(exe), str(ROOT / "tools" / "micro" / f"{name}.hip")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "mismatches: 0" in r.stdout
