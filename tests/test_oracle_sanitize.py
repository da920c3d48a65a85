"""the CPU restatement under AddressSanitizer / UndefinedBehaviorSanitizer (the GPU has no sanitizer on
this pool).  The golden rollouts' inputs run through an instrumented build of oracle/hsr_oracle.c; any report fails the test,
and the instrumented and plain builds must agree on the checksum of the final states."""
import shutil
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
GOLD = Path(__file__).parent / "golden"


@pytest.mark.skipif(not shutil.which("gcc"), reason="gcc not available")
@pytest.mark.parametrize("cfg", ["cfg3", "cfg4", "cupboard"])
def test_oracle_under_asan_ubsan(tmp_path, cfg):
    subprocess.check_call(["make", "-C", str(ROOT / "oracle"), "oracle_asan", "oracle_plain"], stdout=subprocess.DEVNULL)
    g = np.load(GOLD / f"golden_{cfg}.npz")
    q0, ctrl = np.ascontiguousarray(g["qpos0"], np.float64), np.ascontiguousarray(g["ctrl"], np.float64)
    inp = tmp_path / "in.bin"
    inp.write_bytes(struct.pack("<3i", q0.shape[0], q0.shape[1], ctrl.shape[1]) + q0.tobytes() + ctrl.tobytes())
    blob = ROOT / "hsr_env_amd" / "models" / f"{cfg}.hsrm"
    outs = []
    for exe in ("oracle_asan", "oracle_plain"):
        r = subprocess.run([str(ROOT / "oracle" / exe), str(blob), str(inp), "60"], capture_output=True, text=True, timeout=600,
                           env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
        assert r.returncode == 0, r.stderr[-2000:]
        assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
        outs.append(float(r.stdout.strip()))
    assert abs(outs[0] - outs[1]) <= 1e-9 * max(1.0, abs(outs[1]))
