"""The solver's hand-written v_fmac_f32_dpp / v_mov_b32_dpp (csrc/solve_g.h: fmac_bcast, gbcast_after_asm) are invisible to
LLVM's hazard recogniser.  gfx9 needs two wait states between a VALU write of a VGPR and a DPP read of it; this test compiles
the kernels to assembly (no GPU needed) and checks that no DPP instruction reads, as its DPP source, a register that one of
the hand-written instructions wrote fewer than two instructions earlier without an s_nop in between."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not shutil.which(HIPCC), reason="hipcc not available")
def test_no_dpp_read_right_after_inline_asm_write(tmp_path):
    out = tmp_path / "hsrsim.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                           "-Wno-unused-result", "-Wno-unused-value", "-o", str(out), str(ROOT / "hsr_env_amd" / "csrc" / "hsrsim.hip")],
                          stderr=subprocess.DEVNULL)
    lines = out.read_text().split("\n")
    ins = [t.strip() for t in lines if t.strip() and not t.strip().startswith((".", ";", "//")) and not t.strip().endswith(":")]
    n_asm, bad = 0, []
    for k, t in enumerate(ins):
        m = re.match(r"v_(?:fmac_f32|mov_b32)_dpp (v\d+),", t)
        if not m or "row_newbcast" not in t:
            continue
        n_asm += 1
        dst = m.group(1)
        for t2 in ins[k + 1:k + 3]:
            if t2.startswith("s_nop"):
                break
            if "_dpp" in t2:
                ops = [o.strip().split()[0] for o in t2.split(None, 1)[1].split(",")]
                if len(ops) > 1 and ops[1] == dst and not (t2.startswith("v_fmac_f32_dpp") and ops[0] == dst and ops[1] != dst):
                    bad.append((t, t2))
    assert n_asm > 100, "the hand-written DPP instructions were not found in the assembly"
    assert not bad, bad[:3]
