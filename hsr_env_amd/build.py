"""Build libhsrsim.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot)."""
from __future__ import annotations

import subprocess
from pathlib import Path

HERE = Path(__file__).parent
SRC = HERE / "csrc" / "hsrsim.hip"
OUT = HERE / "libhsrsim.so"
HIPCC = "/opt/rocm/bin/hipcc"
# code-generation flags of the product build (tests/test_isa_hazards.py and tools/build_variants.py compile with the same ones).
# -disable-machine-licm (round 6): the straight-line kinematics of kin3.h use ~150 float literals per substep; hoisted out of the substep loop as
# registers they spilled (509 spilled VGPRs); without the hoisting the cfg3 instance has 2 spilled VGPRs and 12 B of scratch per lane
CODEGEN_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-disable-machine-licm"]


def build_lib(force: bool = False, verbose: bool = False, timing: bool = False, life: bool = False) -> Path:
    """timing=True builds the diagnostic variant libhsrsim_timing.so (in-kernel phase stamps; never benchmarked)."""
    out = HERE / "libhsrsim_timing.so" if timing else HERE / "libhsrsim_life.so" if life else OUT
    deps = list((HERE / "csrc").glob("*")) + [HERE.parent / "include" / "hsrsim.h"]
    if not force and out.exists() and all(out.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return out
    cmd = [HIPCC, *CODEGEN_FLAGS, "-shared", "-fPIC", "-Wno-unused-result", "-Wno-unused-value",
           "-o", str(out), str(SRC)]
    if timing:
        cmd.append("-DHSR_PHASE_TIMING")
    elif life:
        cmd.append("-DHSR_BLOCK_LIFE")     # product kernel + two stamps per workgroup (tools/block_life.py)
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
