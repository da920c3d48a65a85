"""Build libhsrsim.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot)."""
from __future__ import annotations

import subprocess
from pathlib import Path

HERE = Path(__file__).parent
SRC = HERE / "csrc" / "hsrsim.hip"
OUT = HERE / "libhsrsim.so"
HIPCC = "/opt/rocm/bin/hipcc"


def build_lib(force: bool = False, verbose: bool = False) -> Path:
    deps = list((HERE / "csrc").glob("*")) + [HERE.parent / "include" / "hsrsim.h"]
    if not force and OUT.exists() and all(OUT.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return OUT
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result", "-Wno-unused-value",
           "-o", str(OUT), str(SRC)]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
