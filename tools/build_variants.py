"""Build diagnostic / experimental variants of libhsrsim in parallel:  python tools/build_variants.py NAME[:-DFLAG[,-DFLAG..]] ...
Each NAME yields hsr_env_amd/var_NAME.so (product flags), var_NAME_t.so (-DHSR_PHASE_TIMING) and var_NAME_l.so (-DHSR_BLOCK_LIFE).
--cfg3 compiles only the cfg3 instance of the persistent kernel (-DHSR_DEV_CFG3); --only p,t,l restricts the kinds."""
import subprocess, sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
HERE = Path(__file__).resolve().parent.parent / "hsr_env_amd"
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from hsr_env_amd.build import CODEGEN_FLAGS  # noqa: E402
BASE = ["/opt/rocm/bin/hipcc", *CODEGEN_FLAGS, "-shared", "-fPIC", "-w"]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
cfg3 = "--cfg3" in sys.argv
kinds = "ptl"
for a in sys.argv[1:]:
    if a.startswith("--only="):
        kinds = a.split("=")[1]
jobs = []
for spec in args:
    name, _, fl = spec.partition(":")
    flags = [f for f in fl.split(",") if f] + (["-DHSR_DEV_CFG3"] if cfg3 else [])
    for k, suf, extra in (("p", "", []), ("t", "_t", ["-DHSR_PHASE_TIMING"]), ("l", "_l", ["-DHSR_BLOCK_LIFE"])):
        if k in kinds:
            jobs.append(BASE + flags + extra + ["-o", str(HERE / f"var_{name}{suf}.so"), str(HERE / "csrc" / "hsrsim.hip")])
with ThreadPoolExecutor(max_workers=6) as ex:
    for cmd, rc in zip(jobs, ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs)):
        print(cmd[-3], "rc", rc.returncode, rc.stderr[-2000:] if rc.returncode else "")
