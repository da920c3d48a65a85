"""BASELINE config 1, literally: the reference's driver command (README.md:5, hsr/control.py:66-86) on the product path - `python -m hsr_env_amd.control`
with the reference's flags, as a child process on the GPU box, once with one env (the reference's own case) and once with BASELINE config 2's 4096
envs and a block."""
import re
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
FLAGS = ["--block-space", "(-.1,.1)(-.2,.2)(.422,.422)(-3.14,3.14)", "--steps-per-action=300", "--geofence=.5", "--goal-space", "(-.1,.1)(-.2,.2)(.422,.422)",
         "--use-dof", "slide_x", "--use-dof", "slide_y", "--env-steps", "3"]


@pytest.mark.parametrize("extra,n", [(["--n-envs", "1"], 1), (["--n-envs", "4096", "--n-blocks", "1"], 4096)])
def test_reference_driver_command_runs_on_the_product_path(extra, n):
    p = subprocess.run([sys.executable, "-m", "hsr_env_amd.control"] + FLAGS + extra, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if "env-steps/s" in l][-1]
    mt = re.fullmatch(r"3 env-steps x (\d+) envs in ([\d.]+) s -> ([\d.]+) env-steps/s", line.strip())
    assert mt and int(mt.group(1)) == n and float(mt.group(3)) > 0, line
    print(line)
