import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def models():
    from hsr_env_amd.compiler import load_config
    return {k: load_config(k) for k in ("cfg1", "cfg2", "cfg3", "cfg4", "cupboard", "cfg3_setxml", "nq18", "nv11", "nv23", "static1", "meshrest4", "meshrest1")}
