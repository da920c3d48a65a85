"""Generates tests/golden/golden_cfg*.npz from the fp64 oracle (oracle/hsr_oracle.c): seeded inputs and the
per-substep qpos/qvel they produce.  The reference itself cannot produce vectors (MuJoCo is absent),
so these pin the *restated* algorithm: any later change of oracle or kernels that moves them is visible.

    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from hsr_env_amd.compiler import load_config  # noqa: E402
from oracle.oracle import OracleSim  # noqa: E402

NENV, NSUB = 6, 80


def inputs(m, seed):
    rng = np.random.default_rng(seed)
    q = np.tile(m.qpos0, (NENV, 1))
    blocks = m.free_joint_qadrs()
    nb = len(blocks)
    for b, a in enumerate(blocks):
        yaw = rng.uniform(-np.pi, np.pi, NENV)
        q[:, a] = rng.uniform(-0.1, 0.1, NENV)
        q[:, a + 1] = rng.uniform(-0.2, 0.2, NENV) if nb == 1 else rng.uniform(-0.04, 0.04, NENV) + 0.13 * (b - (nb - 1) / 2)
        q[:, a + 2] = 0.422 + (0.03 if b == 0 else 0.0) * (np.arange(NENV) % 2)     # every other env drops its block 3 cm
        q[:, a + 3] = np.cos(yaw / 2); q[:, a + 6] = np.sin(yaw / 2)
    ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (NENV, m.nu))
    return q, ctrl


def main():
    for cfg in ("cfg1", "cfg2", "cfg3", "cfg4", "cupboard"):
        m = load_config(cfg)
        q0, ctrl = inputs(m, 1234)
        traj_q = np.zeros((NENV, NSUB, m.nq)); traj_v = np.zeros((NENV, NSUB, m.nv)); ncon = np.zeros((NENV, NSUB), np.int32)
        for e in range(NENV):
            s = OracleSim(m)
            s.qpos[:] = q0[e]; s.ctrl[:] = ctrl[e]
            for k in range(NSUB):
                s.step()
                traj_q[e, k], traj_v[e, k], ncon[e, k] = s.qpos, s.qvel, s.ncon
        np.savez_compressed(Path(__file__).parent / f"golden_{cfg}.npz", qpos0=q0, ctrl=ctrl, qpos=traj_q, qvel=traj_v, ncon=ncon)
        print(cfg, "max ncon", ncon.max(), "bytes", (Path(__file__).parent / f"golden_{cfg}.npz").stat().st_size)


if __name__ == "__main__":
    main()
