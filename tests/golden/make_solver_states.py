"""Generates tests/golden/solver_states.npz: states (qpos, qvel, qacc_warmstart, ctrl) at which the constraint solver works hardest, taken
from oracle rollouts of the regimes that set the launch time (hsr/env.py:123 `self.sim.step()` -> mj_fwdConstraint):
  pinch     cfg3, the block dropped between the fingers (tests/test_gpu_hotpath.py::test_pinched_block_contacts_follow_the_oracle)
  cupboard  the env-step parity inputs of the cupboard scene (seed 12) - incl. every substep at which the line search needed its bisection
            safeguard (the round-4 outliers: before the safeguard the search hopped between its bracket ends and the solve ended unconverged)
  bench     cfg3 under the bench's inputs and ctrl distribution
  cfg4      three blocks
A state is kept when its solve took many Newton iterations or many line-search evaluations.  The file holds inputs only; the expected values
are recomputed by the tests (tests/test_oracle_optimality.py restates the cost in numpy and minimises it with scipy).
    python tests/golden/make_solver_states.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from hsr_env_amd.compiler import load_config          # noqa: E402
from oracle.oracle import OracleSim                    # noqa: E402
from test_gpu_parity import random_states              # noqa: E402
from bench import sample_inputs                        # noqa: E402

CFGS = ["cfg3", "cupboard", "cfg4"]
NQ, NV, NU = 28, 25, 7


def rollout(m, q, v, ctrl, nsub, keep, mocap=None):
    """-> list of (score, state) of one env; keep(niter, ls_max, refused) -> score or None.  ctrl: one vector, or one per env-step of nsub substeps"""
    o = OracleSim(m)
    ctrls = np.atleast_2d(ctrl)
    o.qpos[:] = q; o.qvel[:] = v
    if mocap is not None:
        o.mocap_pos[:] = mocap
    out = []
    for k in range(nsub * len(ctrls)):
        o.ctrl[:] = ctrls[k // nsub]
        st = (o.qpos.copy(), o.qvel.copy(), o.qacc_warmstart.copy(), o.ctrl.copy())
        o.step()
        if o.bad:
            break
        if o.nefc == 0:
            continue
        _, ls_max, refused, _ = o.solver_stats()
        sc = keep(o.solver_niter, ls_max, refused)
        if sc is not None:
            out.append((sc, k, st))
    return out


def pick(cands, n):
    """the n highest scores, at most 6 per env so that one env does not fill the fixture"""
    cands.sort(key=lambda t: -t[0])
    per, out = {}, []
    for sc, e, k, st in cands:
        if per.get(e, 0) >= 6:
            continue
        per[e] = per.get(e, 0) + 1
        out.append((sc, e, k, st))
        if len(out) == n:
            break
    return out


def main():
    models = {c: load_config(c) for c in CFGS}
    rows = []          # (cfg index, regime, env, substep, state)

    # pinch
    m = models["cfg3"]
    n = 96
    rng = np.random.default_rng(3)
    q, v, ctrl = random_states(m, n, rng)
    bl, br = m.body_id("hand_l_distal_link"), m.body_id("hand_r_distal_link")
    a = m.free_joint_qadrs()[0]
    for e in range(n):
        o = OracleSim(m); o.qpos[:] = q[e]; o.forward()
        q[e, a:a + 3] = 0.5 * (o.body_xpos(bl) + o.body_xpos(br)) + rng.uniform(-0.01, 0.01, 3)
        quat = rng.normal(size=4); q[e, a + 3:a + 7] = quat / np.linalg.norm(quat)
    cands = []
    for e in range(n):
        for sc, k, st in rollout(m, q[e], np.zeros(m.nv), ctrl[e], 40, lambda it, ls, rf: it + 0.1 * ls if it >= 3 else None):
            cands.append((sc, e, k, st))
    for sc, e, k, st in pick(cands, 90):
        rows.append((0, 0, e, k, st))

    # cupboard: every substep that needed the bisection safeguard, then the hardest others
    m = models["cupboard"]
    n = 64
    rng = np.random.default_rng(12)
    q, v, ctrl = random_states(m, n, rng)
    safeguard, cands = [], []
    for e in range(n):
        for sc, k, st in rollout(m, q[e], np.zeros(m.nv), ctrl[e], 300, lambda it, ls, rf: (100 + ls if (ls > 6 or rf) else (it + 0.1 * ls if it >= 5 else None))):
            (safeguard if sc >= 100 else cands).append((sc, e, k, st))
    print("cupboard: substeps whose line search went past six evaluations (the bisection safeguard):", len(safeguard), "in envs", sorted({e for _, e, _, _ in safeguard}))
    for sc, e, k, st in pick(safeguard, 40) + pick(cands, 40):
        rows.append((1, 1, e, k, st))

    # bench-like cfg3
    m = models["cfg3"]
    n = 48
    q0, goal = sample_inputs(m, n, 0, 0)
    rng = np.random.Generator(np.random.Philox(key=[1, 0]))
    lo, hi = m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1]
    cands = []
    for e in range(n):
        c = rng.uniform(lo, hi, (4, m.nu))          # four env-steps: nothing is in contact during the first one after a reset
        for sc, k, st in rollout(m, q0[e].astype(np.float64), np.zeros(m.nv), c, 300, lambda it, ls, rf: it + 0.1 * ls + (50 if ls > 6 else 0) if it >= 5 else None, mocap=goal[e]):
            cands.append((sc, e, k, st))
    for sc, e, k, st in pick(cands, 40):
        rows.append((0, 2, e, k, st))

    # cfg4
    m = models["cfg4"]
    n = 16
    rng = np.random.default_rng(5)
    q, v, ctrl = random_states(m, n, rng)
    cands = []
    for e in range(n):
        for sc, k, st in rollout(m, q[e], np.zeros(m.nv), ctrl[e], 200, lambda it, ls, rf: it + 0.1 * ls + (50 if ls > 6 else 0) if it >= 4 else None):
            cands.append((sc, e, k, st))
    for sc, e, k, st in pick(cands, 40):
        rows.append((2, 3, e, k, st))

    N = len(rows)
    cfg = np.zeros(N, np.int32); regime = np.zeros(N, np.int32); env = np.zeros(N, np.int32); sub = np.zeros(N, np.int32)
    qpos = np.zeros((N, NQ)); qvel = np.zeros((N, NV)); warm = np.zeros((N, NV)); ct = np.zeros((N, NU))
    for i, (ci, rg, e, k, st) in enumerate(rows):
        cfg[i], regime[i], env[i], sub[i] = ci, rg, e, k
        qpos[i, :len(st[0])] = st[0]; qvel[i, :len(st[1])] = st[1]; warm[i, :len(st[2])] = st[2]; ct[i, :len(st[3])] = st[3]
    out = ROOT / "tests" / "golden" / "solver_states.npz"
    np.savez_compressed(out, cfg_names=np.array(CFGS), regime_names=np.array(["pinch", "cupboard", "bench", "cfg4"]), cfg=cfg, regime=regime, env=env, substep=sub,
                        qpos=qpos, qvel=qvel, warm=warm, ctrl=ct)
    print(f"{N} states -> {out} ({out.stat().st_size} bytes); per regime {np.bincount(regime).tolist()}")


if __name__ == "__main__":
    main()
