"""Diagnostic (GPU): how far the 300-substep env-step parity statistics are from the bounds of tests/test_gpu_parity.py."""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from oracle.oracle import OracleSim
import test_gpu_parity as t
for cfg in ["cfg2", "cfg3", "cupboard"]:
    m = load_config(cfg); n = 64
    rng = np.random.default_rng(12)
    q, v, ctrl = t.random_states(m, n, rng)
    sim = hs.BatchSim(m, n)
    sim.set_state(np.zeros(n), q, np.zeros_like(v))
    obs, rew, done, ns = sim.step(ctrl, 300)
    errs = []
    for e in range(n):
        o = OracleSim(m); o.qpos[:] = q[e]; o.env_step(ctrl[e], 300)
        errs.append(np.abs(obs[e] - np.concatenate([o.qpos, o.qvel])).max())
    errs = np.array(errs)
    print(cfg, 'median %.2e p75 %.2e p90 %.2e max %.2e' % (np.median(errs), np.percentile(errs, 75), np.percentile(errs, 90), errs.max()))
    sim.close()
