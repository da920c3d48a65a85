"""Diagnostic: one substep from a saved (qpos, qvel, warm start, ctrl) of one env, device vs oracle.  usage: exp_sub.py FILE.npz CFG"""
import sys, os, numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
d = np.load(sys.argv[1]); cfg = sys.argv[2]
m = load_config(cfg); n = 4
sim = hs.BatchSim(m, n)
q = np.tile(d['qpos'], (n, 1)).astype(np.float32); v = np.tile(d['qvel'], (n, 1)).astype(np.float32); w = np.tile(d['warm'], (n, 1)).astype(np.float32)
ctrl = np.tile(d['ctrl'], (n, 1)).astype(np.float32)
sim.reset(qpos0=q, mocap=np.zeros((n, 3), np.float32))
sim.set_state(np.zeros(n, np.float32), q, v); sim.set_warmstart(w)
sim.set_debug(True)
obs, rew, done, ns = sim.step(ctrl, 1, -1, 0.0)
print('HSR_NFB', os.environ.get('HSR_NFB'), 'finite', bool(np.isfinite(obs[0]).all()), 'bad', bool(sim.bad_state()[0][0]), 'trips', int(sim.newton_trips()[0]))
print('qvel after', np.round(obs[0][m.nq:], 4).tolist())
from oracle.oracle import OracleSim
o = OracleSim(m)
o.qpos[:] = d['qpos']; o.qvel[:] = d['qvel']; o.qacc_warmstart[:] = d['warm']; o.ctrl[:] = d['ctrl']
o.step()
print('oracle: ncon', o.ncon(), 'nefc', o.nefc(), 'iterations', o.solver_niter(), 'bad', o.bad())
print('oracle qvel', np.round(np.asarray(o.qvel), 4).tolist())
