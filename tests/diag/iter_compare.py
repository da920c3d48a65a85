"""Diagnostic: Newton iteration counts of the fp32 HIP solver vs the fp64 oracle on the hardest envs of the bench state."""
import sys, numpy as np
sys.path.insert(0, '.')   # run from the repo root: python tests/diag/<name>.py
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs
from oracle.oracle import OracleSim
m = load_config('cfg3'); n = 8192
q0, goal = sample_inputs(m, n, 0, 0)
sim = hs.BatchSim(m, n); sim.set_graph(False)
sim.reset(qpos0=q0, mocap=goal)
rng = np.random.default_rng(1)
for k in range(3):
    ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
    sim.step(ctrl, 300, m.body_id('block0'), 0.05)
sim.set_persistent(False)
t0, qp0, qv0 = sim.get_state(); w0 = sim.get_warmstart()
T = 40
it = np.zeros((T, n)); ne = np.zeros((T, n)); nc = np.zeros((T, n))
for t in range(T):
    sim.step(ctrl, 1, -1, 0.0)
    it[t] = sim.get_field(hs.F_NITER); ne[t] = sim.get_field(hs.F_NEFC); nc[t] = sim.get_field(hs.F_NCON)
mi = it.mean(0)
print('gpu niter/substep: mean %.2f  percentiles 50/90/99/100: %s' % (mi.mean(), np.percentile(mi, [50, 90, 99, 100]).round(2)))
print('hist of per-substep niter:', np.bincount(it.astype(int).ravel())[:30])
top = np.argsort(mi)[-6:]
for e in top:
    o = OracleSim(m)
    o.qpos[:] = qp0[e]; o.qvel[:] = qv0[e]; o.qacc_warmstart[:] = w0[e]; o.ctrl[:] = ctrl[e]; o.mocap_pos[:] = goal[e]
    oi = []; on = []
    for t in range(T):
        o.step(); oi.append(o.solver_niter); on.append(o.nefc)
    print('env', e, 'gpu niter', it[:, e].astype(int).tolist())
    print('        oracle niter', oi)
    print('        gpu nefc', ne[:, e].astype(int).tolist()[:20], 'oracle nefc', on[:20])
rnd = rng.integers(0, n, 200)
tot_o = 0
for e in rnd:
    o = OracleSim(m)
    o.qpos[:] = qp0[e]; o.qvel[:] = qv0[e]; o.qacc_warmstart[:] = w0[e]; o.ctrl[:] = ctrl[e]; o.mocap_pos[:] = goal[e]
    for t in range(T):
        o.step(); tot_o += o.solver_niter
print('200 random envs: gpu mean niter %.3f  oracle mean niter %.3f' % (it[:, rnd].mean(), tot_o / (200 * T)))
