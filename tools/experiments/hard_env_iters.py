"""Newton iterations of the hardest envs of the bench workload: the persistent kernel (fp32) against the fp64 oracle from the same
state.  Diagnostic only (tools/, not the product path)."""
import os, sys
import numpy as np
sys.path.insert(0, '.')
from hsr_env_amd import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs
from oracle.oracle import OracleSim
m = load_config('cfg3'); n = 8192
q0, goal = sample_inputs(m, n, 0, 0)
sim = hs.BatchSim(m, n); sim.set_graph(False)
sim.reset(qpos0=q0, mocap=goal)
rng = np.random.default_rng(1)
gb = m.body_id(m.block_body())
for k in range(2):
    ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
    sim.step(ctrl, 300, gb, 0.05)
st = sim.get_state(); warm = sim.get_warmstart()
ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
sim.step(ctrl, 300, gb, 0.05)
trips = sim.newton_trips()
o = np.argsort(trips)[::-1][:8]
print('state tuple', [np.asarray(x).shape for x in st])
print('gpu trips / substep (last 100):', (trips[o] / 100.).round(2), 'envs', o)
t, qpos, qvel = st[0], st[1], st[2]
np.savez('gpurun_out/r3/hard_states.npz', envs=o, qpos=qpos[o], qvel=qvel[o], warm=warm[o], ctrl=ctrl[o], goal=goal[o], trips=trips[o])
for e in o[:0]:
    od = OracleSim(m)
    od.qpos[:] = qpos[e]; od.qvel[:] = qvel[e]; od.qacc_warmstart[:] = warm[e]; od.ctrl[:] = ctrl[e]; od.mocap_pos[:] = goal[e]
    its = []; ncs = []
    for s in range(300):
        od.step(); its.append(od.solver_niter); ncs.append(od.ncon)
    its = np.array(its)
    print('env %5d gpu %.2f  oracle last100 %.2f  all %.2f  max %d  ncon %.1f  hist %s' % (e, trips[e] / 100., its[200:].mean(), its.mean(), its.max(), np.mean(ncs), np.bincount(its, minlength=12)[:12]))
