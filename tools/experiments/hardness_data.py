"""Round 5: what predicts which envs of an env-step turn hard?  Runs the bench's batch for six env-steps and saves, per env-step, the state the step starts from, its ctrl and the Newton
iterations the step then took over its last 100 substeps (BatchSim.newton_trips) - for offline analysis (gpurun_out/r5/hardness.npz)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs, GEOFENCE
m = load_config('cfg3'); n = 8192
q0, goal = sample_inputs(m, n, 0, 0)
sim = hs.BatchSim(m, n); sim.reset(qpos0=q0, mocap=goal)
rng = np.random.Generator(np.random.Philox(key=[1, 0]))
lo, hi = m.act_ctrlrange[:, 0].astype(np.float32), m.act_ctrlrange[:, 1].astype(np.float32)
bid = m.body_id(m.block_body())
Q, V, C, T, D = [], [], [], [], []
for k in range(6):
    ctrl = rng.uniform(lo, hi, (n, m.nu)).astype(np.float32)
    t, q, v = sim.get_state()
    obs, rew, done, ns = sim.step(ctrl, 300, bid, GEOFENCE)
    Q.append(q.copy()); V.append(v.copy()); C.append(ctrl.copy()); T.append(sim.newton_trips().copy()); D.append(np.asarray(done).copy())
    rq, rg = sample_inputs(m, n, 2 + k, 0)
    sim.reset(mask=np.asarray(done, np.uint8), qpos0=rq, mocap=rg)
np.savez_compressed('gpurun_out/r5/hardness.npz', qpos=np.array(Q), qvel=np.array(V), ctrl=np.array(C), trips=np.array(T), done=np.array(D))
print('saved', np.array(T).shape, 'trips mean', np.array(T).mean(axis=1))
