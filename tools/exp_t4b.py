import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from test_gpu_parity import random_states
m = load_config('cfg4'); n = 256
rng = np.random.default_rng(21)
q, v, ctrl = random_states(m, n, rng)
goal = np.column_stack([rng.uniform(-0.1, 0.1, n), rng.uniform(-0.2, 0.2, n), np.full(n, 0.422)])
sim = hs.BatchSim(m, n); sim.set_persistent(True)
sim.reset(qpos0=q, mocap=goal)
e = 98
for t in range(60):
    obs, rew, done, ns = sim.step(ctrl, 1, -1, 0.0)
    w = sim.get_warmstart()
    print(t, 'qvel max %.3e qacc max %.3e' % (np.abs(obs[e, m.nq:]).max(), np.abs(w[e]).max()), 'bad', sim.bad_state()[0][e])
    if not np.isfinite(obs[e]).all(): break
