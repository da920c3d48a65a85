import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from test_gpu_parity import random_states
m = load_config('cfg4'); n = 256
for rep in range(2):
    rng = np.random.default_rng(21)
    q, v, ctrl = random_states(m, n, rng)
    goal = np.column_stack([rng.uniform(-0.1, 0.1, n), rng.uniform(-0.2, 0.2, n), np.full(n, 0.422)])
    for persistent in (True, False):
        sim = hs.BatchSim(m, n); sim.set_persistent(persistent)
        sim.reset(qpos0=q, mocap=goal)
        obs, rew, done, ns = sim.step(ctrl, 60, m.body_id(m.block_body()), 0.02)
        bad, anyb = sim.bad_state()
        print('persistent', persistent, 'bad envs', np.where(bad)[0], 'finite', np.isfinite(obs).all())
        sim.close()
