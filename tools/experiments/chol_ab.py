"""cfg4: sparse factorisation of the Newton Hessian (default) against the dense one (test hook 128) on the bench's inputs, same binary.
usage: [HSR_LIB=...timing.so] python tools/experiments/chol_ab.py [n_envs]"""
import sys, time, os
import numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs
m = load_config('cfg4'); n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
q0, goal = sample_inputs(m, n, 0, 0)
bid = m.body_id(m.block_body())
for hook, queue in ((0, -1), (128, -1), (0, 0), (128, 0), (0, -1), (128, -1)):
    sim = hs.BatchSim(m, n)
    if queue == 0:
        sim.set_queue(0)
    sim.set_debug(hook)
    sim.reset(qpos0=q0, mocap=goal)
    rng = np.random.default_rng(1)
    ts = []
    for k in range(8):
        ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
        t0 = time.perf_counter(); sim.step(ctrl, 300, bid, 0.05); ts.append(time.perf_counter() - t0)
    print('hook', hook, 'queue', 'auto' if queue < 0 else 'off', 'ms per env-step (steps 3..7):', np.round(1e3 * np.array(ts[3:]), 2), 'mean %.2f' % (1e3 * np.mean(ts[3:])))
    sim.close()
