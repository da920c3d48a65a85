import sys, os, numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs
m = load_config('cfg3'); n = 8192
q0, goal = sample_inputs(m, n, 0, 0)
bid = m.body_id('block0')
sched = int(sys.argv[1])
sim = hs.BatchSim(m, n)
sim.set_schedule(bool(sched))
sim.reset(qpos0=q0, mocap=goal)
rng = np.random.default_rng(1)
sim.set_profiling(True)
ms = []
for k in range(8):
    ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
    sim.step(ctrl, 300, bid, 0.05)
    ms.append(round(sim.last_timing()[1][2], 2))
print(' '.join(sys.argv[1:]), {k: v for k, v in os.environ.items() if k.startswith('HSR_')}, 'kernel ms', ms, 'mean of last 5: %.2f' % np.mean(ms[3:]))
sim.close()
