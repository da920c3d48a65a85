"""Experiment: does a short probe launch predict which envs are hard for the rest of the env-step?  The env-step is run as
probe + rest launches, the rest packed / ordered (k_schedule) by the Newton trips of the probe.  usage: exp_probe.py CFG PROBE SCHED"""
import sys, os, time, numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs
cfg, probe, sched = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
m = load_config(cfg); n = 8192
q0, goal = sample_inputs(m, n, 0, 0)
bid = m.body_id('block0')
sim = hs.BatchSim(m, n)
sim.reset(qpos0=q0, mocap=goal)
rng = np.random.default_rng(1)
sim.set_profiling(True)
ms = []
for k in range(8):
    ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
    t = 0.0
    if probe > 0:
        sim.set_schedule(False)
        sim.step(ctrl, probe, bid, 0.05); t += sim.last_timing()[1][2]
    sim.set_schedule(bool(sched))
    sim.step(ctrl, 300 - probe, bid, 0.05); t += sim.last_timing()[1][2]
    ms.append(round(t, 2))
print(cfg, 'probe', probe, 'sched', sched, 'kernel ms per env-step', ms, 'mean of last 5: %.2f' % np.mean(ms[3:]))
sim.close()
