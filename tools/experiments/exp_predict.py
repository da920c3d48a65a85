"""Experiment: how well do the Newton trips of one stretch of an env-step predict the trips of the next stretches?"""
import sys, numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
m = load_config(cfg); n = 8192
q0, goal = sample_inputs(m, n, 0, 0)
bid = m.body_id('block0')
sim = hs.BatchSim(m, n)
sim.reset(qpos0=q0, mocap=goal)
rng = np.random.default_rng(1)
prev_total = None
for k in range(6):
    ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
    parts = []
    for seg in (20, 30, 50, 100, 100):
        sim.step(ctrl, seg, bid, 0.05)
        parts.append(sim.newton_trips().astype(np.float64) / seg)
    parts = np.array(parts); total = (parts * np.array([20, 30, 50, 100, 100])[:, None]).sum(0) / 300
    if k >= 3:
        rest = (parts[2:] * np.array([50, 100, 100])[:, None]).sum(0) / 250
        top = np.argsort(-rest)[: n // 20]
        def cover(pred):   # share of the 5 % hardest (by the rest of the env-step) among the 40 % the predictor ranks first
            first = set(np.argsort(-pred)[: int(0.4 * n)]); return np.mean([t in first for t in top])
        print('env-step', k, 'trips/substep mean %.2f p95 %.2f max %.2f' % (total.mean(), np.percentile(total, 95), total.max()),
              '| corr(first 20, rest) %.2f  corr(first 50, rest) %.2f  corr(previous env-step, this) %.2f' % (
                  np.corrcoef(parts[0], rest)[0, 1], np.corrcoef((parts[0] * 20 + parts[1] * 30) / 50, rest)[0, 1], np.corrcoef(prev_total, total)[0, 1]),
              '| hardest 5 %% covered by top 40 %% of: first 20: %.2f first 50: %.2f previous step: %.2f' % (cover(parts[0]), cover((parts[0] * 20 + parts[1] * 30) / 50), cover(prev_total)),
              '| segment corr', np.corrcoef(parts)[0].round(2))
    prev_total = total
sim.close()
