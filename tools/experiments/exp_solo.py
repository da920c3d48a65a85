"""Experiment: launch time as a function of how the hard envs are distributed over the waves (host-side permutation)."""
import sys, numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs
m = load_config('cfg3'); n = 8192
q0, goal = sample_inputs(m, n, 0, 0)
sim = hs.BatchSim(m, n)
sim.reset(qpos0=q0, mocap=goal)
rng = np.random.default_rng(1)
bid = m.body_id('block0')
for k in range(3):
    ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
    sim.step(ctrl, 300, bid, 0.05)
t0, qp, qv = sim.get_state(); w0 = sim.get_warmstart()
sim.set_persistent(False)
it = np.zeros(n)
T = 30
for t in range(T):
    sim.step(ctrl, 1, -1, 0.0)
    it += sim.get_field(hs.F_NITER)
it /= T
print('niter percentiles 50/90/97/99/100', np.percentile(it, [50, 90, 97, 99, 100]).round(2))
sim.close()
ctrl2 = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)

def run(order):
    nn = len(order)
    s2 = hs.BatchSim(m, nn)
    s2.reset(qpos0=qp[order], mocap=goal[order])
    s2.set_warmstart(w0[order]); s2.set_state(np.zeros(nn, np.float32), qp[order], qv[order])
    s2.set_profiling(True)
    s2.step(ctrl2[order], 300, bid, 0.05)
    ms = s2.last_timing()[1][2]
    s2.step(ctrl2[order], 300, bid, 0.05)
    ms2 = s2.last_timing()[1][2]
    s2.close()
    return round(ms, 2), round(ms2, 2)

ident = np.arange(n)
print('identity order: kernel ms', run(ident))
srt = np.argsort(-it)
print('sorted by hardness (hard packed 4 per wave, first): kernel ms', run(srt))
# one hard env per wave + three of the easiest: hardest first
nw = n // 4
order = np.empty(n, dtype=int)
order[0::4] = srt[:nw]                        # the nw hardest, one per wave
rest = srt[nw:][::-1]                         # easiest first
order[1::4] = rest[0:nw]; order[2::4] = rest[nw:2 * nw]; order[3::4] = rest[2 * nw:3 * nw]
print('one of the hardest quarter per wave + three easiest-first: kernel ms', run(order))
order2 = order.reshape(nw, 4)[::-1].reshape(-1)
print('same, easy waves dispatched first: kernel ms', run(order2))
