"""Diagnostic: per-phase cycle shares of k_solve_g (needs libhsrsim_timing.so; HSR_LIB points at it)."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
sys.path.insert(0, 'tests')
from bench import sample_inputs
m = load_config('cfg3'); n = 8192
q0, goal = sample_inputs(m, n, 0, 0)
sim = hs.BatchSim(m, n); sim.set_graph(False)
sim.reset(qpos0=q0, mocap=goal)
rng = np.random.default_rng(1)
for k in range(3):
    ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
    sim.step(ctrl, 300, m.body_id('block0'), 0.05)
L = sim._L
L.hsr_batch_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 32)()
L.hsr_batch_phase_cycles(sim._b, buf)
ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
sim.step(ctrl, 300, m.body_id('block0'), 0.05)
L.hsr_batch_phase_cycles(sim._b, buf)
v = np.array(list(buf), dtype=np.float64)[:15]
names = ['A load', 'B/C M+bias', 'D chol M+solve', 'E1-2 limits+compact', 'E3 contact rec', 'E4-5 J rows', 'F0 warm evals', 'F grad',
         'F hess', 'F chol+solve', 'F ls setup+ls', 'F update+eval', 'out', 'G euler', 'tail']
tot = v.sum()
for nm, x in zip(names, v):
    print(f'{nm:22s} {x / tot * 100:6.2f} %   {x / (2048 * 300):9.0f} cyc/block/substep')
print('total cyc/block/substep', tot / (2048 * 300))
