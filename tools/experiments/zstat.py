import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
os.environ['HSR_LIB'] = 'hsr_env_amd/var_zs_l.so'
from hsr_env_amd import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs
m = load_config('cfg3'); n = 8192
q0, goal = sample_inputs(m, n, 0, 0)
sim = hs.BatchSim(m, n); sim.set_graph(False); sim.set_queue(0)
sim.reset(qpos0=q0, mocap=goal)
rng = np.random.default_rng(1)
for k in range(3):
    ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
    sim.step(ctrl, 300, m.body_id(m.block_body()), 0.05)
nb = n // 4
L = sim._L
L.hsr_batch_block_times.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * (40 * nb))()
L.hsr_batch_block_times(sim._b, buf, nb)
a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(nb, 40)
life = (a[:, 1] - a[:, 0]) / 100.0
o = np.argsort(life)
z = a[:, 7]
ncw = (z & 0xfffff) / 300.; top = ((z >> 20) & 0xfffff) / 300.; mid = ((z >> 40) & 0xfffff) / 300.
trips = a[:, 4] / 300.
for nm, idx in (('slowest 20', o[-20:]), ('median 10', o[nb // 2 - 5: nb // 2 + 5])):
    print(nm, 'life ms', (life[idx] / 1e3).round(1)); print('  trips', trips[idx].round(2)); print('  contact-slots/substep', ncw[idx].round(1)); print('  top (active env)', top[idx].round(1)); print('  mid', mid[idx].round(1))
print('all: trips %.2f slots %.2f top %.2f mid %.2f' % (trips.mean(), ncw.mean(), top.mean(), mid.mean()))
