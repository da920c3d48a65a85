"""Diagnostic (lifetime build, HSR_LIB=.../libhsrsim_life.so or a var_*_l.so): when does every ENV finish its env-step, and which envs were
handed over to solo servers when.  The launch ends with its last env; this shows who that is.
    HSR_SOLO=128 HSR_LIB=$PWD/hsr_env_amd/libhsrsim_life.so python tools/env_life.py [N] [cfg]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs, GEOFENCE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cfg = sys.argv[2] if len(sys.argv) > 2 else 'cfg3'
m = load_config(cfg)
q0, goal = sample_inputs(m, n, 0, 0)
sim = hs.BatchSim(m, n)
sim.reset(qpos0=q0, mocap=goal)
rng = np.random.default_rng(1)
bid = m.body_id(m.block_body()) if m.block_body() else -1
L = sim._L
L.hsr_batch_block_times.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
ebuf = (C.c_ulonglong * (2 * 8192))()
for k in range(int(os.environ.get('STEPS', 3))):
    ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
    L.hsr_batch_block_times(sim._b, ebuf, -1)            # clears the env stamps
    sim.step(ctrl, 300, bid, GEOFENCE)
trips = sim.newton_trips()
nb = 4096
bbuf = (C.c_ulonglong * (40 * nb))()
L.hsr_batch_block_times(sim._b, bbuf, nb)
b = np.frombuffer(bbuf, dtype=np.uint64).astype(np.int64).reshape(nb, 40)
used = b[:, 0] > 0
t0 = b[used, 0].min()
L.hsr_batch_block_times(sim._b, ebuf, -1)
e = np.frombuffer(ebuf, dtype=np.uint64).astype(np.int64).reshape(2, 8192)[:, :n]
fin = (e[0] - t0) / 100.0                               # us
ho_sub = e[1] & 0xffff; ho_t = ((e[1] >> 16) - t0) / 100.0
handed = ho_sub > 0
print(f'{cfg} x {n}: launch span {(b[used, 1].max() - t0) / 1e5:.2f} ms; handed over {int(handed.sum())} envs')
pc = [0, 50, 90, 99, 99.9, 100]
print('env finish percentiles (ms)', pc, (np.percentile(fin, pc) / 1e3).round(2))
if handed.any():
    print('  handed-over envs: finish percentiles', (np.percentile(fin[handed], pc) / 1e3).round(2), ' hand-over substep percentiles', np.percentile(ho_sub[handed], [0, 25, 50, 75, 100]))
    print('  others          : finish percentiles', (np.percentile(fin[~handed], pc) / 1e3).round(2))
o = np.argsort(-fin)[:24]
print('last 24 envs: finish ms', (fin[o] / 1e3).round(2))
print('   handed over at substep', ho_sub[o], '\n   at ms', np.where(handed[o], (ho_t[o] / 1e3).round(2), 0))
print('   trips (last 100 substeps)', trips[o])
