"""Diagnostic (timing build): which convex pairs reach the narrowphase of the persistent kernel, and which of them run MPR - the
histograms are kept by a -DHSR_PHASE_TIMING -DHSR_PAIR_HIST build in the unused tail of its per-workgroup stamp buffer."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs
m = load_config('cfg3'); n = 8192
q0, goal = sample_inputs(m, n, 0, 0)
sim = hs.BatchSim(m, n)
sim.reset(qpos0=q0, mocap=goal)
rng = np.random.default_rng(1)
for k in range(3):
    ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
    sim.step(ctrl, 300, m.body_id('block0'), 0.05)
L = sim._L
L.hsr_batch_block_times.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * (40 * 8192))()
L.hsr_batch_block_times(sim._b, buf, 8192)
a = np.array(list(buf), dtype=np.int64)
h = a[40 * 4096: 40 * 4096 + 512]; r = a[40 * 4096 + 512: 40 * 4096 + 1024]
gn = m.names['geom']
tot = 3 * 300 * n
print('convex items per env-substep %.3f, MPR runs %.3f' % (h.sum() / tot, r.sum() / tot))
o = np.argsort(-h)
for p in o[:25]:
    if h[p] == 0: break
    g1, g2 = int(m.pair_geom1[p]), int(m.pair_geom2[p])
    print('pair %3d %-34s %-34s nv %3d %3d  items/env-substep %.4f  mpr runs %.4f' % (p, gn[g1], gn[g2], m.geom_meshnum[g1], m.geom_meshnum[g2], h[p] / tot, r[p] / tot))
