"""Diagnostic: workgroup lifetimes of the PRODUCT kernel (libhsrsim_life.so = product build + two stamps per workgroup), with
the per-env Newton trip counts the product kernel keeps anyway.  The phase profile needs the heavier libhsrsim_timing.so
(tools/block_times.py), whose stamps change the register allocation."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs
import os
cfg = os.environ.get('HSR_CFG', 'cfg3')
m = load_config(cfg); n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8192
q0, goal = sample_inputs(m, n, 0, 0)
sim = hs.BatchSim(m, n); sim.set_graph(False)
if os.environ.get('HSR_LIFE_QUEUE', '0') == '0':
    sim.set_queue(0)          # task lifetimes of the static assignment (with the work queue a workgroup's stamps span all its tasks)
sim.reset(qpos0=q0, mocap=goal)
rng = np.random.default_rng(1)
for k in range(3):
    ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
    sim.step(ctrl, 300, m.body_id(m.block_body()) if m.block_body() else -1, 0.05)
epb = 4 if m.nv <= 16 else 2
nb = n // epb
L = sim._L
L.hsr_batch_block_times.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * (40 * nb))()
L.hsr_batch_block_times(sim._b, buf, nb)
a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(nb, 40)
t0 = a[:, 0].min()
life = (a[:, 1] - a[:, 0]) / 100.0
print('blocks', nb, 'kernel span %.2f ms' % ((a[:, 1].max() - t0) / 1e5))
pc = [0, 10, 25, 50, 75, 90, 99, 100]
print('life percentiles us', pc, np.percentile(life, pc).round(0))
print('p100 / p50 = %.2f   mean / p100 = %.2f' % (life.max() / np.median(life), life.mean() / life.max()))
o = np.argsort(life)
trips = a[:, 4] / 300.0
print('Newton trips / substep: mean %.2f' % trips.mean())
for nm, idx in (('slowest 10', o[-10:]), ('median 10', o[nb // 2 - 5: nb // 2 + 5])):
    print(nm, 'life ms', (life[idx] / 1e3).round(1), '\n   trips', trips[idx].round(2), ' items', (a[idx, 6] / 300).round(1), ' nefc', (a[idx, 7] / 300).round(1))

if '--json' in sys.argv:
    import json
    from pathlib import Path
    out = Path(os.environ.get('HSR_ROUND_DIR', 'gpurun_out/r6')) / ('block_life_%s.json' % cfg)
    out.parent.mkdir(parents=True, exist_ok=True)
    out.write_text(json.dumps({cfg: {"workgroups": int(nb), "p0_ms": float(np.percentile(life, 0) / 1e3), "p50_ms": float(np.median(life) / 1e3),
                                     "p90_ms": float(np.percentile(life, 90) / 1e3), "p99_ms": float(np.percentile(life, 99) / 1e3),
                                     "p100_ms": float(life.max() / 1e3), "p100_over_p50": float(life.max() / np.median(life)),
                                     "mean_over_p100": float(life.mean() / life.max()), "newton_trips_per_substep_mean": float(trips.mean()),
                                     "source": "tools/block_life.py with libhsrsim_life.so (product kernel + two s_memrealtime stamps per workgroup), third env-step"}}))
