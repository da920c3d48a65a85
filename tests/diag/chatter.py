"""Diagnostic (CPU only): which contacts flicker in the envs that need many Newton iterations per substep."""
import sys, numpy as np
sys.path.insert(0, '.')   # run from the repo root: python tests/diag/<name>.py
from hsr_env_amd.compiler import load_config
from bench import sample_inputs
from oracle.oracle import OracleSim
m = load_config('cfg3'); n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
q0, goal = sample_inputs(m, n, 0, 0)
rng = np.random.default_rng(1)
ctrls = [rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (8192, m.nu)) for _ in range(4)]
gname = m.names['geom']
res = []
for e in range(n):
    o = OracleSim(m)
    o.qpos[:] = q0[e]; o.mocap_pos[:] = goal[e]
    its = []
    for k in range(3):
        o.ctrl[:] = ctrls[k][e]
        for t in range(300):
            o.step()
            if k == 2: its.append(o.solver_niter)
    res.append((np.mean(its), e, o))
res.sort(key=lambda r: -r[0])
print('mean niter over envs %.2f; top: %s' % (np.mean([r[0] for r in res]), [(round(r[0], 2), r[1]) for r in res[:8]]))
for mean_it, e, o in res[:3]:
    print('=== env', e, 'mean niter', round(mean_it, 2), 'qpos', np.round(o.qpos, 4))
    o.ctrl[:] = ctrls[3][e]
    for t in range(14):
        o.step()
        c = o.contacts()
        d = {}
        for r in c:
            key = (gname[int(r[13])], gname[int(r[14])])
            d.setdefault(key, []).append(r[12])
        print(' t%02d niter %2d nefc %2d ' % (t, o.solver_niter, o.nefc) + '; '.join('%s-%s x%d dmin %.2e' % (k[0], k[1], len(v), min(v)) for k, v in d.items()))
