"""Diagnostic: replay one saved env (tools/exp_bad.py) substep by substep; prints the first substep that goes non-finite and the
Newton iterations around it.  usage: exp_replay.py FILE.npz CFG"""
import sys, numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
d = np.load(sys.argv[1]); cfg = sys.argv[2]
m = load_config(cfg); n = 64
sim = hs.BatchSim(m, n)
q = np.tile(d['qpos'], (n, 1)).astype(np.float32); v = np.tile(d['qvel'], (n, 1)).astype(np.float32)
sim.reset(qpos0=q, mocap=np.zeros((n, 3), np.float32))
sim.set_state(np.zeros(n, np.float32), q, v)
ctrl = np.tile(d['ctrl'], (n, 1)).astype(np.float32)
hist = []
for k in range(300):
    obs, rew, done, ns = sim.step(ctrl, 1, -1, 0.0)
    tr = sim.newton_trips()
    hist.append((obs[0].copy(), int(tr[0])))
    if not np.isfinite(obs[0]).all() or sim.bad_state()[0][0]:
        print('substep', k, 'non-finite / bad; trips', tr[0], 'same in all copies:', bool((np.isfinite(obs).all(1) == np.isfinite(obs[0]).all()).all()))
        for j in range(max(0, k - 4), k + 1):
            print('  substep', j, 'trips', hist[j][1], 'qvel blocks', np.round(hist[j][0][m.nq + 7:], 3).tolist())
        np.savez('gpurun_out/r2/replay_before.npz', qpos=hist[k - 1][0][:m.nq], qvel=hist[k - 1][0][m.nq:], ctrl=d['ctrl'])
        break
else:
    print('300 substeps finite; final qpos', np.round(obs[0][:m.nq], 4).tolist())
