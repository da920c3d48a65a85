"""Diagnostic: as exp_bad.py, but env-step KSTEP is run as 300 one-substep launches (state, warm start and margin caches carry
over between launches) and env ENV is watched.  usage: exp_bad2.py CFG KSTEP ENV [chunk]"""
import sys, numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs, GEOFENCE, STEPS_PER_ACTION
cfg, kstep, env = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]); chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 1
m = load_config(cfg); n = 8192
q0, goal = sample_inputs(m, n, 0, 0)
rng = np.random.Generator(np.random.Philox(key=[1, 0]))
lo, hi = m.act_ctrlrange[:, 0].astype(np.float32), m.act_ctrlrange[:, 1].astype(np.float32)
ctrl = [rng.uniform(lo, hi, (n, m.nu)).astype(np.float32) for _ in range(kstep + 1)]
resets = [sample_inputs(m, n, 2 + k, 0) for k in range(kstep + 1)]
sim = hs.BatchSim(m, n); sim.reset(qpos0=q0, mocap=goal)
bid = m.body_id(m.block_body()) if m.block_body() else -1
for k in range(kstep):
    obs, rew, done, ns = sim.step(ctrl[k], STEPS_PER_ACTION, bid, GEOFENCE)
    sim.reset(mask=np.asarray(done, np.uint8), qpos0=resets[k][0], mocap=resets[k][1])
print('bad before the watched env-step:', int(sim.bad_state()[0].sum()))
prev = None
for j in range(0, STEPS_PER_ACTION, chunk):
    st = sim.get_state(); wm = sim.get_warmstart()
    obs, rew, done, ns = sim.step(ctrl[kstep], chunk, -1, 0.0)
    tr = sim.newton_trips()[env]
    fin = np.isfinite(obs[env]).all()
    if not fin or sim.bad_state()[0][env] or j % 50 == 0:
        print('substep', j, 'finite', bool(fin), 'bad', bool(sim.bad_state()[0][env]), 'trips', int(tr), 'qvel', np.round(obs[env][m.nq:], 3).tolist())
    if not fin or sim.bad_state()[0][env]:
        np.savez(f'gpurun_out/r2/sub_{cfg}_{env}.npz', qpos=st[1][env], qvel=st[2][env], warm=wm[env], ctrl=ctrl[kstep][env], time=st[0][env])
        break
    prev = obs[env].copy()
