"""Diagnostic: replay the bench's inputs (cfg from argv, 8192 envs) on the host API and report when an env turns bad, with its
state before that env-step saved for a replay against the oracle.  usage: exp_bad.py CFG [steps]"""
import sys, numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs, GEOFENCE, STEPS_PER_ACTION
cfg = sys.argv[1]; total = int(sys.argv[2]) if len(sys.argv) > 2 else 14
m = load_config(cfg); n = 8192
q0, goal = sample_inputs(m, n, 0, 0)
rng = np.random.Generator(np.random.Philox(key=[1, 0]))
lo, hi = m.act_ctrlrange[:, 0].astype(np.float32), m.act_ctrlrange[:, 1].astype(np.float32)
ctrl = [rng.uniform(lo, hi, (n, m.nu)).astype(np.float32) for _ in range(total)]
resets = [sample_inputs(m, n, 2 + k, 0) for k in range(total)]
sim = hs.BatchSim(m, n); sim.reset(qpos0=q0, mocap=goal)
bid = m.body_id(m.block_body()) if m.block_body() else -1
seen = np.zeros(n, bool)
for k in range(total):
    before = sim.get_state()
    obs, rew, done, ns = sim.step(ctrl[k], STEPS_PER_ACTION, bid, GEOFENCE)
    bad = sim.bad_state()[0]
    new = bad & ~seen
    print('env-step', k, 'done', int(np.sum(done)), 'bad', int(bad.sum()), 'new bad envs', np.nonzero(new)[0][:8], flush=True)
    for e in np.nonzero(new)[0][:4]:
        np.savez(f'gpurun_out/r2/bad_{cfg}_{k}_{e}.npz', qpos=before[1][e], qvel=before[2][e], ctrl=ctrl[k][e], step=k, env=e,
                 qpos_after=obs[e, :m.nq], qvel_after=obs[e, m.nq:])
        print('   env', e, 'qpos before', np.round(before[1][e], 4).tolist(), '\n   qvel before', np.round(before[2][e], 3).tolist(), '\n   after', np.round(obs[e], 3).tolist())
    seen |= bad
    sim.reset(mask=np.asarray(done, np.uint8), qpos0=resets[k][0], mocap=resets[k][1])
