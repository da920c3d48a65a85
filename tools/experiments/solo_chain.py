"""Round 4: how much shorter is a hard env's chain on a solo server?  The bench's batch runs two env-steps; the 512 envs with the most Newton
iterations are copied into a batch of their own (state, warm start, goal) and the third env-step is timed three ways on that batch:
plain (one launch, four envs per wave), work queue only, and every env handed over to a solo server after its first round."""
import sys
import numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs, GEOFENCE
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
nh = int(sys.argv[2]) if len(sys.argv) > 2 else 512
m = load_config(cfg); n = 8192
q0, goal = sample_inputs(m, n, 0, 0)
sim = hs.BatchSim(m, n); sim.reset(qpos0=q0, mocap=goal)
rng = np.random.default_rng(1)
bid = m.body_id(m.block_body())
ctrls = [rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32) for _ in range(3)]
for k in range(2):
    sim.step(ctrls[k], 300, bid, GEOFENCE)
trips = sim.newton_trips()
t, q, v = sim.get_state(); w = sim.get_warmstart()
hard = np.argsort(-trips)[:nh]
print('trips of the chosen envs over the last 100 substeps: max %d median %d min %d; batch median %d' % (trips[hard].max(), np.median(trips[hard]), trips[hard].min(), np.median(trips)))
sim.close()
for name, setup in (('plain', lambda s: s.set_queue(0)), ('queue only', lambda s: s.set_queue(1, 20)), ('solo, all envs', lambda s: (s.set_queue(1, 20), s.set_solo(nh, 0.01))),
                    ('solo, trips >= 3.5', lambda s: (s.set_queue(1, 20), s.set_solo(nh, 3.5)))):
    s2 = hs.BatchSim(m, nh)
    setup(s2)
    s2.set_schedule(True)
    s2.set_mocap(goal[hard]); s2.set_warmstart(w[hard]); s2.set_state(t[hard], q[hard], v[hard])
    s2.set_profiling(2)
    obs, rew, done, ns = s2.step(ctrls[2][hard], 300, bid, GEOFENCE)
    ms = s2.kernel_times()
    print(f'{name:20s} launch {ms.sum():7.2f} ms   handovers {s2.solo_handovers():4d}   mean substeps {ns.mean():.1f}  done {int(np.asarray(done).sum())}')
    s2.close()
