"""Round 5: what do the (easiest possible) neighbours in its wave cost a hard env?  The bench's batch runs two env-steps; the 64 envs with the most Newton
iterations are timed over the third env-step (a) with the 192 EASIEST envs of the batch as neighbours, packed by k_schedule as in the bench (one hard env per
wave), and (b) alone in their waves: the three neighbours of every hard env are copies whose goal is met at the first substep, so they leave at once."""
import sys
import numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs, GEOFENCE
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
nh = 64
m = load_config(cfg); n = 8192
q0, goal = sample_inputs(m, n, 0, 0)
sim = hs.BatchSim(m, n); sim.reset(qpos0=q0, mocap=goal)
rng = np.random.default_rng(1)
bid = m.body_id(m.block_body())
a = m.free_joint_qadrs()[0]
ctrls = [rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32) for _ in range(3)]
for k in range(2):
    sim.step(ctrls[k], 300, bid, GEOFENCE)
trips = sim.newton_trips()
t, q, v = sim.get_state(); w = sim.get_warmstart()
sim.set_profiling(2); sim.step(ctrls[2], 300, bid, GEOFENCE); print('full batch, third env-step: %.2f ms' % sim.kernel_times().sum())
trips3 = sim.newton_trips()
sim.close()
order = np.argsort(-trips3)          # hardness DURING the third env-step (what an oracle scheduler would know)
hard, easy = order[:nh], order[-3 * nh:]
print('Newton iterations over the last 100 substeps: hard envs max %d min %d, easy neighbours max %d' % (trips3[hard].max(), trips3[hard].min(), trips3[easy].max()))
def run(idx, gl, sched):
    s2 = hs.BatchSim(m, len(idx)); s2.set_queue(0); s2.set_schedule(sched)
    s2.set_mocap(gl); s2.set_warmstart(w[idx]); s2.set_state(t[idx], q[idx], v[idx])
    s2.set_profiling(2)
    obs, rew, done, ns = s2.step(ctrls[2][idx], 300, bid, GEOFENCE)
    ms = s2.kernel_times().sum(); s2.close()
    return ms, ns
# (a) hard + easiest neighbours, interleaved so that the identity packing puts one hard env into every wave
idx = np.empty(4 * nh, dtype=int); idx[0::4] = hard; idx[1::4] = easy[0::3]; idx[2::4] = easy[1::3]; idx[3::4] = easy[2::3]
ms_a, ns_a = run(idx, goal[idx], False)
# (b) the neighbours leave at the first substep
idx_b = np.repeat(hard, 4)
gl = goal[idx_b].copy()
mask = np.ones(4 * nh, bool); mask[0::4] = False
gl[mask] = q[idx_b][mask][:, a:a + 3]
ms_b, ns_b = run(idx_b, gl, False)
# (c) four hard envs per wave
ms_c, ns_c = run(hard, goal[hard], False)
print('hard env + three easiest neighbours per wave: %.2f ms; alone in its wave: %.2f ms (neighbours ran %.1f substeps); four hard envs per wave: %.2f ms' % (ms_a, ms_b, ns_b[mask].mean(), ms_c))
