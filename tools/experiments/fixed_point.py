"""Round 5: do envs reach a BITWISE fixed point (or a short cycle) of the fp32 substep map inside an env-step?  256 envs of the bench's batch, third env-step, one substep per call."""
import sys
import numpy as np
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs, GEOFENCE
m = load_config(sys.argv[1] if len(sys.argv) > 1 else 'cfg3'); n = 256
q0, goal = sample_inputs(m, n, 0, 0)
sim = hs.BatchSim(m, n); sim.reset(qpos0=q0, mocap=goal)
rng = np.random.default_rng(1)
lo, hi = m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1]
for k in range(2):
    sim.step(rng.uniform(lo, hi, (n, m.nu)).astype(np.float32), 300)
ctrl = rng.uniform(lo, hi, (n, m.nu)).astype(np.float32)
hist = []
first_fix = np.full(n, -1); first_cyc = np.full(n, -1)
for k in range(300):
    sim.step(ctrl, 1)
    t, q, v = sim.get_state(); w = sim.get_warmstart()
    st = np.concatenate([q, v, w], axis=1).view(np.uint32)
    hist.append(st.copy())
    if k >= 1:
        same1 = (hist[-1] == hist[-2]).all(axis=1)
        first_fix[(first_fix < 0) & same1] = k
    if k >= 8:
        cyc = np.zeros(n, bool)
        for p in range(2, 9):
            cyc |= (hist[-1] == hist[-1 - p]).all(axis=1)
        first_cyc[(first_cyc < 0) & cyc] = k
print('envs that reach a bitwise fixed point within 300 substeps: %d of %d; first at substep (percentiles 10/50/90 of those): %s' % ((first_fix >= 0).sum(), n, np.percentile(first_fix[first_fix >= 0], [10, 50, 90]) if (first_fix >= 0).any() else '-'))
print('envs that revisit a state of 2..8 substeps ago: %d; first at %s' % ((first_cyc >= 0).sum(), np.percentile(first_cyc[first_cyc >= 0], [10, 50, 90]) if (first_cyc >= 0).any() else '-'))
vmax = np.abs(v).max(axis=1)
print('|qvel| max at the end: percentiles 10/50/90 %s' % np.percentile(vmax, [10, 50, 90]))
