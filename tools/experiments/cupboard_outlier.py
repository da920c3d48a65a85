"""Round 4 diagnosis: an env of the cupboard scene whose env-step leaves the oracle's by rad/s within ONE substep while the contact lists still agree
(tests/test_gpu_parity.py::test_env_step_300_matches_oracle[cupboard]).  Replays the env on the GPU up to that substep, then runs the
same substep in the oracle from the GPU's own state and compares stage by stage."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from oracle.oracle import OracleSim
from test_gpu_parity import random_states, contact_mismatch
np.set_printoptions(precision=4, linewidth=200, suppress=False)
m = load_config('cupboard')
n = 64
rng = np.random.default_rng(12)
q, v, ctrl = random_states(m, n, rng)
env = int(sys.argv[1]) if len(sys.argv) > 1 else 2
kstop = int(sys.argv[2]) if len(sys.argv) > 2 else 48
sim = hs.BatchSim(m, 1)
sim.set_debug(True)
sim.set_state(np.zeros(1), q[env:env + 1], np.zeros((1, m.nv)))
o = OracleSim(m)
o.qpos[:] = q[env]; o.ctrl[:] = ctrl[env]
c1 = ctrl[env:env + 1]
for k in range(kstop + 2):
    t0, q0, v0 = sim.get_state(); w0 = sim.get_warmstart()
    obs = sim.step(c1, 1)[0]
    o.step()
    dd = np.abs(obs[0] - np.concatenate([o.qpos, o.qvel]))
    if k >= kstop - 3:
        print(f"substep {k}: |dobs| max {dd.max():.3e} at {dd.argmax()}, HIP iters {int(sim.get_field(hs.F_NITER)[0])} ncon {int(sim.get_field(hs.F_NCON)[0])} nefc {int(sim.get_field(hs.F_NEFC)[0])}; oracle iters {o.solver_niter} ncon {o.ncon} nefc {o.nefc}")
    if k == kstop:
        # the same substep in the oracle FROM THE GPU's state and warm start
        o2 = OracleSim(m)
        o2.qpos[:] = q0[0]; o2.qvel[:] = v0[0]; o2.ctrl[:] = ctrl[env]; o2.qacc_warmstart[:] = w0[0]
        o2.forward()
        con = sim.get_field(hs.F_CONTACT)[0]
        print("contact mismatch vs oracle at the GPU state:", contact_mismatch(m, con, o2.contacts()))
        oc = o2.contacts()
        print("oracle contacts (pos, normal, dist, geoms):"); print(oc[:, [0, 1, 2, 3, 4, 5, 12, 13, 14]])
        print("HIP contacts:"); print(con[con[:, 6] <= 0])
        qacc = sim.get_field(hs.F_QACC)[0]
        print("qacc HIP   ", qacc); print("qacc oracle", o2.qacc); print("qacc_smooth oracle", o2.qacc_smooth)
        print("oracle (from GPU state) iters", o2.solver_niter, "nefc", o2.nefc)
        e = o2.efc()
        print("oracle efc force", e['force']); print("oracle efc pos", e['pos'])
sim.close()
