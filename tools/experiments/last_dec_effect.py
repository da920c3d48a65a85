"""What the Newton loop's "last step" threshold (HSR_LAST_DEC) does to one substep: the bench's cfg3 batch after two env-steps, then ONE substep
from that state with the library given by HSR_LIB; qacc and the new state go to an npz for comparison between builds.
usage: HSR_LIB=... python tools/experiments/last_dec_effect.py OUT.npz [STATE.npz]"""
import sys, os
import numpy as np
sys.path.insert(0, '.')
from hsr_env_amd import load_config
from hsr_env_amd import sim as hs
from bench import sample_inputs
m = load_config('cfg3'); n = 8192
out = sys.argv[1]
sim = hs.BatchSim(m, n); sim.set_graph(False)
if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):
    S = np.load(sys.argv[2]); q, v, w, goal, ctrl = S['q'], S['v'], S['w'], S['goal'], S['ctrl']
else:
    q0, goal = sample_inputs(m, n, 0, 0)
    sim.reset(qpos0=q0, mocap=goal)
    rng = np.random.default_rng(1)
    for k in range(2):
        ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
        sim.step(ctrl, 300, m.body_id(m.block_body()), 0.05)
    ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
    sim.step(ctrl, 150, -1, 0.0)
    t, q, v = sim.get_state(); w = sim.get_warmstart()
    if len(sys.argv) > 2: np.savez(sys.argv[2], q=q, v=v, w=w, goal=goal, ctrl=ctrl)
sim.set_mocap(goal); sim.set_debug(True); sim.set_warmstart(w); sim.set_state(np.zeros(n), q, v)
obs, rew, done, ns = sim.step(ctrl, 1)
np.savez(out, qacc=sim.get_field(hs.F_QACC), niter=sim.get_field(hs.F_NITER), obs=obs, ncon=sim.get_field(hs.F_NCON))
print(out, 'mean iterations', sim.get_field(hs.F_NITER).mean())
