"""Round 4 debugging: solo servers on / off on a small batch, what differs per env."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from hsr_env_amd.compiler import load_config
from hsr_env_amd import sim as hs
from test_gpu_parity import random_states
np.set_printoptions(precision=5, linewidth=220)
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
m = load_config(cfg)
rng = np.random.default_rng(51)
q, v, ctrl = random_states(m, n, rng)
goal = np.tile([0.0, 0.0, 0.422], (n, 1)).astype(np.float32)
res = []
for servers in (0, n):
    sim = hs.BatchSim(m, n)
    print('set_solo ->', sim.set_solo(servers, 0.01))
    sim.set_queue(1, 10)
    sim.set_mocap(goal)
    sim.set_state(np.zeros(n), q, v)
    obs, rew, done, ns = sim.step(ctrl, 90, m.body_id(m.block_body()), 0.1)
    t, qq, vv = sim.get_state()
    print('servers', servers, 'handovers', sim.solo_handovers(), 'ns', ns, 'done', done.astype(int), 'time', t)
    res.append((obs, ns, t))
    sim.close()
for e in range(n):
    d = np.abs(res[0][0][e] - res[1][0][e])
    print('env', e, 'max |dobs|', d.max(), 'at', d.argmax(), 'ref', res[0][0][e][:4], 'solo', res[1][0][e][:4])
