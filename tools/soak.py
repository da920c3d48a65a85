"""Long run of the bench workload for the cap statistics: python tools/soak.py CFG ENV_STEPS  (GPU box)."""
import sys, numpy as np, torch, time
sys.path.insert(0, '.')
from hsr_env_amd.compiler import load_config
from hsr_env_amd.sim import BatchSim
from bench import sample_inputs, GEOFENCE, STEPS_PER_ACTION
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'; K = int(sys.argv[2]) if len(sys.argv) > 2 else 300
m = load_config(cfg); n = 8192
q0, goal = sample_inputs(m, n, 0, 0)
sim = BatchSim(m, n); sim.reset(qpos0=q0, mocap=goal); sim.cap_counts(); sim.cap_histogram()
dev = torch.device('cuda', 0)
rng = np.random.Generator(np.random.Philox(key=[1, 0]))
lo, hi = m.act_ctrlrange[:, 0].astype(np.float32), m.act_ctrlrange[:, 1].astype(np.float32)
bid = m.body_id(m.block_body()) if m.block_body() else -1
d_done = torch.empty(n, dtype=torch.uint8, device=dev); d_ns = torch.empty(n, dtype=torch.int32, device=dev)
t0 = time.time(); dones = 0
for k in range(K):
    ctrl = torch.from_numpy(rng.uniform(lo, hi, (n, m.nu)).astype(np.float32)).to(dev)
    rq, rg = sample_inputs(m, n, 2 + k, 0)
    d_rq, d_rg = torch.from_numpy(rq).to(dev), torch.from_numpy(rg).to(dev)
    sim.step_dev(ctrl.data_ptr(), STEPS_PER_ACTION, bid, GEOFENCE, None, None, d_done.data_ptr(), d_ns.data_ptr())
    sim.reset_dev(None, d_rq.data_ptr(), d_rg.data_ptr())
    sim.sync(); dones += int(d_done.sum())
    if k % 50 == 49: print('env-step', k + 1, 'elapsed %.0f s' % (time.time() - t0), flush=True)
c = sim.cap_counts(); h = sim.cap_histogram(); bad = int(sim.bad_state()[0].sum())
print(cfg, 'njmax', int(m.arrays['sizes'][11]), 'env-steps', K, 'env-substeps', c[3], 'contacts beyond nconmax %.3g' % (c[0] / c[3]), 'rows beyond njmax %.3g' % (c[1] / c[3]),
      'items %.3g' % (c[2] / c[3]), 'bad envs', bad, 'dones', dones)
print('row-cap events by rows wanted beyond njmax (bins of 8):', h)
