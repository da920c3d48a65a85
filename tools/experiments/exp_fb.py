"""Diagnostic: per-body vs per-contact Hessian assembly (HSR_NFB=0) on the same actions: state differences and bad envs."""
import sys, os, subprocess, numpy as np
sys.path.insert(0, '.')
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    from hsr_env_amd.compiler import load_config
    from hsr_env_amd import sim as hs
    from bench import sample_inputs
    m = load_config('cfg4'); n = 8192
    q0, goal = sample_inputs(m, n, 0, 0)
    sim = hs.BatchSim(m, n); sim.reset(qpos0=q0, mocap=goal)
    rng = np.random.default_rng(0)
    out = []
    for k in range(14):
        ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
        sim.step(ctrl, 300, m.body_id('block0'), 0.05)
        st = sim.get_state()
        out.append(np.concatenate([st[1], st[2]], 1))
        print(k, 'bad', int(np.sum(sim.bad_state()[0])), 'trips mean', sim.newton_trips().mean() / 100, flush=True)
    np.save(sys.argv[2], np.array(out))
else:
    for tag, nfb in (('fb', '9'), ('nofb', '0')):
        env = dict(os.environ, HSR_NFB=nfb)
        subprocess.check_call([sys.executable, __file__, 'child', f'/tmp/fb_{tag}.npy'], env=env)
    a, b = np.load('/tmp/fb_fb.npy'), np.load('/tmp/fb_nofb.npy')
    for k in range(len(a)):
        d = np.abs(a[k] - b[k]); fin = np.isfinite(d).all(1)
        print('env-step', k, 'non-finite envs', int((~fin).sum()), 'median max-abs diff %.2e  p99 %.2e  max %.2e' % (np.median(d[fin].max(1)), np.percentile(d[fin].max(1), 99), d[fin].max()))
