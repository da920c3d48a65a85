import numpy as np, sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
np.set_printoptions(precision=6, suppress=False, linewidth=220)
from hsr_env_amd.compiler import *
from hsr_env_amd import sim as hs
from oracle.oracle import OracleSim
from test_gpu_parity import random_states
m=load_config('cfg3'); n=8192
rng=np.random.default_rng(16)
q,v,ctrl=random_states(m,n,rng)
goal=np.column_stack([rng.uniform(-0.1,0.1,n),rng.uniform(-0.2,0.2,n),np.full(n,0.422)])
sim=hs.BatchSim(m,n); sim.reset(qpos0=q,mocap=goal)
prev=None
for k in range(300):
    t,qq,vv=sim.get_state(); w=sim.get_warmstart()
    obs,rew,done,ns=sim.step(ctrl,1)
    badenv=np.where(~np.isfinite(obs).all(1))[0]
    if len(badenv):
        e=badenv[0]; print('substep',k,'bad envs',badenv[:10], 'count',len(badenv))
        print('prev q',qq[e]); print('prev v',vv[e]); print('warm',w[e]); print('ctrl',ctrl[e])
        o=OracleSim(m); o.qpos[:]=qq[e]; o.qvel[:]=vv[e]; o.ctrl[:]=ctrl[e]; o.qacc_warmstart[:]=w[e]; o.step()
        print('oracle ncon',o.ncon,'nefc',o.nefc,'niter',o.solver_niter,'q',o.qpos,'v',o.qvel)
        print(o.contacts()[:,[12,13,14,15]])
        np.savez('gpurun_out/badstate.npz',q=qq[e],v=vv[e],w=w[e],ctrl=ctrl[e])
        break
else:
    print('no bad env')
