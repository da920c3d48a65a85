import numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
np.set_printoptions(precision=5, suppress=False, linewidth=220)
from hsr_env_amd.compiler import *
from hsr_env_amd import sim as hs
from oracle.oracle import OracleSim
from bench import sample_inputs
m=load_config('cfg3'); n=8192
q0,goal=sample_inputs(m,n,0,0)
rng=np.random.Generator(np.random.Philox(key=[1,0]))
lo,hi=m.act_ctrlrange[:,0].astype(np.float32),m.act_ctrlrange[:,1].astype(np.float32)
sim=hs.BatchSim(m,n); sim.reset(qpos0=q0,mocap=goal)
bid=m.body_id('block0')
found=False
for k in range(8):
    ctrl=rng.uniform(lo,hi,(n,m.nu)).astype(np.float32)
    for sub in range(0,300,10):
        t,qq,vv=sim.get_state(); w=sim.get_warmstart()
        obs,rew,done,ns=sim.step(ctrl,10,-1,0.0)
        bad,anyb=sim.bad_state()
        if anyb:
            e=np.where(bad)[0][0]; print('env-step',k,'sub',sub,'bad envs',np.where(bad)[0])
            # replay 10 substeps one by one from saved state for that env, vs oracle
            sim2=hs.BatchSim(m,4); sim2.set_warmstart(np.tile(w[e],(4,1))); sim2.set_state(np.zeros(4),np.tile(qq[e],(4,1)),np.tile(vv[e],(4,1)))
            o=OracleSim(m); o.qpos[:]=qq[e]; o.qvel[:]=vv[e]; o.ctrl[:]=ctrl[e]; o.qacc_warmstart[:]=w[e]
            for j in range(10):
                ob=sim2.step(np.tile(ctrl[e],(4,1)),1)[0][0]; o.step()
                ref=np.concatenate([o.qpos,o.qvel])
                print(j,'err %.3e'%np.abs(ob-ref).max(),'ncon',o.ncon,'nefc',o.nefc,'niter',o.solver_niter,'gpu niter',sim2.get_field(hs.F_NITER)[0],'bad',sim2.bad_state()[0][0], 'vmax %.3e'%np.abs(ref[14:]).max())
                if sim2.bad_state()[0][0]:
                    print('q',repr(ob[:14])); print('v',repr(ob[14:])); print('ora q',o.qpos,'v',o.qvel); print(o.contacts()[:,[12,13,14,15]]); break
            found=True; break
    if found: break
print('done', found)
