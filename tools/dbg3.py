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
E=3471
sim=hs.BatchSim(m,n); sim.reset(qpos0=q,mocap=goal)
for k in range(45):
    t,qq,vv=sim.get_state(); w=sim.get_warmstart()
    obs,rew,done,ns=sim.step(ctrl,1)
    o=OracleSim(m); o.qpos[:]=qq[E]; o.qvel[:]=vv[E]; o.ctrl[:]=ctrl[E]; o.qacc_warmstart[:]=w[E]; o.step()
    ref=np.concatenate([o.qpos,o.qvel]); err=np.abs(obs[E]-ref).max()
    print(k,'err %.3e'%err,'ncon',o.ncon,'nefc',o.nefc,'niter',o.solver_niter,'vmax %.3e'%np.abs(obs[E,14:]).max())
    if err>1e-3:
        print('state q',repr(qq[E]));print('v',repr(vv[E]));print('w',repr(w[E]));print('ctrl',repr(ctrl[E]))
        print('gpu',obs[E]);print('ora',ref)
        c=o.contacts(); print(c[:,[0,1,2,3,4,5,12,13,14,15,16]])
        np.savez('gpurun_out/badstate.npz',q=qq[E],v=vv[E],w=w[E],ctrl=ctrl[E])
        # forward introspection
        sim2=hs.BatchSim(m,4); sim2.set_warmstart(np.tile(w[E],(4,1))); sim2.set_state(np.zeros(4),np.tile(qq[E],(4,1)),np.tile(vv[E],(4,1))); sim2.step(np.tile(ctrl[E],(4,1)),0); sim2.forward()
        print('gpu qacc',sim2.get_field(hs.F_QACC)[0]); o2=OracleSim(m); o2.qpos[:]=qq[E]; o2.qvel[:]=vv[E]; o2.ctrl[:]=ctrl[E]; o2.qacc_warmstart[:]=w[E]; o2.forward(); print('ora qacc',o2.qacc)
        print('gpu ncon nefc niter',sim2.get_field(hs.F_NCON)[0],sim2.get_field(hs.F_NEFC)[0],sim2.get_field(hs.F_NITER)[0])
        print('gpu qas',sim2.get_field(hs.F_QACC_SMOOTH)[0]); print('ora qas',o2.qacc_smooth)
        print('gpu qfc',sim2.get_field(hs.F_QFRC_CONSTRAINT)[0]); print('ora qfc',o2.qfrc_constraint)
        break
