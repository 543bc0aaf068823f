import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H
from tests.test_gpu_api import _full_params
from cppflow_amd.robots import get_robot
name='panda'; rb=get_robot(name)
S,T=3,24
rng=np.random.RandomState(7); ch=H.chain(name)
obs=H.PANDA_2CUBES
rb.set_obstacles([c for c,_ in obs],[T_ for _,T_ in obs])
lo,hi=H.box_corners([c for c,_ in obs],[T_ for _,T_ in obs])
cand=H.random_configs(name,4000,seed=11)
m=H.oracle64(name).masks(cand,lo,hi,None,None)
hit=cand[np.flatnonzero((m["self_mask"]|m["env_mask"])>0)[:S]]
x=H.f32(np.clip(hit[:,None,:]+np.cumsum(0.02*rng.randn(S,T,rb.ndof),axis=1),ch.lo,ch.hi).reshape(S*T,rb.ndof))
target=H.f32(H.oracle64(name).fk(x[:T])+np.concatenate([0.002*rng.randn(T,3),np.zeros((T,4))],axis=1))
dev=lambda a: torch.tensor(np.asarray(a),dtype=torch.float32,device='cuda:0')
base=dict(use_pose=True, alpha_position=1.1, alpha_rotation=1.0, alpha_self_collision=0.05, alpha_env_collision=0.03, alpha_differencing=0.01, alpha_differencing_prismatic_scaling=2.0)
variants={
 'all':base,
 'no_coll':{**base,'use_self_collisions':False,'use_env_collisions':False},
 'no_diff':{**base,'use_differencing':False,'use_virtual_configs':False},
 'no_vq':{**base,'use_virtual_configs':False},
 'no_pose':{**base,'use_pose':False},
 'pose_only':{**base,'use_differencing':False,'use_virtual_configs':False,'use_self_collisions':False,'use_env_collisions':False},
}
for useXv in (False,True):
  for k,v in variants.items():
    pm=_full_params(**v)
    xv=H.f32(x+0.01*rng.randn(*x.shape)) if useXv else None
    pm.virtual_configs=dev(xv) if xv is not None else torch.tensor([])
    got=rb.lm_full_step(dev(x),dev(target),pm,virtual_configs=pm.virtual_configs).cpu().numpy().astype(np.float64)
    want=H.oracle64(name).lm_full_step(x,target,pm,S,T,virtual_configs=xv,boxes_lo=lo,boxes_hi=hi)
    want32=H.oracle32(name).lm_full_step(x,target,pm,S,T,virtual_configs=xv,boxes_lo=lo,boxes_hi=hi)
    Js=H.oracle64(name).lm_step(x,H.stacked(target,S),lm_lambda=1e-6,alpha_position=1.1,alpha_rotation=1.0)[1]
    task=np.abs(np.einsum('nij,nj->ni',Js,got-want)).max(axis=1)
    task32=np.abs(np.einsum('nij,nj->ni',Js,want32-want)).max(axis=1)
    print(f'xv={useXv} {k:10s} joint max {np.abs(got-want).max():.3e} task max {task.max():.3e} (per traj {[round(float(task[i*T:(i+1)*T].max()),5) for i in range(S)]}) | oracle32-vs-64: joint {np.abs(want32-want).max():.3e} task {task32.max():.3e} step {np.abs(want-x).max():.3f}')
