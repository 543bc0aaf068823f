"""Developer soak: 150 s of full-size fused launches (problem inputs and gate-heavy random inputs) on four streams with resident
dp_search launches in between; every result compared bit for bit with the first one / the oracle.  Measured (profiles/r3_soak.txt):
725 696 fused launches and 181 424 dp_search runs, no mismatch."""
import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
import bench
from tests import helpers as H
from cppflow_amd.robots import get_robot
DEV="cuda:0"
LM = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)
rb=get_robot("panda")
obs=H.PANDA_2CUBES
rb.set_obstacles([c for c,_ in obs],[T for _,T in obs])
rb.set_joint_limit_padding(float(np.deg2rad(1.5)),0.03)
S,W,K=1024,256,10
x0,target,_=bench.make_inputs_problem(rb,S,W,torch.device(DEV),seed=0)
xr,tr=bench.make_inputs(rb,S,W,torch.device(DEV),seed=1)
n=S*W
streams=[torch.cuda.Stream(device=DEV) for _ in range(4)]
def launch(x,t,st):
    pk=torch.empty(rb.PACKED_BYTES_PER_ROW*n,dtype=torch.uint8,device=DEV); sm=torch.empty((S,8),device=DEV); xo=torch.empty_like(x)
    with torch.cuda.stream(st): rb.lm_pose_steps(x,t,n_steps=K,x_out=xo,packed_out=pk,summary_out=sm,**LM)
    return xo,pk,sm
ref_p=launch(x0,target,streams[0]); ref_r=launch(xr,tr,streams[1]); torch.cuda.synchronize()
k,T=175,128
rng=np.random.RandomState(4); ch=H.chain("panda")
q=H.f32(np.clip(rng.uniform(ch.lo,ch.hi,size=(k,1,7))+np.cumsum(0.05*rng.randn(k,T,7),axis=1),ch.lo,ch.hi))
ext=H.f32(rng.choice([0.0,100.0,1000.0],size=(k,T),p=[0.8,0.1,0.1]))
want_idx,_=H.oracle32("panda").dp_search(q,ext)
qd=torch.tensor(q,dtype=torch.float32,device=DEV); ed=torch.tensor(ext,dtype=torch.float32,device=DEV)
t0=time.time(); bad=0; nl=0; nd=0
while time.time()-t0<150:
    outs=[]
    for i in range(32):
        outs.append((launch(x0,target,streams[i%4]), ref_p) if i%2==0 else (launch(xr,tr,streams[i%4]), ref_r))
    dps=[rb.dp_search(qd,ed,method="resident") for _ in range(8)]
    torch.cuda.synchronize()
    for (a,b) in outs:
        nl+=1
        if not (torch.equal(a[0],b[0]) and torch.equal(a[1],b[1]) and torch.equal(a[2],b[2])): bad+=1
    for p_,i_,c_ in dps:
        nd+=1
        if not np.array_equal(i_.cpu().numpy(),want_idx): bad+=1
    del outs
print("soak: %d fused launches, %d resident dp_search runs in %.0f s, mismatches: %d"%(nl,nd,time.time()-t0,bad))
