"""Developer check: device memory after 40 create / use / destroy cycles of a robot handle (generic and run-time-specialised
in turn) -- the free-memory delta must stay where the first cycle left it (measured: 10.0 MB from the first cycle to the last)."""
import sys, gc; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from tests import helpers as H
from cppflow_amd.robots import Robot, get_robot
DEV="cuda:0"
torch.cuda.init()
x = torch.zeros((1024,7), device=DEV)
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
spec = H.random_chain_spec(7, seed=5)
f0 = free()
for i in range(40):
    rb = Robot(spec, specialize=(i % 2 == 0))
    rb.forward_kinematics(x)
    rb.set_obstacles([H.cuboid_obstacle(0.1,0.1,0.5,0.3,0.3,0.3)[0]],[H.cuboid_obstacle(0.1,0.1,0.5,0.3,0.3,0.3)[1]])
    rb.collision_masks(x.reshape(4,256,7))
    del rb; gc.collect()
    if i in (0, 1, 9, 19, 39): print(i, "free MB delta", (f0 - free())/2**20)
