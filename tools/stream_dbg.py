import os,sys
sys.path.insert(0,"."); sys.path.insert(0,"tests")
from common import gpu_render
from realtimepathtracingresearchframework_amd import abi, backend, scenes
s=scenes.cornell32()
r=backend.RenderHip(); r.initialize(64,64); r.set_scene(s); r.set_frame_schedule(2)
gpu_render(s,64,64,1,1,renderer=r)
print("done", r.frame_schedule())
