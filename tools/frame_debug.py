"""bisects a failing one-launch frame: each configuration in its own process (a GPU fault kills the process)"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SNIPPET = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
from common import gpu_render
from realtimepathtracingresearchframework_amd import abi, backend, scenes
s = getattr(scenes, os.environ.get("DBG_SCENE", "cornell32"))()
W, H, spp = [int(v) for v in os.environ.get("DBG_SIZE", "64,64,1").split(",")]
variant = int(os.environ.get("DBG_VARIANT", "1"))
imgs = []
for one in (False, True):
    r = backend.RenderHip(); r.initialize(W, H); r.set_scene(s); r.set_frame_schedule(int(os.environ.get('DBG_MODE', '1')) if one else 0); r.set_stage_timing(int(os.environ.get('DBG_TIMING', '2')))
    fr = []
    for k in range(int(os.environ.get("DBG_FRAMES", "2"))):
        img, st, _ = gpu_render(s, W, H, spp, variant, reset=(k == 0), renderer=r)
        fr.append((img.copy(), int(st.raw.rays_closest), int(st.raw.rays_shadow)))
    print("  one_launch", one, "rays", fr[-1][1:], "sched", r.frame_schedule(), flush=True)
    imgs.append(fr); r.close()
bad = 0
for k, (a, b) in enumerate(zip(*imgs)):
    d = (a[0].view(np.uint32) != b[0].view(np.uint32)).any(axis=2)
    if d.sum() or a[1:] != b[1:] or k < 2:
        print("  frame", k, "differing pixels", int(d.sum()), "of", d.size, "rays equal", a[1:] == b[1:], flush=True)
    bad += int(d.sum() > 0)
print("  frames", len(imgs[0]), "with differences", bad, flush=True)
''' % (ROOT, ROOT)
cases = [
    {"DBG_MODE": "2", "DBG_SIZE": "256,256,2", "DBG_FRAMES": "60", "DBG_TIMING": "0"},
    {"DBG_MODE": "2", "DBG_SIZE": "256,256,2", "DBG_FRAMES": "60", "DBG_TIMING": "2"},
    {"DBG_MODE": "2", "DBG_SIZE": "1920,1080,4", "DBG_SCENE": "grid_1m", "DBG_VARIANT": "1", "DBG_FRAMES": "30", "DBG_TIMING": "0"},
]
for env in cases:
    e = dict(os.environ); e.update(env)
    print("case", env, flush=True)
    p = subprocess.run([sys.executable, "-c", SNIPPET], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=240)
    out = [l for l in p.stdout.splitlines() if l.startswith("  ") or "fault" in l.lower() or "error" in l.lower() or l.startswith("[stream]")]
    print("\n".join(out[-8:]), "\n  rc", p.returncode, flush=True)
