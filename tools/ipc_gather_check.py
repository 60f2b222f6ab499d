"""The IPC peer-write gather (csrc/host_comm.h COMM_IPC) with ONE PROCESS PER RANK, on whatever devices there are (ranks share cuda:0 when
the box has fewer GPUs than ranks): every rank renders its stripes of a few frames -- single frames and launch sequences of four --,
gathers them through rptr_hip_gather / rptr_hip_gather_batch, and rank 0 compares every assembled frame, bit for bit, with the same frames
rendered by ONE rank. Prints "IPC_GATHER_OK <frames>" on rank 0.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 tools/ipc_gather_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from realtimepathtracingresearchframework_amd import abi, backend, scenes  # noqa: E402
from realtimepathtracingresearchframework_amd.distributed import NativeGather  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ndev = torch.cuda.device_count()
dev = int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1)
dist.init_process_group("gloo", rank=rank, world_size=world)
s = scenes.grid(120, 60, with_emitters=True)
W, H, spp = 200, 120, 2
cam = s.camera_params()


def cfg():
    return backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=True)


want = []
if rank == 0:
    one = backend.RenderHip(device_ordinal=dev)
    one.initialize(W, H)
    one.set_scene(s)
    for k in range(10):
        one.render(cfg(), spp=spp)
        img = np.zeros((H, W, 4), np.float32)
        one.readback_framebuffer(img)
        want.append(img)
    one.close()
r = backend.RenderHip(device_ordinal=dev, rank=rank, world_size=world, stripe_rows=8, frames_in_flight=2)
r.initialize(W, H)
r.set_scene(s)
g = NativeGather(r, rank, world, transport="ipc")
assert r.comm_transport() == "ipc"
got = []
buf = np.zeros((H, W, 4), np.float32)
for k in range(2):                        # two single frames, one gather each
    r.wait(r.render_async(cfg(), spp=spp))
    g.gather()
    if rank == 0:
        g.frame(buf)
        got.append(buf.copy())
for k in range(2):                        # two launch sequences of four frames, ONE gather each
    t = r.render_batch_async(cfg(), spp=spp, n_frames=4, reset_rest=True)
    for x in t:
        r.wait(x)
    g.gather(4)
    if rank == 0:
        for j in range(4):
            g.frame(buf, j)
            got.append(buf.copy())
dist.barrier()
if rank == 0:
    assert len(got) == 10
    bad = [k for k, (a, b) in enumerate(zip(got, want)) if not np.array_equal(a.view(np.uint32), b.view(np.uint32))]
    assert not bad, "frames %s differ from the one-rank frames" % bad
    print("IPC_GATHER_OK %d frames, gathers %d" % (len(got), g.stats()[0]), flush=True)
dist.barrier()
r.close()
dist.destroy_process_group()
