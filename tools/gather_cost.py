#!/usr/bin/env python3
"""What the library's RCCL gather costs the frames it runs beside -- measured on ONE GPU: world 1 with RPTR_COMM_SELF=1 routes rank 0's own
rows through ncclSend / ncclRecv to itself (a full RCCL point-to-point kernel pair on this device) + the assembly kernel, once per frame,
on the communication stream, while the frames of the benchmark's pipelined schedule are in flight. Reports: GPU time of one gather
(rptr_hip_comm_stats), and the pipelined frame time with and without the gathers -- the slowdown is RCCL's CU footprint + the assembly
kernel's share of the machine. Sizes: the whole 1080p frame (33 MB, what rank 0 of an 8-way split receives in total) and 1/8 of it (what
one peer sends).   python tools/gather_cost.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["RPTR_COMM_SELF"] = "1"
from realtimepathtracingresearchframework_amd import abi, backend, scenes  # noqa: E402


def run(s, W, H, with_gather, fif=11, batch=4, steps=200, spp=4, transport="rccl", per_sequence=False):
    """transport "peer": a one-process group of ONE handle with RPTR_COMM_TRANSPORT=peer -- per frame rank 0 scatters its own rows into the
    frame (here: all rows, eight times what rank 0 of an 8-way split owns) and runs NO receive kernels and NO assembly pass: the upper bound
    of what the peer-write transport leaves on rank 0 (the peers' rows arrive as remote writes)"""
    r = backend.RenderHip(frames_in_flight=fif)
    r.initialize(W, H)
    r.set_scene(s)
    if with_gather and transport == "rccl":
        r.comm_init_rank(backend.RenderHip.comm_unique_id())
    elif with_gather:
        os.environ["RPTR_COMM_TRANSPORT"] = transport
        backend.RenderHip.comm_init_all([r])
        del os.environ["RPTR_COMM_TRANSPORT"]
    host = [0.0, 0]

    def gather(n=1):
        t = time.perf_counter()
        if transport == "rccl":
            r.gather(n)
        else:
            backend.RenderHip.gather_all([r], n)
        host[0] += time.perf_counter() - t
        host[1] += 1
    cfg = backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True)

    def collect(tickets):
        for j, t in enumerate(tickets):
            r.wait(t)
            if with_gather and not per_sequence:
                gather()
            elif with_gather and j == len(tickets) - 1:
                gather(len(tickets))  # ONE collective for the frames of the launch sequence (rptr_hip_gather_batch)

    def loop(k):
        q, left = [], k
        while left > 0:
            n = min(batch, left)
            q.append(r.render_batch_async(cfg, spp=spp, n_frames=n, reset_rest=True))
            left -= n
            if len(q) >= fif:
                collect(q.pop(0))
        while q:
            collect(q.pop(0))
    loop(2 * fif * batch)
    if with_gather:
        r.comm_stats()
    t0 = time.perf_counter()
    loop(steps)
    gms = r.comm_stats()[1] if with_gather else 0.0
    ms = (time.perf_counter() - t0) * 1e3 / steps
    r.close()
    if host[1]:
        print("    (host: %.1f us per gather call, %s)" % (host[0] * 1e6 / host[1], transport))
    return ms, gms


if __name__ == "__main__":
    s = scenes.grid_1m()
    for (W, H, what) in ((1920, 1080, "whole 1080p frame (33.2 MB through RCCL)"), (1920, 136, "1/8 of the rows (4.2 MB: one peer's tile)"),
                       (1920, 272, "1/4 of the rows (8.4 MB; a frame time near a peer's 0.19 ms)")):
        a, _ = run(s, W, H, False)
        b, g = run(s, W, H, True)
        print("%-52s frame %.4f ms without, %.4f ms with a gather per frame (+%.1f %%); one gather %.3f ms of GPU time on its stream" % (what, a, b, 100 * (b / a - 1), g))
        b4, g4 = run(s, W, H, True, per_sequence=True)
        print("%-52s ... RCCL, ONE gather per launch sequence of four frames: %.4f ms (+%.1f %%); %.3f ms of GPU time per gather" % ("", b4, 100 * (b4 / a - 1), g4))
        c4, g5 = run(s, W, H, True, transport="peer", per_sequence=True)
        print("%-52s ... peer writes, ONE gather per launch sequence: %.4f ms (+%.1f %%); %.3f ms of GPU time per gather" % ("", c4, 100 * (c4 / a - 1), g5))
        c, g2 = run(s, W, H, True, transport="peer")
        print("%-52s ... with the peer-write transport's work on rank 0 (own rows scattered, no receive, no assembly): %.4f ms (+%.1f %%); %.3f ms of GPU time" % ("", c, 100 * (c / a - 1), g2))
