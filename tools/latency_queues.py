"""tools/latency_queues.py: the frame time of a handle with ONE frame context (+ its side stream) and of one with TWO, against the order in
which the process made its handles -- a HIP stream keeps the hardware queue it was dealt at creation (profiles/r05_notes.md section 21)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from realtimepathtracingresearchframework_amd import abi, backend, scenes
s = scenes.grid_1m()
cfg = backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True)
own = torch.cuda.Stream() if os.environ.get("OWN_STREAM") else None  # (bench.py hands the backend a stream of its own)
stream = own.cuda_stream if own else torch.cuda.current_stream().cuda_stream
def make(depth):
    h = backend.RenderHip(frames_in_flight=depth, stream=stream)
    h.initialize(1920, 1080); h.set_scene(s); h.set_stage_timing(0)
    return h
def measure(h, depth):
    q = []
    def pump(seconds):
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < seconds:
            q.append(h.render_async(cfg, spp=4))
            if len(q) >= depth: h.wait(q.pop(0)); n += 1
        return (time.perf_counter() - t0) * 1e3 / max(n, 1)
    pump(0.2); v = pump(0.4)
    while q: h.wait(q.pop(0))
    return v
order = sys.argv[1] if len(sys.argv) > 1 else "12"
hs = {}
for ch in order:  # '1' / '2': make the handle with that many contexts; 'B<n>': a big one
    if ch in "12": hs[int(ch)] = make(int(ch))
    else: hs["big"] = make(11)
print("handles made in the order %s: one context %s ms per frame, two contexts %s" % (order, "%.3f" % measure(hs[1], 1) if 1 in hs else "-", "%.3f" % measure(hs[2], 2) if 2 in hs else "-"), flush=True)
