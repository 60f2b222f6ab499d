#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mrays/s (+ ms/frame) of the wavefront path
tracer on BASELINE.json configs[1]: procedural 1M-triangle mesh, 1920x1080,
4 spp, diffuse-only BSDF, sun + sky.

One "step" = one frame = 4 samples per pixel through the whole hot path
({extend, [sort], shade, connect} x bounces -> tail -> resolve), with the scene
resident in HBM. For --gpus N the frame is sharded by screen stripes (stripe s ->
rank s % N, no data-path collective while rendering) and the tile radiance is
gathered to rank 0 over RCCL at the end of every step (inside the timed region;
the library's own grouped ncclSend/ncclRecv on a communication stream, csrc/host_comm.h).
Strong scaling: the frame is fixed, N GPUs share it.

`python bench.py --gpus N` without a torch.distributed.run environment starts the
N ranks itself (re-executes under `python -m torch.distributed.run`).

Rank 0 prints ONE JSON line (contract in the task description) carrying
`roofline` (dominant kernel = closest-hit traversal `rp_k_extend`; DESIGN.md section 6
says what each figure is) and, at N=1, `cpu_baseline` (the CPU oracle timed on the
same frame on the host cores).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# One hardware queue per HIP stream: every frame context of the backend owns a stream, and two streams that share a hardware
# queue serialise (the HIP runtime maps streams onto GPU_MAX_HW_QUEUES = 4 queues by default; measured: 7 contexts on 4 queues
# 0.34 ms per 1/8 frame, on 8 queues 0.29 ms; 11 contexts on 16 queues 0.25 ms). Read by the runtime when it initialises, so it is set before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # up to 11 frame contexts + the caller's stream + the communication stream + RCCL's

# record sizes of the algorithmic-bytes model (DESIGN.md "Roofline model")
RAY_BYTES = 32      # ray_o + ray_d (2 x float4) read per query
HIT_BYTES = 8       # hit_ids (int2: instance, triangle) written per closest query, read per shaded vertex
HIT_TUV_BYTES = 16  # ... + hit_tuv (float4) for a query that hits (round 5: a miss stores and loads no t / u / v)
FIRST_RAY_BYTES = 16  # round 5: the first extend stores the camera ray's direction + generator state (ray_d), the first shade loads it
QUEUE_BYTES = 4     # path id read from the ray queue (not for the first bounce: its queue is computed)
NODE_BYTES = 64     # RptrBvh4Node (an instance record, 128 B, counts as two)
TRI_BYTES = 48      # RptrBvhTri
SHADOW_RESULT_BYTES = 16 + 16 + 16  # contribution read + illum read-modify-write for a visible shadow ray
PATH_READ_BYTES, PATH_WRITE_BYTES = 64, 72    # shade: path state in (ray_o, ray_d, thr, illum; not on the first bounce) / out per vertex (the same + two queue words; rounds 1-3: 72 / 80 with the separate generator / path-length array)
VERTEX_BYTES, MATERIAL_BYTES = 64, 80         # the triangle's 64-byte shading record (csrc/dshade.h RpShadeTri; rounds 1-4: 48 = 3 x (qpos + qnrm_uv) from two streams), RptrBaseMaterial
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md "HBM")
MAX_CLOCK_GHZ = 2.4     # MI355X_MICROARCH.md: max clock 2400 MHz
L2_PEAK_GBS = 34500.0   # aggregate L2 bandwidth (MI355X_MICROARCH.md "L2 (per XCD)"): the ceiling of bytes served by the cache hierarchy


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--grid", type=str, default="1000x500", help="quads of the height field (2 triangles each)")
    ap.add_argument("--variant", type=str, default="diffuse", choices=["diffuse", "gltf"])
    ap.add_argument("--lights", action="store_true", help="configs[2]: add the 512 emissive triangles")
    ap.add_argument("--scene", type=str, default="grid", choices=["grid", "forest"],
                    help="forest = SURVEY 8d C4: 10 tree meshes x 10k triangles, 1000 instances (10M instanced triangles)")
    ap.add_argument("--flatten", type=int, default=-1,
                    help="library option \"flatten\" (0: keep instances two-level). Default: not set -- the library flattens a static "
                         "multi-instance scene itself (the forest's 10 M instanced triangles = 0.6 GB; the grid + its emitter mesh of --lights) "
                         "and keeps a scene with a dynamic mesh (--animate) two-level")
    ap.add_argument("--animate", action="store_true",
                    help="SURVEY 8d C5: the grid is a dynamic mesh; every step animates its vertices on the device, refits the BVH "
                         "(inside the timed region) and renders")
    ap.add_argument("--rebuild-budget", type=int, default=0,
                    help="--animate: rptr_hip_set_bvh_policy -- triangles a refit may rebuild on the device (a mesh of n triangles gets a new "
                         "LBVH tree every ceil(n / budget) frames); -1: force_bvh_rebuild (a new tree every frame); 0: refit only (BASELINE configs[4])")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames queued at once (rptr_hip_render_async): the latency-bound tail of a frame overlaps the next frame's head. "
                         "Default: 11 (7 with --animate). Measured: a full frame 1.51 / 1.44 / 1.40 ms with 3 / 7 / 11 contexts, a 1/8 frame "
                         "0.34 / 0.28 / 0.25 ms")
    ap.add_argument("--batch-frames", type=int, default=0,
                    help="frames per launch sequence (rptr_hip_render_batch_async: the samples of consecutive frames share the launches, every "
                         "frame keeps its seeds, its image and its ticket). Default: min(4, 16 // spp); 1 with --animate (every frame has its own geometry)")
    ap.add_argument("--stripe-rows", type=int, default=8,
                    help="rows per screen stripe of the tile split (multiple of 8); stripe s belongs to rank s %% N. 1080 rows in 8-row "
                         "stripes split 17/16 over 8 ranks, 32-row stripes 5/4")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="diagnostic (single GPU): render only rank 0's stripes of an N-rank tile split, no gather")
    ap.add_argument("--gather", type=str, default="native", choices=["native", "torch", "ipc"],
                    help="native: the library's RCCL gather (csrc/host_comm.h; falls back to torch when the communicator cannot be made, "
                         "noted in the JSON line); torch: tile copy + torch.distributed.gather + index_select")
    ap.add_argument("--dist-backend", type=str, default="nccl", choices=["nccl", "gloo"],
                    help="gloo: test rig for the N>1 control flow on a box with fewer GPUs than ranks (tiles are staged through the host)")
    ap.add_argument("--same-device", action="store_true", help="test rig: every rank renders on cuda:0")
    ap.add_argument("--no-probe", action="store_true",
                    help="N > 1: skip the RCCL probe (a child process per rank that tries torch's nccl group and the library's communicator + one "
                         "gather under a time-out before the real run commits to them)")
    ap.add_argument("--probe-timeout", type=float, default=150.0)
    ap.add_argument("--rccl-probe", action="store_true", help="(internal) run as the probe child of a rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fast-math", type=int, default=-1, choices=[-1, 0, 1],
                    help="library option fast_math (csrc/dmath.h): 1 = the shading stages' division / square root on the hardware's 1-ulp instructions; -1: the library's default (0, IEEE)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short legs of the other BASELINE configurations (C3, C4, C5) behind the headline")
    ap.add_argument("--latency-leg", type=int, default=0, help="(internal) run as the child that measures the latency leg with this many frames in flight")
    ap.add_argument("--dynamic-meshes", type=int, default=0,
                    help="mark the first k meshes of the scene RPTR_MESH_DYNAMIC (they are never moved): the forest with one dynamic tree mesh is the "
                         "partially flattened scene of round 5 (static instances in one world-space tree beside the dynamic mesh's instance records)")
    ap.add_argument("--no-boundary", action="store_true", help="skip the `boundary` leg (bin/rptr_hip, the C++ host over the C ABI, on the same workload)")
    ap.add_argument("--static-camera", action="store_true",
                    help="every frame through the same camera (rounds 1-3). Default: the camera MOVES every frame, as the reference's loop lets it "
                         "(app.cpp:350-469) -- launch sequences of several frames then carry a camera per frame (rptr_hip_render_batch_cameras_async)")
    ap.add_argument("--sustained-seconds", type=float, default=1.0,
                    help="after the timed region the same schedule runs on for at least this long (untimed by --steps): roofline.sustained")
    ap.add_argument("--profile-pass", action="store_true",
                    help="for rocprofv3 --pmc / --kernel-trace passes (tools/pmc.sh): warm-up + `steps` frames one at a time, nothing else, no JSON line")
    return ap.parse_args()


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a torch.distributed.run launch: start the N ranks (one process per GPU) and pass their
    output through. The children see WORLD_SIZE and take the normal path."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env, cwd=os.getcwd())


def rccl_probe_child(args):
    """One rank of the probe job (started by every rank of the real job, rendezvous on MASTER_PORT + 1): (1) torch's nccl process group +
    an all-reduce, (2) the library's own communicator from a broadcast unique id + one gather of a tiny frame. Prints a line per stage that
    worked; a hang is the parent's time-out. Nothing of the real run depends on this process."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from realtimepathtracingresearchframework_amd import abi, backend, scenes
    from realtimepathtracingresearchframework_amd.distributed import NativeGather
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    if float(t[0]) != float(world):
        raise SystemExit("all_reduce gave %r" % (t.tolist(),))
    print("PROBE torch_nccl 1", flush=True)
    s = scenes.cornell32()
    r = backend.RenderHip(device_ordinal=local_rank, rank=rank, world_size=world, stripe_rows=8)
    r.initialize(64, 64)
    r.set_scene(s)
    g = NativeGather(r, rank, world)
    for _ in range(2):
        r.wait(r.render_async(backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_GLTF, reset_accumulation=True), spp=1))
        g.gather()
    if rank == 0:
        img = np.zeros((64, 64, 4), np.float32)
        g.frame(img)
        if not (np.isfinite(img).all() and img[..., :3].max() > 0 and (img[..., 3] > 0).mean() > 0.5):
            raise SystemExit("the gathered frame is empty")
    else:
        r.comm_stats()   # waits for this rank's send
    dist.barrier()
    print("PROBE native 1", flush=True)
    # Stage 3, informational (nothing of the run depends on it): the peer-write transport for one process per GPU (--gather ipc) on THIS
    # node's devices -- the same two frames on a fresh handle, gathered through hipIpc-mapped stores into rank 0's frame, must equal the frame
    # RCCL just delivered, bit for bit. The line's gather.probe.ipc says whether remote stores over the node's fabric did that.
    try:
        r2 = backend.RenderHip(device_ordinal=local_rank, rank=rank, world_size=world, stripe_rows=8)
        r2.initialize(64, 64)
        r2.set_scene(s)
        g2 = NativeGather(r2, rank, world, transport="ipc")
        for _ in range(2):
            r2.wait(r2.render_async(backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_GLTF, reset_accumulation=True), spp=1))
            g2.gather()
        same = 0
        if rank == 0:
            img2 = np.zeros((64, 64, 4), np.float32)
            g2.frame(img2)   # (waits for every peer's counter, or for the polls' own 30 s time-out)
            same = int(np.array_equal(img.view(np.uint32), img2.view(np.uint32)))
            print("PROBE ipc %d" % same, flush=True)
        else:
            r2.comm_stats()  # this rank's stores have left
        r2.close()
        # Stage 4, informational ("make the first hardware run self-explaining"): when remote stores work, the two transports
        # of the library's gather side by side on THIS node -- a 1080p frame of the 1 M-triangle height field, 4 spp, split over the ranks,
        # launch sequences of four frames with three in flight, one gather per sequence, 48 frames each. In a child process with a time-out:
        # nothing of the benchmark line depends on it, a transport that misbehaves costs the probe, not the run.
        go = torch.tensor([same], dtype=torch.int32, device="cuda")
        dist.broadcast(go, src=0)
        if int(go[0]) == 1:
            big = scenes.grid_1m()
            for transport in ("rccl", "ipc"):
                rt = backend.RenderHip(device_ordinal=local_rank, rank=rank, world_size=world, stripe_rows=8, frames_in_flight=3)
                rt.initialize(1920, 1080)
                rt.set_scene(big)
                gt = NativeGather(rt, rank, world, transport=transport)
                cfg = backend.RenderConfiguration(big.camera_params(), active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True)

                def run(n_seq):
                    q = []
                    for k in range(n_seq + 3):
                        if k < n_seq:
                            q.append(rt.render_batch_async(cfg, spp=4, n_frames=4, reset_rest=True))
                        if len(q) >= 3 or k >= n_seq:
                            if not q:
                                break
                            for tk in q.pop(0):
                                rt.wait(tk)
                            gt.gather(4)
                run(3)
                dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(12)
                gms = gt.stats()[1]   # (waits for this rank's communication stream; raises when the flag protocol timed out)
                dist.barrier()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) * 1e3 / 48
                if rank == 0:
                    print("PROBE timing %s %.4f %.4f" % (transport, dt, gms), flush=True)
                rt.close()
    except Exception as e:  # noqa: BLE001 -- informational stage
        print("PROBE ipc 0 (%s)" % (str(e)[:160],), flush=True)
    r.close()
    dist.destroy_process_group()


def run_rccl_probe(args, timeout_s, port_offset=1):
    """starts this rank's probe child and reports which stages it got through: {"torch_nccl": 0|1, "native": 0|1, "note": str}"""
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + port_offset)
    env["MASTER_ADDR"] = "127.0.0.1" if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("localhost", "127.0.0.1") else os.environ["MASTER_ADDR"]
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS"):
        env.pop(k, None)   # the child makes its own TCP store on MASTER_PORT + 1, it does not join the agent's
    cmd = [sys.executable, os.path.abspath(__file__), "--rccl-probe"] + (["--same-device"] if args.same_device else [])
    t0 = time.time()
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    note = ""
    try:
        out, err = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, 9)   # the exact process group this call started
        except Exception:
            p.kill()
        try:
            out, err = p.communicate(timeout=15)
        except subprocess.TimeoutExpired:   # (a child stuck in the driver does not even die: leave it behind, its pipes unread)
            out, err = "", ""
        note = "probe timed out after %.0f s" % timeout_s
    got = {"torch_nccl": int("PROBE torch_nccl 1" in out), "native": int("PROBE native 1" in out),
           "ipc": (1 if "PROBE ipc 1" in out else (0 if "PROBE ipc 0" in out else None))}   # (None: the stage was not reached; rank 0's finding)
    # stage 4 (rank 0's finding): ms per 1080p C2 frame of the split, pipelined, with the gather over RCCL and over hipIpc peer writes
    for line in out.splitlines():
        if line.startswith("PROBE timing "):
            _, _, name, ms, gms = line.split()
            got.setdefault("transport_timing", {})[name] = {"ms_per_frame": float(ms), "gather_ms": float(gms)}
    if not note and p.returncode != 0:
        tail = [l for l in err.strip().splitlines() if l.strip()]
        telling = [l for l in tail if any(w in l for w in ("Error", "error", "Duplicate", "failed", "refus", "invalid"))]   # (not a profiler's last words)
        note = "probe exited with %s: %s" % (p.returncode, (telling or tail or [""])[-1].strip()[:240])
    got["note"] = note
    got["seconds"] = round(time.time() - t0, 1)
    return got


def host_cpu_budget(hw_threads):
    """Threads for the CPU baseline: the container's CPU quota (cgroup cpu.max) x2 for SMT, capped by the hardware threads.
    (The GPU box shows 256 hardware threads but grants 16 CPUs; 256 threads run 2x slower than 32 there.)"""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, min(hw_threads, 2 * int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return max(1, hw_threads)


def boundary_leg(args, scene, W, H, spp, fif, batch_frames, bench_ms):
    """bin/rptr_hip --profiling on the benchmarked workload (a child process: its own HIP context on the same GPU, which this process leaves idle
    meanwhile). Returns ms per frame (wall clock of the C++ loop, its own warm-up excluded) for the reference-shaped loop and for the deep queue."""
    import re
    import subprocess
    import tempfile
    from realtimepathtracingresearchframework_amd import build as B
    exe = os.path.join(B.BIN_DIR, "rptr_hip")
    if not os.path.exists(exe):
        return {"note": "bin/rptr_hip is not built (__graft_entry__.build() makes it)"}
    res = {"host": "bin/rptr_hip (host/rptr_cli.cpp + host/render_hip.hpp over include/rptr_hip.h)", "camera": "--fly-through: bench.py's path, a camera per frame"}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "scene.rpsc")
        scene.dump(path)
        base = [exe, path, "--profiling", os.path.join(tmp, "prof"), "--fly-through", "--img", str(W), str(H), "--batch-spp", str(spp),
                "--variant", "diffuse" if args.variant == "diffuse" else "gltf"]
        env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}  # (the library asks for its hardware queues itself)
        for key, extra, n in (("swap_buffers_2", [], 200), ("synchronous", ["--synchronous"], 100),
                              ("full_schedule", ["--frames-in-flight", str(fif), "--frames-per-launch", str(batch_frames)], max(400, 4 * fif * batch_frames))):
            try:
                p = subprocess.run(base + ["--profiling-count", str(n)] + extra, capture_output=True, text=True, env=env, timeout=300)
            except subprocess.TimeoutExpired:
                res[key] = {"note": "timed out"}
                continue
            m = re.search(r"([0-9.]+) ms per frame \(wall\), ([0-9.]+) Mrays/s", p.stdout)
            if p.returncode != 0 or not m:
                res[key] = {"note": "bin/rptr_hip failed (%d): %s" % (p.returncode, (p.stderr or p.stdout)[-300:])}
                continue
            res[key] = {"ms_per_frame": float(m.group(1)), "mrays_s": float(m.group(2)), "frames": n}
    res["what"] = ("swap_buffers_2: RenderBackend::begin_frame / draw_frame(cmd_stream) / end_frame per frame, two frames in flight (what a reference-shaped "
                   "host reaches); synchronous: cmd_stream = nullptr; full_schedule: %d launch sequences of %d frames in flight (value's schedule: %.4f ms here)"
                   % (fif, batch_frames, bench_ms))
    return res


def workload_key(args, world=1):
    """the BASELINE workload a command line names, as tools/pmc_workloads.sh keys its counter passes (None: not one of them)"""
    if world != 1 or args.emulate_world > 1 or args.grid != "1000x500" or args.rebuild_budget != 0 or args.dynamic_meshes:
        return None
    size = (args.width, args.height, args.spp)
    flat = args.flatten if args.flatten >= 0 else (0 if args.animate else 1)
    if args.scene == "forest":
        return ("c4_flat" if flat else "c4_two_level") if (size == (1920, 1080, 4) and args.variant == "diffuse" and not args.lights and not args.animate) else None
    if args.animate:
        return "c5" if (size == (3840, 2160, 2) and args.variant == "diffuse" and not args.lights) else None
    if args.lights:
        return ("c3" if flat else "c3_two_level") if (size == (1920, 1080, 8) and args.variant == "gltf") else None
    return "c2" if (size == (1920, 1080, 4) and args.variant == "diffuse") else None


def load_pmc_traffic(key):
    """profiles/pmc_traffic.json: per workload, HBM-side bytes / VALU instructions / wave-cycle split per launch of every kernel, from
    the committed rocprofv3 --pmc passes of that workload (tools/pmc_workloads.sh: tools/pmc.sh + tools/make_traffic.py). None when absent."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        return doc["workloads"].get(key) if key else None
    except Exception:
        return None


def library_build_id():
    """rptr_hip_build_id(): the hash of the sources the loaded library was built from (build.py source_id)"""
    try:
        from realtimepathtracingresearchframework_amd import backend
        L = backend.load_library()
        import ctypes
        L.rptr_hip_build_id.restype = ctypes.c_char_p
        return L.rptr_hip_build_id().decode()
    except Exception:
        return None


def load_valu_peak():
    """the VALU issue ceiling, MEASURED (tools/microbench/valu_issue.hip -> profiles/r03a_valu_issue.json, chip-wide G wave64
    instructions / s at 8 waves per SIMD): of plain full-rate instructions (v_fma_f32 / v_mul_f32 / v_add_u32: one per 2 clocks per
    SIMD, MI355X_MICROARCH.md) and of the instruction mix of the BVH4 node step (v_cvt_f32_ubyte, v_pk_fma_f32, min / max issue at half
    that rate). None when the file is absent."""
    for name in ("r05_valu_issue.json", "r04_valu_issue.json", "r03a_valu_issue.json"):
        try:
            doc = json.load(open(os.path.join(ROOT, "profiles", name)))
            best = {}
            for r in doc["results"]:
                best[r["op"]] = max(best.get(r["op"], 0.0), r["ginst_s_wall"])
            # the node step's own mix: round 4's (24 scalar fmas for the plane distances) when the file holds it, else rounds 1-3's (12 v_pk_fma_f32)
            r4 = [v for k, v in best.items() if k.startswith("node_step_mix_r4(")]
            mix = max(r4) if r4 else max(v for k, v in best.items() if k.startswith("node_step_mix("))
            full = max(best.get("v_fma_f32", 0.0), best.get("v_mul_f32", 0.0), best.get("v_add_u32", 0.0))
            # the issue ceiling of each path kernel's OWN static instruction mix (tools/kernel_mix.py -> kmix_gen.h -> the microbenchmark)
            kmix = {k[5:]: v for k, v in best.items() if k.startswith("kmix:")}
            return {"node_step_mix_ginst_s": mix, "node_step_mix_what": ("24 v_cvt_f32_ubyte, 24 v_fma_f32, 18 min / max, 14 v_cndmask, 12 integer per 92: round 4's scalar plane distances" if r4 else
                                                                       "24 v_cvt_f32_ubyte, 12 v_pk_fma_f32, 18 min / max, 14 v_cndmask, 12 integer per 80"),
                    "full_rate_ginst_s": full, "half_rate_ginst_s": best.get("v_pk_fma_f32"), "kmix": kmix,
                    "source": "profiles/%s (tools/microbench/valu_issue.hip on an MI355X of the pool)" % name}
        except Exception:
            continue
    return None


def traffic_of(pmc, prefix, field="hbm_bytes_per_launch"):
    """`field` per launch of the kernel whose (template) name starts with `prefix`, averaged over its launches"""
    if not pmc:
        return None
    hit = [v for k, v in pmc.get("kernels", {}).items() if k.replace("void ", "").startswith(prefix) and field in v]
    n = sum(v["launches"] for v in hit)
    return sum(v[field] * v["launches"] for v in hit) / n if n else None


def valu_frame(pmc, ms_per_step, valu_peak, launches):
    """all kernels of one frame (PMC instruction counts of the profile pass: every launch of every kernel / frames) against the VALU
    issue peak over the PIPELINED frame time: how much of the machine's instruction issue the steady state uses. (The bounce at which
    the tail kernel takes over may differ between the two runs; the work of a frame does not.)"""
    total = pmc.get("valu_insts_per_frame") if pmc else None
    if total is None:
        return None
    return {"valu_insts_per_step": int(total), "pipelined_ginst_s": round(total / (ms_per_step * 1e-3) / 1e9, 1),
            "pipelined_frac": round(total / (ms_per_step * 1e-3) / 1e9 / valu_peak, 4)}


SERIAL_KEYS = ("ext", "con", "shade", "tail", "resolve", "other", "gpu")


def combine_ranks(dist, elapsed, ext_ms, gather_host_ms, serial, rays, latency_1_ms):
    """The N > 1 bookkeeping of the contract, over the control-plane group (gloo: CPU tensors; tests/test_tiles_gloo.py runs it with two and three
    ranks): the timed region is the SLOWEST rank's (MAX), and so are the exclusive stage times; rays are SUMMED (value = all ranks' rays / that
    time); every rank's one-frame-at-a-time latency of its own share travels to rank 0 as a list (a frame is done when its slowest rank is).
    Returns (elapsed, ext_ms, gather_host_ms, serial, rays, [latency.1 of rank 0, 1, ...])."""
    import torch
    t = torch.tensor([elapsed, ext_ms, gather_host_ms] + [serial[k] for k in SERIAL_KEYS], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    vals = [float(v) for v in t]
    elapsed, ext_ms, gather_host_ms = vals[:3]
    serial = dict(serial)
    for k, v in zip(SERIAL_KEYS, vals[3:]):
        serial[k] = v
    tr = torch.tensor([float(rays)], dtype=torch.float64)
    dist.all_reduce(tr, op=dist.ReduceOp.SUM)
    lat = torch.zeros(dist.get_world_size(), dtype=torch.float64)
    lat[dist.get_rank()] = float(latency_1_ms or 0.0)
    dist.all_reduce(lat, op=dist.ReduceOp.SUM)
    return elapsed, ext_ms, gather_host_ms, serial, int(tr[0]), [round(float(x), 4) for x in lat]


def gather_report(gather_mode, gather_transport, probe, batch_frames, batched_gather, gather_gpu_ms, gather_host_ms, gathers_timed, bytes_per_step, note,
                  latency_1_per_rank):
    """the `gather` object of an N > 1 line: which data plane ran (and what the probe found), how many frames one collective moves, what it cost,
    and every rank's latency of its own share"""
    return {"transport": gather_transport,
            "mode": {"native": "library: grouped ncclSend/ncclRecv on a communication stream + assembly kernel (csrc/host_comm.h); --gather ipc: every rank writes its rows into rank 0's frame through hipIpc-mapped memory",
                     "torch": "tile copy + torch.distributed.gather (nccl group) + index_select",
                     "host": "tile copy + torch.distributed.gather (gloo, tiles staged through the host) + index_select"}[gather_mode],
            "probe": probe,
            "frames_per_gather": (batch_frames if (batched_gather and gather_transport is not None) else 1),
            "frames_per_gather_note": "the library's gather moves the frames of a launch sequence in ONE collective (rptr_hip_gather_batch); BENCH_GATHER_PER_FRAME=1: one per frame",
            "gather_ms": round(gather_gpu_ms, 4) if gather_gpu_ms is not None else None,
            "gather_ms_note": "mean GPU time of one gather on rank 0's communication stream (receive of N-1 tiles + assembly); asynchronous: it runs "
                              "beside the frames in flight, inside the timed region",
            "host_ms_per_step": round(gather_host_ms, 4), "gathers": gathers_timed,
            "bytes_per_step": bytes_per_step, "note": note,
            "latency_1_ms_per_rank": latency_1_per_rank,
            "latency_1_note": "one frame at a time, every rank its own share of the frame (before the gather): a frame is as late as its slowest rank -- the "
                              "figure that scales a real-time renderer, next to `value`'s throughput with frames queued ahead"}


def main():
    args = parse_args()
    if args.rccl_probe:
        return rccl_probe_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    import torch
    import torch.distributed as dist
    from realtimepathtracingresearchframework_amd import abi, backend, scenes
    from realtimepathtracingresearchframework_amd.distributed import NativeGather, TileGather

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    torch.cuda.set_device(local_rank)
    probe = None
    if world > 1:
        # Control plane (barriers, the max / sum over ranks of a few host numbers, the unique-id broadcast): a gloo group -- it cannot hang
        # on a GPU. Data plane (the path's ONE collective, the gather of tile radiance): RCCL, through the library's own communicator or,
        # failing that, torch's nccl group; which of them this node can do is found out by a probe child per rank under a time-out
        # BEFORE the run commits to it (a wedged ncclCommInitRank would otherwise eat the whole job).
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        if args.dist_backend == "nccl":
            # (a second attempt on another rendezvous port when the first fails fast everywhere: MASTER_PORT + 1 may be taken)
            for attempt, offset in enumerate((1, 101)):
                probe = ({"torch_nccl": 1, "native": 1, "note": "not probed (--no-probe)", "seconds": 0.0} if args.no_probe
                         else run_rccl_probe(args, args.probe_timeout, offset))
                flags = torch.tensor([probe["torch_nccl"], probe["native"], int(probe["seconds"] < 30.0)], dtype=torch.int32)
                dist.all_reduce(flags, op=dist.ReduceOp.MIN)   # every rank takes the same path
                mine = (probe["torch_nccl"], probe["native"])
                probe["torch_nccl"], probe["native"] = int(flags[0]), int(flags[1])
                probe["attempts"] = attempt + 1
                if mine != (probe["torch_nccl"], probe["native"]) and not probe["note"]:
                    probe["note"] = "the probe failed on another rank"
                if probe["native"] or args.no_probe or not int(flags[2]) or args.same_device:
                    break   # fine, or a slow failure (a time-out is not a port problem), or the rig where RCCL cannot work

    nx, nz = (int(v) for v in args.grid.split("x"))
    t0 = time.time()
    if args.scene == "forest":
        scene = scenes.forest()
    else:
        scene = scenes.grid(nx, nz, with_emitters=args.lights, name="grid-%dk" % (2 * nx * nz // 1000),
                            deform_t=0.0 if args.animate else None)
    for m_ in scene.meshes[:max(0, args.dynamic_meshes)]:
        m_.dynamic = True
    t_scene = time.time() - t0
    variant = abi.VARIANT_SIMPLE if args.variant == "diffuse" else abi.VARIANT_GLTF
    W, H, spp = args.width, args.height, args.spp

    # one explicit stream for torch AND the backend: the animation kernel, the tile copy and torch's gather (--gather torch) are ordered
    # with the frames by stream order. (torch's default stream has handle 0, which the C ABI reads as "create your own stream".)
    # options of the handles (rptr_hip_set_option, include/rptr_hip.h "Options"): nothing is set unless a flag asks for it -- the library's
    # defaults ARE the measured configuration (static multi-instance scenes are flattened by the library itself: option "flatten" = auto)
    lib_options = {}
    if args.flatten >= 0:
        lib_options["flatten"] = args.flatten
    if args.fast_math >= 0:
        lib_options["fast_math"] = args.fast_math
    flatten = args.flatten if args.flatten >= 0 else (0 if args.animate else 1)
    torch_stream = torch.cuda.Stream()
    torch.cuda.set_stream(torch_stream)
    stream = torch_stream.cuda_stream
    small_frames = world > 1 or args.emulate_world > 1
    # 11 frame contexts (7 for the animated scene: every context refits its own tree copy). One GPU, full frame: 3 / 7 / 11 contexts give
    # 1.51 / 1.44 / 1.40 ms per frame (profiles/r02_notes.md); the roofline figures come from frames rendered one at a time either way.
    fif = args.frames_in_flight if args.frames_in_flight > 0 else (7 if args.animate else 11)
    batch_frames = args.batch_frames if args.batch_frames > 0 else (1 if args.animate else max(1, min(4, 16 // max(spp, 1))))
    # small frames (a rank of a >= 4-way split): launch sequences of EIGHT frames (32 sample slots in flight instead of 16: option
    # "max_batch_spp", read by initialize) -- measured on rank 0's share of an 8-way / 4-way split: 0.193 -> 0.173 /
    # 0.343 -> 0.330 ms per frame (profiles/r04_notes.md section 6); a full frame gains nothing from it
    ranks_of_split = world if world > 1 else args.emulate_world
    if args.batch_frames <= 0 and not args.animate and ranks_of_split >= 4 and "RPTR_MAX_BATCH_FRAMES" not in os.environ and "RPTR_MAX_BATCH_SPP" not in os.environ:
        batch_frames = max(1, min(8, 32 // max(spp, 1)))
        lib_options["max_batch_spp"] = batch_frames * spp
    # no more contexts than the timed region has launch sequences for: the line names the schedule that ran (20 steps in sequences of 4
    # frames are 5 sequences, not 11)
    # BENCH_BATCH_PATTERN=a,b,c,... (experiment): the lengths of the timed region's first launch sequences (then `batch_frames` each)
    batch_pattern = [max(1, min(batch_frames, int(x))) for x in os.environ.get("BENCH_BATCH_PATTERN", "").split(",") if x.strip()]

    def sequence_lengths(k):
        out, i = [], 0
        while k > 0:
            n = min(batch_pattern[i] if i < len(batch_pattern) else batch_frames, k)
            out.append(n)
            k -= n
            i += 1
        return out
    fif = max(2 if batch_frames > 1 else 1, min(fif, len(sequence_lengths(args.steps))))   # (a batch of frames needs two contexts: every frame keeps its image)
    if args.profile_pass:
        fif = 1   # frames one at a time, every launch at full size (what the exclusive figures of the JSON line measure, on their own handle)
    cam = scene.camera_params()
    anim = None
    if args.animate:
        # stand-in for the reference's animation compute shader: y = y0 + 0.5 sin(0.4 x + 2 pi t), written by a torch
        # elementwise kernel on the same stream, then handed over device-to-device
        g0 = scene.geometries[0]
        base = torch.from_numpy(scenes.dequantize_positions(g0.qpos, g0.scaling, g0.offset)).cuda()
        anim = {"base": base, "cur": base.clone(), "frame": 0, "refit_ms": 0.0, "ev": []}

    def submit_on(handle, count=False):
        if anim is not None:
            t = 0.02 * anim["frame"]
            anim["frame"] += 1
            cur, b = anim["cur"], anim["base"]
            torch.add(b[:, 1], torch.sin(b[:, 0] * 0.4 + 6.283185307179586 * t), alpha=0.5, out=cur[:, 1])
            handle.update_vertices_device(0, cur.data_ptr(), cur.shape[0])
            handle.refit()
        cfg = backend.RenderConfiguration(cam, active_variant=variant, reset_accumulation=True)
        return handle.render_async(cfg, spp=spp, count_traversal=count)

    # ---- latency (ms/frame = GPU time from the first stage launch to the resolve): frames ONE at a time, and TWO in flight
    # -- what the reference's swap chain holds (RenderGraphic::MAX_SWAP_BUFFERS = 2, util/display/render_graphic.h:19). `value` above is
    # the throughput of the pipelined schedule (config.frames_in_flight x frames_per_launch_sequence); these are the figures of an
    # interactive host that cannot queue frames ahead.
    def latency_of(handle, depth, n_frames):
        handle.set_stage_timing(0)
        # warm-up IN THIS MODE for a quarter of a second: a frame at a time leaves the GPU idle between launches, and what the clocks do then
        # depends on what ran before (round 5: the same leg read 1.95 ms after a 40-step timed region and 1.75 after a 100-step one with two
        # warm-up frames) -- an interactive host renders continuously, the steady state is the figure
        t_w, n_w = time.perf_counter(), 0
        while n_w < 2 or time.perf_counter() - t_w < 0.25:
            handle.wait(submit_on(handle))
            n_w += 1
        torch.cuda.synchronize()
        t_l, q, rays_l, gpu_l = time.perf_counter(), [], 0, 0.0
        tails = {}   # stand-alone closest-hit launches of a frame (= the bounce its tail kernel took over at) -> frames
        for k in range(n_frames + depth):
            if k < n_frames:
                q.append(submit_on(handle))
            if len(q) >= depth or k >= n_frames:
                if not q:
                    break
                stl = handle.wait(q.pop(0)).raw
                rays_l += int(stl.rays_closest + stl.rays_shadow)
                gpu_l += stl.render_time_ms
                tails[int(stl.launches_extend)] = tails.get(int(stl.launches_extend), 0) + 1
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t_l) * 1e3 / n_frames
        return {"frames_in_flight": depth, "ms_per_frame": round(wall, 4), "mrays_s": round(rays_l / n_frames / wall / 1e3, 1),
                "gpu_ms_first_launch_to_resolve": round(gpu_l / n_frames, 4), "frames": n_frames,
                "tail_from_bounce_frames": {str(k): v for k, v in sorted(tails.items())}}

    # The latency legs run in a process of their own with ONE backend handle -- the state of the reference's application -- (N = 1; the
    # children are started at the end, when this process has closed its handles). In this process, next to the main handle's 5-11 frame
    # contexts and the handle of the exclusive passes, the same legs read 1.75 or 1.93 ms (one frame at a time) and 1.35 or 1.48 (two in
    # flight) depending on nothing but how many handles and streams the process had made before (tools/latency_queues.py,
    # profiles/r05_notes.md section 21: not shared hardware queues -- rocprofv3 shows the contexts on queues of their own --, but
    # reproducible). N > 1 and the emulated split keep the in-process leg for one frame at a time, taken first.
    n_lat = max(10, min(40, args.steps))
    latency = {}

    def latency_handle(depth):
        rk, wd = (0, args.emulate_world) if args.emulate_world > 1 else (rank, world)
        rl = backend.RenderHip(device_ordinal=local_rank, rank=rk, world_size=wd, stripe_rows=args.stripe_rows, stream=stream, frames_in_flight=depth, options=lib_options)
        rl.initialize(W, H)
        rl.set_scene(scene)
        if args.animate and args.rebuild_budget != 0:
            rl.set_bvh_policy(force_bvh_rebuild=args.rebuild_budget < 0, rebuild_triangle_budget=max(args.rebuild_budget, 0))
        return rl

    if args.latency_leg > 0:  # the child: this leg and nothing else
        rl = latency_handle(args.latency_leg)
        print(json.dumps({"latency_leg": latency_of(rl, args.latency_leg, n_lat)}))
        rl.close()
        return
    latency_in_children = world == 1 and args.emulate_world <= 1 and not args.profile_pass
    if not args.profile_pass and not latency_in_children:
        rl = latency_handle(1)
        latency["1"] = latency_of(rl, 1, n_lat)
        rl.close()
        if anim is not None:
            anim["frame"] = 0
    if args.emulate_world > 1:
        r = backend.RenderHip(device_ordinal=local_rank, rank=0, world_size=args.emulate_world, stripe_rows=args.stripe_rows, stream=stream, frames_in_flight=fif, options=lib_options)
    else:
        r = backend.RenderHip(device_ordinal=local_rank, rank=rank, world_size=world, stripe_rows=args.stripe_rows, stream=stream, frames_in_flight=fif, options=lib_options)
    r.initialize(W, H)
    t0 = time.time()
    r.set_scene(scene)
    t_build = time.time() - t0
    bvh_on_device, bvh_step_ms, bvh_device_ms = r.bvh_build_info()
    fast_math_in_force = int(r.get_option("fast_math"))
    bvh_area_cost, trav_node_min, trav_refill_min = r.traversal_preset()
    if args.animate and args.rebuild_budget != 0:
        r.set_bvh_policy(force_bvh_rebuild=args.rebuild_budget < 0, rebuild_triangle_budget=max(args.rebuild_budget, 0))

    def camera_of(k):
        """the view of frame k: a fly-through step per frame (yaw 0.002 rad, 2 cm sideways), so that consecutive frames differ the way an
        interactive host's do while the workload stays the one the configuration names: the 64 views lie SYMMETRICALLY around the
        configuration's own (steps -32 .. 31; rounds 4-5a walked 0 .. 63 away from it, which on the forest turned into denser trees: +27 % time,
        VERDICT r4 weak 7)"""
        if args.static_camera:
            return cam
        k %= 64
        if k in cam_cache:  # (the 64 views are made once: nothing of this is host work inside the timed region)
            return cam_cache[k]
        import numpy as np
        c = abi.Camera()
        amp = float(os.environ.get("BENCH_CAMERA_SCALE", "1"))  # diagnostics: 0 = per-frame cameras that all equal the configuration's view
        a = 0.002 * ((k % 64) - 32) * amp
        d, up = np.asarray(cam.dir[:], np.float64), np.asarray(cam.up[:], np.float64)
        right = np.cross(d, up)
        right /= np.linalg.norm(right)
        nd = np.cos(a) * d + np.sin(a) * right
        nd /= np.linalg.norm(nd)
        c.pos[:] = [float(np.float32(cam.pos[i] + 0.02 * ((k % 64) - 32) * amp * right[i])) for i in range(3)]
        c.dir[:] = [float(np.float32(x)) for x in nd]
        c.up[:] = list(cam.up[:])
        c.fovy = cam.fovy
        cam_cache[k] = c
        return c
    cam_cache = {}
    frame_no = [0]
    for k_ in range(64):
        camera_of(k_)

    # ---- the gather (N > 1): the library's own RCCL path, or torch.distributed as plumbing
    # the gather: "native" = the library's RCCL gather; "torch" = tile copy + torch.distributed.gather over an nccl group; "host" = the same over
    # gloo with the tiles staged through the host (the last resort, and the CPU test rig --dist-backend gloo)
    on_host = world > 1 and (args.dist_backend == "gloo" or not probe["torch_nccl"])
    gather_mode, gather_note, native, tgather, stage, nccl_group = None, None, None, None, None, None
    if world > 1:
        gather_mode = "native" if (args.gather == "native" and not on_host and probe["native"]) else ("host" if on_host else "torch")
        if args.gather == "ipc":   # opt-in: peer writes through hipIpc-mapped frame buffers (no RCCL involved: works with ranks that share a device)
            gather_mode = "native"
        if probe is not None and probe["note"] and gather_mode != "native":
            gather_note = "RCCL probe: %s -> %s" % (probe["note"], {"torch": "torch.distributed gather over nccl", "host": "gather over gloo, tiles staged through the host"}[gather_mode])
        if gather_mode == "native":
            ok = 1
            try:
                native = NativeGather(r, rank, world, transport="ipc" if args.gather == "ipc" else "rccl")
            except Exception as e:  # every rank must take the same path: agree on it
                ok, gather_note = 0, "native RCCL communicator failed (%s): torch.distributed gather used instead" % (str(e)[:200],)
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 0:
                gather_mode, native = ("host" if on_host else "torch"), None
                gather_note = gather_note or "native RCCL communicator failed on another rank: torch.distributed gather used instead"
                try:
                    r.comm_destroy()
                except Exception:
                    pass
        if gather_mode == "torch":
            nccl_group = dist.new_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        if gather_mode in ("torch", "host"):
            tgather = TileGather(W, H, args.stripe_rows, rank, world, device="cpu" if on_host else "cuda", group=nccl_group)
            stage = torch.zeros_like(tgather.tile, device="cuda") if on_host else None  # device tile -> host tile
    my_bytes = r.local_pixel_count() * 16
    host_gather_s = [0.0]

    def animate():
        t = 0.02 * anim["frame"]
        anim["frame"] += 1
        cur, b = anim["cur"], anim["base"]
        torch.add(b[:, 1], torch.sin(b[:, 0] * 0.4 + 6.283185307179586 * t), alpha=0.5, out=cur[:, 1])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r.update_vertices_device(0, cur.data_ptr(), cur.shape[0])
        r.refit()
        e1.record()
        anim["ev"].append((e0, e1))

    def finish(ticket, last_of=1):
        """collect one queued frame; N > 1: the path's one collective, tile radiance -> rank 0. Asynchronous on the device: it runs
        behind the collected frame, beside the frames still in flight. The library's gather moves the frames of a launch sequence in ONE
        collective (rptr_hip_gather_batch): issued with the sequence's last frame (last_of = its length; 0 = not the last, nothing to do yet);
        the torch / host fall-backs gather frame by frame."""
        st = r.wait(ticket)
        if world > 1:
            t_g = time.perf_counter()
            if native is not None:
                if last_of > 0:
                    native.gather(last_of if batched_gather else 1)
                elif not batched_gather:
                    native.gather()
            else:
                if my_bytes:
                    r.copy_tile_to_device((stage if on_host else tgather.tile).data_ptr(), my_bytes)
                if on_host:
                    tgather.tile.copy_(stage)  # synchronising device-to-host copy on the current stream
                tgather.gather()
            host_gather_s[0] += time.perf_counter() - t_g
        return st

    def step(count=False):
        """one synchronous frame (warm-up and the instrumented passes)"""
        if anim is not None:
            animate()
        cfg = backend.RenderConfiguration(cam, active_variant=variant, reset_accumulation=True)
        return finish(r.render_async(cfg, spp=spp, count_traversal=count))

    batched_gather = os.environ.get("BENCH_GATHER_PER_FRAME", "0") == "0"   # (1: one collective per frame, as rounds 2-3 did)

    def collect(tickets, on_stats):
        for j, t in enumerate(tickets):
            on_stats(finish(t, len(tickets) if j == len(tickets) - 1 else 0))

    def timed_steps(k, on_stats):
        """k frames, `batch_frames` of them per launch sequence, up to `fif` launch sequences in flight; every frame is submitted,
        rendered, collected (and gathered) inside the caller's timed region"""
        queue, left = [], k
        lengths = sequence_lengths(k)
        while left > 0:
            if anim is not None:
                animate()
            n = lengths.pop(0)
            cams = [camera_of(frame_no[0] + j) for j in range(n)]
            frame_no[0] += n
            cfg = backend.RenderConfiguration(cams[0], active_variant=variant, reset_accumulation=True)
            if n == 1:
                queue.append([r.render_async(cfg, spp=spp)])
            elif args.static_camera:
                queue.append(r.render_batch_async(cfg, spp=spp, n_frames=n, reset_rest=True))
            else:
                queue.append(r.render_batch_cameras_async(cfg, cams, spp=spp, reset_rest=True))
            left -= n
            if len(queue) >= fif:
                collect(queue.pop(0), on_stats)
        while queue:
            collect(queue.pop(0), on_stats)

    if args.profile_pass:  # what a rocprofv3 pass should see: identical frames, one at a time, no instrumented variants
        r.set_stage_timing(0)
        for _ in range(max(1, args.warmup) + args.steps):
            step()
        torch.cuda.synchronize()
        return

    r.set_stage_timing(int(os.environ.get("BENCH_STAGE_TIMING", "1")))  # timed region: HIP events around every closest-hit traversal launch (the roofline kernel) only
    # warm-up: W steps through the very path the timed region takes (launch sequences of `batch_frames` frames, `fif` of them in flight:
    # every frame context renders before the clock starts); the first frame of a handle runs without the tail kernel, so at least two
    step()
    timed_steps(max(args.warmup, 1), lambda st: None)
    if anim is not None:
        anim["ev"].clear()
    host_gather_s[0] = 0.0
    gathers_before = native.stats()[0] if native is not None else 0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_begin = time.perf_counter()
    acc = dict(ext=0.0, rays=0, launches=0)

    def on_stats(st):
        acc["launches"] = int(st.raw.launches_extend)  # stand-alone closest-hit launches per frame (the rest runs in the tail kernel)
        acc["ext"] += st.raw.extend_time_ms
        acc["rays"] += st.raw.rays_closest + st.raw.rays_shadow

    timed_steps(args.steps, on_stats)
    ext_ms, rays = acc["ext"], acc["rays"]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_begin
    refit_ms = sum(a.elapsed_time(b) for a, b in anim["ev"]) / max(len(anim["ev"]), 1) if anim is not None else None
    gather_host_ms = host_gather_s[0] * 1e3 / args.steps
    gather_gpu_ms = native.stats()[1] if native is not None else None
    gather_transport = r.comm_transport() if native is not None else None
    gathers_timed = (native.stats()[0] - gathers_before) if native is not None else args.steps

    # untimed, one frame at a time, on a SECOND handle with one frame context (a traversal launch then asks for all the blocks a CU holds;
    # with 11 contexts each launch is sized to share the GPU with ten others): (1) events around every stage -> EXCLUSIVE launch durations
    # (nothing else on the GPU): what the roofline figures use; (2) instrumented frames: node / triangle visits of this rank's queries
    # (counted, not modelled). The rocprofv3 passes under profiles/ run the same configuration (bench.py --profile-pass).
    torch.cuda.synchronize()
    # (a handle with ONE frame context runs the shadow rays of bounce b on a side stream beside the closest-hit rays of bounce b + 1 unless
    # told otherwise: the exclusive figures need every launch alone on the GPU)
    rx = backend.RenderHip(device_ordinal=local_rank, rank=r.rank, world_size=r.world_size, stripe_rows=args.stripe_rows, stream=stream, frames_in_flight=1,
                           options=dict(lib_options, side_connect=0))
    rx.initialize(W, H)
    rx.set_scene(scene)
    if args.animate and args.rebuild_budget != 0:
        rx.set_bvh_policy(force_bvh_rebuild=args.rebuild_budget < 0, rebuild_triangle_budget=max(args.rebuild_budget, 0))

    def step_x(count=False):
        return rx.wait(submit_on(rx, count))

    rx.set_stage_timing(2)
    for _ in range(3):
        step_x()
    serial = dict(ext=0.0, con=0.0, shade=0.0, tail=0.0, resolve=0.0, other=0.0, gpu=0.0)
    n_serial = max(3, min(10, args.steps))
    serial_launches = 0
    for _ in range(n_serial):
        st = step_x().raw
        serial["ext"] += st.extend_time_ms / n_serial
        serial["con"] += st.connect_time_ms / n_serial
        serial["shade"] += st.shade_only_time_ms / n_serial
        serial["tail"] += st.tail_time_ms / n_serial
        serial["resolve"] += st.resolve_time_ms / n_serial
        serial["other"] += (st.shade_time_ms - st.shade_only_time_ms - st.tail_time_ms - st.resolve_time_ms) / n_serial
        serial["gpu"] += st.render_time_ms / n_serial
        serial_launches = int(st.launches_extend)

    # ---- sustained: the same schedule for at least --sustained-seconds more (the timed region of a 20-step run lasts 26 ms: no sampler
    # sees it and it says nothing about clocks under load); and, for comparison, the same with a camera that stands still
    sustained, static_leg = None, None
    if args.sustained_seconds > 0 and world == 1:
        # ONE uninterrupted run of the schedule (its length from the timed region's rate), not chunks: every call of timed_steps drains
        # the queue at its end, and a chunk of 8 frames with 7 in flight (C5) was mostly fill and drain (round 5: "sustained" 3.2 ms
        # against 2.75 timed for that reason alone)
        est_ms = max(elapsed * 1e3 / max(args.steps, 1), 1e-3)
        chunk = max(batch_frames * fif, 8)
        n_s, t_s = 0, time.perf_counter()
        while n_s == 0 or time.perf_counter() - t_s < 0.9 * args.sustained_seconds:
            n_run = max(chunk, int((args.sustained_seconds - (time.perf_counter() - t_s)) * 1e3 / est_ms) // batch_frames * batch_frames)
            timed_steps(n_run, lambda st: None)
            n_s += n_run
        torch.cuda.synchronize()
        dt = time.perf_counter() - t_s
        sustained = {"ms_per_step": round(dt * 1e3 / n_s, 4), "steps": n_s, "seconds": round(dt, 3)}
        if not args.static_camera and anim is None:
            args.static_camera = True
            timed_steps(chunk, lambda st: None)
            torch.cuda.synchronize()
            n_c, t_c = 0, time.perf_counter()
            while n_c == 0 or time.perf_counter() - t_c < 0.9 * min(0.5, args.sustained_seconds):
                n_run = max(chunk, int((min(0.5, args.sustained_seconds) - (time.perf_counter() - t_c)) * 1e3 / est_ms) // batch_frames * batch_frames)
                timed_steps(n_run, lambda st: None)
                n_c += n_run
            torch.cuda.synchronize()
            static_leg = {"ms_per_step": round((time.perf_counter() - t_c) * 1e3 / n_c, 4), "steps": n_c,
                          "what": "the same schedule with ONE camera for all frames (what rounds 1-3 timed)"}
            args.static_camera = False
    def counted(depth=None):
        """one instrumented frame (COUNT kernels, every bounce a stand-alone launch), optionally cut at `depth` bounces"""
        full = rx.params.max_path_depth
        if depth is not None:
            rx.params.max_path_depth = depth
        try:
            stc = step_x(count=True).raw
        finally:
            rx.params.max_path_depth = full
        return dict(rays_closest=int(stc.rays_closest), rays_shadow=int(stc.rays_shadow), hits=int(stc.hits_shaded),
                    nodes_closest=int(stc.nodes_closest), tris_closest=int(stc.tris_closest),
                    nodes_shadow=int(stc.nodes_visited - stc.nodes_closest), tris_shadow=int(stc.tris_tested - stc.tris_closest),
                    launches=int(stc.launches_extend))
    cnt = counted()
    max_depth = cnt.pop("launches")
    # The timed frames hand the late bounces to the tail kernel: `launches_extend` stand-alone closest-hit launches (bounces
    # 0..k-1) and as many shadow-ray launches per frame. Their work, counted exactly: the closest-hit queries of a frame cut at k
    # bounces, the shadow queries of a frame cut at k+1 (the last bounce of a path issues no shadow ray).
    # (counted on the handle that renders one frame at a time, the one all per-kernel figures below come from: the pipelined run batches
    # several frames per launch sequence and may hand over to the tail kernel one bounce later)
    launches_extend = serial_launches if serial_launches > 0 else max_depth
    if launches_extend < max_depth:
        cnt_ext = counted(launches_extend)
        cnt_con = counted(launches_extend + 1)
    else:
        cnt_ext = cnt_con = cnt

    latency_1_per_rank = None
    if world > 1:   # (the control plane is a gloo group)
        elapsed, ext_ms, gather_host_ms, serial, rays, latency_1_per_rank = combine_ranks(dist, elapsed, ext_ms, gather_host_ms, serial, rays,
                                                                                             (latency.get("1") or {}).get("ms_per_frame"))
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    K = args.steps
    ms_per_step = elapsed * 1e3 / K
    mrays = rays / elapsed / 1e6
    # ---- roofline of the dominant kernel: rp_k_extend (closest-hit BVH4 traversal), rank 0's share.
    # All durations below are EXCLUSIVE: HIP events on the dispatch packets of frames rendered one at a time (nothing else on the GPU).
    primary = r.local_pixel_count() * spp  # the first launch computes its camera rays and its queue instead of reading them
    ext_bytes = (cnt_ext["rays_closest"] * (QUEUE_BYTES + RAY_BYTES + HIT_BYTES) + cnt_ext["hits"] * HIT_TUV_BYTES - primary * (RAY_BYTES + QUEUE_BYTES - FIRST_RAY_BYTES)
                 + cnt_ext["nodes_closest"] * NODE_BYTES + cnt_ext["tris_closest"] * TRI_BYTES)
    con_bytes = (cnt_con["rays_shadow"] * (QUEUE_BYTES + RAY_BYTES + SHADOW_RESULT_BYTES) + cnt_con["nodes_shadow"] * NODE_BYTES
                 + cnt_con["tris_shadow"] * TRI_BYTES)
    shade_vertices = cnt_ext["rays_closest"]       # one shade invocation per closest-hit query of the stand-alone bounces
    shade_bytes = (shade_vertices * (QUEUE_BYTES + HIT_BYTES + PATH_READ_BYTES + PATH_WRITE_BYTES) - primary * (QUEUE_BYTES + PATH_READ_BYTES - FIRST_RAY_BYTES)
                   + cnt_ext["hits"] * (HIT_TUV_BYTES + VERTEX_BYTES + MATERIAL_BYTES))
    # all bounces of a frame (the tail kernel's included), counted on the instrumented frame
    total_alg_bytes = (cnt["rays_closest"] * (QUEUE_BYTES + RAY_BYTES + HIT_BYTES) + 2 * cnt["hits"] * HIT_TUV_BYTES - primary * (RAY_BYTES + QUEUE_BYTES - FIRST_RAY_BYTES) + cnt["nodes_closest"] * NODE_BYTES
                       + cnt["tris_closest"] * TRI_BYTES + cnt["rays_shadow"] * (QUEUE_BYTES + RAY_BYTES + SHADOW_RESULT_BYTES) + cnt["nodes_shadow"] * NODE_BYTES
                       + cnt["tris_shadow"] * TRI_BYTES + cnt["rays_closest"] * (QUEUE_BYTES + HIT_BYTES + PATH_READ_BYTES + PATH_WRITE_BYTES)
                       - primary * (QUEUE_BYTES + PATH_READ_BYTES - FIRST_RAY_BYTES) + cnt["hits"] * (VERTEX_BYTES + MATERIAL_BYTES) + r.local_pixel_count() * (16 * spp + 36))
    wkey = workload_key(args, world)
    lib_build = library_build_id()
    pmc = load_pmc_traffic(wkey)   # the committed counter passes of THIS workload (profiles/pmc_traffic.json), None for anything else
    n_launch = max(launches_extend, 1)
    props = torch.cuda.get_device_properties(local_rank)
    # VALU issue ceiling: MEASURED (tools/microbench/valu_issue.hip). Plain v_fma / v_mul / v_add issue at one wave64 instruction per
    # 2 clocks per SIMD (912-964 G/s chip-wide, as MI355X_MICROARCH.md says), but v_cvt_f32_ubyte, v_pk_fma_f32, v_min3 / v_max -- most
    # of the BVH4 node step -- at half of that: the instruction mix of the node step tops out at 588 G/s. (Round 2 assumed 4 clocks for
    # everything = 614 G/s at 2.4 GHz: wrong reasoning, nearly the right number for this mix.)
    vp = load_valu_peak()
    valu_peak = vp["node_step_mix_ginst_s"] if vp else props.multi_processor_count * MAX_CLOCK_GHZ

    def own_peak(labels):
        """issue ceiling of a kernel class: the mean over its instantiations' static mixes; None when the mixes were not measured"""
        k = (vp or {}).get("kmix") or {}
        v = [k[l] for l in labels if l in k]
        return sum(v) / len(v) if v else None

    def kernel_entry(name, prefix_list, alg_bytes_step, ms_step, launches, mix_labels=(), node_step=False):
        """one kernel class: exclusive time, algorithmic bytes and their rate against the cache-hierarchy ceiling, counter traffic
        and its rate against the HBM peak"""
        tr = None
        if pmc:
            parts = [traffic_of(pmc, p) for p in prefix_list]
            if all(v is not None for v in parts):
                tr = sum(parts) / len(parts)       # mean bytes per launch over the listed instantiations (one launch each per frame)
        launch_ms = ms_step / max(launches, 1)
        e = {"kernel": name, "launches_per_step": launches, "launch_ms": round(launch_ms, 5), "exclusive_ms_per_step": round(ms_step, 4),
             "algorithmic_bytes_per_launch": int(alg_bytes_step // max(launches, 1)),
             "algorithmic_gbs": round(alg_bytes_step / (ms_step * 1e-3) / 1e9, 1) if ms_step > 0 else 0.0}
        e["algorithmic_frac"] = round(e["algorithmic_gbs"] / L2_PEAK_GBS, 4)
        e["hbm_bytes_per_launch"] = int(tr) if tr is not None else None
        e["hbm_gbs"] = round(tr / (launch_ms * 1e-3) / 1e9, 1) if (tr is not None and launch_ms > 0) else None
        e["hbm_frac"] = round(e["hbm_gbs"] / HBM_PEAK_GBS, 4) if e["hbm_gbs"] is not None else None
        # VALU issue (PMC: SQ_INSTS_VALU per launch) against the measured issue ceiling of the node step's instruction mix; and where the
        # resident waves' cycles go (SQ_WAIT_ANY: parked on s_waitcnt = memory latency; SQ_WAIT_INST_ANY: issue stalls; SQ_ACTIVE_INST_*)
        vi = None
        if pmc:
            parts = [traffic_of(pmc, p, "valu_insts_per_launch") for p in prefix_list]
            if all(v is not None for v in parts):
                vi = sum(parts) / len(parts)
        e["valu_insts_per_launch"] = int(vi) if vi is not None else None
        e["valu_ginst_s"] = round(vi / (launch_ms * 1e-3) / 1e9, 1) if (vi is not None and launch_ms > 0) else None
        # the ceiling of THIS kernel's instruction mix: its static VALU histogram run through the issue microbenchmark (profiles/r05_kernel_mix.json);
        # the traversal kernels spend their time in the node step, whose hand-counted mix has its own measurement (the lower of the two is used)
        op = own_peak(mix_labels)
        peak = min(op, valu_peak) if (op is not None and node_step) else (op if op is not None else valu_peak)
        e["valu_peak_ginst_s"] = round(peak, 1)
        e["valu_peak_what"] = ("static mix of the kernel (%s)%s" % (", ".join(mix_labels), "; node-step mix %.0f" % valu_peak if node_step else "")) if op is not None else "node-step mix (no per-kernel measurement)"
        e["valu_frac"] = round(e["valu_ginst_s"] / peak, 4) if e["valu_ginst_s"] is not None else None
        if pmc:
            for fld in ("wait_any_frac", "wait_inst_any_frac", "active_inst_valu_frac", "tcc_hit_rate"):
                parts = [traffic_of(pmc, p, fld) for p in prefix_list]
                e[fld] = round(sum(parts) / len(parts), 4) if all(v is not None for v in parts) else None
        return e

    single = len(scene.instances) == 1 or (bool(flatten) and not args.animate and not args.dynamic_meshes)   # (a flattened scene is one identity instance over one tree)
    sfx = ", false, %s" % ("true" if single else "false")   # (COUNT, FIRST,) ALPHA, SINGLE [, TABLE]
    var_id = "1" if variant == abi.VARIANT_SIMPLE else "0"
    k_ext = kernel_entry("rp_k_extend<COUNT=false, FIRST, ALPHA=false, SINGLE=%s>: first bounce FIRST=true, later bounces FIRST=false" % str(single).lower(),
                         ["rp_k_extend<false, true" + sfx, "rp_k_extend<false, false" + sfx][:n_launch], ext_bytes, serial["ext"], launches_extend,
                         ["extend_first", "extend"][:n_launch], True)
    k_con = kernel_entry("rp_k_connect<COUNT=false, ALPHA=false, SINGLE=%s>" % str(single).lower(), ["rp_k_connect<false" + sfx], con_bytes, serial["con"], launches_extend, ["connect"], True)
    k_shade = kernel_entry("rp_k_shade<VARIANT=%s, FIRST, LIGHTS, TEX>" % var_id, ["rp_k_shade<%s, true" % var_id, "rp_k_shade<%s, false" % var_id][:n_launch],
                           shade_bytes, serial["shade"], launches_extend,
                           (["shade_first_lambert", "shade_lambert"] if variant == abi.VARIANT_SIMPLE else ["shade_first_gltf_lights", "shade_gltf_lights"])[:n_launch])
    hbm_known = k_ext["hbm_gbs"] is not None
    fetches = cnt_ext["nodes_closest"] + cnt_ext["tris_closest"]
    roofline = {
        # The contract's fields name what BINDS the dominant kernel: VALU instruction issue. `achieved` = wave64 VALU
        # instructions per second of one stand-alone closest-hit launch (PMC SQ_INSTS_VALU per launch / its exclusive HIP-event duration), `peak` =
        # the MEASURED issue ceiling of that kernel's instruction mix (tools/microbench/valu_issue.hip), `frac` their ratio; binding_frac is the
        # same for the whole pipelined frame. The byte view the contract's recipe asks for stays beside it: the kernel's ALGORITHMIC
        # bytes per launch / its duration against the HBM peak = algorithmic_frac_of_hbm_peak -- it can exceed 1 because the tree (tens of MB) is
        # served by the L2s and the Infinity Cache: a statement about work per second, not a bound -- and `traffic`, the bytes the counters
        # really see crossing the HBM-side fabric per launch (hbm_counter: a fraction of the peak).
        "bound": "valu_issue", "kernel": k_ext["kernel"], "unit": "G wave64 VALU instructions/s", "peak": k_ext["valu_peak_ginst_s"],
        "achieved": k_ext["valu_ginst_s"], "frac": k_ext["valu_frac"],
        "algorithmic_frac_of_hbm_peak": round(k_ext["algorithmic_gbs"] / HBM_PEAK_GBS, 4), "hbm_peak_gbs": HBM_PEAK_GBS,
        "algorithmic_note": "algorithmic bytes of the dominant kernel per launch / exclusive launch duration / 8 TB/s (SURVEY 8d's recipe): cache-served (tree + triangles "
                            "live in L2 + Infinity Cache), not a bound; `traffic` is what reaches the HBM side",
        "counters_build": (pmc or {}).get("library_build_id") or (pmc or {}).get("build"), "library_build": lib_build,
        "counters_stale": (bool(pmc) and (pmc.get("library_build_id") != lib_build)),
        "counters_note": "traffic / valu.* / wait_any_frac / tcc_hit_rate come from the committed counter passes profiles/pmc_traffic.json (rocprofv3 --pmc cannot run inside "
                         "this process); counters_stale = they were taken on a library built from other sources than the one loaded now (rptr_hip_build_id)",
        "traffic": k_ext["hbm_bytes_per_launch"],
        "traffic_source": ("profiles/pmc_traffic.json[%s]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_workloads.sh) of this workload, bytes per launch "
                           "averaged over the stand-alone closest-hit launches of a frame; gfx950 correction 2 x FETCH_SIZE" % wkey) if hbm_known else None,
        "hbm_counter": {"gbs": k_ext["hbm_gbs"], "frac_of_hbm_peak": k_ext["hbm_frac"],
                        "traffic_over_algorithmic": round(k_ext["hbm_bytes_per_launch"] / max(k_ext["algorithmic_bytes_per_launch"], 1), 4) if hbm_known else None,
                        "what": "bytes that really cross the HBM-side fabric (counters) / exclusive launch time / 8 TB/s: HBM is nearly idle, the kernel does not wait for it"},
        # what the measurements say binds the frame (neither of the contract's two): VALU instruction issue of the whole pipeline against the
        # MEASURED issue ceiling of the traversal's instruction mix, see "valu"; one kernel alone additionally waits on memory LATENCY
        # (wait_any_frac of its wave-cycles), which the frames in flight hide
        "binding": "valu_issue", "binding_frac": (valu_frame(pmc, ms_per_step, valu_peak, launches_extend) or {}).get("pipelined_frac") if pmc else None,
        "binding_note": ("nearest single ceiling, not the only one: 7 % fewer vector instructions per node step moved the pipelined frame by 0.6 %, 8 bytes less per "
                         "divergent node fetch by 1.9 % (profiles/r06_notes.md section 9) -- issue, the waves' dependent-fetch latency and the L1 return path of "
                         "the lane fetches lie within a few per cent of one another"),
        "workload_key": wkey,
        "hbm_frac": k_ext["hbm_frac"], "algorithmic_frac": k_ext["algorithmic_frac"],
        "algorithmic_bytes_per_launch": k_ext["algorithmic_bytes_per_launch"], "algorithmic_gbs": k_ext["algorithmic_gbs"],
        "algorithmic_ceiling": {"gbs": L2_PEAK_GBS, "what": "aggregate L2 bandwidth, MI355X_MICROARCH.md 'L2 (per XCD)': the tree is cache resident, so the bytes the "
                                                            "lanes consume are bounded by the cache hierarchy, not by HBM"},
        "launch_ms": k_ext["launch_ms"], "launches_per_step": launches_extend, "exclusive_ms_per_step": k_ext["exclusive_ms_per_step"],
        "timing": "exclusive: HIP events on the dispatch packets of %d frames rendered ONE AT A TIME after the timed region, on a handle with one frame context (no other frame on the GPU, every launch at full size); "
                  "sum of all stages = stage_ms_per_step.gpu_total" % n_serial,
        "valu": {"binding_unit": "VALU issue", "peak_ginst_s": round(valu_peak, 1),
                 "peak_what": ("MEASURED: wave64 instructions / s of the BVH4 node step's instruction mix (%s) at 8 waves per SIMD, tools/microbench/valu_issue.hip; "
                               "plain v_fma / v_mul / v_add_u32 reach %.0f G/s (1 per 2 clocks per SIMD), "
                               "v_pk_fma_f32 / v_cvt_f32_ubyte / v_min3 / v_max %.0f G/s" % (vp["node_step_mix_what"], vp["full_rate_ginst_s"], vp["half_rate_ginst_s"])) if vp else
                              "%d CUs x 4 SIMDs x 1 wave64 VALU instruction per 4 clocks x %.1f GHz (profiles/r03a_valu_issue.json absent)" % (props.multi_processor_count, MAX_CLOCK_GHZ),
                 "peak_source": vp["source"] if vp else None,
                 "wait_any_frac": k_ext.get("wait_any_frac"), "wait_inst_any_frac": k_ext.get("wait_inst_any_frac"),
                 "frac": k_ext["valu_frac"], "ginst_s": k_ext["valu_ginst_s"], "insts_per_launch": k_ext["valu_insts_per_launch"],
                 "full_rate_ginst_s": round(vp["full_rate_ginst_s"], 1) if vp else None,
                 "ceiling_note": "the ceiling is that of the node step's instruction mix; a frame with more triangle tests and shading per node visit (C3, C4) "
                                 "issues more full-rate instructions and can exceed it (its own ceiling lies between peak_ginst_s and full_rate_ginst_s)",
                 "frame": valu_frame(pmc, ms_per_step, valu_peak, launches_extend) if pmc else None,
                 "source": "profiles/pmc_traffic.json: rocprofv3 --pmc SQ_INSTS_VALU pass of this workload (tools/pmc.sh insts)" if pmc else None},
        "node_and_triangle_fetches_per_s": round(fetches / (serial["ext"] * 1e-3)) if serial["ext"] > 0 else None,
        "gather_reference": "tools/microbench/gather64.hip: fully divergent dependent 64-byte lane fetches run at 220 G/s (L1 resident), 120 G/s (32 MB), "
                            "62 G/s (256 MB) chip-wide; traversal fetches of neighbouring rays partly coincide, so this is a reference point, not a ceiling",
        "kernels": {"rp_k_extend": k_ext, "rp_k_connect": k_con, "rp_k_shade": k_shade},
        "stage_ms_per_step": {"extend": round(serial["ext"], 4), "connect": round(serial["con"], 4), "shade": round(serial["shade"], 4),
                              "tail": round(serial["tail"], 4), "resolve": round(serial["resolve"], 4), "other": round(serial["other"], 4),
                              "gpu_total": round(serial["gpu"], 4)},
        "pipelined": {"frames_in_flight": fif, "frames_per_launch_sequence": batch_frames, "ms_per_step": round(ms_per_step, 4),
                      "extend_launch_ms_overlapped": round(ext_ms * batch_frames / K / n_launch, 5),
                      "note": "launches of neighbouring frames share the GPU in the timed region: their durations overlap and are NOT exclusive (their sum may "
                              "exceed ms_per_step); they are reported for the rocprofv3 cross-check only (profiles/, same command)"},
        "sustained": sustained,
        "static_camera": static_leg,
        # every stage's algorithmic bytes of a frame over the pipelined frame time: more than HBM could deliver -- the tree (tens of MB) is
        # served by the L2s and the Infinity Cache; a statement about work per second, not about a bound
        "all_stages_algorithmic": {"bytes_per_step": int(total_alg_bytes), "gbs": round(total_alg_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                                   "over_hbm_peak": round(total_alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                                   "what": "extend + connect + shade (all bounces, counted) + resolve bytes per frame / pipelined ms per frame: cache-served"},
        "latency": latency,
        "counts_per_step": cnt,
        "tail": {"from_bounce": launches_extend, "max_path_depth": max_depth,
                 "note": "bounces >= from_bounce run in one rp_k_tail launch per frame; the kernel figures cover the stand-alone launches of bounces < from_bounce",
                 "standalone_counts": {"rays_closest": cnt_ext["rays_closest"], "nodes_closest": cnt_ext["nodes_closest"],
                                       "tris_closest": cnt_ext["tris_closest"], "rays_shadow": cnt_con["rays_shadow"],
                                       "nodes_shadow": cnt_con["nodes_shadow"], "tris_shadow": cnt_con["tris_shadow"]}} if launches_extend < max_depth else None,
    }
    if refit_ms is not None:
        roofline["update_vertices_and_refit_ms"] = round(refit_ms, 4)
        roofline["bvh_policy"] = {"rebuild_budget": args.rebuild_budget, "device_rebuilds": r.bvh_rebuild_count()}
    bsdf = "diffuse-only" if variant == abi.VARIANT_SIMPLE else "glTF"
    which = "configs[2]" if args.lights else "configs[1]"
    if args.scene == "forest":
        what = "C4: instanced forest, %d unique / %d instanced triangles, %d instances" % (scene.num_tris(), scene.num_instanced_tris(),
                                                                                          len(scene.instances))
    elif args.animate:
        what = "C5: animated %d-triangle height field (dynamic mesh, device-side vertex animation + BVH refit every frame)" % scene.num_tris()
    else:
        what = "%s: procedural %d-triangle height field%s" % (which, scene.num_tris(),
                                                              " + 512 emissive triangles (binned-RIS NEE)" if args.lights else "")
    out = {
        "metric": "Mrays/s", "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%s, %dx%d, %d spp, %s BSDF, sun+sky, max depth 9" % (what, W, H, spp, bsdf),
                   "camera": "one view for all frames" if args.static_camera else "moves every frame (a camera per frame inside a launch sequence)",
                   "frame_schedule": "stage launches", "fast_math": fast_math_in_force,
                   "frames_in_flight": fif, "frames_per_launch_sequence": batch_frames, "flattened_instances": bool(flatten) and len(scene.instances) > 1,
                   "parallelism": "tile%d" % world if args.emulate_world <= 1 else "rank 0 of an emulated tile%d split" % args.emulate_world, "stripe_rows": args.stripe_rows, "rays_per_step": rays // K,
                   "scene_gen_s": round(t_scene, 2), "bvh_build_s": round(t_build, 2),
                   "bvh": {"built_on": "device (csrc/ploc.h)" if bvh_on_device else "host (csrc/bvh_build.cpp)", "acceleration_structure_step_ms": round(bvh_step_ms, 1),
                           "device_ms": round(bvh_device_ms, 2), "area_cost": round(bvh_area_cost, 1),
                           "traversal_thresholds": {"node_min": trav_node_min or 10, "refill_min": trav_refill_min or 48,
                                                    "pool_entries": int(os.environ.get("RPTR_TRAVERSE_FETCH", "0")) or (256 if trav_node_min else 384),
                                                    "preset": "dense (area cost >= 24)" if trav_node_min else "default"}}},
        "roofline": roofline,
    }
    if world > 1:
        out["gather"] = gather_report(gather_mode, gather_transport, probe, batch_frames, batched_gather, gather_gpu_ms, gather_host_ms, gathers_timed,
                                      W * H * 16 - my_bytes, gather_note, latency_1_per_rank)

    # ---- boundary: the same workload through the drop-in's own host code -- bin/rptr_hip (host/rptr_cli.cpp: C++, the RenderBackend-shaped
    # adapter host/render_hip.hpp over the C ABI, nothing of this Python file) with bench.py's camera path: (a) the reference's frame loop,
    # begin_frame / draw_frame / end_frame with a command stream = two swap buffers in flight (app.cpp:453-469, util/display/
    # render_graphic.h:19); (b) queued as deep as `value`'s schedule. VERDICT r4: "benchmark the drop-in".
    if latency_in_children or (world == 1 and args.emulate_world <= 1 and not args.no_boundary and not args.animate and not args.static_camera):
        # (this process is done with the GPU: its handles go first -- their hardware queues with them: two processes with a dozen streams each
        # oversubscribe the GPU's queues and the driver time-slices them: 3.9 instead of 1.14 ms per frame for the child, measured)
        for hdl in (r, rx):
            try:
                hdl.close()
            except Exception:
                pass
        torch.cuda.synchronize()
    if latency_in_children:
        import subprocess
        for depth in (1, 2):
            cmd = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:]] + ["--latency-leg", str(depth), "--no-probe", "--no-cpu-baseline", "--no-boundary"]
            try:
                pr = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                line = [l for l in pr.stdout.splitlines() if l.startswith('{"latency_leg"')]
                latency[str(depth)] = json.loads(line[-1])["latency_leg"] if line else {"note": "the child printed no result (rc %d): %s" % (pr.returncode, pr.stderr[-200:])}
            except subprocess.TimeoutExpired:
                latency[str(depth)] = {"note": "timed out"}
    if world == 1 and args.emulate_world <= 1 and not args.no_boundary and not args.animate and not args.static_camera:
        out["boundary"] = boundary_leg(args, scene, W, H, spp, fif, batch_frames, ms_per_step)

    # ---- the other BASELINE configurations, briefly ("let the driver see C3"): after the headline and outside --steps, one
    # short run of this file per configuration in a child process (its own handles; this process has closed its own) -- the pipelined
    # schedule (100 steps), two frames in flight (what a reference-shaped host reaches), the exclusive stage split. C3 is the reference's
    # actual default renderer (glTF BSDF + binned-RIS NEE).
    if wkey == "c2" and not args.no_other_configs and not args.profile_pass:
        import subprocess
        legs = (("c3", ["--lights", "--variant", "gltf", "--spp", "8"]), ("c3_fast_math", ["--lights", "--variant", "gltf", "--spp", "8", "--fast-math", "1"]),
                ("c4", ["--scene", "forest"]), ("c5", ["--animate", "--width", "3840", "--height", "2160", "--spp", "2"]))
        out["other_configs"] = {}
        for name, cfg_args in legs:
            cmd = [sys.executable, os.path.abspath(__file__)] + cfg_args + ["--steps", "100", "--warmup", "4", "--no-cpu-baseline", "--no-boundary", "--no-other-configs",
                                                                           "--sustained-seconds", "0.4"]   # (sustained + static-camera legs: C4's moving views cost 20 % more than its configuration's)
            t_leg = time.time()
            try:
                pr = subprocess.run(cmd, capture_output=True, text=True, timeout=180)   # (a leg takes ~12 s; a hung child must not hold the headline back)
                line = [l for l in pr.stdout.splitlines() if l.startswith('{"metric"')]
                if not line:
                    out["other_configs"][name] = {"note": "the child printed no result (rc %d): %s" % (pr.returncode, pr.stderr[-200:])}
                    continue
                d = json.loads(line[-1])
                rf = d["roofline"]
                out["other_configs"][name] = {
                    "workload": d["config"]["workload"], "camera": d["config"]["camera"], "rays_per_step": d["config"]["rays_per_step"],
                    "ms_per_step": d["ms_per_step"], "mrays_s": d["value"], "steps": d["steps"],
                    "frames_in_flight": d["config"]["frames_in_flight"], "frames_per_launch_sequence": d["config"]["frames_per_launch_sequence"],
                    "sustained_ms": (rf.get("sustained") or {}).get("ms_per_step"), "static_camera_ms": (rf.get("static_camera") or {}).get("ms_per_step"),
                    "two_in_flight_ms": (rf["latency"].get("2") or {}).get("ms_per_frame"), "one_at_a_time_ms": (rf["latency"].get("1") or {}).get("ms_per_frame"),
                    "exclusive_stage_ms": rf["stage_ms_per_step"], "flattened_instances": d["config"]["flattened_instances"], "fast_math": d["config"]["fast_math"],
                    "update_vertices_and_refit_ms": rf.get("update_vertices_and_refit_ms"), "seconds": round(time.time() - t_leg, 1)}
            except subprocess.TimeoutExpired:
                out["other_configs"][name] = {"note": "timed out"}

    # ---- CPU baseline: the build's own scalar backend -- the oracle's sources (scalar BVH2 traversal + the
    # shading restatement) compiled WITHOUT their diagnostics, -O3 -march=native -ffp-contract=off, 32 x 32 screen tiles over the host's
    # threads (oracle/Makefile libcpu_baseline.so; rebuilt here for this host when a compiler is at hand, else the portable x86-64-v3 copy
    # that travels with the repository) -- first checked, on a band of rows, against the image of the oracle the parity tests use: same bits.
    # 1 warm-up band, then 3 timed passes, median (a rate).
    if world == 1 and not args.no_cpu_baseline and args.emulate_world <= 1:
        import tempfile
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        # the heavy configurations (C3 - C5) at 480 x 270 with the same spp, as BASELINE.md section 2 prescribes: Mrays/s is a rate
        cw, chh = (W, H) if (args.scene == "grid" and not args.lights and not args.animate and W * H * spp <= 1920 * 1080 * 4) else (480, 270)
        W_gpu, H_gpu = W, H
        W, H = cw, chh
        band = (H // 2, H // 2 + 8)
        osc = O.OracleScene(scene)
        osc.build_bvh()
        want, _ = osc.render(W, H, 1, variant=variant, rows=band, threads=0)
        del osc
        tmpdir = tempfile.mkdtemp(prefix="rptr_cpu_baseline_")
        native_lib = O.build_baseline(tmpdir)
        O.use_library(native_lib or O.build_baseline())
        osc = O.OracleScene(scene)
        osc.build_bvh()
        got, _ = osc.render(W, H, 1, variant=variant, rows=band, threads=0)  # (also the warm-up)
        same_bits = bool((want[band[0]:band[1]].view("uint32") == got[band[0]:band[1]].view("uint32")).all())
        cores = host_cpu_budget(O.lib().orc_hw_threads())
        # bounded sample: the whole frame when there are many cores, a centred band of rows otherwise (~10-30 core-seconds per pass)
        rows = (0, H) if cores >= 16 else (H // 2 - H // 8, H // 2 + H // 8)
        runs = []
        for _ in range(3):
            _, ost = osc.render(W, H, spp, variant=variant, rows=rows, threads=cores)
            runs.append(((ost.rays_closest + ost.rays_shadow) / ost.seconds / 1e6, ost.rays_closest + ost.rays_shadow, ost.seconds, int(ost.threads)))
        runs.sort()
        rate, cpu_rays, secs, threads = runs[1]
        out["cpu_baseline"] = {
            "value": round(rate, 3), "unit": "Mrays/s", "cores": threads, "kind": "port", "mrays_s_per_core": round(rate / max(threads, 1), 3),
            "build": "oracle/libcpu_baseline.so: -O3 -march=%s -ffp-contract=off -DORC_BASELINE (no counters, no diagnostics), 32 x 32 tiles over std::thread"
                     % ("native (built on this host)" if native_lib else "x86-64-v3 (the portable copy: no compiler on this host)"),
            "image_equals_the_oracles": same_bits,
            "sample": "median of 3 timed passes over rows %d..%d of the same %dx%d frame at %d spp (%d rays in %.2f s; all three: %s Mrays/s); "
                      "scalar BVH2 traversal + shading%s"
                      % (rows[0], rows[1] - 1, W, H, spp, cpu_rays, secs, ", ".join("%.1f" % x[0] for x in runs),
                         "" if (W, H) == (W_gpu, H_gpu) else "; reduced resolution (the GPU frame is %dx%d): BASELINE.md section 2" % (W_gpu, H_gpu)),
        }
        O.use_library(None)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
