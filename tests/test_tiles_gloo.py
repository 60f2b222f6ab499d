"""Multi-rank path on CPU: stripe assignment covers the frame exactly once and the
tile gather (the path's only collective) reassembles the frame, world_size 2 and 3, gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from realtimepathtracingresearchframework_amd import distributed as D


@pytest.mark.parametrize("height,stripe,world", [(1080, 32, 8), (1080, 32, 1), (37, 8, 3), (16, 32, 4), (2160, 32, 8)])
def test_stripes_partition_the_frame(height, stripe, world):
    seen = np.zeros(height, int)
    for r in range(world):
        rows = D.tile_rows(height, stripe, r, world)
        for first, cnt in rows:
            assert cnt > 0 and first % stripe == 0
            seen[first:first + cnt] += 1
        assert D.local_rows(height, stripe, r, world) == sum(c for _, c in rows)
    assert (seen == 1).all()
    # balance: tiles differ by at most one stripe
    sizes = [D.local_rows(height, stripe, r, world) for r in range(world)]
    assert max(sizes) - min(sizes) <= stripe


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, stripe, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = D.TileGather(W, H, stripe, rank, world, device="cpu")
    # every rank fills its packed rows with a function of (global row, column, channel)
    off = 0
    for first, cnt in g.layout[rank]:
        ys = torch.arange(first, first + cnt, dtype=torch.float32).view(cnt, 1, 1)
        xs = torch.arange(W, dtype=torch.float32).view(1, W, 1)
        cs = torch.arange(4, dtype=torch.float32).view(1, 1, 4)
        g.tile[off:off + cnt * W] = (ys * 1000 + xs + cs * 0.25).reshape(cnt * W, 4)
        off += cnt * W
    frame = g.gather()
    if rank == 0:
        torch.save(frame, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_gather_reassembles_the_frame_gloo(tmp_path, world):
    W, H, stripe = 24, 44, 8
    out = str(tmp_path / "frame.pt")
    mp.spawn(_worker, args=(world, _free_port(), W, H, stripe, out), nprocs=world, join=True)
    frame = torch.load(out)
    ys = torch.arange(H, dtype=torch.float32).view(H, 1, 1)
    xs = torch.arange(W, dtype=torch.float32).view(1, W, 1)
    cs = torch.arange(4, dtype=torch.float32).view(1, 1, 4)
    assert torch.equal(frame, ys * 1000 + xs + cs * 0.25)


# ---------------------------------------------------------------- bench.py's N > 1 bookkeeping (no GPU: the control plane is gloo)
def _bench_worker(rank, world, port, out_path):
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    serial = {k: 0.1 * (rank + 1) for k in bench.SERIAL_KEYS}
    elapsed, ext_ms, host_ms, serial, rays, lat = bench.combine_ranks(dist, 0.010 * (rank + 1), 3.0 + rank, 0.2 * (rank + 1), serial, 1000 * (rank + 1),
                                                                       0.5 + 0.25 * rank)
    if rank == 0:
        g = bench.gather_report("native", "rccl", {"torch_nccl": 1, "native": 1, "note": "", "seconds": 1.0}, 4, True, 0.05, host_ms, 5, 123456, None, lat)
        json.dump({"n_gpus": world, "elapsed": elapsed, "ext_ms": ext_ms, "serial": serial, "rays": rays, "gather": g}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bench_rank_bookkeeping_and_gather_report(tmp_path, world):
    """What the first real SCALE record will carry, on the CPU: rank 0's line names n_gpus, the transport, how many frames one
    collective moves, and every rank's one-frame-at-a-time latency; the timed region is the slowest rank's, rays are summed."""
    import json
    out = str(tmp_path / "line.json")
    mp.spawn(_bench_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    d = json.load(open(out))
    assert d["n_gpus"] == world and d["rays"] == 1000 * world * (world + 1) // 2
    assert d["elapsed"] == pytest.approx(0.010 * world) and d["ext_ms"] == pytest.approx(3.0 + world - 1)
    assert all(v == pytest.approx(0.1 * world) for v in d["serial"].values())
    g = d["gather"]
    assert g["transport"] == "rccl" and g["mode"].startswith("library") and g["frames_per_gather"] == 4 and g["gathers"] == 5
    assert g["latency_1_ms_per_rank"] == [pytest.approx(0.5 + 0.25 * r) for r in range(world)]
    assert g["host_ms_per_step"] == pytest.approx(0.2 * world) and g["probe"]["native"] == 1
