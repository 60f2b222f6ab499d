"""Multi-GPU plumbing: screen-space stripe assignment and the one collective of
the path, a gather of tile radiance to rank 0. torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests) is used as
plumbing only; rendering itself involves no communication: every rank holds a
replica of the scene and renders the rows it owns.
"""
from typing import List, Tuple


def tile_rows(height: int, stripe_rows: int, rank: int, world: int) -> List[Tuple[int, int]]:
    """(first_row, num_rows) of the stripes owned by `rank`: stripe s -> rank s % world
    (same rule as rptr_hip_tile_rows in csrc/rptr_hip.hip)."""
    n_stripes = (height + stripe_rows - 1) // stripe_rows
    return [(s * stripe_rows, min(stripe_rows, height - s * stripe_rows)) for s in range(rank, n_stripes, world)]


def local_rows(height: int, stripe_rows: int, rank: int, world: int) -> int:
    return sum(c for _, c in tile_rows(height, stripe_rows, rank, world))


class TileGather:
    """Gathers every rank's packed rows (float32 RGBA) into the full frame on rank 0.

    Tiles differ in size by at most one stripe, so each rank sends a buffer padded
    to the largest tile: one `gather` per frame (RCCL has no native gather; torch
    lowers it to grouped send/recv, each peer using its own xGMI link to rank 0)."""

    def __init__(self, width, height, stripe_rows, rank, world, device, group=None):
        import torch
        self.torch = torch
        self.group = group   # the process group the tiles travel through (None: the default group)
        self.width, self.height, self.rank, self.world = width, height, rank, world
        self.layout = [tile_rows(height, stripe_rows, k, world) for k in range(world)]
        self.tile_rows = [sum(c for _, c in rows) for rows in self.layout]
        self.tile_pixels = [r * width for r in self.tile_rows]
        self.max_rows = max(self.tile_rows) if self.tile_rows else 0
        self.max_tile = self.max_rows * width
        self.tile = torch.zeros((max(self.max_tile, 1), 4), dtype=torch.float32, device=device)
        root = rank == 0
        # rank 0 receives into one contiguous buffer [world, max_rows * width, 4]; the per-rank receive buffers are views
        self.recv_all = torch.zeros((world, max(self.max_tile, 1), 4), dtype=torch.float32, device=device) if root else None
        self.recv = [self.recv_all[k] for k in range(world)] if (world > 1 and root) else None
        self.frame = torch.zeros((height, width, 4), dtype=torch.float32, device=device) if root else None
        # frame row y comes from packed row row_index[y] of the receive buffer: the whole frame is assembled by ONE
        # index_select instead of one copy per stripe (34 stripes at 1080p would be 34 launches per frame on rank 0)
        self.row_index = None
        if root:
            idx = torch.zeros((height,), dtype=torch.int64)
            for k in range(world):
                off = 0
                for first, cnt in self.layout[k]:
                    idx[first:first + cnt] = torch.arange(cnt, dtype=torch.int64) + (k * max(self.max_rows, 1) + off)
                    off += cnt
            self.row_index = idx.to(device)

    def assemble(self):
        """packed rows of every rank (self.recv_all) -> self.frame, one launch"""
        rows = self.recv_all.view(self.world * max(self.max_rows, 1), self.width * 4)
        self.torch.index_select(rows, 0, self.row_index, out=self.frame.view(self.height, self.width * 4))
        return self.frame

    def gather(self):
        """self.tile holds this rank's packed rows; returns the frame on rank 0 (None elsewhere)."""
        if self.world == 1:
            self.recv_all[0].copy_(self.tile)
            return self.assemble()
        import torch.distributed as dist
        dist.gather(self.tile, self.recv, dst=0, group=self.group)
        return self.assemble() if self.rank == 0 else None


def exchange_unique_id(make_id, rank, world):
    """rank 0 calls make_id() (128 bytes, rptr_hip_comm_get_unique_id); every rank returns the same bytes. The bytes travel through
    torch.distributed's own store-backed object broadcast -- plumbing only, any side channel would do (rptr_cli uses a file)."""
    import torch.distributed as dist
    box = [make_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    return box[0]


class NativeGather:
    """The gather done by the library itself (include/rptr_hip.h "multi-GPU", csrc/host_comm.h): grouped ncclSend / ncclRecv on a
    communication stream of the backend, rows sent straight from the frame context's image, stripes interleaved by one HIP kernel
    on rank 0. Python only hands the RCCL unique id around once; per frame it makes ONE ctypes call (`gather()`), which returns as
    soon as the work is queued -- with frames in flight the gather of frame i runs beside the rendering of frames i+1.. ."""

    def __init__(self, renderer, rank, world, transport="rccl"):
        """transport "ipc": peer writes instead of RCCL -- rank 0 exports its frame buffers (hipIpcGetMemHandle), every rank maps them and
        scatters its rows into rank 0's frame itself (csrc/host_comm.h COMM_IPC)"""
        from .backend import RenderHip
        self.r, self.rank, self.world = renderer, rank, world
        if transport == "ipc":
            blob = exchange_unique_id(renderer.comm_ipc_export, rank, world)
            renderer.comm_ipc_init(blob)
        else:
            uid = exchange_unique_id(RenderHip.comm_unique_id, rank, world)
            renderer.comm_init_rank(uid)

    def gather(self, n_frames=1):
        """n_frames > 1: one collective for the frames of the launch sequence whose last ticket was just waited for"""
        self.r.gather(n_frames)

    def frame(self, out, index=None):
        """rank 0: waits for the last gather and copies the assembled frame (frame `index` of a batched gather; default its last)
        into `out` (float32, height x width x 4)"""
        return self.r.readback_gathered(out, index)

    def stats(self):
        return self.r.comm_stats()
