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

    def __init__(self, width, height, stripe_rows, rank, world, device):
        import torch
        self.torch = torch
        self.width, self.height, self.rank, self.world = width, height, rank, world
        self.layout = [tile_rows(height, stripe_rows, k, world) for k in range(world)]
        self.tile_pixels = [sum(c for _, c in rows) * width for rows in self.layout]
        self.max_tile = max(self.tile_pixels) if self.tile_pixels else 0
        self.tile = torch.zeros((max(self.max_tile, 1), 4), dtype=torch.float32, device=device)
        self.recv = [torch.zeros_like(self.tile) for _ in range(world)] if (world > 1 and rank == 0) else None
        self.frame = torch.zeros((height, width, 4), dtype=torch.float32, device=device) if rank == 0 else None

    def scatter_rows_into_frame(self, k, packed):
        off = 0
        for first, cnt in self.layout[k]:
            self.frame[first:first + cnt] = packed[off:off + cnt * self.width].view(cnt, self.width, 4)
            off += cnt * self.width

    def gather(self):
        """self.tile holds this rank's packed rows; returns the frame on rank 0 (None elsewhere)."""
        if self.world == 1:
            self.scatter_rows_into_frame(0, self.tile)
            return self.frame
        import torch.distributed as dist
        dist.gather(self.tile, self.recv, dst=0)
        if self.rank == 0:
            for k in range(self.world):
                self.scatter_rows_into_frame(k, self.recv[k])
        return self.frame
