"""
Multi-GPU image-tile partition: one process per GPU, round-robin row stripes, one RCCL all-gather.

The reference has no distributed code at all; this is the new part the north star
asks for.  Pixels are independent, so the only exchange step of a frame is assembling it:

  rank r renders the rows {y : (y // stripe) % world == r}  (fvsrn_render_stripes, include/fvsrn.h) into a
  compact (8, rows_r, W) image; all compact images have the same size when H % (stripe*world) == 0, so ONE
  ``all_gather_into_tensor`` (RCCL over xGMI with backend "nccl", gloo on CPU) moves 8*rows*W*4 bytes per
  rank and a view/permute puts the stripes back in image order.

Round-robin stripes (instead of H/world contiguous blocks) balance empty-space rows against dense rows.
"""
from __future__ import annotations

from typing import List

import torch


def owned_rows(height: int, stripe: int, rank: int, world: int) -> List[int]:
    """Image rows of `rank`, in the order they appear in its compact image."""
    rows = []
    for y0 in range(rank * stripe, height, stripe * world):
        rows.extend(range(y0, min(y0 + stripe, height)))
    return rows


def check_even_partition(height: int, stripe: int, world: int) -> None:
    if stripe <= 0 or stripe % 8 != 0:
        raise ValueError("stripe must be a positive multiple of 8 (the pixel tile of one wave)")
    if height % (stripe * world) != 0:
        raise ValueError("height %d is not a multiple of stripe*world = %d: ranks would own different row counts"
                         % (height, stripe * world))


def all_gather_frame(local: torch.Tensor, gathered: torch.Tensor = None, group=None) -> torch.Tensor:
    """local (8, rows, W) of every rank -> (world, 8, rows, W) on every rank."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    # concatenation along dim 0 is the layout every backend (RCCL, gloo) accepts
    dist.all_gather_into_tensor(gathered.view((world * local.shape[0],) + tuple(local.shape[1:])), local.contiguous(), group=group)
    return gathered


def assemble(gathered: torch.Tensor, height: int, stripe: int) -> torch.Tensor:
    """(world, 8, rows, W) compact stripe images -> (1, 8, H, W) frame."""
    world, ch, rows, width = gathered.shape
    check_even_partition(height, stripe, world)
    assert rows * world == height
    return gathered.view(world, ch, rows // stripe, stripe, width).permute(1, 2, 0, 3, 4).reshape(1, ch, height, width)
