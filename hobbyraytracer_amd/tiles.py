"""Image-tile partition and gather used when the film is split over several GPUs (one process per GPU).

The partition itself is defined by the C ABI (``hrt_stripe_rows`` / ``hrt_stripe_row_index`` in
include/hrt.h): interleaved blocks of ``rows_per_block`` film rows, block ``b`` owned by rank
``b % n_ranks`` (contiguous row ranges would load-imbalance sky rows against geometry
rows).  This module only holds the torch.distributed side: every rank contributes its rows padded to
the largest share, one ``all_gather_into_tensor`` (RCCL over xGMI on GPUs, gloo in the CPU tests) moves
them, and a row permutation restores film order.  There is no other collective on the data path.
"""
import numpy as np

from . import api


class StripeLayout:
    def __init__(self, height, width, rows_per_block, n_ranks):
        self.height, self.width, self.rows_per_block, self.n_ranks = height, width, rows_per_block, n_ranks
        self.rows = [api.stripe_rows(height, rows_per_block, r, n_ranks) for r in range(n_ranks)]
        self.max_rows = max(self.rows) if self.rows else 0
        # film row -> index into the gathered (n_ranks * max_rows) row array
        perm = np.full(height, -1, dtype=np.int64)
        for r in range(n_ranks):
            idx = api.stripe_row_indices(height, rows_per_block, r, n_ranks)
            perm[idx] = r * self.max_rows + np.arange(len(idx))
        if (perm < 0).any():
            raise ValueError("stripe partition does not cover the film")
        self.perm = perm

    def row_indices(self, rank):
        return api.stripe_row_indices(self.height, self.rows_per_block, rank, self.n_ranks)


def gather_film(tile, layout, dist=None, out=None, gathered=None, force_collective=False):
    """tile: (max_rows, W, 3) tensor holding this rank's rows (rows beyond its share are padding).
    Returns the (H, W, 3) film in row order (on every rank).  `dist` = torch.distributed when n_ranks > 1;
    force_collective: run the all_gather with one rank too (exercises the RCCL path on a one-GPU box)."""
    import torch
    if layout.n_ranks > 1 or (force_collective and dist is not None):
        if gathered is None:
            gathered = torch.empty((layout.n_ranks * layout.max_rows, layout.width, 3), dtype=tile.dtype, device=tile.device)
        dist.all_gather_into_tensor(gathered, tile)   # concatenation along dim 0 (the form gloo and RCCL both accept)
        src = gathered
    else:
        src = tile
    key = str(tile.device)
    cache = layout.__dict__.setdefault("_perm_on", {})
    perm = cache.get(key)
    if perm is None:                                   # uploaded once per device, not once per frame
        perm = cache[key] = torch.as_tensor(layout.perm, device=tile.device)
    if out is None:
        return torch.index_select(src, 0, perm)
    torch.index_select(src, 0, perm, out=out)
    return out
