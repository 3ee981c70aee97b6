"""world_size-2 (and 3) test of the N > 1 path on CPU with the gloo backend: the stripe partition of
include/hrt.h + the all_gather + row permutation of hobbyraytracer_amd/tiles.py reassemble the film exactly.
No GPU here, so each rank fills its rows with a deterministic function of the ABSOLUTE pixel (exactly what
the RNG keying guarantees for the real kernel: a row's content does not depend on which rank renders it)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pixel_function(rows, width):
    """fp32 'film' content of absolute rows: any rank must produce the same values for the same row."""
    y = rows.astype(np.float64)[:, None, None]
    x = np.arange(width, dtype=np.float64)[None, :, None]
    c = np.arange(3, dtype=np.float64)[None, None, :]
    return (np.sin(0.37 * x + 1.3 * c) * np.cos(0.11 * y) + y * 1e-3 + c).astype(np.float32)


def _worker(rank, world, port, H, W, R, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from hobbyraytracer_amd import tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    layout = tiles.StripeLayout(H, W, R, world)
    rows = layout.row_indices(rank)
    tile = torch.full((layout.max_rows, W, 3), float("nan"), dtype=torch.float32)   # padding must never reach the film
    tile[:len(rows)] = torch.from_numpy(_pixel_function(rows, W))
    film = tiles.gather_film(tile, layout, dist)
    # the one real exchange step besides the gather: the 4 x u64 statistics all-reduce
    stats = torch.tensor([len(rows) * W, rank + 1, 0, 0], dtype=torch.int64)
    dist.all_reduce(stats)
    np.save(os.path.join(out_dir, f"film_{rank}.npy"), film.numpy())
    np.save(os.path.join(out_dir, f"stats_{rank}.npy"), stats.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W,R", [(2, 64, 40, 8), (2, 37, 16, 5), (3, 50, 8, 4)])
def test_stripe_gather_reassembles_film(built, tmp_path, world, H, W, R):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, H, W, R, str(tmp_path)), nprocs=world, join=True)
    expect = _pixel_function(np.arange(H), W)
    for r in range(world):
        film = np.load(tmp_path / f"film_{r}.npy")
        assert film.shape == (H, W, 3)
        assert np.array_equal(film.view(np.uint32), expect.view(np.uint32)), f"rank {r}"
        stats = np.load(tmp_path / f"stats_{r}.npy")
        assert stats[0] == H * W and stats[1] == world * (world + 1) // 2


def test_single_rank_layout_is_identity(built):
    import torch

    from hobbyraytracer_amd import tiles
    layout = tiles.StripeLayout(21, 6, 8, 1)
    assert layout.rows == [21] and (layout.perm == np.arange(21)).all()
    t = torch.arange(21 * 6 * 3, dtype=torch.float32).view(21, 6, 3)
    assert torch.equal(tiles.gather_film(t, layout), t)
