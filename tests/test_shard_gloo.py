"""N > 1 path on CPU: two processes over gloo partition a tile mosaic the way bench.py / a C5 job does
(lerc_amd/shard.py), each encodes only its own tiles (the oracle stands in for the HIP codec here -- this is a
test of the partitioning, not of the kernels) and the manifests must agree with a single-process run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from lerc_amd import shard  # noqa: E402

N_TILES, TILE = 11, 32


def _tile(t):
    rng = np.random.default_rng(1000 + t)
    y, x = np.mgrid[0:TILE, 0:TILE]
    return (1000 + 3 * t + 0.5 * x + 0.25 * y + rng.normal(0, 1, (TILE, TILE))).astype(np.float32)


def _encode(t):
    import capi
    rc, blob = capi.oracle().encode(_tile(t), 0.01)
    assert rc == 0
    return blob


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, count = shard.tile_range(rank, world, N_TILES)
        blobs = [_encode(t) for t in range(first, first + count)]
        sizes, offsets = shard.gather_manifest([len(b) for b in blobs], N_TILES)
        slowest = shard.max_over_ranks(0.25 + rank)
        np.save(os.path.join(out_dir, f"sizes{rank}.npy"), sizes.numpy())
        np.save(os.path.join(out_dir, f"offsets{rank}.npy"), offsets.numpy())
        with open(os.path.join(out_dir, f"blobs{rank}.bin"), "wb") as f:
            f.write(b"".join(blobs))
        with open(os.path.join(out_dir, f"t{rank}.txt"), "w") as f:
            f.write(repr((first, count, slowest)))
    finally:
        dist.destroy_process_group()


def test_tile_range_covers_everything():
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 8, 65536):
            got = [shard.tile_range(r, world, n) for r in range(world)]
            assert sum(c for _, c in got) == n
            assert all(got[i][0] + got[i][1] == got[i + 1][0] for i in range(world - 1))
            assert max(c for _, c in got) - min(c for _, c in got) <= 1
    with pytest.raises(ValueError):
        shard.tile_range(2, 2, 5)


def test_single_process_manifest():
    sizes, offsets = shard.gather_manifest([5, 7, 11], 3)
    assert sizes.tolist() == [5, 7, 11] and offsets.tolist() == [0, 5, 12, 23]
    assert shard.max_over_ranks(1.5) == 1.5


def test_two_ranks_over_gloo(tmp_path):
    import capi
    if capi.oracle() is None:
        pytest.skip("oracle not built")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    want = [_encode(t) for t in range(N_TILES)]
    want_sizes = [len(b) for b in want]
    for r in range(2):
        assert np.load(tmp_path / f"sizes{r}.npy").tolist() == want_sizes
        assert np.load(tmp_path / f"offsets{r}.npy").tolist() == [0] + np.cumsum(want_sizes).tolist()
        first, count, slowest = eval(open(tmp_path / f"t{r}.txt").read())
        assert slowest == 1.25                        # max over ranks of 0.25 + rank
        assert (first, count) == shard.tile_range(r, 2, N_TILES)
    # the ranks' arenas, concatenated in rank order, are the single-process mosaic byte for byte
    mosaic = open(tmp_path / "blobs0.bin", "rb").read() + open(tmp_path / "blobs1.bin", "rb").read()
    assert mosaic == b"".join(want)
