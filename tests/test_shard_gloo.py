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


def _worker(rank, world, port, out_dir, n_tiles=N_TILES, scramble=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N_TILES = n_tiles    # noqa: N806 -- (shadows the module's default on purpose)
        first, count = shard.tile_range(rank, world, N_TILES)
        blobs = [_encode(t) for t in range(first, first + count)]
        sizes, offsets = shard.gather_manifest([len(b) for b in blobs], N_TILES)
        slowest = shard.max_over_ranks(0.25 + rank)
        # the exchange step proper: every rank's arena (blobs at 16-byte aligned offsets, as the tile batch call leaves
        # them) travels to rank 0 through the collective; rank 0 keeps the gathered mosaic for the parent to check
        # (scramble: odd ranks place their blobs back to front, as a batch does with tiles it handed to the general path --
        # offsets[] then is not monotonic)
        order = list(range(len(blobs)))
        if scramble and rank % 2 == 1:
            order.reverse()
        loc_off, at = [0] * len(blobs), 0
        for i in order:
            loc_off[i] = at
            at += (len(blobs[i]) + 15) & ~15
        arena = torch.zeros(max(at, 1), dtype=torch.uint8)
        for o, b in zip(loc_off, blobs):
            arena[o:o + len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8)
        # the two halves of the step: what a rank does between them (decoding its own tiles) runs beside the transfers
        flight = shard.gather_arenas_start(arena, at, loc_off, [len(b) for b in blobs], root=0)
        mosaic, t_off, t_size, bases = flight.finish()
        if rank == 0:
            np.save(os.path.join(out_dir, "mosaic.npy"), mosaic.numpy())
            np.save(os.path.join(out_dir, "mosaic_off.npy"), t_off.numpy())
            np.save(os.path.join(out_dir, "mosaic_size.npy"), t_size.numpy())
        else:
            assert mosaic is None and t_off is None
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
    # ... and so is what rank 0 holds after the gather collective: every tile's blob, cut out of the GATHERED tensor by the
    # gathered offset / size tables, is the blob a single process makes of that tile; it decodes to the tile
    g = np.load(tmp_path / "mosaic.npy")
    g_off, g_size = np.load(tmp_path / "mosaic_off.npy"), np.load(tmp_path / "mosaic_size.npy")
    assert g_size.tolist() == want_sizes and len(g_off) == N_TILES and (g_off % 16 == 0).all()
    assert (np.diff(g_off) >= g_size[:-1]).all()                      # tile order, no overlap
    for t in range(N_TILES):
        blob = g[int(g_off[t]):int(g_off[t]) + int(g_size[t])].tobytes()
        assert blob == want[t], t
    rc, dec, _ = capi.oracle().decode(g[int(g_off[7]):int(g_off[7]) + int(g_size[7])].tobytes())
    assert rc == 0 and float(np.abs(dec.reshape(TILE, TILE).astype(np.float64) - _tile(7)).max()) <= 0.0101


@pytest.mark.parametrize("world,n_tiles", [(3, 11), (8, 21), (8, 5)])
def test_more_ranks_uneven_shares_empty_ranks_and_scrambled_arenas(tmp_path, world, n_tiles):
    """World sizes 3 and 8 over gloo: tile counts that do not divide (the first ranks take one more), ranks without a tile
    (5 tiles on 8 ranks), and arenas whose blobs do not lie in tile order (odd ranks place theirs back to front: offsets[] is
    not monotonic, as after a batch that handed tiles to the general path).  What rank 0 holds after the gather is, tile by
    tile, the blob a single process makes."""
    import capi
    if capi.oracle() is None:
        pytest.skip("oracle not built")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path), n_tiles, True), nprocs=world, join=True)
    want = [_encode(t) for t in range(n_tiles)]
    g = np.load(tmp_path / "mosaic.npy")
    g_off, g_size = np.load(tmp_path / "mosaic_off.npy"), np.load(tmp_path / "mosaic_size.npy")
    assert g_size.tolist() == [len(b) for b in want] and len(g_off) == n_tiles and (g_off % 16 == 0).all()
    for t in range(n_tiles):
        assert g[int(g_off[t]):int(g_off[t]) + int(g_size[t])].tobytes() == want[t], t
    spans = sorted((int(o), int(o) + int(z)) for o, z in zip(g_off, g_size))
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))    # no two blobs overlap
    for r in range(world):
        first, count, slowest = eval(open(tmp_path / f"t{r}.txt").read())
        assert (first, count) == shard.tile_range(r, world, n_tiles) and slowest == 0.25 + world - 1


def test_single_process_gather_is_the_arena_itself():
    arena = torch.arange(64, dtype=torch.uint8)
    mosaic, off, size, bases = shard.gather_arenas(arena, 40, [0, 16], [10, 24])
    assert mosaic.tolist() == list(range(40)) and off.tolist() == [0, 16] and size.tolist() == [10, 24] and bases == [0]


def _group_of_one(port, path):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        arena = torch.arange(64, dtype=torch.uint8)
        mosaic, off, size, bases = shard.gather_arenas(arena, 40, [16, 0], [24, 10], force_collective=True)
        # (the collective steps ran: the lengths' all-gather, the root's own copy into a buffer of its own, the tables)
        ok = (mosaic.tolist() == list(range(40)) + [v for v in mosaic.tolist()[40:]] and mosaic.data_ptr() != arena.data_ptr()
              and int(mosaic.numel()) == 48 and off.tolist() == [16, 0] and size.tolist() == [24, 10] and bases == [0])
        open(path, "w").write("ok" if ok else repr((mosaic.tolist(), off.tolist(), size.tolist(), bases)))
    finally:
        dist.destroy_process_group()


def test_collective_steps_in_a_group_of_one(tmp_path):
    """force_collective: a process group of ONE rank goes through the lengths' all-gather, the root's own copy and the tables -- what
    the GPU suite runs over RCCL (tests/test_gpu_parity.py: test_blob_gather_over_rccl_in_a_group_of_one), here over gloo."""
    import multiprocessing as mp
    port = 29600 + os.getpid() % 200
    path = str(tmp_path / "one.txt")
    p = mp.get_context("spawn").Process(target=_group_of_one, args=(port, path))
    p.start()
    p.join(120)
    assert p.exitcode == 0 and open(path).read() == "ok", open(path).read() if os.path.exists(path) else p.exitcode
