"""Multi-GPU partitioning of LERC work.

A band blob is ONE sequential block stream with one global zMin / zMax and one checksum, so a single raster does
not split across GPUs.  What shards is the unit above it: independent rasters / mosaic tiles, each its own blob.
Ranks take contiguous tile ranges and compress them without talking to each other; the one exchange step of a
mosaic job is the GATHER OF THE COMPRESSED BLOBS on the rank that writes the container (gather_arenas below):
an all-gather of the arena lengths (8 bytes per rank), then every rank sends its arena -- the blobs of its tiles as
lerc_amd_encode_tiles_device left them, device memory -- straight into its slice of the root's buffer with one
grouped send / receive batch (RCCL: one ncclGroup of point-to-point transfers, each rank's bytes travelling over
its own xGMI link to the root; no padding to the longest arena, no staging on the host), and the per-tile
(offset, size) tables follow the same way.  The other collective is the max-over-ranks of the elapsed time that
bench.py reports.  One process per GPU, torch.distributed ("nccl" is RCCL on ROCm; the CPU tests run the same code
over "gloo").

Under RCCL the exchange itself runs BELOW Python: lerc_amd_gather_blobs (include/lerc_amd_device.h, csrc/gather_rccl.cpp) on a
communicator of this module's own (RcclComm: ncclGetUniqueId on rank 0, handed round by torch.distributed, ncclCommInitRank) --
one message a rank, its (count, bytes) header and per-tile table in front of its arena; a sender posts its ncclSend without
waiting for anybody's length, the root's only host wait is for the lengths' 8 bytes a rank.  LERC_AMD_GATHER=torch keeps the
torch.distributed form (which the gloo tests run).
"""
import ctypes as ct
import os

import torch
import torch.distributed as dist


def tile_range(rank, world, n_tiles):
    """Contiguous share of `n_tiles` for `rank`: (first, count).  The first n_tiles % world ranks take one more."""
    if world <= 0 or not 0 <= rank < world or n_tiles < 0:
        raise ValueError((rank, world, n_tiles))
    base, extra = divmod(n_tiles, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def max_over_ranks(seconds, device="cpu"):
    """The slowest rank's time: what a whole-job throughput has to be computed from."""
    if _world() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_manifest(local_sizes, n_tiles, device="cpu"):
    """All ranks learn every tile's blob size.  local_sizes: this rank's sizes in tile order (its tile_range).
    Returns (sizes[n_tiles], offsets[n_tiles + 1]) as int64 CPU tensors: tile t lives at offsets[t] of the mosaic."""
    world = _world()
    rank = dist.get_rank() if world > 1 else 0
    first, count = tile_range(rank, world, n_tiles)
    if len(local_sizes) != count:
        raise ValueError(f"rank {rank} owns {count} tiles, got {len(local_sizes)} sizes")
    sizes = torch.zeros(n_tiles, dtype=torch.int64, device=device)
    sizes[first:first + count] = torch.as_tensor(list(local_sizes), dtype=torch.int64, device=device)
    if world > 1:
        dist.all_reduce(sizes, op=dist.ReduceOp.SUM)    # disjoint ranges: the sum is the concatenation
    sizes = sizes.cpu()
    offsets = torch.zeros(n_tiles + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(sizes, 0)
    return sizes, offsets


class RcclComm:
    """This process's own RCCL communicator over the ranks of the default process group, for lerc_amd_gather_blobs: the same
    librccl the C library opens (by soname: inside a PyTorch process the loader hands both the RCCL torch.distributed uses), and a
    HIP stream of the exchange's own."""

    class _UID(ct.Structure):
        _fields_ = [("internal", ct.c_char * 128)]    # ncclUniqueId (rccl.h:40-43)

    def __init__(self, device):
        self.lib = ct.CDLL(os.environ.get("LERC_AMD_RCCL") or "librccl.so.1")
        world = _world()
        rank = dist.get_rank() if world > 1 else 0
        uid = RcclComm._UID()
        if rank == 0:
            self.lib.ncclGetUniqueId.argtypes = [ct.POINTER(RcclComm._UID)]
            rc = self.lib.ncclGetUniqueId(ct.byref(uid))
            if rc != 0:
                raise RuntimeError(f"ncclGetUniqueId: {rc}")
        raw = [bytes(uid) if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(raw, src=0)
        ct.memmove(ct.byref(uid), raw[0], 128)
        self.comm = ct.c_void_p()
        self.lib.ncclCommInitRank.argtypes = [ct.POINTER(ct.c_void_p), ct.c_int, RcclComm._UID, ct.c_int]
        with torch.cuda.device(device):
            rc = self.lib.ncclCommInitRank(ct.byref(self.comm), world, uid, rank)
        if rc != 0 or not self.comm.value:
            raise RuntimeError(f"ncclCommInitRank: {rc}")
        self.world, self.rank, self.device = world, rank, device
        self.stream = torch.cuda.Stream(device)

    def close(self):
        if self.comm is not None and self.comm.value:
            self.lib.ncclCommDestroy.argtypes = [ct.c_void_p]
            self.lib.ncclCommDestroy(self.comm)
        self.comm = None


_RCCL = {}


def rccl_comm(device):
    """the module's communicator for `device` (made on first use; every rank has to get here together)"""
    key = (str(device), _world())
    if key not in _RCCL:
        _RCCL[key] = RcclComm(device)
    return _RCCL[key]


def _gather_c(arena, used, offsets, sizes, root, after, codec, front, root_capacity):
    """gather_arenas_start's exchange through lerc_amd_gather_blobs: a message a rank = [tiles, bytes | per-tile offsets | sizes | arena]"""
    from . import api
    dev = arena.device
    R = rccl_comm(dev)
    own = None
    if codec is None:
        own = codec = api.DeviceCodec(R.stream.cuda_stream)
    n_local = int(offsets.numel())
    need = 16 + 16 * n_local
    head = torch.empty(2 + 2 * n_local, dtype=torch.int64)
    head[0], head[1] = n_local, used
    head[2:2 + n_local] = offsets
    head[2 + n_local:] = sizes
    head = head.view(torch.uint8)
    if after is not None:
        if isinstance(after, torch.cuda.Event):
            R.stream.wait_event(after)
        else:
            R.stream.wait_stream(after)
    keep = [arena, head]
    with torch.cuda.stream(R.stream):
        gap = arena.data_ptr() - front.data_ptr() if front is not None else -1
        if front is not None and gap >= need and gap % 16 == 0 and front.numel() >= gap + used:
            # the table goes where the caller left room in front of the arena: the message is sent as it lies, no pass over the blobs
            front[gap - need:gap].copy_(head, non_blocking=True)
            msg_ptr = arena.data_ptr() - need
            keep.append(front)
        else:
            msg = torch.empty(need + used, dtype=torch.uint8, device=dev)
            msg[:need].copy_(head, non_blocking=True)
            msg[need:].copy_(arena[:used])
            msg_ptr = msg.data_ptr()
            keep.append(msg)
        rootbuf = None
        if R.rank == root:
            cap = int(root_capacity) if root_capacity else R.world * (need + int(arena.numel()) + 4096)
            rootbuf = torch.empty(cap, dtype=torch.uint8, device=dev)
        lens = (ct.c_ulonglong * R.world)()
        offs = (ct.c_ulonglong * (R.world + 1))()
        fn = codec.lib.lerc_amd_gather_blobs
        fn.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_ulonglong, ct.c_void_p, ct.c_ulonglong,
                       ct.POINTER(ct.c_ulonglong), ct.POINTER(ct.c_ulonglong), ct.c_void_p]
        fn.restype = ct.c_uint
        rc = fn(codec.h, R.comm, root, msg_ptr, need + used, rootbuf.data_ptr() if rootbuf is not None else None,
                rootbuf.numel() if rootbuf is not None else 0, lens, offs, R.stream.cuda_stream)
    if rc != 0:
        err = codec.last_error()
        if own is not None:
            own.close()
        raise RuntimeError(f"lerc_amd_gather_blobs: status {rc}: {err}")

    def build():
        R.stream.synchronize()
        keep.clear()
        if own is not None:
            own.close()
        if R.rank != root:
            return None, None, None, None
        total = int(offs[R.world])
        all_off, all_size, bases = [], [], []
        for r in range(R.world):
            at = int(offs[r])
            hd = rootbuf[at:at + 16].cpu().view(torch.int64)
            n_r, used_r = int(hd[0]), int(hd[1])
            if 16 + 16 * n_r + used_r != int(lens[r]):
                raise RuntimeError(f"rank {r}'s message does not add up: {n_r} tiles, {used_r} bytes, {int(lens[r])} received")
            tab = rootbuf[at + 16:at + 16 + 16 * n_r].cpu().view(torch.int64)
            base = at + 16 + 16 * n_r
            bases.append(base)
            all_off.append(tab[:n_r] + base)
            all_size.append(tab[n_r:])
        cat = (lambda v: torch.cat(v) if v else torch.zeros(0, dtype=torch.int64))
        return rootbuf[:total], cat(all_off), cat(all_size), bases
    return GatherInFlight(None, (), build)


class GatherInFlight:
    """The exchange step between its two halves: the transfers are enqueued (under RCCL they run on the collective's own
    stream, beside whatever the caller enqueues next -- a rank decodes its own tiles while its blobs travel), finish() waits
    for them and returns what gather_arenas returns."""

    def __init__(self, result=None, reqs=(), build=None):
        self._result, self._reqs, self._build = result, list(reqs), build

    def finish(self):
        for req in self._reqs:
            req.wait()
        self._reqs = []
        if self._build is not None:
            self._result, self._build = self._build(), None
        return self._result


def gather_arenas(arena, used, offsets, sizes, root=0, after=None, force_collective=False, **kw):
    """The exchange step of a mosaic job in one call: gather_arenas_start(...).finish()."""
    return gather_arenas_start(arena, used, offsets, sizes, root, after, force_collective, **kw).finish()


def gather_arenas_start(arena, used, offsets, sizes, root=0, after=None, force_collective=False, codec=None, front=None, root_capacity=0):
    """The exchange step of a mosaic job: the blobs of all ranks' tiles end up on `root`, in rank (= tile) order.

    arena    this rank's blob arena (uint8 tensor on the job's device: HBM under RCCL, host memory under gloo)
    used     bytes of it in use
    offsets  int64 / uint64 array-like [nLocalTiles]: where each local tile's blob starts in `arena`
    sizes    array-like [nLocalTiles]: its length

    after    the stream the codec wrote `arena` on (a torch.cuda.Stream), or an event recorded there behind the encode: the
             collective's stream is ordered behind it explicitly (RCCL's transfers run on a stream of their own, which torch
             orders behind the CURRENT stream only -- the codec's need not be that one)
    force_collective  run the collective steps in a process group of ONE rank too (tests: the lengths' all-gather on RCCL with
             device tensors, the root's own copy between HBM slices; there is nobody to send to)
    codec    (RCCL) the DeviceCodec whose library makes the exchange (lerc_amd_gather_blobs); None: one of the call's own
    front    (RCCL) a uint8 tensor whose memory `arena` lies in, with room in front of it (16 + 16 bytes a tile, a multiple of 16):
             the per-tile table is written there and the message leaves as it lies; without it the message is put together by a copy
    root_capacity  (RCCL) bytes of the root's buffer (the lengths are only known inside the call); 0: ranks x this rank's arena size

    offsets need not be monotonic (tiles a batch handed back to the general path sit behind the batch's own), a rank may
    hold no tile at all.

    finish() returns on root (mosaic, tile_offsets, tile_sizes, rank_bases): `mosaic` a uint8 tensor on the same device holding
    the ranks' arenas back to back (each starting at a multiple of 16 bytes, as inside an arena), tile_offsets /
    tile_sizes int64 CPU tensors over ALL tiles in tile order (offsets into `mosaic`), rank_bases where each rank's
    arena starts.  Other ranks get (None, None, None, None).  Works for a single process (no process group) too.
    """
    import numpy as np
    world = _world()
    rank = dist.get_rank() if world > 1 else 0
    dev = arena.device
    offsets = torch.as_tensor(np.asarray(offsets).astype(np.int64), dtype=torch.int64)
    sizes = torch.as_tensor(np.asarray(sizes).astype(np.int64), dtype=torch.int64)
    n_local = int(offsets.numel())
    used = int(used)
    grouped = dist.is_available() and dist.is_initialized()
    if world == 1 and not (force_collective and grouped):
        return GatherInFlight((arena[:used], offsets.clone(), sizes.clone(), [0]))
    if dev.type == "cuda" and dist.get_backend() == "nccl" and os.environ.get("LERC_AMD_GATHER", "c") != "torch":
        return _gather_c(arena, used, offsets, sizes, root, after, codec, front, root_capacity)
    if after is not None and dev.type == "cuda":
        cur = torch.cuda.current_stream(dev)
        if isinstance(after, torch.cuda.Event):
            cur.wait_event(after)
        else:
            cur.wait_stream(after)

    # 1. how much everybody has: [bytes in use, tiles] (on the device under RCCL; gloo moves host memory)
    tdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor([used, n_local], dtype=torch.int64, device=tdev)
    table = torch.empty(2 * world, dtype=torch.int64, device=tdev)
    dist.all_gather_into_tensor(table, mine)
    table = table.cpu().view(world, 2)    # (16 bytes a rank: the root cannot size its buffer, nor anybody post a transfer, without them)
    lens = [int(v) for v in table[:, 0]]
    tiles = [int(v) for v in table[:, 1]]
    bases, at = [], 0
    for n in lens:
        bases.append(at)
        at += (n + 15) & ~15
    total = at

    # 2. the arenas and the per-tile tables, rank -> root, one grouped batch of point-to-point transfers
    meta = torch.stack([offsets, sizes]).to(dev).contiguous() if n_local else torch.empty((2, 0), dtype=torch.int64, device=dev)
    ops = []
    if rank == root:
        mosaic = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
        metas = [torch.empty((2, tiles[r]), dtype=torch.int64, device=dev) for r in range(world)]
        for r in range(world):
            if r == root:
                continue
            if lens[r]:
                ops.append(dist.P2POp(dist.irecv, mosaic[bases[r]:bases[r] + lens[r]], r))
            if tiles[r]:
                ops.append(dist.P2POp(dist.irecv, metas[r], r))
        mosaic[bases[root]:bases[root] + used].copy_(arena[:used])
        metas[root] = meta
    else:
        if used:
            ops.append(dist.P2POp(dist.isend, arena[:used], root))
        if n_local:
            ops.append(dist.P2POp(dist.isend, meta, root))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if rank != root:
        keep = (arena, meta)    # (what is being sent stays alive until finish())
        return GatherInFlight((None, None, None, None), reqs, lambda: (keep and None, None, None, None))

    def build():
        all_off = torch.cat([metas[r][0].cpu() + bases[r] for r in range(world)]) if sum(tiles) else torch.zeros(0, dtype=torch.int64)
        all_size = torch.cat([metas[r][1].cpu() for r in range(world)]) if sum(tiles) else torch.zeros(0, dtype=torch.int64)
        return mosaic[:total], all_off, all_size, bases
    return GatherInFlight(None, reqs, build)
