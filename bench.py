#!/usr/bin/env python3
"""bench.py -- LERC encode + decode round trip on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N = 1 (default workload c2, BASELINE configs[1] -- the configuration the metric is quoted on): one step = lerc_encode +
lerc_decode of one 8192 x 8192 float32 raster (1 band, MaxZError 0.01) that is already resident in HBM, through the
device-pointer C ABI of liblerc_amd.so.

N > 1 (default workload c5, BASELINE configs[4]): a mosaic of 65 536 independent 256 x 256 float32 tiles sharded over the
N GPUs (lerc_amd/shard.py: contiguous tile ranges); one step = every rank encodes its tiles with one batched call, the
compressed blobs are GATHERED on rank 0 over RCCL (the one exchange step of the job: lengths all-gather + one grouped
send / receive batch, rank -> root over xGMI), and every rank decodes its own tiles again.  The total work is fixed
("scaling": "strong").  `--workload c2` keeps one C2 raster per rank instead (weak scaling, no data-path collective).

Rank 0 prints ONE JSON line.
  value        whole-job MPix/s = pixels of all ranks * K / (max-over-ranks time of K steps)
  roofline     dominant kernel of the step, timed live with HIP events inside the library on the stream the
               kernels run on: achieved = algorithmic bytes of that launch / its average duration
               (encode-side launches B_enc = raw + blob bytes, decode-side B_dec = blob + raw)
  cache_cold   the same K steps again over `--rotate` distinct rasters / blob buffers / outputs in rotation, so that
               neither the 256 MiB Infinity Cache nor an L2 holds a step's input when it starts (the plain loop
               re-encodes one raster into one buffer: "MALL-warm")
  cpu_baseline the reference CPU codec (oracle/_ref, else the oracle port) on the GPU box's host: c2 -- the same raster,
               1 core (the library is single threaded); c5 -- tiles on ALL host cores, one process per core
"""
import argparse
import ctypes as ct
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
XGMI_LINK_GBS = 153.0    # one xGMI link, one direction (the task's figure: 7 links x ~153 GB/s per GPU)
MOSAIC_TILES = 65536     # BASELINE configs[4]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=8192, help="raster edge (default: the BASELINE 8192)")
    ap.add_argument("--max-z-err", type=float, default=0.01)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the c3 / c4 sub-objects (BASELINE configs[2], configs[3])")
    ap.add_argument("--workload", choices=("auto", "c2", "c5"), default="auto",
                    help="auto (default): c2 on one GPU -- the BASELINE metric: one 8192^2 raster --, c5 on several: the 65 536-tile "
                         "mosaic sharded over the ranks with the RCCL gather of the blobs (BASELINE configs[4])")
    ap.add_argument("--tiles", type=int, default=0, help="tiles of the whole mosaic for --workload c5 (default: all 65536, on one GPU too -- 17 GB of pixels)")
    ap.add_argument("--no-c5-anchor", action="store_true",
                    help="default c2 line on one GPU: leave out `c5_1gpu`, the N = 1 point of the mosaic's strong-scaling curve")
    ap.add_argument("--rotate", type=int, default=3, help="buffer sets of the cache-cold pass (0: skip it)")
    ap.add_argument("--mode", choices=("async", "sync"), default="async",
                    help="c2: async (default) -- every step's encode and decode are ENQUEUED on the HIP stream (lerc_amd_*_device_async; the decode "
                         "reads the blob's size from its header on the device) and the host waits once, at the end of the K steps, the way any "
                         "stream of GPU work is driven; sync -- lerc_amd_encode_device / lerc_amd_decode_device, each waiting for its own result "
                         "(reported beside it as `sync_per_call` in async mode)")
    return ap.parse_args()


def cpu_baseline(raster_np, max_z_err):
    """Reference CPU codec on the GPU box's host: 1 thread (the library is single threaded)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import capi
    lib, kind = capi.ref(), "reference"
    if lib is None:
        lib, kind = capi.oracle(), "port"
    if lib is None:
        return None
    n_pix = raster_np.shape[0] * raster_np.shape[1]
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        rc, blob = lib.encode(raster_np, max_z_err)
        t1 = time.perf_counter()
        rc2, dec, _ = lib.decode(blob)
        t2 = time.perf_counter()
        assert rc == 0 and rc2 == 0
        if best is None or (t2 - t0) < best[0]:
            best = (t2 - t0, t1 - t0, t2 - t1)
    import hashlib
    return {
        "value": round(n_pix / best[0] / 1e6, 2), "unit": "MPix/s", "cores": 1, "kind": kind,
        "sample": f"{raster_np.shape[0]}x{raster_np.shape[1]} float32 full raster, best of 2 round trips "
                  f"(lerc_computeCompressedSize+lerc_encode {best[1]*1e3:.0f} ms, lerc_decode {best[2]*1e3:.0f} ms)",
        "blob_bytes": len(blob), "blob_sha256": hashlib.sha256(bytes(blob)).hexdigest(),
    }


def _cpu_tiles_worker(args):
    """One host core: round trips over the sample tiles until the time is up (no torch in here: numpy + the C library)."""
    path, max_z_err, seconds = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import capi
    lib = capi.ref() or capi.oracle()
    tiles = np.load(path, mmap_mode="r")
    tiles = [np.ascontiguousarray(tiles[t]) for t in range(tiles.shape[0])]
    done, t0 = 0, time.perf_counter()
    while True:
        for t in tiles:
            rc, blob = lib.encode(t, max_z_err)
            rc2, dec, _ = lib.decode(blob)
            assert rc == 0 and rc2 == 0
        done += len(tiles)
        if time.perf_counter() - t0 >= seconds:
            break
    return done, time.perf_counter() - t0


def cpu_baseline_tiles(tiles_np, max_z_err, seconds=8.0):
    """SURVEY 8(d) CPU baseline (ii): the reference codec on all host cores, one process per core, 256 x 256 tiles
    (every process works through the same 64 sample tiles of the mosaic, again and again, for `seconds`)."""
    import subprocess
    import tempfile
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import capi
    kind = "reference" if capi.ref() is not None else ("port" if capi.oracle() is not None else None)
    if kind is None:
        return None
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "tiles.npy")
        np.save(path, np.ascontiguousarray(tiles_np[:64]))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", path, str(max_z_err), str(seconds)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for _ in range(cores)]
        res = []
        for p in procs:
            try:
                out, _ = p.communicate(timeout=seconds * 6 + 60)
                done, el = out.decode().split()
                res.append((int(done), float(el)))
            except Exception:
                p.kill()
    if not res:
        return None
    tiles = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    return {"value": round(tiles * 65536 / wall / 1e6, 2), "unit": "MPix/s", "cores": len(res), "kind": kind,
            "sample": f"{tiles} round trips of 256x256 float32 tiles (64 sample tiles of the mosaic, repeated for {seconds:.0f} s), "
                      f"one process per core on {len(res)} cores, lerc_computeCompressedSize+lerc_encode+lerc_decode each"}


def other_config(torch, api, codec, name, x, max_z_err, n_depth, steps=5, warmup=2, reference=True, mask=None):
    """BASELINE configs[2] / configs[3] on this GPU, device resident: `steps` round trips (lerc_amd_encode_device +
    lerc_amd_decode_device, the host waits for each call), per-kernel HIP-event times, the blob compared with the reference's.
    mask: a uint8 validity mask on the device (the configuration of SURVEY.md section 8 "next" row 1)"""
    import hashlib
    L = codec.lib
    out = torch.empty(x.numel() * x.element_size() + (1 << 20), dtype=torch.uint8, device=x.device)
    dec = torch.empty_like(x)
    dec_mask = torch.empty_like(mask) if mask is not None else None
    n_pix = int(x.shape[0]) * int(x.shape[1])
    raw = x.numel() * x.element_size()
    dt = api._torch_dt_code(x)
    n_rows, n_cols = int(x.shape[0]), int(x.shape[1])
    enc_s = dec_s = 0.0

    def one():
        nonlocal enc_s, dec_s
        t0 = time.perf_counter()
        if mask is None:
            rc, nb = api.encode_device(codec, x, max_z_err, out, n_depth)
        else:
            rc, nb = codec.encode(x.data_ptr(), dt, n_depth, n_cols, n_rows, 1, max_z_err, out.data_ptr(), out.numel(), mask.data_ptr(), 1)
        t1 = time.perf_counter()
        if mask is None:
            rc2 = api.decode_device(codec, out, nb, dec, n_depth)
        else:
            rc2 = codec.decode(out.data_ptr(), nb, dt, n_depth, n_cols, n_rows, 1, dec.data_ptr(), dec_mask.data_ptr(), 1)
        t2 = time.perf_counter()
        enc_s += t1 - t0
        dec_s += t2 - t1
        if rc != 0 or rc2 != 0:
            raise RuntimeError(f"{name}: encode / decode failed: status {rc} / {rc2}: {codec.last_error()}")
        return nb

    for _ in range(warmup):
        nb = one()
    torch.cuda.synchronize()
    L.lerc_amd_profile_enable(codec.h, 1)
    enc_s = dec_s = 0.0
    forms0, refus0, paths0 = codec.decode_forms(), codec.decode_refusals(), codec.path_counters()
    t0 = time.perf_counter()
    for _ in range(steps):
        nb = one()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    L.lerc_amd_profile_enable(codec.h, 0)
    forms1, refus1, paths1 = codec.decode_forms(), codec.decode_refusals(), codec.path_counters()
    buf = ct.create_string_buffer(1 << 16)
    L.lerc_amd_profile_read(codec.h, buf, len(buf), 1)
    kern = {}
    for line in buf.value.decode().splitlines():
        k, ms, cnt = line.split()
        kern[k] = {"avg_ms": round(float(ms) / max(int(cnt), 1), 5), "launches": int(cnt)}
    ms = el / steps * 1e3
    kms = sum(v["avg_ms"] * v["launches"] for v in kern.values()) / steps
    b_rt = 2 * (raw + nb)
    same = bool(torch.equal(dec.view(torch.uint8), x.view(torch.uint8))) if (max_z_err == 0 and mask is None) else None
    if mask is not None:
        same = bool(torch.equal(dec_mask, mask)) and float(((dec.double() - x.double()).abs() * mask).max().item()) <= max_z_err + 6.2e-5
    res = {"workload": name, "value": round(n_pix * steps / el / 1e6, 2), "unit": "MPix/s", "steps": steps, "ms_per_step": round(ms, 4),
           "encode_ms": round(enc_s / steps * 1e3, 4), "decode_ms": round(dec_s / steps * 1e3, 4),
           "kernel_ms_per_step": round(kms, 4), "blob_bytes": int(nb), "compression_ratio": round(raw / max(nb, 1), 3),
           "algorithmic_bytes": b_rt, "frac_of_hbm_peak_wall": round(b_rt / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5),
           "frac_of_hbm_peak_kernels": round(b_rt / (max(kms, 1e-9) / 1e3) / 1e9 / HBM_PEAK_GBS, 5),
           "host": "every call waits for its own result", "lossless_round_trip": same, "kernels": kern,
           # which tier served the timed decodes, and what was thrown away on the way (a fall to a lower tier is a number here)
           "decode_forms": {"masked_scan": forms1[0] - forms0[0], "two_launches": forms1[1] - forms0[1], "walking": forms1[2] - forms0[2],
                            "scanning": forms1[3] - forms0[3]},
           "decode_refusals": {"masked_offsets_refused": refus1[0] - refus0[0], "masked_scan_handed_on": refus1[1] - refus0[1],
                               "tier_handed_on": refus1[2] - refus0[2]},
           "path_counters": {"encode_streaming": paths1[0] - paths0[0], "encode_general": paths1[1] - paths0[1],
                             "decode_streaming": paths1[2] - paths0[2], "decode_general": paths1[3] - paths0[3]}}
    sha = hashlib.sha256(out[:nb].cpu().numpy().tobytes()).hexdigest()
    res["blob_sha256"] = sha
    if reference:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import capi
        lib = capi.ref() or capi.oracle()
        if lib is not None:
            xn = x.cpu().numpy()
            t0 = time.perf_counter()
            kw = {"mask": mask.cpu().numpy()} if mask is not None else {}
            rc, blob = lib.encode(xn, max_z_err, n_depth=n_depth, **kw)
            t1 = time.perf_counter()
            res["blob_matches_reference"] = bool(rc == 0 and len(blob) == nb and hashlib.sha256(bytes(blob)).hexdigest() == sha)
            res["reference_encode_s"] = round(t1 - t0, 3)
    res["verified"] = bool((same is not False) and res.get("blob_matches_reference", True))
    # The same round trips QUEUED, like the headline's: encode and decode enqueued back to back on the stream (the decode is given
    # the buffer's capacity and reads the blob's size from its header on the device), one wait behind all of them -- for the
    # configurations the streaming kernels take blind (one value a pixel, no mask, 16-bit and wider types).
    if mask is None and n_depth == 1 and x.element_size() >= 2:
        def pair():
            rc, t1 = api.encode_device_async(codec, x, max_z_err, out)
            rc2, t2 = api.decode_device_async(codec, out, out.numel(), dec)
            if rc != 0 or rc2 != 0:
                raise RuntimeError(f"{name}: enqueue failed: status {rc} / {rc2}")
            return t1, t2
        for _ in range(warmup):
            t1, t2 = pair()
        codec.finish(t1); codec.finish(t2)
        torch.cuda.synchronize()
        dec.zero_()
        L.lerc_amd_profile_enable(codec.h, 1)
        refq0 = codec.decode_refusals()
        tq0 = time.perf_counter()
        tickets = [pair() for _ in range(steps)]
        bad = [codec.finish(t)[0] for pr in tickets for t in pr]
        torch.cuda.synchronize()
        elq = time.perf_counter() - tq0
        L.lerc_amd_profile_enable(codec.h, 0)
        L.lerc_amd_profile_read(codec.h, buf, len(buf), 1)
        kq = {}
        for line in buf.value.decode().splitlines():
            k, ms_, cnt = line.split()
            kq[k] = {"avg_ms": round(float(ms_) / max(int(cnt), 1), 5), "launches": int(cnt)}
        msq = elq / steps * 1e3
        okq = not any(bad) and (bool(torch.equal(dec.view(torch.uint8), x.view(torch.uint8))) if max_z_err == 0 else True)
        res["queued"] = {"value": round(n_pix * steps / elq / 1e6, 2), "ms_per_step": round(msq, 4),
                         "frac_of_hbm_peak_wall": round(b_rt / (msq / 1e3) / 1e9 / HBM_PEAK_GBS, 5), "kernels": kq, "verified": bool(okq),
                         # (a decode whose streaming form hands the band on is repeated at finish time, and so is everything enqueued behind it)
                         "tier_handed_on": codec.decode_refusals()[2] - refq0[2], "last_note": codec.last_note(),
                         "host": "steps enqueued on the stream, one wait at the end (lerc_amd_encode_device_async / lerc_amd_decode_device_async)"}
    if mask is not None:
        res["mask_and_error_bound_hold"] = res.pop("lossless_round_trip")
    del out, dec
    return res


def csrc_digest():
    """sha256 over the kernel and host sources of lerc_amd/csrc (what a committed HBM-traffic file was measured on)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "lerc_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".cpp", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def traffic_files():
    """the committed HBM-traffic files, newest first -- only those measured on THESE sources (a file that carries no digest,
    or another one, is stale: the kernels changed since the PMC passes)"""
    import glob
    good, stale = [], []
    now = csrc_digest()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        (good if d.get("csrc_digest") == now else stale).append(path)
    return good, stale


def measured_traffic(kernel_group, size):
    """HBM bytes per launch of a kernel group from the committed PMC passes (profiles/*hbm_traffic.json, written
    by tools/profile_run.sh: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, plus WRITE_SIZE).
    Counters cannot be read from inside the timed process, so this is the figure of the last profiled build of
    the same workload -- None when no such file or kernel group exists."""
    import glob
    for path in traffic_files()[0]:
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("size") == size and kernel_group in t.get("bytes_per_launch", {}):
                return t["bytes_per_launch"][kernel_group]
        except (OSError, ValueError):
            pass
    return None


def measured_traffic_table(size):
    """All kernel groups of the newest committed traffic file for this raster size (see measured_traffic), or None."""
    import glob
    for path in traffic_files()[0]:
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("size") == size and t.get("bytes_per_launch"):
                return os.path.basename(path), t["bytes_per_launch"]
        except (OSError, ValueError):
            pass
    return None


def measured_ceiling(torch, x, y, reps=10):
    """What this box's HBM delivers to the plainest streaming kernel there is, in the same process: a device-to-device copy
    of the raster (hipMemcpyDtoD: every byte read once and written once), timed with events on the stream, best of `reps`."""
    best = None
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        y.copy_(x)
        b.record()
        b.synchronize()
        ms = a.elapsed_time(b)
        best = ms if best is None or ms < best else best
    nbytes = 2 * x.numel() * x.element_size()
    return {"kind": "hipMemcpyDtoD of the raster (bytes read + bytes written), best of %d" % reps, "bytes": nbytes,
            "ms": round(best, 5), "GBps": round(nbytes / (best / 1e3) / 1e9, 1)}


TILE_SLOT_BYTES = (256 * 256 * 4 // 2 + 4096 + 15) // 16 * 16    # a buffer per 256 x 256 float32 tile, as a caller of lerc_encode() would size it for lossy floats


def c5_single_gpu(torch, api, synth, codec, dev, max_z_err, total_tiles, steps=3, warmup=1):
    """All tiles of the mosaic on ONE GPU (the N = 1 anchor of `bench.py --gpus N`'s strong-scaling curve): batched encode +
    batched decode per step, in slabs of 8192 tiles (what one rank of eight holds) so that the buffers stay at 2 GB each."""
    slab = 8192
    n_slabs = (total_tiles + slab - 1) // slab
    xs = []
    for k in range(n_slabs):
        count = min(slab, total_tiles - k * slab)
        rows_of_tiles = (count + 255) // 256
        big = synth.c2_float32(256 * rows_of_tiles, 65536, row0=256 * (k * (slab // 256)), col0=0, virt_cols=65536, device=dev)
        xs.append(big.reshape(rows_of_tiles, 256, 256, 256).permute(0, 2, 1, 3).contiguous().reshape(rows_of_tiles * 256, 256, 256)[:count].contiguous())
        del big
    out = torch.empty(xs[0].numel() * 4 + slab * 256, dtype=torch.uint8, device=dev)
    y = torch.empty_like(xs[0])
    slot_bytes = TILE_SLOT_BYTES
    n_pix = total_tiles * 65536

    def one_pass(slots):
        nbytes = 0
        for x in xs:
            if slots:
                rc, sizes = api.encode_tiles_device_slots(codec, x, max_z_err, out, slot_bytes)
            else:
                rc, offs, sizes, used = api.encode_tiles_device(codec, x, max_z_err, out)
            if rc != 0:
                raise RuntimeError(f"tile encode failed: status {rc}: {codec.last_error()}")
            nbytes += int(sizes.sum())
            if slots:
                rc = api.decode_tiles_device_slots(codec, out, slot_bytes, sizes, y[:x.shape[0]])
            else:
                rc = api.decode_tiles_device(codec, out, offs, sizes, y[:x.shape[0]])
            if rc != 0:
                raise RuntimeError(f"tile decode failed: status {rc}: {codec.last_error()}")
        return nbytes

    def timed(slots):
        for _ in range(warmup):
            one_pass(slots)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            nbytes = one_pass(slots)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
        err = float((y[:xs[-1].shape[0]].double() - xs[-1].double()).abs().max().item())
        b_rt = 2 * (n_pix * 4 + nbytes)
        return {"value": round(n_pix / el / 1e6, 2), "unit": "MPix/s", "ms_per_step": round(el * 1e3, 3), "steps": steps,
                "blob_bytes": nbytes, "frac_of_hbm_peak_wall": round(b_rt / el / 1e9 / HBM_PEAK_GBS, 5), "max_abs_error": err,
                "verified": bool(err <= max_z_err * (1 + 1e-6) + 6.2e-5)}

    res = {"tiles": total_tiles}
    res.update(timed(True))
    res["note"] = ("the whole 65 536-tile mosaic on one GPU, %d batched calls of %d tiles each way per step, a buffer of %d bytes per tile "
                   "(lerc_amd_encode_tiles_device_slots: the encode kernel writes every blob where it stays) -- what `--workload c5` "
                   "runs on one GPU; no exchange step" % (n_slabs, slab, slot_bytes))
    res["packed"] = timed(False)
    res["packed"]["note"] = ("the same with the blobs packed into one arena (lerc_amd_encode_tiles_device: one more pass over the blobs) -- "
                             "what every rank of a multi-GPU job does before the gather")
    return res


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: lerc_amd has no CPU path")
    # (LERC_BENCH_ONE_DEVICE=1: a dry run of the N > 1 code path on a one-GPU box -- all ranks on device 0, gloo instead of RCCL)
    one_device = os.environ.get("LERC_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)    # "nccl" is RCCL on ROCm

    from lerc_amd import api, shard, synth

    workload = args.workload if args.workload != "auto" else ("c2" if world == 1 else "c5")
    tiles_mode = workload == "c5"
    n = args.size
    stream = torch.cuda.current_stream().cuda_stream
    codec = api.DeviceCodec(stream)

    def make_set(k):
        """Input, blob buffer and output of one buffer set (set 0 is the plain loop's)."""
        if tiles_mode:
            total = args.tiles or MOSAIC_TILES
            first, count = shard.tile_range(rank, world, total)
            # this rank's tiles: rows of 256 tiles of the 65536-wide virtual raster, cut into 256 x 256 tiles
            rows_of_tiles = (count + 255) // 256
            r0 = first // 256 + k * 1024            # (another set: another part of the virtual raster)
            big = synth.c2_float32(256 * rows_of_tiles, 65536, row0=256 * r0, col0=0, virt_cols=65536, device=dev)
            x = big.reshape(rows_of_tiles, 256, 256, 256).permute(0, 2, 1, 3).contiguous().reshape(rows_of_tiles * 256, 256, 256)[:count].contiguous()
            del big
            # (room for the rank's table in front of the arena: its message to the root leaves as it lies, lerc_amd/shard.py)
            room = (16 + 16 * count + 255) & ~255
            front = torch.empty(room + x.numel() * 4 + count * 256, dtype=torch.uint8, device=dev)
            out = front[room:]
            fronts[k] = front
        else:
            # every rank compresses its own window of one large virtual raster (independent blobs)
            x = synth.c2_float32(n, n, row0=k * n, col0=rank * n, virt_cols=max(world, 1) * n, device=dev)
            out = torch.empty(n * n * 4 + (1 << 20), dtype=torch.uint8, device=dev)
        return x, out, torch.empty_like(x)

    fronts = {}
    sets = [make_set(0)]
    n_pix = sets[0][0].numel()
    torch.cuda.synchronize()

    state = {"blob_bytes": 0, "gather_s": 0.0, "gather_bytes": 0, "gather_steps": 0, "tickets": [],
             "async": args.mode == "async" and not tiles_mode}

    def step(k=0):
        x, out, y = sets[k]
        if tiles_mode and world == 1:
            # one GPU, no exchange step: every tile's blob stays in a slot of its own (a buffer per tile, as with lerc_encode());
            # the ranks of a multi-GPU job pack theirs into an arena, which is what the gather sends
            rc, sizes = api.encode_tiles_device_slots(codec, x, args.max_z_err, out, TILE_SLOT_BYTES)
            if rc != 0:
                raise RuntimeError(f"tile encode failed: status {rc}: {codec.last_error()}")
            state["blob_bytes"] = int(sizes.sum())
            rc = api.decode_tiles_device_slots(codec, out, TILE_SLOT_BYTES, sizes, y)
            if rc != 0:
                raise RuntimeError(f"tile decode failed: status {rc}: {codec.last_error()}")
            return
        if tiles_mode:
            te0 = time.perf_counter()
            rc, offs, sizes, used = api.encode_tiles_device(codec, x, args.max_z_err, out)
            if rc != 0:
                raise RuntimeError(f"tile encode failed: status {rc}: {codec.last_error()}")
            te1 = time.perf_counter()
            state["blob_bytes"] = int(sizes.sum())
            flight = None
            if world > 1:
                # the exchange step: all ranks' blobs on rank 0.  The transfers are enqueued here and run beside what this rank
                # does next -- the decode of its own tiles does not hang on them (timed inside the step: start to arrival, and
                # what of it was still to wait for behind the decode)
                flight = shard.gather_arenas_start(out, used, offs, sizes, root=0, after=torch.cuda.current_stream(), codec=codec, front=fronts.get(k))    # (after: the codec's stream)
            tg0 = time.perf_counter()
            rc = api.decode_tiles_device(codec, out, offs, sizes, y)
            if rc != 0:
                raise RuntimeError(f"tile decode failed: status {rc}: {codec.last_error()}")
            td1 = time.perf_counter()
            if flight is not None:
                mosaic, t_off, t_size, _ = flight.finish()
                torch.cuda.synchronize()
                tg1 = time.perf_counter()
                state["gather_s"] += tg1 - te1
                state["gather_wait_s"] = state.get("gather_wait_s", 0.0) + (tg1 - td1)
                state["gather_steps"] += 1
                if rank == 0:
                    state["gather_bytes"] = int(mosaic.numel()) - int(used)    # what arrived over the links
                    state["mosaic_tiles"] = int(t_off.numel())
                del mosaic
            state["encode_s"] = state.get("encode_s", 0.0) + (te1 - te0)
            state["decode_s"] = state.get("decode_s", 0.0) + (td1 - tg0)
            return
        if state["async"]:
            rc, t1 = api.encode_device_async(codec, x, args.max_z_err, out)
            rc2, t2 = api.decode_device_async(codec, out, out.numel(), y)
            if rc != 0 or rc2 != 0:
                raise RuntimeError(f"enqueue failed: status {rc} / {rc2}: {codec.last_error()}")
            state["tickets"].append((t1, t2))
            if len(state["tickets"]) >= 24:
                drain()
            return
        rc, nb = api.encode_device(codec, x, args.max_z_err, out)
        if rc != 0:
            raise RuntimeError(f"encode failed: status {rc}: {codec.last_error()}")
        state["blob_bytes"] = nb
        rc = api.decode_device(codec, out, nb, y)
        if rc != 0:
            raise RuntimeError(f"decode failed: status {rc}: {codec.last_error()}")

    def drain():
        """async mode: wait for what is in flight and look at every operation's status"""
        for t1, t2 in state["tickets"]:
            rc, nb = codec.finish(t1)
            rc2, _ = codec.finish(t2)
            if rc != 0 or rc2 != 0:
                raise RuntimeError(f"encode / decode failed: status {rc} / {rc2}: {codec.last_error()}")
            state["blob_bytes"] = nb
        state["tickets"] = []

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    codec.lib.lerc_amd_profile_enable.argtypes = [ct.c_void_p, ct.c_int]
    codec.lib.lerc_amd_profile_read.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_int, ct.c_int]

    def timed(n_sets):
        """W warm-up steps, then exactly K timed steps between barriers; per-kernel HIP-event times of the timed steps."""
        for i in range(args.warmup):
            step(i % n_sets)
        drain()
        state["gather_s"], state["gather_steps"] = 0.0, 0
        state["gather_wait_s"], state["encode_s"], state["decode_s"] = 0.0, 0.0, 0.0
        barrier()
        codec.lib.lerc_amd_profile_enable(codec.h, 1)
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i % n_sets)
        barrier()
        elapsed = time.perf_counter() - t0
        drain()    # (the stream is idle: this only reads the verdicts)
        codec.lib.lerc_amd_profile_enable(codec.h, 0)
        buf = ct.create_string_buffer(1 << 16)
        codec.lib.lerc_amd_profile_read(codec.h, buf, len(buf), 1)
        prof = {}
        for line in buf.value.decode().splitlines():
            name, ms, cnt = line.split()
            prof[name] = (float(ms), int(cnt))
        return shard.max_over_ranks(elapsed, device=dev), prof

    elapsed, prof = timed(1)
    gather = dict(state)
    per_rank = None
    if tiles_mode and world > 1:
        # every rank's share of a step, so that the first run on hardware can be read from one line
        mine = torch.tensor([state.get("encode_s", 0.0), state.get("decode_s", 0.0), state["gather_s"], state.get("gather_wait_s", 0.0),
                             float(state["blob_bytes"]), float(sets[0][0].shape[0])], dtype=torch.float64, device=dev if not one_device else "cpu")
        table = torch.empty(6 * world, dtype=torch.float64, device=mine.device)
        dist.all_gather_into_tensor(table, mine)
        table = table.cpu().view(world, 6)
        k = max(args.steps, 1)
        per_rank = [{"rank": r, "tiles": int(table[r, 5]), "blob_bytes": int(table[r, 4]), "encode_ms": round(float(table[r, 0]) / k * 1e3, 4),
                     "decode_ms": round(float(table[r, 1]) / k * 1e3, 4), "gather_ms_start_to_arrival": round(float(table[r, 2]) / k * 1e3, 4),
                     "gather_ms_waited_behind_decode": round(float(table[r, 3]) / k * 1e3, 4)} for r in range(world)]
    timed_blob_bytes = int(state["blob_bytes"])    # of the TIMED pass (buffer set 0); later passes must not overwrite it
    # the blob the timed pass left in set 0's buffer (its last step's), before anything else writes there
    import hashlib
    timed_blob_sha = None
    if not tiles_mode:
        timed_blob_sha = hashlib.sha256(sets[0][1][:timed_blob_bytes].cpu().numpy().tobytes()).hexdigest()

    # correctness of what was timed (outside the timed region)
    x0, _, y0 = sets[0]
    err = float((y0.double() - x0.double()).abs().max().item())
    verified = err <= args.max_z_err * (1 + 1e-6) + 6.2e-5    # + 1/2 ulp of an f32 near 1000

    # the same steps once more with an event behind every step (outside the timed region: an event record between kernels is
    # not free): the spread of the single steps -- median and fastest next to the timed region's mean
    step_ms = None
    if not tiles_mode:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        evs[0].record()
        for i in range(args.steps):
            step(0)
            evs[i + 1].record()
        torch.cuda.synchronize()
        drain()
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
        step_ms = {"median": round(per[len(per) // 2], 4), "min": round(per[0], 4), "max": round(per[-1], 4),
                   "note": "one event per step on the stream, measured in a pass of its own behind the timed region"}

    sync_run = None
    if state["async"]:
        state["async"] = False
        sync_run = timed(1)
        state["async"] = True
    ceiling = measured_ceiling(torch, x0, torch.empty_like(x0)) if not tiles_mode else None

    cold = None
    if args.rotate >= 2:
        try:
            while len(sets) < args.rotate:
                sets.append(make_set(len(sets)))
            torch.cuda.synchronize()
            cold_elapsed, cold_prof = timed(len(sets))
            cold = (cold_elapsed, cold_prof, len(sets))
        except torch.OutOfMemoryError:
            cold = None

    # the N = 1 point of the mosaic's strong-scaling curve (BASELINE configs[4]: all 65 536 tiles on this one GPU), so that a
    # driver record holds it next to the c2 line
    c5_anchor = None
    if world == 1 and not tiles_mode and not args.no_c5_anchor and n == 8192:
        try:
            del sets[1:]
            torch.cuda.empty_cache()
            c5_anchor = c5_single_gpu(torch, api, synth, codec, dev, args.max_z_err, MOSAIC_TILES)
        except (torch.OutOfMemoryError, RuntimeError) as e:
            c5_anchor = {"error": str(e)[:200]}

    # BASELINE configs[2] and configs[3] beside it (parity-test cases with their own timing: not the line's value)
    others = None
    if world == 1 and not tiles_mode and not args.no_other_configs and n == 8192:
        others = {}
        del sets[1:]
        torch.cuda.empty_cache()
        for key, name, make, depth in (("c3", "16384x16384 uint16 DEM, MaxZError=0 (lossless bit-stuff path)", synth.c3_uint16, 1),
                                       ("c4", "4096x4096 nDepth=3 byte RGB, MaxZError=0 (8-bit Huffman path)", synth.c4_rgb_u8, 3)):
            try:
                xo = make(device=dev)
                others[key] = other_config(torch, api, codec, name, xo, 0, depth, reference=not args.no_cpu_baseline)
                del xo
                torch.cuda.empty_cache()
            except Exception as e:    # noqa: BLE001 -- a sub-object must not take the line with it
                others[key] = {"error": repr(e)[:200]}
        # ... and the C2 raster with a validity mask (SURVEY.md section 8 "next" row 1): rectangles of invalid pixels, 10 % of the raster
        try:
            xo = synth.c2_float32(n, n, device=dev)
            ii = torch.arange(n, device=dev).view(-1, 1)
            jj = torch.arange(n, device=dev).view(1, -1)
            mk = (((ii // 97) + (jj // 131)) % 10 != 0).to(torch.uint8).contiguous()
            others["c2_masked"] = other_config(torch, api, codec, "8192x8192 float32 DEM, MaxZError=0.01, 10 % of the pixels invalid (general path)",
                                               xo, args.max_z_err, 1, reference=not args.no_cpu_baseline, mask=mk)
            del xo, mk
            torch.cuda.empty_cache()
        except Exception as e:    # noqa: BLE001
            others["c2_masked"] = {"error": repr(e)[:200]}
        # ... and two rasters shaped like the ones the reference was benchmarked on (none of them has sides that are
        # multiples of 8, DEMs have lakes and sea): a ragged one, 8190 x 8190, and the C2 raster with 15 % of its area flat (a few
        # rectangles of one value: runs of constant blocks).  decode_forms says which tier served them.
        try:
            xo = synth.c2_float32(8190, 8190, device=dev)
            others["c2_ragged"] = other_config(torch, api, codec, "8190x8190 float32 DEM, MaxZError=0.01 (rows / columns no multiples of 8: edge blocks)",
                                               xo, args.max_z_err, 1, steps=8, reference=not args.no_cpu_baseline)
            del xo
            torch.cuda.empty_cache()
        except Exception as e:    # noqa: BLE001
            others["c2_ragged"] = {"error": repr(e)[:200]}
        try:
            xo = synth.c2_float32(n, n, device=dev)
            for (r0, r1, c0, c1, val) in ((512, 2560, 1024, 3584, 1017.25), (3000, 5048, 4096, 7168, 733.5), (6000, 7024, 256, 2304, 1500.0),
                                          (5120, 5632, 0, 2048, 0.0)):
                xo[r0:r1, c0:c1] = val          # 5.2 + 6.3 + 2.1 + 1.0 M pixels of 67.1 M: 15 %
            others["c2_flat"] = other_config(torch, api, codec, "8192x8192 float32 DEM, MaxZError=0.01, 15 % of the area flat (rectangles of one value: runs of constant blocks)",
                                             xo, args.max_z_err, 1, steps=8, reference=not args.no_cpu_baseline)
            del xo
            torch.cuda.empty_cache()
        except Exception as e:    # noqa: BLE001
            others["c2_flat"] = {"error": repr(e)[:200]}

    if rank == 0:
        blob_bytes = timed_blob_bytes
        raw_bytes = n_pix * 4
        b_enc = raw_bytes + blob_bytes
        b_dec = blob_bytes + raw_bytes
        # an encode-side launch is priced at B_enc = raw + blob bytes, a decode-side launch at
        # B_dec = blob + raw bytes, whatever part of them that launch really touches (extra passes only lower frac)
        dec_side = ("decode", "discover", "resolve", "gather", "walk", "fletcher_dec", "huff_dec")
        alg = {k: (b_dec if any(t in k for t in dec_side) else b_enc) for k in prof}

        def table(p):
            return {k: {"avg_ms": round(v[0] / max(v[1], 1), 5), "launches": v[1]} for k, v in p.items()}

        dom = max(prof.items(), key=lambda kv: kv[1][0])[0] if prof else None
        roofline = None
        if dom:
            avg_s = prof[dom][0] / max(prof[dom][1], 1) / 1e3
            ach = alg[dom] / avg_s / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": measured_traffic(dom, n) if not tiles_mode else None,
                        "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": round(avg_s * 1e3, 5)}
            if ceiling:
                roofline["measured_ceiling"] = ceiling
                roofline["frac_of_measured_ceiling"] = round(ach / ceiling["GBps"], 5)
            if roofline["traffic"] is None and not tiles_mode:
                stale = traffic_files()[1]
                roofline["traffic_note"] = ("no committed PMC passes of these kernel sources (lerc_amd/csrc digest " + csrc_digest()
                                            + (("; stale: " + ", ".join(os.path.basename(q) for q in stale[:2])) if stale else "") + ")")
            tt = measured_traffic_table(n) if not tiles_mode else None
            if tt and all(k in tt[1] for k in prof):
                step_traffic = sum(tt[1][k] * prof[k][1] for k in prof) / max(args.steps, 1)
                roofline["step_traffic"] = {"bytes": int(step_traffic), "algorithmic_bytes": b_enc + b_dec,
                                            "ratio": round(step_traffic / (b_enc + b_dec), 4), "source": "profiles/" + tt[0],
                                            "note": "HBM bytes all launches of one step move (PMC passes of the last profiled build) "
                                                    "over the step's algorithmic bytes B_enc + B_dec"}
        ms_per_step = elapsed / args.steps * 1e3
        kernel_ms = sum(v[0] for v in prof.values()) / max(args.steps, 1)

        def roundtrip(ms, kms):
            return {"algorithmic_bytes": b_enc + b_dec, "ms_per_step": round(ms, 4), "kernel_ms_per_step": round(kms, 4),
                    "frac_of_hbm_peak_wall": round((b_enc + b_dec) / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5),
                    "frac_of_hbm_peak_kernels": round((b_enc + b_dec) / (max(kms, 1e-9) / 1e3) / 1e9 / HBM_PEAK_GBS, 5)}

        if tiles_mode:
            total_tiles = state.get("mosaic_tiles", sets[0][0].shape[0])
            metric = (f"MPix/s encode+decode round-trip, mosaic of {total_tiles} independent 256^2 float32 tiles MaxZError=0.01"
                      + (f" sharded across {world} GPUs, RCCL gather of the blobs" if world > 1 else " (batched calls, 1 GPU)"))
            wl = (f"{total_tiles} tiles of 256x256 float32, MaxZError={args.max_z_err}: {sets[0][0].shape[0]} per rank, one batched encode call, "
                  + ("gather of the compressed blobs on rank 0 (lengths all-gather + grouped send/recv over RCCL), " if world > 1 else "")
                  + "one batched decode call per rank"
                  + ("" if world > 1 else f"; every tile's blob in a slot of {TILE_SLOT_BYTES} bytes (a buffer per tile; the ranks of a multi-GPU job "
                                          "pack theirs into one arena for the gather)"))
        else:
            metric = "MPix/s encode+decode round-trip, 8192^2 float32 MaxZError=0.01"
            wl = (f"{n}x{n} float32 1-band, MaxZError={args.max_z_err}, lerc encode+decode on HBM-resident data"
                  + (", one raster per rank (independent blobs, no data-path collective)" if world > 1 else ""))
        res = {
            "metric": metric,
            "value": round(world * n_pix * args.steps / elapsed / 1e6, 2),
            "unit": "MPix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong" if (tiles_mode and world > 1) else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl, "blob_bytes": blob_bytes, "compression_ratio": round(raw_bytes / max(blob_bytes, 1), 3),
                       "max_abs_error": err, "verified": bool(verified), "cache_state": "MALL-warm (one buffer set re-used every step)"},
            "roofline": roofline,
            "roundtrip": dict(roundtrip(ms_per_step, kernel_ms), **({"step_ms": step_ms} if step_ms else {})),
            "kernels": table(prof),
        }
        if tiles_mode and world > 1 and gather["gather_steps"]:
            g_s = gather["gather_s"] / gather["gather_steps"]
            gbps = gather["gather_bytes"] / max(g_s, 1e-9) / 1e9
            res["gather"] = {"bytes_into_root": gather["gather_bytes"], "ms_per_step": round(g_s * 1e3, 4), "GBps": round(gbps, 2),
                             "links": world - 1, "frac_of_xgmi": round(gbps / ((world - 1) * XGMI_LINK_GBS), 4),
                             "peak": f"{world - 1} links x {XGMI_LINK_GBS} GB/s into the root",
                             "ms_waited_behind_decode": round(gather.get("gather_wait_s", 0.0) / gather["gather_steps"] * 1e3, 4),
                             "note": "the transfers are enqueued behind the batched encode and run beside the rank's own batched decode; "
                                     "ms_per_step is start to arrival on rank 0, ms_waited_behind_decode what was left to wait for"}
        if per_rank is not None:
            res["per_rank"] = per_rank
        res["config"]["host"] = ("K steps enqueued on the stream, one wait at the end (lerc_amd_encode_device_async / lerc_amd_decode_device_async)"
                                 if state["async"] else "every call waits for its own result")
        if sync_run is not None:
            s_ms = sync_run[0] / args.steps * 1e3
            s_kms = sum(v[0] for v in sync_run[1].values()) / max(args.steps, 1)
            res["sync_per_call"] = {"value": round(world * n_pix * args.steps / sync_run[0] / 1e6, 2), "roundtrip": roundtrip(s_ms, s_kms),
                                    "note": "lerc_amd_encode_device + lerc_amd_decode_device, the host waits for each call's result"}
        if cold is not None:
            c_ms = cold[0] / args.steps * 1e3
            c_kms = sum(v[0] for v in cold[1].values()) / max(args.steps, 1)
            cdom = max(cold[1].items(), key=lambda kv: kv[1][0])[0] if cold[1] else None
            cc = {"buffer_sets": cold[2], "value": round(world * n_pix * args.steps / cold[0] / 1e6, 2), "roundtrip": roundtrip(c_ms, c_kms),
                  "kernels": table(cold[1]),
                  "note": "inputs, blob buffers and outputs rotate over distinct allocations, so a step's input is in neither the "
                          "Infinity Cache nor an L2 when the step starts"}
            if cdom:
                avg_s = cold[1][cdom][0] / max(cold[1][cdom][1], 1) / 1e3
                cc["roofline"] = {"kernel": cdom, "achieved": round(alg.get(cdom, b_enc) / avg_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(alg.get(cdom, b_enc) / avg_s / 1e9 / HBM_PEAK_GBS, 5)}
            res["cache_cold"] = cc
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_tiles(x0[:64].cpu().numpy(), args.max_z_err) if tiles_mode else cpu_baseline(x0.cpu().numpy(), args.max_z_err)
            cb = res["cpu_baseline"]
            if cb and not tiles_mode and cb.get("blob_sha256"):
                # the blob the TIMED (asynchronous) pass wrote against the reference's blob of the same raster: size and bytes
                same = cb["blob_bytes"] == blob_bytes and cb["blob_sha256"] == timed_blob_sha
                res["config"]["blob_matches_reference"] = bool(same)
                res["config"]["blob_sha256"] = timed_blob_sha
                if not same:
                    res["config"]["verified"] = False
        if c5_anchor is not None:
            res["c5_1gpu"] = c5_anchor
        if others:
            res.update(others)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) == 5 and sys.argv[1] == "--cpu-worker":    # one process of cpu_baseline_tiles
        print(*_cpu_tiles_worker((sys.argv[2], float(sys.argv[3]), float(sys.argv[4]))))
    else:
        main()
