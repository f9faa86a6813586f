#!/usr/bin/env python3
"""bench.py -- LERC encode + decode round trip on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = lerc_encode + lerc_decode of one 8192 x 8192 float32 raster (1 band, MaxZError 0.01, BASELINE
configs[1]) that is already resident in HBM, through the device-pointer C ABI of liblerc_amd.so.  With
N > 1 every rank owns one such raster (weak scaling: a band blob is one sequential block stream, so rasters /
tiles shard as independent blobs and the data path has no exchange step); the only collectives are
the barrier and the max-over-ranks of the elapsed time.  Rank 0 prints ONE JSON line.

  value        whole-job MPix/s = N * nPix * K / (max-over-ranks time of K steps)
  roofline     dominant kernel of the step, timed live with HIP events inside the library on the stream the
               kernels run on: achieved = algorithmic bytes of that launch / its average duration
               (encode-side launches B_enc = raw + blob bytes, decode-side B_dec = blob + raw)
  cpu_baseline the reference CPU codec (oracle/_ref, else the oracle port) on the same raster, 1 host core
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=8192, help="raster edge (default: the BASELINE 8192)")
    ap.add_argument("--max-z-err", type=float, default=0.01)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=("c2", "c5"), default="c2",
                    help="c2 (default, the BASELINE metric): one 8192^2 raster per rank; c5: a mosaic of 256^2 tiles per rank, one "
                         "batched call each way (BASELINE configs[4] in miniature, reported for DESIGN.md, not the headline)")
    ap.add_argument("--tiles", type=int, default=1024, help="tiles per rank for --workload c5")
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
    return {
        "value": round(n_pix / best[0] / 1e6, 2), "unit": "MPix/s", "cores": 1, "kind": kind,
        "sample": f"{raster_np.shape[0]}x{raster_np.shape[1]} float32 full raster, best of 2 round trips "
                  f"(lerc_computeCompressedSize+lerc_encode {best[1]*1e3:.0f} ms, lerc_decode {best[2]*1e3:.0f} ms)",
        "blob_bytes": len(blob),
    }


def measured_traffic(kernel_group, size):
    """HBM bytes per launch of a kernel group from the committed PMC passes (profiles/*hbm_traffic.json, written
    by tools/profile_run.sh: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, plus WRITE_SIZE).
    Counters cannot be read from inside the timed process, so this is the figure of the last profiled build of
    the same workload -- None when no such file or kernel group exists."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("size") == size and kernel_group in t.get("bytes_per_launch", {}):
                return t["bytes_per_launch"][kernel_group]
        except (OSError, ValueError):
            pass
    return None


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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)    # "nccl" is RCCL on ROCm

    from lerc_amd import api, shard, synth

    n = args.size
    n_pix = n * n
    tiles_mode = args.workload == "c5"
    if tiles_mode:
        # this rank's contiguous tile range of the mosaic (lerc_amd/shard.py), cut from the virtual raster
        side = max(1, int(round(args.tiles ** 0.5)))
        n_tiles = side * side
        first, _ = shard.tile_range(rank, world, n_tiles * world)
        big = synth.c2_float32(256 * side, 256 * side, row0=0, col0=(first // side) * 256, virt_cols=65536, device=dev)
        x = big.reshape(side, 256, side, 256).permute(0, 2, 1, 3).contiguous().reshape(n_tiles, 256, 256)
        n_pix = x.numel()
        out = torch.empty(n_pix * 4 + n_tiles * 256, dtype=torch.uint8, device=dev)
    else:
        # every rank compresses its own window of one large virtual raster (independent blobs)
        x = synth.c2_float32(n, n, row0=0, col0=rank * n, virt_cols=max(world, 1) * n, device=dev)
        out = torch.empty(n_pix * 4 + (1 << 20), dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    torch.cuda.synchronize()

    stream = torch.cuda.current_stream().cuda_stream
    codec = api.DeviceCodec(stream)

    blob_bytes = 0

    def step():
        nonlocal blob_bytes
        if tiles_mode:
            rc, offs, sizes, used = api.encode_tiles_device(codec, x, args.max_z_err, out)
            if rc != 0:
                raise RuntimeError(f"tile encode failed: status {rc}: {codec.last_error()}")
            blob_bytes = int(sizes.sum())
            rc = api.decode_tiles_device(codec, out, offs, sizes, y)
            if rc != 0:
                raise RuntimeError(f"tile decode failed: status {rc}: {codec.last_error()}")
            return
        rc, nb = api.encode_device(codec, x, args.max_z_err, out)
        if rc != 0:
            raise RuntimeError(f"encode failed: status {rc}: {codec.last_error()}")
        blob_bytes = nb
        rc = api.decode_device(codec, out, nb, y)
        if rc != 0:
            raise RuntimeError(f"decode failed: status {rc}: {codec.last_error()}")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    codec.lib.lerc_amd_profile_enable.argtypes = [ct.c_void_p, ct.c_int]
    codec.lib.lerc_amd_profile_read.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_int, ct.c_int]
    codec.lib.lerc_amd_profile_enable(codec.h, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    codec.lib.lerc_amd_profile_enable(codec.h, 0)
    buf = ct.create_string_buffer(1 << 16)
    codec.lib.lerc_amd_profile_read(codec.h, buf, len(buf), 1)
    prof = {}
    for line in buf.value.decode().splitlines():
        name, ms, cnt = line.split()
        prof[name] = (float(ms), int(cnt))

    elapsed = shard.max_over_ranks(elapsed, device=dev)

    # correctness of what was timed (outside the timed region)
    err = float((y.double() - x.double()).abs().max().item())
    verified = err <= args.max_z_err * (1 + 1e-6) + 6.2e-5    # + 1/2 ulp of an f32 near 1000

    if rank == 0:
        raw_bytes = n_pix * 4
        b_enc = raw_bytes + blob_bytes
        b_dec = blob_bytes + raw_bytes
        # an encode-side launch is priced at B_enc = raw + blob bytes, a decode-side launch at
        # B_dec = blob + raw bytes, whatever part of them that launch really touches (extra passes only lower frac)
        dec_side = ("decode", "discover", "resolve", "gather", "walk", "fletcher_dec", "huff_dec")
        alg = {k: (b_dec if any(t in k for t in dec_side) else b_enc) for k in prof}
        kernels = {k: {"avg_ms": round(v[0] / max(v[1], 1), 5), "launches": v[1]} for k, v in prof.items()}
        dom = max(prof.items(), key=lambda kv: kv[1][0])[0] if prof else None
        roofline = None
        if dom:
            avg_s = prof[dom][0] / max(prof[dom][1], 1) / 1e3
            ach = alg[dom] / avg_s / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": measured_traffic(dom, n),
                        "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": round(avg_s * 1e3, 5)}
        ms_per_step = elapsed / args.steps * 1e3
        kernel_ms = sum(v[0] for v in prof.values()) / max(args.steps, 1)
        res = {
            "metric": "MPix/s encode+decode round-trip, 8192^2 float32 MaxZError=0.01" if not tiles_mode
                      else "MPix/s encode+decode round-trip, 256^2 float32 tiles MaxZError=0.01 (batched calls)",
            "value": round(world * n_pix * args.steps / elapsed / 1e6, 2),
            "unit": "MPix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{n}x{n} float32 1-band, MaxZError={args.max_z_err}, lerc encode+decode on HBM-resident data" if not tiles_mode
                                    else f"{x.shape[0]} tiles of 256x256 float32 per rank, MaxZError={args.max_z_err}, one batched encode + one batched decode call")
                                   + (", one raster per rank (independent blobs, no data-path collective)" if world > 1 else ""),
                       "blob_bytes": blob_bytes, "compression_ratio": round(raw_bytes / max(blob_bytes, 1), 3),
                       "max_abs_error": err, "verified": bool(verified)},
            "roofline": roofline,
            "roundtrip": {"algorithmic_bytes": b_enc + b_dec, "kernel_ms_per_step": round(kernel_ms, 4),
                          "frac_of_hbm_peak_wall": round((b_enc + b_dec) / (ms_per_step / 1e3) / 1e9 / HBM_PEAK_GBS, 5),
                          "frac_of_hbm_peak_kernels": round((b_enc + b_dec) / (max(kernel_ms, 1e-9) / 1e3) / 1e9 / HBM_PEAK_GBS, 5)},
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline and not tiles_mode:
            res["cpu_baseline"] = cpu_baseline(x.cpu().numpy(), args.max_z_err)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
