"""Tuning aid: C5-style tile mosaics -- per-tile calls versus one batched call.   gpurun -- 'python tools/time_tiles.py [nSide]'"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lerc_amd import api, synth  # noqa: E402


def main():
    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda:0")
    big = synth.c2_float32(256 * n_side, 256 * n_side, virt_cols=65536, device=dev)
    tiles = big.reshape(n_side, 256, n_side, 256).permute(0, 2, 1, 3).contiguous().reshape(n_side * n_side, 256, 256)
    n_tiles = tiles.shape[0]
    codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
    arena = torch.empty(tiles.numel() * 4 + n_tiles * 256, dtype=torch.uint8, device=dev)
    out = torch.empty_like(tiles)
    for _ in range(2):
        rc, offs, sizes, used = api.encode_tiles_device(codec, tiles, float(os.environ.get("MZE", "0.01")), arena)
        assert rc == 0
        assert api.decode_tiles_device(codec, arena, offs, sizes, out) == 0
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        rc, offs, sizes, used = api.encode_tiles_device(codec, tiles, float(os.environ.get("MZE", "0.01")), arena)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps):
        api.decode_tiles_device(codec, arena, offs, sizes, out)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    assert float((out - tiles).abs().max()) <= float(os.environ.get("MZE", "0.01")) * 1.01
    import ctypes as ct
    lib = codec.lib
    lib.lerc_amd_profile_enable.argtypes = [ct.c_void_p, ct.c_int]
    lib.lerc_amd_profile_read.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_int, ct.c_int]
    lib.lerc_amd_path_counters.argtypes = [ct.c_void_p, ct.POINTER(ct.c_ulonglong)]
    c0 = (ct.c_ulonglong * 4)()
    lib.lerc_amd_path_counters(codec.h, c0)
    lib.lerc_amd_profile_enable(codec.h, 1)
    api.encode_tiles_device(codec, tiles, float(os.environ.get("MZE", "0.01")), arena)
    api.decode_tiles_device(codec, arena, offs, sizes, out)
    lib.lerc_amd_profile_enable(codec.h, 0)
    buf = ct.create_string_buffer(1 << 16)
    lib.lerc_amd_profile_read(codec.h, buf, len(buf), 1)
    c1 = (ct.c_ulonglong * 4)()
    lib.lerc_amd_path_counters(codec.h, c1)
    lib.lerc_amd_last_note.argtypes = [ct.c_void_p]
    lib.lerc_amd_last_note.restype = ct.c_char_p
    print("note:", lib.lerc_amd_last_note(codec.h).decode())
    print("paths (enc stream, enc general, dec stream, dec general):", [int(b - a) for a, b in zip(c0, c1)])
    for line in buf.value.decode().splitlines():
        name, ms, cnt = line.split()
        print(f"   {name:22s} {float(ms) * 1e3:9.1f} us  x{cnt}")
    px = tiles.numel()
    print(f"batched : {n_tiles} tiles, blobs {used} B; encode {(t1 - t0) / reps * 1e3:.3f} ms, decode {(t2 - t1) / reps * 1e3:.3f} ms, "
          f"round trip {px / ((t2 - t0) / reps) / 1e6:.0f} MPix/s")
    # a slot per tile: the encode kernel writes every blob where it stays (no packing pass)
    slot = (256 * 256 * 4 // 2 + 4096 + 15) // 16 * 16
    mze = float(os.environ.get("MZE", "0.01"))
    for _ in range(2):
        rc, sz = api.encode_tiles_device_slots(codec, tiles, mze, arena, slot)
        assert rc == 0 and api.decode_tiles_device_slots(codec, arena, slot, sz, out) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        rc, sz = api.encode_tiles_device_slots(codec, tiles, mze, arena, slot)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps):
        api.decode_tiles_device_slots(codec, arena, slot, sz, out)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    assert float((out - tiles).abs().max()) <= mze * 1.01
    print(f"slotted : {n_tiles} tiles, a slot of {slot} B each; encode {(t1 - t0) / reps * 1e3:.3f} ms, decode {(t2 - t1) / reps * 1e3:.3f} ms, "
          f"round trip {px / ((t2 - t0) / reps) / 1e6:.0f} MPix/s")
    one = torch.empty(256 * 256 * 4 + 4096, dtype=torch.uint8, device=dev)
    y = torch.empty_like(tiles[0])
    m = min(n_tiles, 256)
    t0 = time.perf_counter()
    for t in range(m):
        rc, nb = api.encode_device(codec, tiles[t], float(os.environ.get("MZE", "0.01")), one)
        api.decode_device(codec, one, nb, y)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"per tile: {m} tiles, {(t1 - t0) / m * 1e6:.0f} us per tile round trip, {m * 65536 / (t1 - t0) / 1e6:.0f} MPix/s")


if __name__ == "__main__":
    main()
