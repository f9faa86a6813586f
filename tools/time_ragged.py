"""Tuning tool: rasters whose sides are no multiples of 8 (8190 x 8190, 257 x 257 elevation tiles) on the device API."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from lerc_amd import api, synth  # noqa: E402
import capi  # noqa: E402

dev = torch.device("cuda:0")
codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
O = capi.oracle()
for rows, cols in ((8190, 8190), (8188, 8188), (8191, 8191), (4300, 4600), (3612, 3612), (3601, 3601), (257, 257), (1000, 1201)):
    x = synth.c2_float32(rows + 8, cols + 8, device=dev)[:rows, :cols].contiguous()
    blob = torch.empty(x.numel() * 4 + 4096, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    pc0 = codec.path_counters() if hasattr(codec, "path_counters") else None
    rc, n = api.encode_device(codec, x, 0.01, blob)
    assert rc == 0, rc
    rc = api.decode_device(codec, blob, n, y)
    assert rc == 0, rc
    torch.cuda.synchronize()
    best_e, best_d = 1e9, 1e9
    for _ in range(5):
        t0 = time.perf_counter(); rc, n = api.encode_device(codec, x, 0.01, blob); torch.cuda.synchronize(); t1 = time.perf_counter()
        rc2 = api.decode_device(codec, blob, n, y); torch.cuda.synchronize(); t2 = time.perf_counter()
        best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1)
    err = float((y.double() - x.double()).abs().max())
    same = None
    if rows * cols <= 9000 * 9000:
        r, b = O.encode(x.cpu().numpy(), 0.01)
        same = bytes(blob[:n].cpu().numpy().tobytes()) == b
    forms = codec.decode_forms() if hasattr(codec, "decode_forms") else None
    print(f"{rows} x {cols} f32: forms {forms} blob {n} B, encode {best_e*1e3:.3f} ms, decode {best_d*1e3:.3f} ms, round trip {rows*cols/(best_e+best_d)/1e6:.0f} MPix/s, max err {err:.5f}, blob == oracle: {same}")
