"""Tuning aid: device-resident decode latency by raster size (one call at a time, the host waits for each), to place the
threshold between the one-launch streaming decoder and the two-launch form.
    gpurun -- 'python tools/time_sizes.py; LERC_AMD_DECODE_LAUNCHES=2 python tools/time_sizes.py'"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lerc_amd import api, synth  # noqa: E402

dev = torch.device("cuda:0")
codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
print("launches =", os.environ.get("LERC_AMD_DECODE_LAUNCHES", "default"))
for n in (256, 512, 1024, 2048, 4096, 8192):
    x = synth.c2_float32(n, n, device=dev)
    blob = torch.empty(x.numel() * 4 + 4096, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    rc, nb = api.encode_device(codec, x, 0.01, blob)
    assert rc == 0
    best = 1e9
    for rep in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = api.decode_device(codec, blob, nb, y)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    assert rc == 0 and float((y - x).abs().max()) <= 0.0101
    print(f"  {n:5d}^2 f32: blob {nb:10d} B, decode {1e6 * best:8.1f} us")
