"""Tuning tool: N encode + decode round trips of the C2 raster on the device, return codes ignored -- to be run under
`rocprofv3 --kernel-trace` with experimental library builds (LERC_AMD_LIBRARY) whose results need not be valid.
    LERC_AMD_LIBRARY=$PWD/lerc_amd/csrc/_var/x.so rocprofv3 --kernel-trace --stats -d /tmp/p -o t -- python tools/roundtrip_loop.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lerc_amd import api, synth  # noqa: E402

dev = torch.device("cuda:0")
x = synth.c2_float32(8192, 8192, device=dev)
codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
blob = torch.empty(x.numel() * 4 + 4096, dtype=torch.uint8, device=dev)
y = torch.empty_like(x)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    rc, n = api.encode_device(codec, x, 0.01, blob)
    try:
        api.decode_device(codec, blob, n, y)
    except Exception as e:  # noqa: BLE001
        print("decode:", e)
torch.cuda.synchronize()
