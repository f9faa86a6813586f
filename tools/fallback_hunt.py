"""Rasters of many sizes (rows, columns: multiples of 8 and not), float32 and uint16, through the device API on fresh and on
warm contexts: every call should be served by the streaming kernels (synthetic terrain has nothing that sends a band
elsewhere); prints the calls that were not, with the library's note.
    python tools/fallback_hunt.py [seconds] [seed]"""
import sys, time, ctypes as ct
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from lerc_amd import api, synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
big = synth.c2_float32(4096, 4608, device=dev)
def note(codec):
    codec.lib.lerc_amd_last_note.restype = ct.c_char_p
    codec.lib.lerc_amd_last_note.argtypes = [ct.c_void_p]
    return codec.lib.lerc_amd_last_note(codec.h)
t0 = time.time(); n = odd = 0
warm = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
while time.time() - t0 < budget:
    r, c = int(rng.integers(32, 4096)), int(rng.integers(64, 4608))
    if rng.random() < 0.6: r -= r % 8; c -= c % 8
    x = big[:r, :c].contiguous()
    e = 0.01
    if rng.random() < 0.4:
        x = (x * 8).to(torch.int32).clamp(0, 65535).to(torch.uint16) if hasattr(torch, "uint16") else x
        e = 0
    if x.dtype not in (torch.float32,):
        try:
            api._torch_dt_code(x)
        except Exception:
            x = big[:r, :c].contiguous(); e = 0.01
    blob = torch.empty(x.numel() * x.element_size() + 8192, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    for fresh in (True, False):
        codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream) if fresh else warm
        c0 = codec.path_counters()
        rc, nb = api.encode_device(codec, x, e, blob)
        rc2 = api.decode_device(codec, blob, nb, y)
        c1 = codec.path_counters()
        n += 1
        if rc or rc2 or c1[1] != c0[1] or c1[3] != c0[3]:
            odd += 1
            if odd <= 12: print("  ", tuple(x.shape), x.dtype, "fresh" if fresh else "warm", "status", rc, rc2, "counters", [int(b - a) for a, b in zip(c0, c1)], note(codec), codec.last_error())
        if fresh: codec.close()
print("round trips", n, "not served by the streaming kernels alone:", odd)
