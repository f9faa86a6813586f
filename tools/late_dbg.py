import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from lerc_amd import api, synth
dev = torch.device("cuda:0")
big = synth.c2_float32(2048, 2304, device=dev)
rng = np.random.default_rng(20260928)
bad = 0
for it in range(400):
    r, c = int(rng.integers(32, 2048)), int(rng.integers(64, 2304))
    x = big[:r, :c].contiguous()
    e = 0.01
    kind = it % 3
    if kind == 1: x = (x * 8).to(torch.int32).contiguous(); e = 0
    if kind == 2: x = (x * 8).to(torch.int32).to(torch.uint16).contiguous(); e = 0
    codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
    blob = torch.empty(x.numel() * x.element_size() + 8192, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    rc, nb = api.encode_device(codec, x, e, blob)
    rc2 = api.decode_device(codec, blob, nb, y)
    f1 = codec.decode_forms()
    if f1[3] != 1:
        bad += 1
        if bad <= 12: print((r, c), str(x.dtype), nb, f1, codec.decode_refusals(), codec.last_note())
    codec.close()
print("bad", bad, "of 400")
