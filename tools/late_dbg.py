import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from lerc_amd import api, synth
dev = torch.device("cuda:0")
shape = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8190, 8190)
x = synth.c2_float32(shape[0], shape[1], device=dev)
codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
out = torch.empty(x.numel() * 4 + (1 << 20), dtype=torch.uint8, device=dev)
y = torch.empty_like(x)
for rnd in range(3):
    tickets = []
    f0, q0 = codec.decode_forms(), codec.decode_refusals()
    for _ in range(6):
        rc, t1 = api.encode_device_async(codec, x, 0.01, out)
        rc2, t2 = api.decode_device_async(codec, out, out.numel(), y)
        tickets.append((t1, t2))
    res = [(codec.finish(a), codec.finish(b)) for a, b in tickets]
    torch.cuda.synchronize()
    f1, q1 = codec.decode_forms(), codec.decode_refusals()
    print("round", rnd, [b - a for a, b in zip(f0, f1)], [b - a for a, b in zip(q0, q1)], codec.last_note(), "err", float((y.double() - x.double()).abs().max()))
