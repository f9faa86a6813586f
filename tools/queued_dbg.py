"""Tuning tool: queued round trips of the C2 raster with flat stretches (or 8190^2: `ragged`), refusals and notes per round."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lerc_amd import api, synth
dev = torch.device("cuda:0")
rag = len(sys.argv) > 1 and sys.argv[1] == "ragged"
x = synth.c2_float32(8190, 8190, device=dev) if rag else synth.c2_float32(8192, 8192, device=dev)
if not rag:
    for (r0, r1, c0, c1, val) in ((512, 2560, 1024, 3584, 1017.25), (3000, 5048, 4096, 7168, 733.5), (6000, 7024, 256, 2304, 1500.0), (5120, 5632, 0, 2048, 0.0)):
        x[r0:r1, c0:c1] = val
codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
out = torch.empty(x.numel() * 4 + (1 << 20), dtype=torch.uint8, device=dev)
y = torch.empty_like(x)
for _ in range(10):
    rc, nb = api.encode_device(codec, x, 0.01, out); rc2 = api.decode_device(codec, out, nb, y)
print("per call:", codec.decode_forms(), codec.decode_refusals(), codec.last_note())
for rnd in range(4):
    tickets = []
    f0, q0 = codec.decode_forms(), codec.decode_refusals()
    for _ in range(8):
        rc, t1 = api.encode_device_async(codec, x, 0.01, out)
        rc2, t2 = api.decode_device_async(codec, out, out.numel(), y)
        tickets.append((t1, t2))
    res = [(codec.finish(a), codec.finish(b)) for a, b in tickets]
    torch.cuda.synchronize()
    f1, q1 = codec.decode_forms(), codec.decode_refusals()
    print("round", rnd, [b - a for a, b in zip(f0, f1)], [b - a for a, b in zip(q0, q1)], codec.last_note(), "err", float((y.double() - x.double()).abs().max()))
