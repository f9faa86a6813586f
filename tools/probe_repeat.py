"""Diagnostic: repeated device-resident decodes of one blob -- which path served each call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lerc_amd import api, synth
dev = torch.device("cuda:0")
codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
for n in (256, 1024, 2048, 4096):
    x = synth.c2_float32(n, n, device=dev)
    blob = torch.empty(x.numel() * 4 + 4096, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    rc, nb = api.encode_device(codec, x, 0.01, blob)
    out = []
    for rep in range(4):
        c0 = codec.path_counters()
        rc = api.decode_device(codec, blob, nb, y)
        c1 = codec.path_counters()
        out.append((rc, [int(b) - int(a) for a, b in zip(c0, c1)]))
    print(n, nb, out, codec.last_error())
