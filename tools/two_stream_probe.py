"""Tuning probe: do two decodes on two HIP streams overlap (discovery of one filling the gaps of the other's decode)?
K decodes of the C2 blob on one stream against the same K split over two streams (two contexts, two blobs, two outputs)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lerc_amd import api, synth  # noqa: E402

dev = torch.device("cuda:0")
x = synth.c2_float32(8192, 8192, device=dev)
s = [torch.cuda.Stream(), torch.cuda.Stream()]
codecs = [api.DeviceCodec(st.cuda_stream) for st in s]
blobs = [torch.empty(x.numel() * 4 + 4096, dtype=torch.uint8, device=dev) for _ in range(2)]
ys = [torch.empty_like(x) for _ in range(2)]
ns = []
for i in range(2):
    with torch.cuda.stream(s[i]):
        rc, n = api.encode_device(codecs[i], x, 0.01, blobs[i])
        assert rc == 0
        ns.append(n)
torch.cuda.synchronize()
K = 16


def run(two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tickets = []
    for k in range(K):
        i = (k % 2) if two else 0
        rc, t = api.decode_device_async(codecs[i], blobs[i], ns[i], ys[i])
        assert rc == 0
        tickets.append((i, t))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    for i, t in tickets:
        rc, _ = codecs[i].finish(t)
        assert rc == 0
    return el / K * 1e6


for _ in range(2):
    print("one stream: %.1f us per decode;  two streams: %.1f us per decode" % (run(False), run(True)))
