"""Tuning tool: per-workgroup time lines of the Huffman packing kernel on C4 (probe build with -DLERC_PROBE -DLERC_PROBE_TRACE_ONLY).
    gpurun -- 'PROBE_LIB=$PWD/lerc_amd/csrc/_var/trace.so python tools/trace_huff.py'"""
import ctypes as ct
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["LERC_AMD_LIBRARY"] = os.environ["PROBE_LIB"]
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from lerc_amd import api, synth  # noqa: E402

lib = api.load_library()
dev = torch.device("cuda:0")
x = synth.c4_rgb_u8().to(dev)    # [rows, cols, 3]: nDepth 3
codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
blob = torch.empty(x.numel() + (1 << 20), dtype=torch.uint8, device=dev)
for _ in range(3):
    rc, n = api.encode_device(codec, x, 0, blob, 3)
    assert rc == 0, rc
torch.cuda.synchronize()
rows = 8192
buf = (ct.c_ulonglong * (8 * rows))()
lib.lerc_amd_probe_trace_huff(buf, 8 * rows)
t = np.frombuffer(buf, dtype=np.uint64).reshape(rows, 8).astype(np.int64)
tt = t[t[:, 0] > 0]
t0 = tt[:, 0].min()
us = (tt[:, :6] - t0) / 100.0
print(f"huff_pack: {len(tt)} workgroups traced, blob {n} B, span {us[:, 5].max():.1f} us, mean life {(us[:, 5] - us[:, 0]).mean():.2f} us")
for k, nm in enumerate(["symbols staged", "bits counted", "look-back, span cleared", "packed", "span stored"]):
    d = us[:, k + 1] - us[:, k]
    print("   %-28s mean %7.2f  p50 %7.2f  p90 %7.2f" % (nm, d.mean(), np.median(d), np.percentile(d, 90)))
st = np.sort(us[:, 0])
print("   starts: p10 %.1f p50 %.1f p90 %.1f us; concurrently alive (mean): %.0f" % (np.percentile(st, 10), np.median(st), np.percentile(st, 90),
      (us[:, 5] - us[:, 0]).sum() / us[:, 5].max()))
