"""Tuning tool: per-workgroup time lines of the two streaming decode kernels (probe build with -DLERC_PROBE_TRACE_ONLY).
    gpurun -- 'PROBE_LIB=$PWD/lerc_amd/csrc/_var/trace.so python tools/trace_decode.py'"""
import ctypes as ct
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["LERC_AMD_LIBRARY"] = os.environ.get("PROBE_LIB") or os.path.join(ROOT, "lerc_amd", "csrc", "_probe", "liblerc_amd_probe.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from lerc_amd import api, synth  # noqa: E402

lib = api.load_library()
dev = torch.device("cuda:0")
x = synth.c2_float32(8192, 8192, device=dev)
codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
blob = torch.empty(x.numel() * 4 + 4096, dtype=torch.uint8, device=dev)
y = torch.empty_like(x)
rc, n = api.encode_device(codec, x, 0.01, blob)
for _ in range(3):
    rc = api.decode_device(codec, blob, n, y)
    assert rc == 0
torch.cuda.synchronize()
rows = 32768
buf = (ct.c_ulonglong * (8 * rows))()
lib.lerc_amd_probe_trace_decode(buf, 8 * rows)
t = np.frombuffer(buf, dtype=np.uint64).reshape(rows, 8).astype(np.int64)
n_chunks = (n + 2047) // 2048
for name, lo, cnt, names in (("discover", 0, (n_chunks + 15) // 16, ["start->staged (+ Fletcher)", "pattern scan", "heads", "walk", ]),
                             ("decode", 8192, min((n_chunks + 3) // 4, rows - 8192), ["start->cells, lists, bytes", "stage, first-block cells", "parse", "pixels"])):
    tt = t[lo:lo + cnt]
    tt = tt[tt[:, 0] > 0]
    t0 = tt[:, 0].min()
    us = (tt[:, :5] - t0) / 100.0
    print(f"{name}: {len(tt)} workgroups traced, span {us[:, 4].max():.1f} us, mean life {(us[:, 4] - us[:, 0]).mean():.2f} us")
    for k, nm in enumerate(names):
        d = us[:, k + 1] - us[:, k]
        print("   %-30s mean %6.2f  p50 %6.2f  p90 %6.2f" % (nm, d.mean(), np.median(d), np.percentile(d, 90)))
    if name == "discover" and (tt[:, 5] > 0).all():
        al = (tt[:, :8] - t0) / 100.0
        for nm, a, b in (("  scan loop (wave 0)", 1, 5), ("  barrier behind it", 5, 6), ("  offset types + barrier", 6, 2), ("  walk inside the chunk (wave 0)", 3, 7), ("  behind the chunk + barrier", 7, 4)):
            d = al[:, b] - al[:, a]
            print("   %-34s mean %6.2f  p50 %6.2f  p90 %6.2f" % (nm, d.mean(), np.median(d), np.percentile(d, 90)))
