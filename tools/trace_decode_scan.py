"""Tuning tool: per-workgroup time lines of the scanning decoder (build with -DLERC_PROBE -DLERC_PROBE_TRACE_ONLY).
    gpurun -- 'PROBE_LIB=$PWD/lerc_amd/csrc/_var/trace.so python tools/trace_decode_scan.py [c3]'"""
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
c3 = len(sys.argv) > 1 and sys.argv[1] == "c3"
piece = int(os.environ.get("SCAN_PIECE", "32768"))
rag = len(sys.argv) > 1 and sys.argv[1] == "ragged"
x = synth.c3_uint16().to(dev) if c3 else synth.c2_float32(8190, 8190, device=dev) if rag else synth.c2_float32(8192, 8192, device=dev)
codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
blob = torch.empty(x.numel() * x.element_size() + 4096, dtype=torch.uint8, device=dev)
y = torch.empty_like(x)
rc, n = api.encode_device(codec, x, 0 if c3 else 0.01, blob)
for _ in range(3):
    rc = api.decode_device(codec, blob, n, y)
    assert rc == 0
torch.cuda.synchronize()
rows = 8192
buf = (ct.c_ulonglong * (16 * rows))()
lib.lerc_amd_probe_trace_decode_scan(buf, 16 * rows)
t = np.frombuffer(buf, dtype=np.uint64).reshape(rows, 16).astype(np.int64)
n_wg = (n + piece - 1) // piece
tt = t[:min(n_wg, rows)]
tt = tt[tt[:, 0] > 0]
t0 = tt[:, 0].min()
us = (tt[:, :7] - t0) / 100.0  # (items in ticket order)
names = ["start -> staged, Fletcher, scan", "candidates", "survivors -> list", "headers + tiling check (+ mending)", "cells of the pieces in front", "places + pixels"]
print(f"decode_scan: {len(tt)} workgroups traced, span {us[:, 6].max():.1f} us, mean life {(us[:, 6] - us[:, 0]).mean():.2f} us")
for k, nm in enumerate(names):
    d = us[:, k + 1] - us[:, k]
    print("   %-44s mean %6.2f  p50 %6.2f  p90 %6.2f" % (nm, d.mean(), np.median(d), np.percentile(d, 90)))
st = np.sort(us[:, 0])
print("   start times: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % (np.percentile(st, 10), np.median(st), np.percentile(st, 90), st.max()))
# the wait for the cells by position in the stream (a chain through the groups would make it grow), and by position in the group of 64
wait = us[:, 5] - us[:, 4]
n_t = len(wait)
dec = [wait[i * n_t // 10:(i + 1) * n_t // 10].mean() for i in range(10)]
print("   wait for the cells by tenth of the stream: " + " ".join("%.2f" % v for v in dec))
pos = np.arange(n_t) % 64
print("   wait by position in the group (0, 1, 8, 32, 62, 63): " + " ".join("%.2f" % wait[pos == q].mean() for q in (0, 1, 8, 32, 62, 63)))
life = us[:, 6] - us[:, 0]
print("   life by tenth of the stream: " + " ".join("%.1f" % life[i * n_t // 10:(i + 1) * n_t // 10].mean() for i in range(10)))
# pieces that were mended (TRACEV slots: 7 = pass | frontBad << 4 | broken links << 8 | entries << 16; 11 = over | bad << 1 | verdict bad << 2 | mended << 3; 14 / 15 = runs entered / entries struck)
flags = tt[:, 11]
mended = np.nonzero(flags & 8)[0]
print("   pieces mended: %d of %d; first: %s" % (len(mended), len(tt), [(int(i), int(tt[i, 7]) >> 8 & 0xFF, int(tt[i, 7]) >> 4 & 1, int(tt[i, 14]), int(tt[i, 15])) for i in mended[:8]]), "(piece, broken links, front gap, runs, struck)")
print("   pieces flagged bad: %d" % int(np.count_nonzero(flags & 6)))
