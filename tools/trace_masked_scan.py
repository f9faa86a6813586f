"""Tuning tool: per-workgroup time lines of the scan for a masked band's block offsets (build with -DLERC_PROBE -DLERC_PROBE_TRACE_ONLY).
    gpurun -- 'PROBE_LIB=$PWD/lerc_amd/csrc/_var/trace.so python tools/trace_masked_scan.py'"""
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
n = 8192
x = synth.c2_float32(n, n, device=dev)
ii = torch.arange(n, device=dev).view(-1, 1)
jj = torch.arange(n, device=dev).view(1, -1)
mk = (((ii // 97) + (jj // 131)) % 10 != 0).to(torch.uint8).contiguous()
codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
blob = torch.empty(x.numel() * 4 + 4096, dtype=torch.uint8, device=dev)
y = torch.empty_like(x)
om = torch.empty_like(mk)
rc, nb = codec.encode(x.data_ptr(), 6, 1, n, n, 1, 0.01, blob.data_ptr(), blob.numel(), mk.data_ptr(), 1)
assert rc == 0
for _ in range(2):
    f0 = codec.decode_forms()
    rc = codec.decode(blob.data_ptr(), nb, 6, 1, n, n, 1, y.data_ptr(), om.data_ptr(), 1)
    assert rc == 0
    print("forms delta", [b - a for a, b in zip(f0, codec.decode_forms())], "note:", codec.last_note())
torch.cuda.synchronize()
rows = 8192
buf = (ct.c_ulonglong * (16 * rows))()
lib.lerc_amd_probe_trace_decode_scan(buf, 16 * rows)
t = np.frombuffer(buf, dtype=np.uint64).reshape(rows, 16).astype(np.int64)
n_wg = (nb + 32767) // 32768
tt = t[:min(n_wg, rows)]
tt = tt[tt[:, 0] > 0]
t0 = tt[:, 0].min()
us = (tt[:, :6] - t0) / 100.0
names = ["start -> staged, scan", "candidates", "survivors -> list", "headers + tiling check", "flood, second look, mending; cells of the pieces in front"]
print(f"scan_offsets: {len(tt)} workgroups traced, span {us[:, 5].max():.1f} us, mean life {(us[:, 5] - us[:, 0]).mean():.2f} us")
for k, nm in enumerate(names):
    d = us[:, k + 1] - us[:, k]
    print("   %-58s mean %6.2f  p50 %6.2f  p90 %6.2f  max %7.2f" % (nm, d.mean(), np.median(d), np.percentile(d, 90), d.max()))

v = tt[:, 8:16]
odd = np.nonzero((v[:, 3] & 7) != 0)[0]
w7 = tt[:, 7]
print("at the mending's door: flooded", int(((w7 & 15) >= 1).sum()), "front mismatch", int((((w7 >> 4) & 15) != 0).sum()), "broken links left: mean %.2f" % ((w7 >> 8) & 255).mean(), "pieces with any", int((((w7 >> 8) & 255) != 0).sum()))
w6 = tt[:, 6]
print("flood, mean per piece: bytes that read like a one-byte block %.1f, of them going on from the byte in front %.1f, seeds (a survivor ends there) %.1f, blocks found %.1f" % ((w6 & 0xFFFF).mean(), ((w6 >> 16) & 0xFFFF).mean(), ((w6 >> 32) & 0xFFFF).mean(), ((w6 >> 48) & 0xFFFF).mean()))
print("pieces with a flag:", len(odd), "of", len(tt), "; mended:", int(((v[:, 3] & 8) != 0).sum()))
print("   first broken links: mean %.1f max %d; entries struck: mean %.1f max %d; blocks entered: mean %.1f max %d" %
      (v[:, 1].mean(), v[:, 1].max(), v[:, 7].mean(), v[:, 7].max(), v[:, 6].mean(), v[:, 6].max()))
for i in odd[:12]:
    print("   piece", int(i), "total", int(v[i, 0]), "broken links", int(v[i, 1]), int(v[i, 2]), "flags", int(v[i, 3]), "first", int(v[i, 4]), "expected", int(v[i, 5]), "entered", int(v[i, 6]), "struck", int(v[i, 7]))
