"""Tuning tool: per-workgroup time line of the one-launch encoder (probe build): when each workgroup started, had its size
published, had its payload packed, knew where its span goes and was done -- and on which XCD / CU it ran.
    make -C lerc_amd/csrc probe;  gpurun -- 'python tools/trace_encode1.py'"""
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
for _ in range(3):
    rc, n = api.encode_device(codec, x, 0.01, blob)
    assert rc == 0
torch.cuda.synchronize()
nwg = int(os.environ.get('NWG', '8192'))
buf = (ct.c_ulonglong * (8 * nwg))()
lib.lerc_amd_probe_trace(buf, 8 * nwg)
t = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 8).astype(np.int64)
t0 = t[:, 0].min()
us = (t[:, :5] - t0) / 100.0    # wall_clock64: 100 MHz
xcc = (t[:, 7] >> 32) & 0xF
hw = t[:, 7] & 0xFFFFFFFF
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
print("kernel span %.1f us" % us[:, 4].max())
print("phase durations (us): mean / p50 / p90")
names = ["start->size published", "first payload", "wait for the base", "flush (+ other units)"]
for k in range(4):
    d = us[:, k + 1] - us[:, k]
    print("  %-24s %6.2f %6.2f %6.2f" % (names[k], d.mean(), np.median(d), np.percentile(d, 90)))
t5 = (t[:, 5] - t0) / 100.0
if t[:, 5].max() > 0:
    d = t5 - us[:, 0]
    print("  %-24s %6.2f %6.2f %6.2f" % ("start->statistics done", d.mean(), np.median(d), np.percentile(d, 90)))
print("life %.2f us mean" % (us[:, 4] - us[:, 0]).mean())
print("xcc of blocks 0..23:", xcc[:24].tolist())
# is the start order the index order?
s = us[:, 0]
print("start time by index, every 1024th:", np.round(s[::1024], 1).tolist())
inv = (np.diff(s) < 0).mean()
print("fraction of neighbours starting in reverse order: %.3f" % inv)
# how late is the latest-starting predecessor within the window of 512, relative to own start
late = np.array([s[max(0, i - 511):i].max() - s[i] if i else 0.0 for i in range(nwg)])
print("latest start among the 511 predecessors minus own start (us): mean %.2f p50 %.2f p90 %.2f max %.2f" % (late.mean(), np.median(late), np.percentile(late, 90), late.max()))
pub = us[:, 1]
latep = np.array([pub[max(0, i - 511):i].max() - us[i, 2] if i else 0.0 for i in range(nwg)])
print("latest size publication among the predecessors minus own payload end (us): mean %.2f p50 %.2f p90 %.2f" % (latep.mean(), np.median(latep), np.percentile(latep, 90)))
for xx in range(8):
    m = xcc == xx
    print("xcc %d: blocks %d, mean start %.1f, mean life %.2f, mean wait %.2f" % (xx, m.sum(), s[m].mean(), (us[m, 4] - us[m, 0]).mean(), (us[m, 3] - us[m, 2]).mean()))
np.save(os.path.join(ROOT, "gpurun_out", "trace_encode1.npy"), t)
