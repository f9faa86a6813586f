"""Tuning / soak tool: which streaming form serves repeated decodes, and why a form hands a blob on.
    gpurun -- 'python tools/scan_diag.py [n] [size]'"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lerc_amd import api, synth  # noqa: E402
import ctypes as ct  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
size = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dev = torch.device("cuda:0")
codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
L = codec.lib
L.lerc_amd_decode_forms.argtypes = [ct.c_void_p, ct.POINTER(ct.c_ulonglong)]
L.lerc_amd_last_note.argtypes = [ct.c_void_p]
L.lerc_amd_last_note.restype = ct.c_char_p


def forms():
    out = (ct.c_ulonglong * 4)()
    L.lerc_amd_decode_forms(codec.h, out)
    return tuple(int(v) for v in out)


for name, x, e in (("c2", synth.c2_float32(size, size, device=dev), 0.01), ("c3", synth.c3_uint16(size, size, device=dev), 0)):
    blob = torch.empty(x.numel() * x.element_size() + 4096, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    rc, nb = api.encode_device(codec, x, e, blob)
    assert rc == 0
    notes = collections.Counter()
    f0 = forms()
    fb = f0
    for i in range(n):
        y.zero_()
        rc = api.decode_device(codec, blob, nb, y)
        assert rc == 0
        fa = forms()
        served = [k for k in (1, 2, 3) if fa[k] != fb[k]]
        fb = fa
        notes[(tuple(served), L.lerc_amd_last_note(codec.h).decode() if served != [3] else "")] += 1
        err = float((y.double() - x.double()).abs().max().item())
        assert err <= e * (1 + 1e-6) + 6.2e-5, (i, err)
    f1 = forms()
    print(name, size, "blob", nb, "forms served (two-launch, walk, scan):", tuple(b - a for a, b in zip(f0, f1))[1:], flush=True)
    for k, v in notes.items():
        print("   %4d x %r" % (v, k))

# the queued form: encode and decode enqueued back to back, the decode given the buffer's capacity; one wait per pair / per K pairs
for per_wait in (1, 10):
    x = synth.c2_float32(size, size, device=dev)
    blob = torch.empty(x.numel() * 4 + 4096, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    notes = collections.Counter()
    fb = forms()
    for i in range(0, n, per_wait):
        tk = []
        for j in range(per_wait):
            rc, t1 = api.encode_device_async(codec, x, 0.01, blob)
            rc2, t2 = api.decode_device_async(codec, blob, blob.numel(), y)
            assert rc == 0 and rc2 == 0
            tk.append((t1, t2))
        for t1, t2 in tk:
            assert codec.finish(t1)[0] == 0
            assert codec.finish(t2)[0] == 0
        fa = forms()
        notes[(tuple(b - a for a, b in zip(fb, fa))[1:], L.lerc_amd_last_note(codec.h).decode())] += 1
        fb = fa
    print("queued, %d pairs a wait:" % per_wait)
    for k, v in notes.items():
        print("   %4d x %r" % (v, k))
