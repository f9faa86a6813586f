"""A fresh context's first streaming calls, again and again (the persistent cells are allocated and wiped right in front of the
first kernel that uses them): counts calls that fell back to the general kernels.  Run several copies side by side.
    python tools/fresh_context_stress.py [iterations]"""
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from lerc_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
x = synth.c2_float32(2048, 4096, device=dev)
blob = torch.empty(x.numel() * 4 + 4096, dtype=torch.uint8, device=dev)
y = torch.empty_like(x)
fell = bad = tier = 0
import ctypes as ct
def note(codec):
    codec.lib.lerc_amd_last_note.restype = ct.c_char_p
    codec.lib.lerc_amd_last_note.argtypes = [ct.c_void_p]
    return codec.lib.lerc_amd_last_note(codec.h)
for i in range(n):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        codec = api.DeviceCodec(s.cuda_stream)
        rc, nb = api.encode_device(codec, x, 0.01, blob)
        rc2 = api.decode_device(codec, blob, nb, y)
        c = codec.path_counters()
        nt = note(codec)
        if nt and b"decode" in nt:
            tier += 1
            if tier <= 3: print("  iteration", i, nt)
        if rc or rc2: bad += 1
        if c[1] or c[3]:
            fell += 1
            if fell <= 4: print("  iteration", i, "counters", list(c), "note:", note(codec), "error:", codec.last_error())
        codec.close()
print("iterations", n, "decodes whose first tier gave the band back:", tier, "with a call through the general kernels:", fell, "failed:", bad, os.environ.get("LERC_AMD_LIBRARY", "product"))
