"""Tuning tool: per-phase shader cycles of the streaming kernels (needs `make -C lerc_amd/csrc probe` and a GPU).

Loads lerc_amd/csrc/_probe/liblerc_amd_probe.so (the product sources compiled with -DLERC_PROBE: thread 0 of every
workgroup accumulates clock64() deltas between phase markers), runs the C2 workload a few times and prints, per
kernel file, the total cycles per phase slot.  The sums include the time a workgroup waits while other resident
workgroups run, so read them as proportions.
    gpurun -- 'python tools/probe_phases.py [rows cols]'
"""
import ctypes as ct
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["LERC_AMD_LIBRARY"] = os.environ.get("PROBE_LIB") or os.path.join(ROOT, "lerc_amd", "csrc", "_probe", "liblerc_amd_probe.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lerc_amd import api, synth  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    cols = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    lib = api.load_library()
    dev = torch.device("cuda:0")
    x = synth.c2_float32(rows, cols, device=dev)
    codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
    blob = torch.empty(rows * cols * 4 + 4096, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    out = (ct.c_ulonglong * 32)()
    readers = {}
    for tag in ("fast_encode", "fast_decode"):
        try:
            readers[tag] = getattr(lib, "lerc_amd_probe_" + tag)
        except AttributeError:
            pass

    def once():
        rc, n = api.encode_device(codec, x, 0.01, blob)
        assert rc == 0, rc
        rc = api.decode_device(codec, blob, n, y)
        assert rc == 0, rc

    once()
    for f in readers.values():
        f(out, 1)
    reps = 5
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    assert float((y - x).abs().max()) <= 0.0101
    for tag, f in readers.items():
        f(out, 1)
        vals = [int(v) / reps for v in out]
        tot = sum(vals) or 1
        print(tag, "Mcycles per call, summed over workgroups (thread 0):")
        for i, v in enumerate(vals):
            if v:
                print(f"  slot {i:2d}: {v / 1e6:10.2f} Mcyc  {100 * v / tot:5.1f} %")


if __name__ == "__main__":
    main()
