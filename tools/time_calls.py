"""Tuning aid: wall-clock time of the device-API encode and decode calls (each returns after its own stream
sync) next to the sum of their kernels' HIP-event times.   gpurun -- 'python tools/time_calls.py'"""
import ctypes as ct
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lerc_amd import api, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    dev = torch.device("cuda:0")
    x = synth.c2_float32(n, n, device=dev)
    out = torch.empty(n * n * 4 + (1 << 20), dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
    lib = codec.lib
    lib.lerc_amd_profile_enable.argtypes = [ct.c_void_p, ct.c_int]
    lib.lerc_amd_profile_read.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_int, ct.c_int]
    for _ in range(3):
        rc, nb = api.encode_device(codec, x, 0.01, out)
        api.decode_device(codec, out, nb, y)
    torch.cuda.synchronize()
    lib.lerc_amd_profile_enable(codec.h, 1)
    reps = 20
    te = td = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        rc, nb = api.encode_device(codec, x, 0.01, out)
        t1 = time.perf_counter()
        rc2 = api.decode_device(codec, out, nb, y)
        t2 = time.perf_counter()
        assert rc == 0 and rc2 == 0
        te += t1 - t0
        td += t2 - t1
    buf = ct.create_string_buffer(1 << 16)
    lib.lerc_amd_profile_read(codec.h, buf, len(buf), 1)
    ke = kd = 0.0
    for line in buf.value.decode().splitlines():
        name, ms, cnt = line.split()
        if any(t in name for t in ("decode", "candidates", "chains", "resolve", "emit")):
            kd += float(ms)
        else:
            ke += float(ms)
    print(f"encode: wall {te / reps * 1e6:7.1f} us   kernels {ke / reps * 1e3:7.1f} us")
    print(f"decode: wall {td / reps * 1e6:7.1f} us   kernels {kd / reps * 1e3:7.1f} us")


if __name__ == "__main__":
    main()
