"""Tuning aid: the stock host-pointer entry points (what a drop-in user calls) on the C2 raster: wall time of the C calls
alone (buffers allocated and touched beforehand), i.e. PCIe staging + kernels.   gpurun -- 'python tools/time_host_api.py'"""
import ctypes as ct
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import capi  # noqa: E402
from lerc_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    x = synth.c2_float32(n, n).numpy()
    L = capi.product().lib
    size = ct.c_uint(0)
    written = ct.c_uint(0)
    out = np.zeros(x.nbytes // 2, np.uint8)
    dec = np.zeros_like(x)
    for rep in range(4):
        t0 = time.perf_counter()
        rc0 = L.lerc_computeCompressedSize(x.ctypes.data, 6, 1, n, n, 1, 0, None, 0.01, ct.byref(size))
        t1 = time.perf_counter()
        rc1 = L.lerc_encode(x.ctypes.data, 6, 1, n, n, 1, 0, None, 0.01, out.ctypes.data, size.value, ct.byref(written))
        t2 = time.perf_counter()
        rc2 = L.lerc_decode(out.ctypes.data, written.value, 0, None, 1, n, n, 1, 6, dec.ctypes.data)
        t3 = time.perf_counter()
        assert rc0 == rc1 == rc2 == 0, (rc0, rc1, rc2)
        print(f"lerc_computeCompressedSize {1e3 * (t1 - t0):7.2f} ms   lerc_encode {1e3 * (t2 - t1):7.2f} ms   lerc_decode {1e3 * (t3 - t2):7.2f} ms   "
              f"({n * n / (t3 - t1) / 1e6:.0f} MPix/s encode+decode incl. PCIe, blob {written.value} B)")
    assert float(np.abs(dec.astype(np.float64) - x).max()) <= 0.0101


if __name__ == "__main__":
    main()
