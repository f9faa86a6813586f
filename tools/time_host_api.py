"""Tuning aid: the stock host-pointer entry points (what a drop-in user calls) on the C2 raster: wall time per call, i.e.
PCIe staging + kernels.   gpurun -- 'python tools/time_host_api.py'"""
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
    P = capi.product()
    for rep in range(3):
        t0 = time.perf_counter()
        rc, size = P.compute_size(x, 0.01)
        t1 = time.perf_counter()
        rc2, blob = P.encode(x, 0.01, buf_size=size)
        t2 = time.perf_counter()
        rc3, dec, _ = P.decode(blob)
        t3 = time.perf_counter()
        assert rc == rc2 == rc3 == 0
        print(f"computeCompressedSize {1e3 * (t1 - t0):7.1f} ms   encode {1e3 * (t2 - t1):7.1f} ms   decode {1e3 * (t3 - t2):7.1f} ms   "
              f"({n * n / (t3 - t1) / 1e6:.0f} MPix/s encode+decode, blob {len(blob)} B)")
    assert float(np.abs(dec.reshape(n, n).astype(np.float64) - x).max()) <= 0.0101


if __name__ == "__main__":
    main()
