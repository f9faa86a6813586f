"""Tuning aid: the other BASELINE configurations, device resident, wall time per call and time per kernel:
C3 16384^2 uint16 lossless, C4 4096^2 x 3 uint8 lossless (8-bit Huffman mode), C2 raster at maxZErr 0 (lossless float).
    gpurun -- 'python tools/time_configs.py [c3 c4 c2lossless]'"""
import ctypes as ct
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lerc_amd import api, synth  # noqa: E402


def run(name, x, max_z_err, n_depth, mask=None):
    dev = torch.device("cuda:0")
    x = x.to(dev)
    d_mask = mask.to(dev) if mask is not None else None
    out_mask = torch.empty_like(d_mask) if mask is not None else None
    codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
    L = codec.lib
    L.lerc_amd_profile_enable.argtypes = [ct.c_void_p, ct.c_int]
    L.lerc_amd_profile_read.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_int, ct.c_int]
    out = torch.empty(x.numel() * x.element_size() + (1 << 20), dtype=torch.uint8, device=dev)
    dec = torch.empty_like(x)
    n_pix = x.shape[0] * x.shape[1]
    best_e = best_d = 1e9
    for rep in range(4):
        if rep == 3:
            L.lerc_amd_profile_enable(codec.h, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mask is None:
            rc, n = api.encode_device(codec, x, max_z_err, out, n_depth)
        else:
            rc, n = codec.encode(x.data_ptr(), api._torch_dt_code(x), n_depth, int(x.shape[1]), int(x.shape[0]), 1, max_z_err,
                                 out.data_ptr(), out.numel(), d_mask.data_ptr(), 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if mask is None:
            rc2 = api.decode_device(codec, out, n, dec, n_depth)
        else:
            rc2 = codec.decode(out.data_ptr(), n, api._torch_dt_code(x), n_depth, int(x.shape[1]), int(x.shape[0]), 1, dec.data_ptr(),
                               out_mask.data_ptr(), 1)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        assert rc == 0 and rc2 == 0, (rc, rc2, codec.last_error())
        best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1)
    buf = ct.create_string_buffer(1 << 16)
    L.lerc_amd_profile_read(codec.h, buf, len(buf), 1)
    same = bool(torch.equal(dec.view(torch.uint8), x.view(torch.uint8))) if max_z_err == 0 else None
    print(f"{name}: blob {n} B (ratio {x.numel() * x.element_size() / n:.2f}), encode {1e3 * best_e:.3f} ms, decode {1e3 * best_d:.3f} ms, "
          f"round trip {n_pix / (best_e + best_d) / 1e6:.0f} MPix/s, lossless round trip: {same}")
    for line in buf.value.decode().strip().splitlines():
        f = line.split()
        print(f"    {f[0]:28s} {1e3 * float(f[1]) / max(int(f[2]), 1):9.1f} us x{f[2]}")


def main():
    which = sys.argv[1:] or ["c3", "c4", "c2lossless"]
    if "c3" in which:
        run("C3 16384^2 uint16 lossless", synth.c3_uint16(), 0, 1)
    if "c4" in which:
        run("C4 4096^2 x3 uint8 lossless", synth.c4_rgb_u8(), 0, 3)
    if "general" in which:    # what the streaming kernels do not take: a mask, a width that is no multiple of 512
        x = synth.c2_float32()
        i = torch.arange(8192).view(-1, 1)
        j = torch.arange(8192).view(1, -1)
        m = (((i // 97) + (j // 131)) % 10 != 0).to(torch.uint8).contiguous()
        run("C2 raster with a 10 % mask (general path)", x, 0.01, 1, mask=m)
        run("8000 x 8000 float32 (general path)", x[:8000, :8000].contiguous(), 0.01, 1)
    if "c2lossless" in which:
        run("C2 8192^2 float32 maxZErr 0", synth.c2_float32(), 0, 1)


if __name__ == "__main__":
    main()
