"""Tuning tool: the flat-stretch round trip's kernels (bench.py's c2_ragged object alone).   gpurun -- 'python tools/time_c2_ragged.py'"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lerc_amd import api, synth
import bench
dev = torch.device("cuda:0")
codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
codec.lib.lerc_amd_profile_enable.argtypes = [bench.ct.c_void_p, bench.ct.c_int]
codec.lib.lerc_amd_profile_read.argtypes = [bench.ct.c_void_p, bench.ct.c_char_p, bench.ct.c_int, bench.ct.c_int]
xo = synth.c2_float32(8192, 8192, device=dev)
for (r0, r1, c0, c1, val) in ((512, 2560, 1024, 3584, 1017.25), (3000, 5048, 4096, 7168, 733.5), (6000, 7024, 256, 2304, 1500.0), (5120, 5632, 0, 2048, 0.0)):
    xo[r0:r1, c0:c1] = val
r = bench.other_config(torch, api, codec, "c2 flat", xo, 0.01, 1, steps=8, reference=False)
print("ms", r["ms_per_step"], "frac", r["frac_of_hbm_peak_wall"], "ok", r["verified"], {k: round(v["avg_ms"] * 1000, 1) for k, v in r["kernels"].items()}, r["decode_forms"], r["decode_refusals"],
      "queued", r["queued"]["ms_per_step"], r["queued"]["frac_of_hbm_peak_wall"], {k: round(v["avg_ms"] * 1000, 1) for k, v in r["queued"]["kernels"].items()})
