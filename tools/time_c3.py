"""Tuning tool: the C3 round trip's kernels (bench.py's c3 object alone).   gpurun -- 'python tools/time_c3.py'"""
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
xo = synth.c3_uint16(device=dev)
r = bench.other_config(torch, api, codec, "c3", xo, 0, 1, steps=8, reference=False)
print("ms", r["ms_per_step"], "frac", r["frac_of_hbm_peak_wall"], "ok", r["verified"], {k: round(v["avg_ms"] * 1000, 1) for k, v in r["kernels"].items()},
      "queued", r["queued"]["ms_per_step"], {k: round(v["avg_ms"] * 1000, 1) for k, v in r["queued"]["kernels"].items()})
