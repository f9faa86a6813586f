"""Tuning tool: the C4 round trip's kernels (bench.py's c4 object alone).   gpurun -- 'python tools/time_c4.py'"""
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
xo = synth.c4_rgb_u8(device=dev)
r = bench.other_config(torch, api, codec, "c4", xo, 0, 3, steps=8, reference=False)
print("ms", r["ms_per_step"], "enc", r["encode_ms"], "dec", r["decode_ms"], "ok", r["verified"], {k: round(v["avg_ms"] * 1000, 1) for k, v in r["kernels"].items()})
