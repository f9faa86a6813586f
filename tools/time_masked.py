"""Tuning tool: the masked C2 round trip's kernels (bench.py's c2_masked object alone).   gpurun -- 'python tools/time_masked.py'"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lerc_amd import api, synth
import bench
dev = torch.device("cuda:0")
codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
codec.lib.lerc_amd_profile_enable.argtypes = [bench.ct.c_void_p, bench.ct.c_int]
codec.lib.lerc_amd_profile_read.argtypes = [bench.ct.c_void_p, bench.ct.c_char_p, bench.ct.c_int, bench.ct.c_int]
n = 8192
xo = synth.c2_float32(n, n, device=dev)
ii = torch.arange(n, device=dev).view(-1, 1); jj = torch.arange(n, device=dev).view(1, -1)
mk = (((ii // 97) + (jj // 131)) % 10 != 0).to(torch.uint8).contiguous()
r = bench.other_config(torch, api, codec, "c2 masked", xo, 0.01, 1, reference=False, mask=mk)
print("ms", r["ms_per_step"], "enc", r["encode_ms"], "dec", r["decode_ms"], "ok", r["verified"], {k: round(v["avg_ms"] * 1000, 1) for k, v in r["kernels"].items()})
