"""Tuning tool: 16 384 tiles of the mosaic on one GPU, slots and packed (bench.py's c5_1gpu object at a quarter of its size).
    gpurun -- 'python tools/time_c5_packed.py'"""
import os, sys, ctypes as ct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lerc_amd import api, synth
import bench
dev = torch.device("cuda:0")
codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
r = bench.c5_single_gpu(torch, api, synth, codec, dev, 0.01, 16384, steps=3, warmup=1)
print("slots", r["ms_per_step"], r["frac_of_hbm_peak_wall"], "packed", r["packed"]["ms_per_step"], r["packed"]["frac_of_hbm_peak_wall"], r["packed"]["verified"],
      "paths", codec.path_counters(), "forms", codec.decode_forms(), "refusals", codec.decode_refusals(), codec.last_note())
