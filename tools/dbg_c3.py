import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch, ctypes as ct
import capi
from lerc_amd import api, synth
O = capi.oracle(); P = capi.product()
codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
for (r, c) in ((4096, 4096), (4104, 4096), (4608, 4096), (8192, 2048), (8192, 512)):
    x = synth.c3_uint16(r, c, device="cuda:0")
    torch.cuda.synchronize()
    xh = x.cpu().numpy()
    out = torch.empty(x.numel() * 2 + (1 << 20), dtype=torch.uint8, device="cuda:0")
    rc, nb = api.encode_device(codec, x, 0.0, out)
    rc2, b2 = O.encode(xh, 0)
    rc3, b3 = P.encode(xh, 0)
    # a second, different generator: numpy noise
    print(r, c, "device-api", nb, "host-api", len(b3), "oracle", len(b2), flush=True)
rng = np.random.default_rng(1)
for (r, c) in ((4096, 512), (4104, 512), (8192, 512), (8200, 512)):
    xh = (1500 + 3 * rng.standard_normal((r, c))).astype(np.uint16)
    rc2, b2 = O.encode(xh, 0); rc3, b3 = P.encode(xh, 0)
    print("numpy", r, c, "host-api", len(b3), "oracle", len(b2), b3 == b2, flush=True)
