"""Soak on the GPU box for the round-6 paths: rasters of random shapes (mostly ragged) with random flat rectangles (constant and zero blocks in runs), float32 / uint16 /
int32 / int16 / float64, through the product's device API and the oracle; blob and pixels compared, the serving tier counted.
    gpurun -- 'python tools/fuzz_flat_ragged.py [seed] [seconds]'"""
import sys, os, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch, capi, cases
from lerc_amd import api
O = capi.oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120
dev = torch.device("cuda:0")
codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
t0 = time.time(); n = 0; bad = 0; forms0 = codec.decode_forms()
TD = {np.float32: torch.float32, np.int32: torch.int32, np.int16: torch.int16, np.float64: torch.float64}
while time.time() - t0 < budget:
    dt = [np.float32, np.float32, np.uint16, np.int32, np.int16, np.float64][int(rng.integers(0, 6))]
    r, c = int(rng.integers(16, 1400)), int(rng.integers(16, 2600))
    if rng.random() < 0.25: r -= r % 8; c -= c % 8
    r = max(r, 8); c = max(c, 8)
    x = cases.terrain(r, c, rng, amp=float(rng.choice([5, 50, 500])), base=float(rng.choice([0, 100, 1000])), sigma=float(rng.choice([0.3, 1.5, 3])))
    for _ in range(int(rng.integers(0, 6))):
        i0, j0 = int(rng.integers(0, r)), int(rng.integers(0, c))
        x[i0:i0 + int(rng.integers(1, 400)), j0:j0 + int(rng.integers(1, 1200))] = float(rng.choice([0, 7, 1017.25, 1500, 33000.5]))
    if rng.random() < 0.3: x[:, -int(rng.integers(1, 40)):] = 12.0      # flat up to the edge column
    if rng.random() < 0.3: x[-int(rng.integers(1, 20)):, :] = 0.0       # and the last rows
    x = cases._cast(x, dt)
    e = float(rng.choice([0.001, 0.01, 0.5])) if np.dtype(dt).kind == 'f' else float(rng.choice([0, 0, 1]))
    r0, b0 = O.encode(x, e)
    if dt == np.uint16:
        xt = torch.from_numpy(x.view(np.int16)).to(dev).view(torch.uint16)
    else:
        xt = torch.from_numpy(x).to(dev)
    out = torch.empty(x.nbytes + (1 << 16), dtype=torch.uint8, device=dev)
    y = torch.zeros_like(xt)
    rc, nb = api.encode_device(codec, xt, e, out)
    blob = out[:nb].cpu().numpy().tobytes() if rc == 0 else b""
    ok = rc == r0 and (rc != 0 or blob == bytes(b0))
    if ok and rc == 0:
        rc2 = api.decode_device(codec, out, nb, y)
        torch.cuda.synchronize()
        d = O.decode(b0)[1].reshape(x.shape)
        yh = y.view(torch.int16).cpu().numpy().view(np.uint16) if dt == np.uint16 else y.cpu().numpy()
        ok = rc2 == 0 and np.array_equal(np.ascontiguousarray(d).view(np.uint8), np.ascontiguousarray(yh).view(np.uint8))
    n += 1
    if not ok:
        bad += 1
        print("MISMATCH", np.dtype(dt).name, (r, c), e, rc, r0, codec.last_note(), codec.last_error())
f1 = codec.decode_forms()
print("cases", n, "mismatches", bad, "forms", [b - a for a, b in zip(forms0, f1)], "paths", codec.path_counters(), "refusals", codec.decode_refusals())
