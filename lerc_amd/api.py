"""ctypes binding of liblerc_amd.so (see include/lerc_amd.h).

Host-pointer functions follow the reference wrapper OtherLanguages/Python/lerc/_lerc.py:
  encode(npArr, nValuesPerPixel, bHasMask, npValidMask, maxZErr, nBytesHint)  (_lerc.py:333-470)
      -> (result, nBytesWritten, npBuffer)       nBytesHint == 0: size query -> (result, nBytesNeeded)
  decode(lercBlob)                                                               (_lerc.py:610-700)
      -> (result, npArr, npValidMask)
  getLercBlobInfo(lercBlob)                                                      (_lerc.py:520-560)
      -> (result, codecVersion, dataType, nValuesPerPixel, nCols, nRows, nBands, nValidPixels, blobSize, nMasks,
          zMin, zMax, maxZErrUsed, nUsesNoData)
  getLercDataRanges(lercBlob, nValuesPerPixel, nBands) -> (result, npMins, npMaxs)  (_lerc.py:565-600)
Array layout as in the reference: [nBands,] nRows, nCols [, nValuesPerPixel].
"""
import ctypes as ct
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_DT_NP = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.float32, np.float64]


class LercError(RuntimeError):
    pass


def library_path():
    # LERC_AMD_LIBRARY: another build of the same HIP sources (e.g. csrc/_probe/liblerc_amd_probe.so, the tuning build)
    return os.environ.get("LERC_AMD_LIBRARY") or os.path.join(_HERE, "csrc", "liblerc_amd.so")


def load_library():
    """Loads the HIP library; raises (never falls back) when it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise LercError(f"{path} not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU fallback)")
    lib = ct.CDLL(path, mode=ct.RTLD_LOCAL)
    u32p, dblp = ct.POINTER(ct.c_uint), ct.POINTER(ct.c_double)
    enc = [ct.c_void_p, ct.c_uint, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_void_p, ct.c_double]
    lib.lerc_computeCompressedSize.argtypes = enc + [u32p]
    lib.lerc_encode.argtypes = enc + [ct.c_void_p, ct.c_uint, u32p]
    lib.lerc_getBlobInfo.argtypes = [ct.c_void_p, ct.c_uint, u32p, dblp, ct.c_int, ct.c_int]
    lib.lerc_getDataRanges.argtypes = [ct.c_void_p, ct.c_uint, ct.c_int, ct.c_int, dblp, dblp]
    lib.lerc_decode.argtypes = [ct.c_void_p, ct.c_uint, ct.c_int, ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, ct.c_int,
                                ct.c_uint, ct.c_void_p]
    lib.lerc_amd_create.argtypes = [ct.c_void_p]
    lib.lerc_amd_create.restype = ct.c_void_p
    lib.lerc_amd_destroy.argtypes = [ct.c_void_p]
    lib.lerc_amd_set_stream.argtypes = [ct.c_void_p, ct.c_void_p]
    lib.lerc_amd_last_error.argtypes = [ct.c_void_p]
    lib.lerc_amd_last_error.restype = ct.c_char_p
    lib.lerc_amd_encode_device.argtypes = [ct.c_void_p] + enc + [ct.c_void_p, ct.c_uint, u32p]
    lib.lerc_amd_decode_device.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_uint, ct.c_int, ct.c_void_p, ct.c_int, ct.c_int,
                                           ct.c_int, ct.c_int, ct.c_uint, ct.c_void_p]
    lib.lerc_amd_encode_tiles_device.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_uint, ct.c_int, ct.c_int, ct.c_int, ct.c_double, ct.c_void_p,
                                                 ct.c_ulonglong, ct.c_void_p, ct.c_void_p, ct.POINTER(ct.c_ulonglong)]
    lib.lerc_amd_decode_tiles_device.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, ct.c_uint,
                                                 ct.c_void_p]
    lib.lerc_amd_encode_tiles_device_slots.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_uint, ct.c_int, ct.c_int, ct.c_int, ct.c_double, ct.c_void_p,
                                                       ct.c_ulonglong, ct.c_void_p]
    lib.lerc_amd_decode_tiles_device_slots.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_ulonglong, ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, ct.c_uint,
                                                       ct.c_void_p]
    lib.lerc_amd_build_info.restype = ct.c_char_p
    lib.lerc_amd_encode_device_async.argtypes = [ct.c_void_p] + enc + [ct.c_void_p, ct.c_uint, u32p]
    lib.lerc_amd_decode_device_async.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_uint, ct.c_int, ct.c_void_p, ct.c_int, ct.c_int,
                                                 ct.c_int, ct.c_int, ct.c_uint, ct.c_void_p, u32p]
    lib.lerc_amd_finish.argtypes = [ct.c_void_p, ct.c_uint, u32p]
    for n in ("lerc_computeCompressedSize", "lerc_encode", "lerc_getBlobInfo", "lerc_getDataRanges", "lerc_decode",
              "lerc_amd_encode_device", "lerc_amd_decode_device", "lerc_amd_encode_tiles_device", "lerc_amd_decode_tiles_device",
              "lerc_amd_encode_device_async", "lerc_amd_decode_device_async", "lerc_amd_finish", "lerc_amd_encode_tiles_device_slots",
              "lerc_amd_decode_tiles_device_slots"):
        getattr(lib, n).restype = ct.c_uint
    _LIB = lib
    return lib


def build_info():
    return load_library().lerc_amd_build_info().decode()


def _dt_code(dtype):
    dtype = np.dtype(dtype)
    for i, t in enumerate(_DT_NP):
        if np.dtype(t) == dtype:
            return i
    raise LercError(f"unsupported dtype {dtype}")


def _shape(arr, nValuesPerPixel):
    shape = list(arr.shape)
    n_depth = int(nValuesPerPixel)
    if n_depth > 1:
        if shape[-1] != n_depth:
            raise LercError("last axis must be nValuesPerPixel")
        shape = shape[:-1]
    if len(shape) == 2:
        return 1, shape[0], shape[1]
    if len(shape) == 3:
        return shape[0], shape[1], shape[2]
    raise LercError(f"unsupported array shape {arr.shape}")


def _mask_args(bHasMask, npValidMask, n_bands, n_rows, n_cols):
    if not bHasMask or npValidMask is None:
        return 0, None, None
    m = np.ascontiguousarray(npValidMask, dtype=np.uint8)
    n_masks = n_bands if (m.ndim == 3 and n_bands > 1 and m.shape[0] == n_bands) else 1
    if m.size != n_masks * n_rows * n_cols:
        raise LercError("mask shape does not match the raster")
    return n_masks, m.ctypes.data, m


def computeCompressedSize(npArr, nValuesPerPixel, bHasMask, npValidMask, maxZErr):
    lib = load_library()
    a = np.ascontiguousarray(npArr)
    n_bands, n_rows, n_cols = _shape(a, nValuesPerPixel)
    n_masks, mptr, _keep = _mask_args(bHasMask, npValidMask, n_bands, n_rows, n_cols)
    out = ct.c_uint(0)
    rc = lib.lerc_computeCompressedSize(a.ctypes.data, _dt_code(a.dtype), int(nValuesPerPixel), n_cols, n_rows, n_bands,
                                        n_masks, mptr, float(maxZErr), ct.byref(out))
    return rc, out.value


def encode(npArr, nValuesPerPixel, bHasMask, npValidMask, maxZErr, nBytesHint):
    if nBytesHint == 0:
        return computeCompressedSize(npArr, nValuesPerPixel, bHasMask, npValidMask, maxZErr)
    lib = load_library()
    a = np.ascontiguousarray(npArr)
    n_bands, n_rows, n_cols = _shape(a, nValuesPerPixel)
    n_masks, mptr, _keep = _mask_args(bHasMask, npValidMask, n_bands, n_rows, n_cols)
    buf = np.empty(int(nBytesHint), np.uint8)
    written = ct.c_uint(0)
    rc = lib.lerc_encode(a.ctypes.data, _dt_code(a.dtype), int(nValuesPerPixel), n_cols, n_rows, n_bands, n_masks, mptr,
                         float(maxZErr), buf.ctypes.data, int(nBytesHint), ct.byref(written))
    return rc, written.value, buf[:written.value]


def getLercBlobInfo(lercBlob):
    lib = load_library()
    b = np.frombuffer(lercBlob, np.uint8)
    info = (ct.c_uint * 11)()
    rng = (ct.c_double * 3)()
    rc = lib.lerc_getBlobInfo(b.ctypes.data, len(b), info, rng, 11, 3)
    if rc:
        return (rc,) + (0,) * 13
    return (rc, info[0], info[1], info[2], info[3], info[4], info[5], info[6], info[7], info[8], rng[0], rng[1], rng[2], info[10])


def getLercDataRanges(lercBlob, nValuesPerPixel, nBands):
    lib = load_library()
    b = np.frombuffer(lercBlob, np.uint8)
    n = int(nValuesPerPixel) * int(nBands)
    mins = (ct.c_double * n)()
    maxs = (ct.c_double * n)()
    rc = lib.lerc_getDataRanges(b.ctypes.data, len(b), int(nValuesPerPixel), int(nBands), mins, maxs)
    shape = (nBands, nValuesPerPixel)
    return rc, np.array(mins).reshape(shape), np.array(maxs).reshape(shape)


def decode(lercBlob):
    lib = load_library()
    r = getLercBlobInfo(lercBlob)
    if r[0]:
        return r[0], None, None
    _, _, dt, n_depth, n_cols, n_rows, n_bands, _, _, n_masks = r[:10]
    b = np.frombuffer(lercBlob, np.uint8)
    shape = ((n_bands,) if n_bands > 1 else ()) + (n_rows, n_cols) + ((n_depth,) if n_depth > 1 else ())
    out = np.empty(shape, _DT_NP[dt])
    mask = np.empty(((n_masks,) if n_masks > 1 else ()) + (n_rows, n_cols), np.uint8) if n_masks > 0 else None
    rc = lib.lerc_decode(b.ctypes.data, len(b), n_masks, mask.ctypes.data if mask is not None else None, n_depth, n_cols,
                         n_rows, n_bands, dt, out.ctypes.data)
    return rc, out, mask


# ---- device-pointer extension (torch tensors or raw device addresses) --------------------------------
class DeviceCodec:
    """One lerc_amd context bound to a HIP stream (None / 0 = the HIP default stream)."""

    def __init__(self, stream_ptr=None):
        self.lib = load_library()
        self.h = self.lib.lerc_amd_create(ct.c_void_p(stream_ptr) if stream_ptr else None)
        if not self.h:
            raise LercError("lerc_amd_create failed: no usable HIP device (this library has no CPU path)")

    def close(self):
        if self.h:
            self.lib.lerc_amd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def path_counters(self):
        """[encodes by the streaming kernels, encodes by the general kernels, decodes streaming, decodes general] of this context"""
        out = (ct.c_ulonglong * 4)()
        self.lib.lerc_amd_path_counters.argtypes = [ct.c_void_p, ct.POINTER(ct.c_ulonglong)]
        self.lib.lerc_amd_path_counters.restype = None
        self.lib.lerc_amd_path_counters(self.h, out)
        return list(out)

    def decode_forms(self):
        """bands / tiles of this context decoded by [-, discovery + decode in two launches, the walking one-launch decoder, the scanning decoder]"""
        out = (ct.c_ulonglong * 4)()
        self.lib.lerc_amd_decode_forms.argtypes = [ct.c_void_p, ct.POINTER(ct.c_ulonglong)]
        self.lib.lerc_amd_decode_forms.restype = None
        self.lib.lerc_amd_decode_forms(self.h, out)
        return list(out)

    def decode_refusals(self):
        """attempts thrown away on the way down the tiers: [the decode kernels refused the masked scan's block offsets, the masked scan handed
        a band on, a streaming decode tier handed a band on, -]"""
        out = (ct.c_ulonglong * 4)()
        self.lib.lerc_amd_decode_refusals.argtypes = [ct.c_void_p, ct.POINTER(ct.c_ulonglong)]
        self.lib.lerc_amd_decode_refusals.restype = None
        self.lib.lerc_amd_decode_refusals(self.h, out)
        return list(out)

    def last_note(self):
        """why the last call that left the streaming kernels (or went down a streaming tier) did so; "" if none did"""
        self.lib.lerc_amd_last_note.argtypes = [ct.c_void_p]
        self.lib.lerc_amd_last_note.restype = ct.c_char_p
        return self.lib.lerc_amd_last_note(self.h).decode()

    def last_error(self):
        return self.lib.lerc_amd_last_error(self.h).decode()

    def encode(self, d_data, dt_code, n_depth, n_cols, n_rows, n_bands, max_z_err, d_out, out_cap, d_mask=0, n_masks=0):
        written = ct.c_uint(0)
        rc = self.lib.lerc_amd_encode_device(self.h, d_data, dt_code, n_depth, n_cols, n_rows, n_bands, n_masks,
                                             d_mask or None, float(max_z_err), d_out or None, out_cap, ct.byref(written))
        return rc, written.value

    def encode_async(self, d_data, dt_code, n_depth, n_cols, n_rows, n_bands, max_z_err, d_out, out_cap, d_mask=0, n_masks=0):
        """-> (status, ticket): enqueued on the context's stream; finish(ticket) waits and returns (status, nBytes)"""
        ticket = ct.c_uint(0)
        rc = self.lib.lerc_amd_encode_device_async(self.h, d_data, dt_code, n_depth, n_cols, n_rows, n_bands, n_masks,
                                                   d_mask or None, float(max_z_err), d_out or None, out_cap, ct.byref(ticket))
        return rc, ticket.value

    def decode_async(self, d_blob, size_bound, dt_code, n_depth, n_cols, n_rows, n_bands, d_out, d_mask=0, n_masks=0):
        ticket = ct.c_uint(0)
        rc = self.lib.lerc_amd_decode_device_async(self.h, d_blob, size_bound, n_masks, d_mask or None, n_depth, n_cols, n_rows,
                                                   n_bands, dt_code, d_out, ct.byref(ticket))
        return rc, ticket.value

    def finish(self, ticket=0):
        n = ct.c_uint(0)
        rc = self.lib.lerc_amd_finish(self.h, int(ticket), ct.byref(n))
        return rc, n.value

    def decode(self, d_blob, blob_size, dt_code, n_depth, n_cols, n_rows, n_bands, d_out, d_mask=0, n_masks=0):
        return self.lib.lerc_amd_decode_device(self.h, d_blob, blob_size, n_masks, d_mask or None, n_depth, n_cols, n_rows,
                                               n_bands, dt_code, d_out)


_TORCH_DT = None


def _torch_dt_code(t):
    import torch
    global _TORCH_DT
    if _TORCH_DT is None:
        _TORCH_DT = {torch.int8: 0, torch.uint8: 1, torch.int16: 2, torch.int32: 4, torch.float32: 6, torch.float64: 7}
        if hasattr(torch, "uint16"):
            _TORCH_DT[torch.uint16] = 3
        if hasattr(torch, "uint32"):
            _TORCH_DT[torch.uint32] = 5
    return _TORCH_DT[t.dtype]


def encode_device(codec, tensor, max_z_err, out, n_depth=1):
    """tensor: CUDA(HIP) tensor [nRows, nCols] (or [nRows, nCols, nDepth]); out: uint8 CUDA tensor.  -> (status, nBytes)"""
    n_rows, n_cols = int(tensor.shape[0]), int(tensor.shape[1])
    return codec.encode(tensor.data_ptr(), _torch_dt_code(tensor), n_depth, n_cols, n_rows, 1, max_z_err, out.data_ptr(), out.numel())


def decode_device(codec, blob, n_bytes, out, n_depth=1):
    n_rows, n_cols = int(out.shape[0]), int(out.shape[1])
    return codec.decode(blob.data_ptr(), int(n_bytes), _torch_dt_code(out), n_depth, n_cols, n_rows, 1, out.data_ptr())


def encode_device_async(codec, tensor, max_z_err, out, n_depth=1):
    """Like encode_device, without the wait -> (status, ticket); codec.finish(ticket) -> (status, nBytes)."""
    n_rows, n_cols = int(tensor.shape[0]), int(tensor.shape[1])
    return codec.encode_async(tensor.data_ptr(), _torch_dt_code(tensor), n_depth, n_cols, n_rows, 1, max_z_err, out.data_ptr(), out.numel())


def decode_device_async(codec, blob, size_bound, out, n_depth=1):
    """Like decode_device, without the wait; size_bound: the blob's size or the capacity of its buffer."""
    n_rows, n_cols = int(out.shape[0]), int(out.shape[1])
    return codec.decode_async(blob.data_ptr(), int(size_bound), _torch_dt_code(out), n_depth, n_cols, n_rows, 1, out.data_ptr())


def encode_tiles_device(codec, tiles, max_z_err, arena):
    """tiles: CUDA(HIP) tensor [nTiles, nRows, nCols]; arena: uint8 CUDA tensor.  One blob per tile, each exactly what
    encode() makes of that tile.  -> (status, offsets uint64[nTiles], sizes uint32[nTiles], arena bytes used)"""
    n_tiles, n_rows, n_cols = (int(v) for v in tiles.shape)
    offsets = np.zeros(n_tiles, np.uint64)
    sizes = np.zeros(n_tiles, np.uint32)
    used = ct.c_ulonglong(0)
    rc = codec.lib.lerc_amd_encode_tiles_device(codec.h, tiles.data_ptr(), _torch_dt_code(tiles), n_cols, n_rows, n_tiles, float(max_z_err),
                                                arena.data_ptr(), arena.numel(), offsets.ctypes.data, sizes.ctypes.data, ct.byref(used))
    return rc, offsets, sizes, int(used.value)


def encode_tiles_device_slots(codec, tiles, max_z_err, slots, slot_bytes):
    """The same with a slot per tile: tile t's blob at slots[t * slot_bytes :] (slot_bytes: a multiple of 16), nothing is packed
    afterwards -- the way a caller of lerc_encode() hands every tile a buffer of its own.  -> (status, sizes uint32[nTiles])"""
    n_tiles, n_rows, n_cols = (int(v) for v in tiles.shape)
    assert slots.numel() >= n_tiles * slot_bytes
    sizes = np.zeros(n_tiles, np.uint32)
    rc = codec.lib.lerc_amd_encode_tiles_device_slots(codec.h, tiles.data_ptr(), _torch_dt_code(tiles), n_cols, n_rows, n_tiles, float(max_z_err),
                                                      slots.data_ptr(), int(slot_bytes), sizes.ctypes.data)
    return rc, sizes


def decode_tiles_device_slots(codec, slots, slot_bytes, sizes, out):
    n_tiles, n_rows, n_cols = (int(v) for v in out.shape)
    sizes = np.ascontiguousarray(sizes, np.uint32)
    return codec.lib.lerc_amd_decode_tiles_device_slots(codec.h, slots.data_ptr(), int(slot_bytes), sizes.ctypes.data, n_tiles, n_cols, n_rows,
                                                        _torch_dt_code(out), out.data_ptr())


def decode_tiles_device(codec, arena, offsets, sizes, out):
    """out: CUDA(HIP) tensor [nTiles, nRows, nCols] to fill from the blobs arena[offsets[t] : offsets[t] + sizes[t]]."""
    n_tiles, n_rows, n_cols = (int(v) for v in out.shape)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    sizes = np.ascontiguousarray(sizes, np.uint32)
    return codec.lib.lerc_amd_decode_tiles_device(codec.h, arena.data_ptr(), offsets.ctypes.data, sizes.ctypes.data, n_tiles, n_cols, n_rows,
                                                  _torch_dt_code(out), out.data_ptr())
