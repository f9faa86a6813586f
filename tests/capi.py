"""ctypes harness over the stock 12-function LERC C ABI (reference: src/LercLib/include/Lerc_c_api.h:126-380).

The same class drives three different shared objects, all loaded RTLD_LOCAL because they export
identical symbol names:

  * ref()     -- oracle/_ref/libLercRef.so  : the real Esri/lerc reference, compiled by `make -C oracle ref`
  * oracle()  -- oracle/liblerc_oracle.so   : our CPU restatement (test infrastructure)
  * product() -- lerc_amd/csrc/liblerc_amd.so: the MI355X product library (needs a GPU to *run*)

argtypes follow the reference's own Python binding (OtherLanguages/Python/lerc/_lerc.py:277-312).
"""
import ctypes as ct
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DT_NP = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.float32, np.float64]


def dt_code(dtype):
    dtype = np.dtype(dtype)
    for i, t in enumerate(DT_NP):
        if np.dtype(t) == dtype:
            return i
    raise ValueError(dtype)


class LercLib:
    def __init__(self, path):
        self.path = path
        self.lib = ct.CDLL(path, mode=ct.RTLD_LOCAL)
        L = self.lib
        u8p, u32p, dblp = ct.POINTER(ct.c_ubyte), ct.POINTER(ct.c_uint), ct.POINTER(ct.c_double)
        enc_common = [ct.c_void_p, ct.c_uint, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_void_p, ct.c_double]
        L.lerc_computeCompressedSize.argtypes = enc_common + [u32p]
        L.lerc_encode.argtypes = enc_common + [ct.c_void_p, ct.c_uint, u32p]
        L.lerc_computeCompressedSize_4D.argtypes = enc_common + [u32p, ct.c_void_p, ct.c_void_p]
        L.lerc_encode_4D.argtypes = enc_common + [ct.c_void_p, ct.c_uint, u32p, ct.c_void_p, ct.c_void_p]
        L.lerc_computeCompressedSizeForVersion.argtypes = [ct.c_void_p, ct.c_int] + enc_common[1:] + [u32p]
        L.lerc_encodeForVersion.argtypes = [ct.c_void_p, ct.c_int] + enc_common[1:] + [ct.c_void_p, ct.c_uint, u32p]
        L.lerc_getBlobInfo.argtypes = [ct.c_void_p, ct.c_uint, u32p, dblp, ct.c_int, ct.c_int]
        L.lerc_getDataRanges.argtypes = [ct.c_void_p, ct.c_uint, ct.c_int, ct.c_int, dblp, dblp]
        dec_common = [ct.c_void_p, ct.c_uint, ct.c_int, ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, ct.c_int]
        L.lerc_decode.argtypes = dec_common + [ct.c_uint, ct.c_void_p]
        L.lerc_decode_4D.argtypes = dec_common + [ct.c_uint, ct.c_void_p, ct.c_void_p, ct.c_void_p]
        L.lerc_decodeToDouble.argtypes = dec_common + [ct.c_void_p]
        L.lerc_decodeToDouble_4D.argtypes = dec_common + [ct.c_void_p, ct.c_void_p, ct.c_void_p]
        for name in ("lerc_computeCompressedSize", "lerc_encode", "lerc_computeCompressedSize_4D", "lerc_encode_4D",
                     "lerc_computeCompressedSizeForVersion", "lerc_encodeForVersion", "lerc_getBlobInfo",
                     "lerc_getDataRanges", "lerc_decode", "lerc_decode_4D", "lerc_decodeToDouble",
                     "lerc_decodeToDouble_4D"):
            getattr(L, name).restype = ct.c_uint

    def path_counters(self):
        """lerc_amd only: (encode streaming, encode general, decode streaming, decode general) call counts of
        this thread's context behind the stock entry points."""
        out = (ct.c_ulonglong * 4)()
        self.lib.lerc_amd_path_counters.argtypes = [ct.c_void_p, ct.POINTER(ct.c_ulonglong)]
        self.lib.lerc_amd_path_counters.restype = None
        self.lib.lerc_amd_path_counters(None, out)
        return tuple(int(v) for v in out)

    def decode_forms(self):
        """lerc_amd only: bands / tiles decoded by (unused, the two-launch form, the walking one-launch decoder, the scanning decoder)"""
        out = (ct.c_ulonglong * 4)()
        self.lib.lerc_amd_decode_forms.argtypes = [ct.c_void_p, ct.POINTER(ct.c_ulonglong)]
        self.lib.lerc_amd_decode_forms.restype = None
        self.lib.lerc_amd_decode_forms(None, out)
        return tuple(int(v) for v in out)

    def decode_refusals(self):
        """lerc_amd only: attempts thrown away on the way down the tiers: (the decode kernels refused the masked scan's offsets, the masked
        scan handed a band on, a streaming decode tier handed a band on, unused)"""
        out = (ct.c_ulonglong * 4)()
        self.lib.lerc_amd_decode_refusals.argtypes = [ct.c_void_p, ct.POINTER(ct.c_ulonglong)]
        self.lib.lerc_amd_decode_refusals.restype = None
        self.lib.lerc_amd_decode_refusals(None, out)
        return tuple(int(v) for v in out)

    def last_note(self):
        self.lib.lerc_amd_last_note.argtypes = [ct.c_void_p]
        self.lib.lerc_amd_last_note.restype = ct.c_char_p
        return self.lib.lerc_amd_last_note(None).decode()

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _dims(arr, n_depth, n_bands):
        """arr is [nBands?][nRows][nCols][nDepth?] C-contiguous."""
        a = np.ascontiguousarray(arr)
        shape = list(a.shape)
        if n_bands > 1:
            assert shape[0] == n_bands
            shape = shape[1:]
        if n_depth > 1:
            assert shape[-1] == n_depth
            shape = shape[:-1]
        assert len(shape) == 2, shape
        return a, shape[0], shape[1]

    def compute_size(self, arr, max_z_err, n_depth=1, n_bands=1, mask=None, no_data=None):
        a, n_rows, n_cols = self._dims(arr, n_depth, n_bands)
        n_masks, mptr, m = self._mask(mask, n_bands)
        out = ct.c_uint(0)
        if no_data is None:
            rc = self.lib.lerc_computeCompressedSize(a.ctypes.data, dt_code(a.dtype), n_depth, n_cols, n_rows, n_bands,
                                                     n_masks, mptr, float(max_z_err), ct.byref(out))
        else:
            uses, vals = self._nodata(no_data, n_bands)
            rc = self.lib.lerc_computeCompressedSize_4D(a.ctypes.data, dt_code(a.dtype), n_depth, n_cols, n_rows,
                                                        n_bands, n_masks, mptr, float(max_z_err), ct.byref(out),
                                                        uses.ctypes.data, vals.ctypes.data)
        return rc, out.value

    @staticmethod
    def _mask(mask, n_bands):
        if mask is None:
            return 0, None, None
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        n_masks = n_bands if (m.ndim == 3 and m.shape[0] == n_bands and n_bands > 1) else 1
        return n_masks, m.ctypes.data, m

    @staticmethod
    def _nodata(no_data, n_bands):
        vals = np.zeros(n_bands, np.float64)
        uses = np.zeros(n_bands, np.uint8)
        nd = np.atleast_1d(np.asarray(no_data, dtype=object))
        for i in range(n_bands):
            v = nd[i] if len(nd) > 1 else nd[0]
            if v is not None:
                uses[i] = 1
                vals[i] = float(v)
        return uses, vals

    def encode(self, arr, max_z_err, n_depth=1, n_bands=1, mask=None, buf_size=None, no_data=None):
        """returns (status, blob bytes)"""
        a, n_rows, n_cols = self._dims(arr, n_depth, n_bands)
        n_masks, mptr, m = self._mask(mask, n_bands)
        if buf_size is None:
            rc, buf_size = self.compute_size(arr, max_z_err, n_depth, n_bands, mask, no_data)
            if rc != 0:
                return rc, b""
        buf = np.empty(max(int(buf_size), 1), np.uint8)
        written = ct.c_uint(0)
        if no_data is None:
            rc = self.lib.lerc_encode(a.ctypes.data, dt_code(a.dtype), n_depth, n_cols, n_rows, n_bands, n_masks, mptr,
                                      float(max_z_err), buf.ctypes.data, int(buf_size), ct.byref(written))
        else:
            uses, vals = self._nodata(no_data, n_bands)
            rc = self.lib.lerc_encode_4D(a.ctypes.data, dt_code(a.dtype), n_depth, n_cols, n_rows, n_bands, n_masks,
                                         mptr, float(max_z_err), buf.ctypes.data, int(buf_size), ct.byref(written),
                                         uses.ctypes.data, vals.ctypes.data)
        return rc, buf[:written.value].tobytes()

    def encode_for_version(self, arr, version, max_z_err, n_depth=1, n_bands=1, mask=None):
        """lerc_computeCompressedSizeForVersion + lerc_encodeForVersion; returns (status, size status, size, blob bytes)"""
        a, n_rows, n_cols = self._dims(arr, n_depth, n_bands)
        n_masks, mptr, m = self._mask(mask, n_bands)
        size = ct.c_uint(0)
        rc0 = self.lib.lerc_computeCompressedSizeForVersion(a.ctypes.data, int(version), dt_code(a.dtype), n_depth, n_cols,
                                                            n_rows, n_bands, n_masks, mptr, float(max_z_err), ct.byref(size))
        # (64 bytes of slack: the reference's codec 2 packer clears whole 32-bit words and so writes up to 3 bytes behind
        # the blob it announced, BitStuffer2.cpp:292-300 -- with an exact-size buffer that is a heap overrun)
        cap = (int(size.value) if rc0 == 0 else a.nbytes + 4096) + 64
        buf = np.empty(max(cap, 1), np.uint8)
        written = ct.c_uint(0)
        rc = self.lib.lerc_encodeForVersion(a.ctypes.data, int(version), dt_code(a.dtype), n_depth, n_cols, n_rows, n_bands,
                                            n_masks, mptr, float(max_z_err), buf.ctypes.data, cap, ct.byref(written))
        return rc, rc0, size.value, buf[:written.value].tobytes()

    def blob_info(self, blob):
        b = np.frombuffer(blob, np.uint8)
        info = (ct.c_uint * 11)()
        rng = (ct.c_double * 3)()
        rc = self.lib.lerc_getBlobInfo(b.ctypes.data, len(blob), info, rng, 11, 3)
        return rc, list(info), list(rng)

    def data_ranges(self, blob, n_depth, n_bands):
        b = np.frombuffer(blob, np.uint8)
        mins = (ct.c_double * (n_depth * n_bands))()
        maxs = (ct.c_double * (n_depth * n_bands))()
        rc = self.lib.lerc_getDataRanges(b.ctypes.data, len(blob), n_depth, n_bands, mins, maxs)
        return rc, list(mins), list(maxs)

    def decode(self, blob, want_masks=None, n_bands=None, to_double=False, with_nodata=False):
        """returns (status, array [nBands?][nRows][nCols][nDepth?], mask or None[, uses, vals])"""
        rc, info, _ = self.blob_info(blob)
        if rc != 0:
            return (rc, None, None) + ((None, None) if with_nodata else ())
        _, dt, n_depth, n_cols, n_rows, nb, _, _, n_masks_blob, _, _ = info
        if n_bands is None:
            n_bands = nb
        n_masks = n_masks_blob if want_masks is None else want_masks
        np_dt = np.float64 if to_double else DT_NP[dt]
        out = np.full((n_bands, n_rows, n_cols, n_depth), 0, np_dt)
        out.view(np.uint8)[...] = 0xCD    # the library must overwrite every byte
        mask = np.full((max(n_masks, 1), n_rows, n_cols), 0xCD, np.uint8) if n_masks > 0 else None
        b = np.frombuffer(blob, np.uint8)
        mptr = mask.ctypes.data if mask is not None else None
        uses = np.zeros(n_bands, np.uint8)
        vals = np.zeros(n_bands, np.float64)
        if to_double:
            if with_nodata:
                rc = self.lib.lerc_decodeToDouble_4D(b.ctypes.data, len(blob), n_masks, mptr, n_depth, n_cols, n_rows,
                                                     n_bands, out.ctypes.data, uses.ctypes.data, vals.ctypes.data)
            else:
                rc = self.lib.lerc_decodeToDouble(b.ctypes.data, len(blob), n_masks, mptr, n_depth, n_cols, n_rows,
                                                  n_bands, out.ctypes.data)
        elif with_nodata:
            rc = self.lib.lerc_decode_4D(b.ctypes.data, len(blob), n_masks, mptr, n_depth, n_cols, n_rows, n_bands, dt,
                                         out.ctypes.data, uses.ctypes.data, vals.ctypes.data)
        else:
            rc = self.lib.lerc_decode(b.ctypes.data, len(blob), n_masks, mptr, n_depth, n_cols, n_rows, n_bands, dt,
                                      out.ctypes.data)
        if with_nodata:
            return rc, out, mask, uses, vals
        return rc, out, mask


_cache = {}


def _load(key, path):
    if key not in _cache:
        if not os.path.exists(path):
            return None
        _cache[key] = LercLib(path)
    return _cache[key]


def ref():
    return _load("ref", os.path.join(ROOT, "oracle", "_ref", "libLercRef.so"))


def oracle():
    return _load("oracle", os.path.join(ROOT, "oracle", "liblerc_oracle.so"))


def product():
    return _load("product", os.path.join(ROOT, "lerc_amd", "csrc", "liblerc_amd.so"))


def sim():
    """The product sources compiled for the CPU SIMT emulator (tools/hipsim) -- kernel-logic tests only."""
    return _load("sim", os.path.join(ROOT, "tests", "_sim", "liblerc_amd_sim.so"))
