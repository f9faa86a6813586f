"""lerc_amd -- MI355X-native LERC (Lerc2 v6) encode / decode.

Python host side of the C ABI in include/lerc_amd.h.  It mirrors the reference's own Python binding
(OtherLanguages/Python/lerc/_lerc.py: encode / decode / getLercBlobInfo / getLercDataRanges ...,
same argument meaning and return tuples) on top of liblerc_amd.so, and adds the device-pointer calls
used with torch tensors.  There is no CPU codec in this package: without the HIP library or without a
GPU every call fails loudly.
"""
from .api import (  # noqa: F401
    LercError,
    build_info,
    computeCompressedSize,
    decode,
    decode_device,
    encode,
    encode_device,
    getLercBlobInfo,
    getLercDataRanges,
    library_path,
    load_library,
)

__all__ = ["encode", "decode", "computeCompressedSize", "getLercBlobInfo", "getLercDataRanges", "encode_device",
           "decode_device", "load_library", "library_path", "build_info", "LercError"]
