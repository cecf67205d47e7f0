"""nafcodec_amd -- MI355X-native decode path for Nucleotide Archive Format files.

Python mirror of the reference's public surface for the decode path
(nafcodec-py/nafcodec/lib.pyi:18-67 and __init__.py:3-9): `Decoder`, `Record`, `open`.
All decoding happens on the GPU inside libnafgpu.so (include/nafgpu.h); there is no CPU path."""
from .decoder import Decoder, Record, open  # noqa: F401
from .encoder import Encoder  # noqa: F401
from ._ffi import NafError  # noqa: F401

__version__ = "0.1.0"
__all__ = ["Decoder", "Encoder", "Record", "open", "NafError", "trim_device_memory"]


def trim_device_memory(device=-1):
    """Give the device memory the library keeps from closed decoders (mapped ranges, small buffers) back to the driver
    (nafgpu_trim_device_memory; no counterpart in the reference)."""
    from . import _ffi as _f
    rc = _f.default().c.nafgpu_trim_device_memory(device)
    if rc != 0:
        raise OSError("nafgpu_trim_device_memory failed (%d)" % rc)
