"""ffpic_amd -- MI355X (gfx950) back-end for the post-entropy reconstruction stage
of the ffpic image decoder.

The product is the C-ABI shared library ``ffpic_amd/libffpic_hip.so`` (sources in
``ffpic_amd/csrc``, declarations in ``include/ffpic_hip.h``).  This package is the
thin Python host side used by the tests and bench.py: a ctypes binding
(:mod:`ffpic_amd.capi`), an operator-level mirror of the reference interface
(:mod:`ffpic_amd.ops`) and the synthetic input generator (:mod:`ffpic_amd.synth`).
There is no CPU fallback: every operator raises if the HIP library or a gfx950
device is missing.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
