"""manta_amd -- MI355X-native assemble+align hot path of Manta's GenerateSVCandidates.

The product is the C-ABI shared library ``libmanta_amd.so`` (HIP kernels for gfx950, built by
``__graft_entry__.build()`` / ``manta_amd/build.py``).  This Python package is only a thin ctypes binding used by
the tests and ``bench.py``; there is no CPU implementation and nothing here falls back to one.
"""
from ._capi import Lib, MantaError, default_library_path  # noqa: F401
