"""The C-ABI library loads and exports every symbol include/manta_amd.h declares (no compute, no GPU needed)."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "manta_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(manta_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_by_hip_library():
    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "manta_amd", "libmanta_amd.so"))
    syms = declared_symbols()
    assert len(syms) >= 5
    for s in syms:
        assert hasattr(lib, s), s


def test_library_contains_gfx950_code_objects():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", os.path.join(ROOT, "manta_amd", "libmanta_amd.so")],
                         capture_output=True, text=True).stdout
    assert ".hip_fatbin" in out


def test_product_package_has_no_cpu_path():
    """without a GPU, creating a context must fail loudly instead of computing anything on the host"""
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    from manta_amd import Lib, MantaError
    with pytest.raises(MantaError):
        Lib()
