"""C++ host adapter (manta_amd/host/manta_amd.hpp: the reference's own object interface over the C ABI)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _build(lib_dir, lib_name, out):
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", out, os.path.join(CPP, "test_host_adapter.cpp"), "-L" + lib_dir,
                           "-l" + lib_name, "-Wl,-rpath," + lib_dir])
    return out


def test_host_adapter_on_emulator(emu):
    exe = _build(os.path.join(ROOT, "tests", "emu"), "manta_amd_emu", os.path.join(CPP, "test_host_adapter_emu"))
    subprocess.check_call([exe])


@pytest.mark.gpu
def test_host_adapter_on_gpu(gpu):
    exe = _build(os.path.join(ROOT, "manta_amd"), "manta_amd", os.path.join(CPP, "test_host_adapter_gpu"))
    subprocess.check_call([exe])
