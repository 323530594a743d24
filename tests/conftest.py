import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import OracleLib
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return OracleLib()


@pytest.fixture(scope="session")
def reflib():
    from oracle_lib import RefLib, have_ref
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not have_ref():
        pytest.skip("oracle/_ref/libmanta_ref.so not built (reference sources unavailable)")
    return RefLib()


@pytest.fixture(scope="session")
def emu():
    """The product sources compiled against the lock-step wave emulator (tests/emu, test infrastructure)."""
    from manta_amd._capi import Lib
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", d])
    return Lib(path=os.path.join(d, "libmanta_amd_emu.so"))


@pytest.fixture(scope="session")
def gpu():
    """The real thing: manta_amd/libmanta_amd.so on cuda:0.  No fallback: missing library or device is an error."""
    from manta_amd._capi import Lib
    return Lib()
