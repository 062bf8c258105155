"""The C-ABI library loads and exports every symbol include/dove_hip.h declares (no compute calls: CPU-safe)."""
import os
import re

import pytest

from dove_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return L.load()


def header_symbols():
    with open(os.path.join(ROOT, "include", "dove_hip.h")) as f:
        src = f.read()
    return sorted(set(re.findall(r"\b(dove_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/dove_hip.h but not exported"


def test_binding_covers_header():
    assert set(header_symbols()) == set(L.SIGNATURES) | set(L.PLAIN)


def test_version_and_error_channel(lib):
    assert lib.dove_abi_version() == 2
    # argument validation happens before any HIP call, so it is safe without a GPU
    rc = lib.dove_axpby(None, None, None, 0, 0, 1.0, 1.0, None)
    assert rc == -1 and b"axpby" in lib.dove_last_error()
