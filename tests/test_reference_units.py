"""Bit-exact checks of the building blocks against the reference's own translation units that compile on their own
(random.c, decomposition.c, parallel.c -> oracle/_ref/libcomd_ref.so, built by oracle/Makefile from /root/reference).
Skipped when the prebuilt library is absent (it is git-ignored but travels to the GPU box)."""
import ctypes
import os

import numpy as np
import pytest

REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libcomd_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libcomd_ref.so not built (reference checkout absent)")


@pytest.fixture(scope="module")
def ref():
    L = ctypes.CDLL(REF)
    L.lcg61.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    L.lcg61.restype = ctypes.c_double
    L.gasdev.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    L.gasdev.restype = ctypes.c_double
    L.mkSeed.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    L.mkSeed.restype = ctypes.c_uint64
    return L


def _streams(L, names, ids):
    lcg, gas, mk = (getattr(L, n) for n in names)
    out = []
    for gid in ids:
        for site in (123, 457):
            s = ctypes.c_uint64(mk(gid, site))
            seed0 = s.value
            vals = [lcg(ctypes.byref(s)) for _ in range(4)] + [gas(ctypes.byref(s)) for _ in range(3)]
            out.append((seed0, s.value, tuple(vals)))
    return out


IDS = [0, 1, 2, 3, 31999, 2047999, 16383999, 67108863, 2**31 - 1, 2**32 - 1]


def test_oracle_rng_is_the_reference_rng(ref, orc):
    got = _streams(orc.lib(), ("oracle_lcg61", "oracle_gasdev", "oracle_mkSeed"), IDS)
    assert got == _streams(ref, ("lcg61", "gasdev", "mkSeed"), IDS)


def test_product_rng_is_the_reference_rng(ref, pkg):
    host = pkg.lib_host()
    host.lcg61.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    host.lcg61.restype = ctypes.c_double
    host.gasdev.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    host.gasdev.restype = ctypes.c_double
    host.mkSeed.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    host.mkSeed.restype = ctypes.c_uint64
    assert _streams(host, ("lcg61", "gasdev", "mkSeed"), IDS) == _streams(ref, ("lcg61", "gasdev", "mkSeed"), IDS)
