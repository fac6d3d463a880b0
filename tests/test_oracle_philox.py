"""Pins the counter-based RNG: Random123 known-answer vectors for Philox4x32-10
(Salmon et al., SC'11; kat_vectors of the Random123 distribution), for the oracle and for the
product's device header (through the host harness)."""
import ctypes as C

import numpy as np

import util as U
from util import O

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_oracle_philox_kat():
    for ctr, key, want in KAT:
        got = O.philox4x32(*ctr, *key)
        assert tuple(int(x) for x in got) == want


def test_product_philox_kat():
    lib = U.emul_lib()
    out = (C.c_uint32 * 4)()
    for ctr, key, want in KAT:
        lib.emul_philox(*[C.c_uint32(c) for c in ctr], *[C.c_uint32(k) for k in key], out)
        assert tuple(out) == want


def test_u01_range_and_exactness():
    x = np.array([0, 255, 256, 0xFFFFFFFF], np.uint32)
    u = O.u01(x, np.float32)
    assert u[0] == 0 and u[1] == 0 and u[2] == np.float32(2.0 ** -24) and u[3] < 1.0
    assert (O.u01(x, np.float64) == u.astype(np.float64)).all()


def test_sincos_product_equals_oracle_bitwise_and_accurate():
    lib = U.emul_lib()
    th = np.concatenate([np.linspace(-np.pi, np.pi, 200001), np.array([0.0, np.pi / 2, -np.pi / 2, 9.3, -9.3])]).astype(
        np.float32)
    s = np.empty_like(th)
    c = np.empty_like(th)
    lib.emul_sincos(C.c_void_p(th.ctypes.data), C.c_int(th.size), C.c_void_p(s.ctypes.data), C.c_void_p(c.ctypes.data))
    so, co = O.sincos(th, np.float32)
    assert (s.view(np.uint32) == so.view(np.uint32)).all() and (c.view(np.uint32) == co.view(np.uint32)).all()
    assert np.abs(s - np.sin(th.astype(np.float64))).max() < 2e-7
    assert np.abs(c - np.cos(th.astype(np.float64))).max() < 2e-7
