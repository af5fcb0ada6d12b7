"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol include/marlin_hip.h
declares, and its host-only entry points (partition, reciprocal axes, error paths) follow the reference."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest
import torch

from oracle import marlin_oracle as mo
from tests.conftest import ROOT


def test_library_exports_every_declared_symbol():
    from marlin_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "marlin_hip.h")).read()
    declared = set(re.findall(r"\b(mrl_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in marlin_hip.h but not exported"
    assert set(_lib.SIGNATURES) == declared
    assert lib.mrl_abi_version() == 3


def test_partition_matches_reference_helper():
    from marlin_amd.api import partition
    for total, w in [(20, [1, 1]), (21, [1, 1, 1]), (22, [1, 1, 1]), (512, [1] * 8), (10, [3, 1]), (257, [1] * 8), (7, [5, 1, 1])]:
        assert partition(total, len(w), w) == mo.partition_helper(total, w)


@pytest.mark.parametrize("n,dx,rfft", [(20, 0.15, False), (20, 0.15, True), (9, 1.0 / 9, False), (9, 1.0 / 9, True),
                                       (256, 8 * math.pi / 200, True), (1, 1.0, False)])
def test_reciprocal_axis_bit_exact(n, dx, rfft):
    """k = 2*pi*fftfreq / rfftfreq with the reference's rounding sequence (DomainAction.C:268-293)"""
    from marlin_amd.api import reciprocal_axis
    f = (torch.fft.rfftfreq if rfft else torch.fft.fftfreq)(n, dx, dtype=torch.float64) * 2.0 * math.pi
    assert np.array_equal(np.array(reciprocal_axis(n, dx, rfft)), f.numpy())


def test_error_paths_without_gpu():
    from marlin_amd import _lib
    from marlin_amd._lib import MrlDomain
    lib = _lib.load()
    h = C.c_void_p()
    d = MrlDomain()
    d.dim = 4
    assert lib.mrl_ctx_create(C.byref(h), C.byref(d)) == -1
    assert b"Unsupported mesh dimension" in lib.mrl_last_error(None)
    d.dim = 2
    d.n[0], d.n[1] = 8, 8
    d.max[0], d.max[1] = 1.0, 0.0
    d.nranks, d.rank = 1, 0
    assert lib.mrl_ctx_create(C.byref(h), C.byref(d)) == -1
    assert b"Max coordinate" in lib.mrl_last_error(None)
    out = (C.c_int64 * 4)()
    assert lib.mrl_partition(2, 4, None, out) == -1
    if not torch.cuda.is_available():
        d.max[1] = 1.0
        d.device = -1
        rc = lib.mrl_ctx_create(C.byref(h), C.byref(d))
        assert rc == -3 and b"no HIP device" in lib.mrl_last_error(None)   # fails loudly: no CPU fallback
