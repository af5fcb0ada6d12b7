"""HIP FFT service vs the oracle (DomainAction::fft / ifft), through the C ABI.  Needs a GPU."""
import os

import pytest
import torch

from oracle import marlin_oracle as mo
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

# the any-length path (odd / prime / mixed sizes, 1-D ... 3-D) and every planned length of fft_pow2.h once per plan family, each on the
# axis roles it can take (z: r2c kernels, x / y: strided passes) -- co-dimensions kept small: the oracle's CPU transforms dominate
SHAPES = [
    (9,), (10,), (16,), (127,), (200,),
    (7, 9), (8, 6), (20, 20), (33, 17), (64, 128), (200, 100), (256, 512), (1024, 64),
    (5, 7, 9), (6, 8, 4), (5, 6, 7), (16, 16, 16), (12, 20, 30), (32, 64, 128), (40, 40, 40),
    (100, 50, 64), (64, 200, 100), (100, 32, 400),                        # register-radix path, radix 10 / 5 / 2 plans
    (96, 96, 96), (192, 64, 100), (64, 96, 384),                          # radix 12 / 4 / 2 plans
    (250, 48, 40), (500, 32, 80), (144, 768, 48), (1000, 32), (32, 1000, 40),
    (2048, 64), (32, 4096), (2048, 32, 40),                               # long lines of 2-D problems (planned lengths only)
    (120, 90, 60), (150, 150), (240, 40, 150), (270, 300, 60), (180, 360), (450, 64, 90), (600, 40),   # radix-30 plans (2 x 3 x 5 lengths)
    (160, 40, 64), (64, 320, 32), (32, 48, 640), (1280, 64),              # radix-20 plans (2^a 5, a >= 5)
    (240, 120, 34), (150, 180, 32), (160, 240), (120, 160, 18), (180, 150),   # two-stage plans on the strided axes (fft_two.h)
    (72, 216, 40), (64, 432, 32), (576, 96), (864, 32), (1152, 48), (40, 800, 32), (288, 32, 216),   # further plain plans
]


def _ctx(shape, **kw):
    from marlin_amd.api import Context
    L = [1.0 + 0.5 * i for i in range(len(shape))]
    return Context(len(shape), list(shape), L, **kw), mo.Domain(len(shape), list(shape), L)


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_inverse_match_oracle(shape):
    ctx, dom = _ctx(shape)
    torch.manual_seed(len(shape) * 100 + shape[0])
    a = torch.rand(shape, dtype=torch.float64)
    ref = dom.fft(a)
    got = ctx.fft(a.cuda()).cpu()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 1e-13 * scale, (got - ref).abs().max().item()
    # inverse of an arbitrary (not hermitian-consistent) half spectrum must match irfftn too
    spec = torch.randn(ref.shape, dtype=torch.complex128)
    ref_r = dom.ifft(spec)
    got_r = ctx.ifft(spec.cuda()).cpu()
    assert (got_r - ref_r).abs().max().item() <= 1e-13 * max(1.0, ref_r.abs().max().item())
    # round trip (test/tests/tensor_compute/backandforth.i)
    back = ctx.ifft(ctx.fft(a.cuda())).cpu()
    assert (back - a).abs().max().item() <= 1e-14


def _planned_lengths():
    import re
    src = open(os.path.join(ROOT, "marlin_amd", "csrc", "fft_pow2.h")).read()
    return sorted({int(m) for m in re.findall(r"^MRL_PLAN\((\d+),", src, flags=re.M)})


@pytest.mark.parametrize("axis", ["z", "y", "x"])
def test_every_planned_length_on_every_axis(axis):
    """every length with a register-radix plan (MRL_PLAN in fft_pow2.h, 46 of them) as the contiguous axis (r2c / c2r kernels, staged
    twiddles), as y and as x (strided passes; two-stage plans where fft_two.h has one), forward and inverse against the oracle's
    transforms.  Round 5 found the 240-point z transform wrong (a miscompiled modulo in stage()): until then no test had that length on
    the contiguous axis."""
    from marlin_amd.api import Context
    lengths = _planned_lengths()
    assert len(lengths) >= 46 and 240 in lengths
    worst = []
    for n in lengths:
        if n > 1280 and axis != "z":
            continue                      # (the longest lines are 2-D / 1-D plans)
        shape = {"z": (32, n), "y": (32, n, 32), "x": (n, 32)}[axis]      # (every extent planned: no generic stages)
        L = [1.0 + 0.5 * i for i in range(len(shape))]
        ctx = Context(len(shape), list(shape), L)
        torch.manual_seed(n)
        x = torch.rand(shape, dtype=torch.float64)
        X = torch.fft.rfftn(x)
        got = ctx.fft(x.cuda()).cpu()
        e1 = (got - X).abs().max().item() / (X.abs().max().item() * 2e-15 * n)
        back = ctx.ifft(X.cuda().contiguous()).cpu()
        e2 = (back - x).abs().max().item() / 1e-14
        worst.append((max(e1, e2), n))
    bad = [(round(e, 2), n) for e, n in worst if e > 1.0]
    assert not bad, bad


def test_random_mixes_of_planned_lengths():
    """60 random 3-D grids whose three extents are drawn independently from the planned lengths (every plan family next to every other:
    uniform, radix-30 / radix-20, two-stage), forward and inverse against libTorch's transforms on the device (rocFFT: an independent
    implementation, here only as the checker)"""
    import random
    from marlin_amd.api import Context
    rng = random.Random(20261005)
    lengths = [n for n in _planned_lengths() if n <= 640]
    small = [n for n in lengths if n <= 100]
    bad = []
    for it in range(60):
        big_axis = rng.randrange(3)
        shape = [rng.choice(lengths) if a == big_axis else rng.choice(small) for a in range(3)]
        if (shape[0] * shape[1]) % 2:
            continue
        ctx = Context(3, shape, [1.0, 1.5, 2.0])
        g = torch.Generator(device="cuda").manual_seed(it)
        x = torch.rand(shape, dtype=torch.float64, device="cuda", generator=g)
        X = torch.fft.rfftn(x)
        e1 = (ctx.fft(x) - X).abs().max().item() / (X.abs().max().item() * 2e-15 * max(shape))
        e2 = (ctx.ifft(X.contiguous()) - x).abs().max().item() / 1e-14
        if max(e1, e2) > 1.0:
            bad.append((shape, e1, e2))
    assert not bad, bad


@pytest.mark.parametrize("shape", [(6, 8, 4), (16, 16, 16), (9, 10), (12, 10, 14)])
def test_value_major_batch(shape):
    """trailing value dimensions are batch (mechanics [n,n,n,3,3]); SURVEY A.2"""
    ctx, dom = _ctx(shape)
    d = len(shape)
    torch.manual_seed(5)
    a = torch.rand(list(shape) + [d, d], dtype=torch.float64)
    ref = dom.fft_batched(a)
    got = ctx.fft(a.cuda()).cpu()
    assert (got - ref).abs().max().item() <= 1e-13 * ref.abs().max().item()
    back = ctx.ifft(got.cuda()).cpu()
    assert (back - a).abs().max().item() <= 1e-14


def test_field_major_batch():
    ctx, dom = _ctx((8, 12, 10))
    a = torch.rand(3, 8, 12, 10, dtype=torch.float64)
    got = ctx.fft_fields(a.cuda()).cpu()
    for b in range(3):
        assert (got[b] - dom.fft(a[b])).abs().max().item() <= 1e-12
    back = ctx.ifft_fields(got.cuda()).cpu()
    assert (back - a).abs().max().item() <= 1e-14


def test_reciprocal_axis_bit_exact():
    from marlin_amd.api import Context
    ctx = Context(3, [12, 9, 10], [3.0, 2.0, 7.0])
    dom = mo.Domain(3, [12, 9, 10], [3.0, 2.0, 7.0])
    for d in range(3):
        assert torch.equal(ctx.reciprocal_axis(d), dom.kaxis[d].reshape(-1))


def test_errors_are_reported():
    from marlin_amd.api import Context, MarlinHipError
    with pytest.raises(MarlinHipError, match="Max coordinate must be larger"):
        Context(2, [8, 8], [0.0, 1.0])
    with pytest.raises(MarlinHipError, match="Unsupported mesh dimension"):
        Context(4, [8, 8, 8], [1.0, 1.0, 1.0])


@pytest.mark.parametrize("shape", [(64, 64, 64), (128, 32, 256), (64, 128, 64), (256, 64, 128), (64, 512, 64)])
def test_fast_path_pow2(shape):
    """power-of-two fast path (fft_pow2*.h) vs the oracle, incl. non-hermitian-consistent inverse input"""
    ctx, dom = _ctx(shape)
    torch.manual_seed(shape[0] + shape[2])
    a = torch.rand(shape, dtype=torch.float64)
    ref = dom.fft(a)
    got = ctx.fft(a.cuda()).cpu()
    assert (got - ref).abs().max().item() <= 1e-13 * ref.abs().max().item()
    spec = torch.randn(ref.shape, dtype=torch.complex128)
    ref_r = dom.ifft(spec)
    got_r = ctx.ifft(spec.cuda()).cpu()
    assert (got_r - ref_r).abs().max().item() <= 1e-13 * max(1.0, ref_r.abs().max().item())
    b = torch.rand((2,) + tuple(shape), dtype=torch.float64)
    gb = ctx.fft_fields(b.cuda())
    for i in range(2):
        assert (gb[i].cpu() - dom.fft(b[i])).abs().max().item() <= 1e-13 * ref.abs().max().item()
    assert (ctx.ifft_fields(gb).cpu() - b).abs().max().item() <= 1e-14
