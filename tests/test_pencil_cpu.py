"""FFT_PENCIL without a GPU: the oracle's restatement of DomainAction::partitionPencils / fftPencil / ifftPencil (src/actions/
DomainAction.C:568-742, 1021-1047, 1105-1404) against the serial torch transform, the library's host-side choice of the process grid
against the reference's rule, and the reference's 4-rank spec (test/tests/gradient/tests:21-29) through the oracle's staged transforms."""
import math

import pytest
import torch

from oracle import marlin_oracle as mo


@pytest.mark.parametrize("n,P", [([40, 40, 40], 4), ([16, 12, 10], 4), ([9, 8, 7], 6), ([12, 10, 8], 8), ([64, 64, 64], 8), ([7, 9, 11], 4)])
def test_oracle_pencil_stages_equal_the_serial_transform(n, P):
    """every rank's reciprocal block of the staged pencil transform == the same slice of fftn(rfft(x, dim 0), dims 1 2) of the global
    array (rounding only), and the inverse stages give the real blocks back; uneven partitions included (9 / 8 / 7 over 2 x 3)"""
    d = mo.PencilDomain(n, [2 * math.pi, 4 * math.pi, 6 * math.pi], P)
    assert d.Py * d.Pz == P and d.Py >= 2 and d.Pz >= 2
    torch.manual_seed(3)
    g = torch.rand(n, dtype=torch.float64)
    full = torch.fft.fftn(torch.fft.rfft(g, dim=0), dim=(1, 2))
    blocks = d.split(g)
    spec = d.fft(blocks)
    tol = 2e-15 * full.abs().max().item() * max(n)
    covered = torch.zeros(full.shape, dtype=torch.int32)
    for r in range(P):
        ks = d.recip_slices(r)
        assert list(spec[r].shape) == [s.stop - s.start for s in ks]
        assert (spec[r] - full[ks]).abs().max().item() <= tol
        covered[ks] += 1
    assert bool((covered == 1).all())          # the reciprocal blocks tile the half spectrum exactly once
    back = d.ifft(spec)
    for r in range(P):
        assert (back[r] - g[d.real_slices(r)]).abs().max().item() <= 1e-14
    # local reciprocal axes: rfftfreq along x (DomainAction.C:282-284), fftfreq along y and z
    assert d.kaxis[0].numel() == n[0] // 2 + 1 and d.kaxis[1].numel() == n[1] and d.kaxis[2].numel() == n[2]


def test_pencil_process_grid_follows_the_reference_rule():
    """partitionPencils (DomainAction.C:574-618): the library's host-side mrl_pencil_factors == the oracle's restatement for every
    rank count up to 64 on several grids, including the counts nothing fits (primes, 2 x 1): MRL_ERR_INVALID with the reference's text"""
    from marlin_amd import api
    for n in ([40, 40, 40], [9, 8, 7], [256, 256, 256], [4, 3, 2], [64, 8, 4]):
        for P in range(1, 65):
            want = mo.pencil_factors(P, n)
            if want is None:
                with pytest.raises(api.MarlinHipError) as e:
                    api.pencil_factors(P, n)
                assert e.value.code == -1 and "FFT_PENCIL requires factoring" in e.value.message
            else:
                assert api.pencil_factors(P, n) == want
    assert mo.pencil_factors(4, [40, 40, 40]) == (2, 2) and mo.pencil_factors(8, [40, 40, 40]) == (2, 4)


def test_gradient_spec_on_four_pencil_ranks_through_the_oracle():
    """test/tests/gradient/tests:21-29 (gradient_cpu_pencil): gradient.i with parallel_mode = FFT_PENCIL on 4 ranks; FFTGradient
    (src/tensor_computes/FFTGradient.C:36-40: ifft(fft(s) * i k_d)) per rank block through the staged transforms, the integral of
    |grad - analytic| summed over the ranks: round-off, like the gold value 7.65e-12 of gradient_out.csv"""
    from tests.conftest import load_golden
    gold = load_golden("fft_gold.npz")["gradient_out"][1, 1]
    n, L, P = [40, 40, 40], [2 * math.pi, 4 * math.pi, 6 * math.pi], 4
    d = mo.PencilDomain(n, L, P)
    ax = [torch.linspace(L[i] / n[i] / 2.0, L[i] - L[i] / n[i] / 2.0, n[i], dtype=torch.float64) for i in range(3)]
    X, Y, Z = torch.meshgrid(*ax, indexing="ij")
    s = torch.sin(X) + torch.sin(Y) + torch.sin(Z)
    spec = d.fft(d.split(s))
    total = 0.0
    exact = [torch.cos(X), torch.cos(Y), torch.cos(Z)]
    diffs = [torch.zeros(1, dtype=torch.float64) for _ in range(P)]
    for dirn in range(3):
        g_hat = []
        for r in range(P):
            ks = d.recip_slices(r)
            shape = [1, 1, 1]
            shape[dirn] = -1
            k = d.kaxis[dirn][ks[dirn]].reshape(shape)
            g_hat.append(spec[r] * (k * 1j))
        grads = d.ifft(g_hat)
        for r in range(P):
            diffs[r] = diffs[r] + (grads[r] - exact[dirn][d.real_slices(r)]).abs()
    cell = (L[0] / n[0]) * (L[1] / n[1]) * (L[2] / n[2])
    total = sum(float(x.sum()) for x in diffs) * cell
    assert 0.0 <= total <= 10.0 * gold
