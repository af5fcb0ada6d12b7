"""CPU oracle for the Marlin FFT spectral hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement of the reference's algorithm for the hot path named in
BASELINE.json (`north_star`).  Nothing in the product (`marlin_amd/`, `include/`) may
import, call, link or execute it; only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` do, and there only as the checker / the reported CPU
baseline.

Why torch: the reference performs *all* arithmetic through libTorch/ATen calls
(README.md:3,28 of the reference); it is a third-party dependency that the reference
does not pin (it takes "all its dependencies provided by MOOSE", README.md:29).  This
image ships libTorch 2.10.0 (CPU/MKL kernels); the Python API dispatches to the same
ATen CPU kernels as the C++ API the reference calls, so calling the same ops in the same
order *is* the reference's CPU path for these functions.  The reference itself cannot be
compiled here (every TU needs MOOSE/libMesh headers and `moose/` is an empty submodule).

Pinning (see tests/test_oracle_golden.py, runs with -m "not gpu"):
  * cahnhilliard.h5 gold  (reference test/tests/cahnhilliard/tests:46-57, abs_tol 1e-13)
  * cahnhilliard.rank0001.h5 gold (2-rank FFT_SLAB, tests:58-70)
  * map_to_aux_3d.e (Exodus; cahnhilliard.i in 3-D, 5^3: nodal c, elemental mu; tests:13-22)
  * mech3d.h5 / mech.h5 gold (test/tests/mechanics/tests:2-21, abs_tol 1e-10): F_*, disp_* (ComputeDisplacements), sV
  * test/tests/solvers/gold/*.csv: diagonal_* (ABM orders 1-4, AM corrector), coupled_* (AdamsBashforthMoultonCoupled),
    nl_coupled_*, etdrk4_diffusion_rmse
  * cahnhilliard/gold/sharp.e, houli.e (Exodus: explicit Euler Cahn-Hilliard with DeAliasingTensor SHARP / HOULI, 1000 substeps)
  * rotating_grain_secant.h5 (SecantSolver + SwiftHohenbergLinear + iteration-adaptive dt, abs_tol 1e-10)
  * typed_tensors/gradient.h5 (GradientTensor), backandforth / gradient(_square) CSV gold, ConjugateGradientTest iteration counts.
NOT pinned (parity unpinned): class BroydenSolver -- the reference ships no regression test, golden vector or fixture for
it; it restates src/tensor_solver/BroydenSolver.C line by line and is only compared with the HIP kernels.
Likewise unpinned: SecantSolver.add_predictor (LinearTensorPredictor; no reference test uses a predictor), quasistatic_elasticity /
elastic_chemical_potential / class CoupledPFMech (FFTQuasistaticElasticity.C, FFTElasticChemicalPotential.C):
their only input file, test/tests/tensor_compute/coupled_pf_mech.i, is not part of any test spec and has no gold data.

All file:line citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import torch

F64 = torch.float64
C128 = torch.complex128


# --------------------------------------------------------------------------------------
# Domain: axes, reciprocal axes, k^2, k-grid, FFT service
# --------------------------------------------------------------------------------------
def _align(t: torch.Tensor, dim: int, ndim: int) -> torch.Tensor:
    """DomainAction::align (src/actions/DomainAction.C:1406-1434): make a 1-D axis broadcastable."""
    shape = [1] * ndim
    shape[dim] = t.numel()
    return t.reshape(shape)


class Domain:
    """Serial (`parallel_mode = NONE`) restatement of DomainAction's math service.

    src/actions/DomainAction.C:226-338 (gridChanged), :853-867 (fftSerial), :1049-1063 (ifft),
    :1479-1509 (k-grid, k-square).  `slab_c2c=True` reproduces the reciprocal axes of FFT_SLAB
    mode (all axes `fftfreq`, full c2c transform, :279-281) for a *global* (all ranks
    concatenated) field.
    """

    def __init__(self, dim: int, n: Sequence[int], mx: Sequence[float], mn: Sequence[float] = (0.0, 0.0, 0.0),
                 slab_c2c: bool = False):
        self.dim = dim
        self.n = [int(n[d]) if d < dim else 1 for d in range(3)]
        self.min = [float(mn[d]) for d in range(3)]
        self.max = [float(mx[d]) if d < dim else 1.0 for d in range(3)]
        self.slab_c2c = slab_c2c
        self.dx = [(self.max[d] - self.min[d]) / self.n[d] for d in range(3)]      # :241
        self.shape = self.n[:dim]
        # real-space axes: linspace(min+dx/2, max-dx/2, n)  (:246-251)
        self.axis = []
        for d in range(3):
            if d < dim:
                a = torch.linspace(self.min[d] + self.dx[d] / 2.0, self.max[d] - self.dx[d] / 2.0, self.n[d], dtype=F64)
                self.axis.append(_align(a, d, dim))
            else:
                self.axis.append(torch.tensor([0.0], dtype=F64))
        # reciprocal axes (:259-293): rfftfreq on the last active axis in NONE mode
        self.kaxis = []
        for d in range(3):
            if d < dim:
                use_rfft = (d == dim - 1) and not slab_c2c
                f = (torch.fft.rfftfreq if use_rfft else torch.fft.fftfreq)(self.n[d], self.dx[d], dtype=F64)
                self.kaxis.append(_align(f * 2.0 * math.pi, d, dim))
            else:
                self.kaxis.append(torch.tensor([0.0], dtype=F64))
        self.rshape = [self.kaxis[d].numel() for d in range(dim)]

    # :1503-1509
    def k_square(self) -> torch.Tensor:
        return self.kaxis[0] * self.kaxis[0] + self.kaxis[1] * self.kaxis[1] + self.kaxis[2] * self.kaxis[2]

    # :1479-1501
    def k_grid(self) -> torch.Tensor:
        if self.dim == 1:
            return self.kaxis[0]
        return torch.stack([self.kaxis[d].expand(self.rshape) for d in range(self.dim)], -1)

    # :853-867 (and the c2c stages of :869-938 when slab_c2c)
    def fft(self, t: torch.Tensor) -> torch.Tensor:
        axes = list(range(self.dim))
        if self.slab_c2c:
            return torch.fft.fftn(t, dim=axes)
        if self.dim == 1:
            return torch.fft.rfft(t, dim=0)
        if self.dim == 2:
            return torch.fft.rfft2(t, dim=(0, 1))
        return torch.fft.rfftn(t, dim=(0, 1, 2))

    # :1049-1063 (and :940-1019 when slab_c2c: ifft + torch::real)
    def ifft(self, t: torch.Tensor) -> torch.Tensor:
        if self.slab_c2c:
            return torch.real(torch.fft.ifftn(t, dim=list(range(self.dim))))
        if self.dim == 1:
            return torch.fft.irfft(t, self.shape[0], dim=0)
        if self.dim == 2:
            return torch.fft.irfft2(t, self.shape, dim=(0, 1))
        return torch.fft.irfftn(t, self.shape, dim=(0, 1, 2))

    def value_shape(self, extra: Sequence[int]) -> List[int]:
        return list(self.shape) + list(extra)

    # DomainAction::average (:1558-1574): mean over the spatial axes
    def average(self, t: torch.Tensor) -> torch.Tensor:
        return t.sum(dim=list(range(self.dim))) / float(torch.tensor(self.shape).prod().item())


def partition_helper(total: int, weights: Sequence[int]) -> List[int]:
    """DomainAction::partitionHepler (include/actions/DomainAction.h:247-280)."""
    ns: List[int] = []
    remaining = sum(weights)
    for w in weights:
        if remaining == 0:
            raise RuntimeError("Internal partitioning error. remaining_total_weight == 0")
        n = max((total * w) // remaining, 1)
        ns.append(n)
        remaining -= w
        if total < n:
            raise RuntimeError("Internal partitioning error.")
        total -= n
    ns[-1] += total
    return ns


# --------------------------------------------------------------------------------------
# FFT_PENCIL (src/actions/DomainAction.C:568-742 partitionPencils, :1021-1047 fftPencil / ifftPencil, :1105-1404 the staged
# exchanges).  The reference runs one MPI rank per block; here every rank's block is an entry of a Python list and an MPI
# message is an assignment between entries -- the arithmetic (torch rfft / fft / ifft / irfft per block, in the reference's
# order) and the data placement (who sends which slice to whom, where it lands) are the reference's.
# --------------------------------------------------------------------------------------
def pencil_factors(nranks: int, n: Sequence[int]):
    """:574-618 -- (pencil_y_partitions, pencil_z_partitions): py, pz >= 2, py <= min(ny, nx/2+1), pz <= min(nz, ny), smallest
    |py - pz|, first found wins; None when nothing fits (the reference's paramError)."""
    nxc = n[0] // 2 + 1
    best = None
    best_cost = None

    def consider(px, pz):
        nonlocal best, best_cost
        if px < 2 or pz < 2 or px > n[1] or px > nxc or pz > n[2] or pz > n[1]:
            return
        cost = abs(px - pz)
        if best is None or cost < best_cost:
            best, best_cost = (px, pz), cost

    max_divisor = max(2, int(math.sqrt(nranks)))
    for d in range(2, max_divisor + 1):
        if nranks % d == 0:
            consider(d, nranks // d)
            consider(nranks // d, d)
    return best


class PencilDomain:
    """The blocks of every rank of an FFT_PENCIL job on an nx x ny x nz grid (:620-698): rank r = (py, pz) = (r % Py, r // Py)
    holds real [nx][y block py][z block pz] and reciprocal [kx block py][ky block pz][nz]; kx = rfftfreq (the r2c transform runs
    along x, :282-284), ky, kz = fftfreq."""

    def __init__(self, n: Sequence[int], mx: Sequence[float], nranks: int, mn: Sequence[float] = (0.0, 0.0, 0.0)):
        self.n = [int(v) for v in n]
        self.nranks = nranks
        f = pencil_factors(nranks, self.n)
        if f is None:
            raise RuntimeError("FFT_PENCIL requires factoring the number of MPI ranks into two integers greater than one that fit the domain")
        self.Py, self.Pz = f
        self.dx = [(float(mx[d]) - float(mn[d])) / self.n[d] for d in range(3)]
        ones = lambda k: [1] * k
        self.y_counts = partition_helper(self.n[1], ones(self.Py))
        self.z_counts = partition_helper(self.n[2], ones(self.Pz))
        self.x_sizes = partition_helper(self.n[0] // 2 + 1, ones(self.Py))      # _pencil_x_sizes
        self.y2_sizes = partition_helper(self.n[1], ones(self.Pz))              # _pencil_stage2_y_sizes
        off = lambda c: [sum(c[:i]) for i in range(len(c))]
        self.y_off, self.z_off, self.x_off, self.y2_off = off(self.y_counts), off(self.z_counts), off(self.x_sizes), off(self.y2_sizes)
        self.kaxis = [torch.fft.rfftfreq(self.n[0], self.dx[0], dtype=F64) * 2.0 * math.pi,
                      torch.fft.fftfreq(self.n[1], self.dx[1], dtype=F64) * 2.0 * math.pi,
                      torch.fft.fftfreq(self.n[2], self.dx[2], dtype=F64) * 2.0 * math.pi]

    def real_slices(self, r):
        py, pz = r % self.Py, r // self.Py
        return (slice(0, self.n[0]), slice(self.y_off[py], self.y_off[py] + self.y_counts[py]),
                slice(self.z_off[pz], self.z_off[pz] + self.z_counts[pz]))

    def recip_slices(self, r):
        px, pyf = r % self.Py, r // self.Py
        return (slice(self.x_off[px], self.x_off[px] + self.x_sizes[px]), slice(self.y2_off[pyf], self.y2_off[pyf] + self.y2_sizes[pyf]),
                slice(0, self.n[2]))

    def split(self, g: torch.Tensor) -> List[torch.Tensor]:
        return [g[self.real_slices(r)].contiguous() for r in range(self.nranks)]

    # :1021-1034 with pencilStage1Forward (:1105-1180) and pencilStage2Forward (:1182-1256)
    def fft(self, blocks: List[torch.Tensor]) -> List[torch.Tensor]:
        Py, Pz = self.Py, self.Pz
        after_x = [torch.fft.rfft(b, dim=0) for b in blocks]
        stage1 = []
        for r in range(self.nranks):
            px, base = r % Py, (r // Py) * Py
            res = torch.empty((self.x_sizes[px], self.n[1], blocks[r].shape[2]), dtype=after_x[r].dtype)
            for py_src in range(Py):           # message from rank base + py_src: its slice [x_off[px] : +x_sizes[px]]
                src = base + py_src
                chunk = after_x[src][self.x_off[px]:self.x_off[px] + self.x_sizes[px]].contiguous()
                res[:, self.y_off[py_src]:self.y_off[py_src] + self.y_counts[py_src], :] = chunk
            stage1.append(res)
        after_y = [torch.fft.fft(t, dim=1) for t in stage1]
        out = []
        for r in range(self.nranks):
            px, yf = r % Py, r // Py
            res = torch.empty((self.x_sizes[px], self.y2_sizes[yf], self.n[2]), dtype=after_y[r].dtype)
            for z_src in range(Pz):            # message from rank z_src * Py + px: its slice [:, y2_off[yf] : +y2_sizes[yf]]
                src = z_src * Py + px
                chunk = after_y[src][:, self.y2_off[yf]:self.y2_off[yf] + self.y2_sizes[yf]].contiguous()
                res[:, :, self.z_off[z_src]:self.z_off[z_src] + self.z_counts[z_src]] = chunk
            out.append(torch.fft.fft(res, dim=2))
        return out

    # :1036-1047 with pencilStage2Inverse (:1258-1329) and pencilStage1Inverse (:1331-1404)
    def ifft(self, spec: List[torch.Tensor]) -> List[torch.Tensor]:
        Py, Pz = self.Py, self.Pz
        after_z = [torch.fft.ifft(t, dim=2) for t in spec]
        stage2 = []
        for r in range(self.nranks):
            px, zi = r % Py, r // Py
            res = torch.empty((self.x_sizes[px], self.n[1], self.z_counts[zi]), dtype=after_z[r].dtype)
            for py_src in range(Pz):           # message from rank py_src * Py + px: its z slice of THIS rank
                src = py_src * Py + px
                chunk = after_z[src][:, :, self.z_off[zi]:self.z_off[zi] + self.z_counts[zi]].contiguous()
                res[:, self.y2_off[py_src]:self.y2_off[py_src] + self.y2_sizes[py_src], :] = chunk
            stage2.append(res)
        after_y = [torch.fft.ifft(t, dim=1) for t in stage2]
        out = []
        for r in range(self.nranks):
            py, base = r % Py, (r // Py) * Py
            res = torch.empty((self.n[0] // 2 + 1, self.y_counts[py], after_y[r].shape[2]), dtype=after_y[r].dtype)
            for px_src in range(Py):           # message from rank base + px_src: its y slice of THIS rank
                src = base + px_src
                chunk = after_y[src][:, self.y_off[py]:self.y_off[py] + self.y_counts[py], :].contiguous()
                res[self.x_off[px_src]:self.x_off[px_src] + self.x_sizes[px_src]] = chunk
            out.append(torch.fft.irfft(res, n=self.n[0], dim=0))
        return out


# --------------------------------------------------------------------------------------
# Cahn-Hilliard operators
# --------------------------------------------------------------------------------------
def reciprocal_laplacian_factor(dom: Domain, factor: float) -> torch.Tensor:
    """src/tensor_computes/ReciprocalLaplacianFactor.C:28-31:  -k^2 * factor."""
    return -dom.k_square() * factor


def reciprocal_laplacian_square_factor(dom: Domain, factor: float) -> torch.Tensor:
    """src/tensor_computes/ReciprocalLaplacianSquareFactor.C:28-32:  k^2 * k^2 * factor."""
    k2 = dom.k_square()
    return k2 * k2 * factor


def float32_operators(dom: "Domain", mobility: float, kappa_factor: float):
    """Mbar = -k^2 M and Lbar = k^2 k^2 f as a float32 run of the reference builds them: every tensor of the run is created with
    MooseTensor::floatTensorOptions() (src/utils/MarlinUtils.C:39-44, selected in src/actions/DomainAction.C:81,201), so the
    reciprocal axes are fftfreq / rfftfreq in float32 times 2.0 times pi (DomainAction.C:284-293) and k^2 = kx*kx + ky*ky + kz*kz is
    accumulated in float32 (ReciprocalLaplacianFactor.C:28-31, ReciprocalLaplacianSquareFactor.C:28-32).  Returns (Mbar, Lbar)."""
    assert dom.dim == 3
    n, dx = dom.n, dom.dx
    kx = torch.fft.fftfreq(n[0], d=dx[0], dtype=torch.float32) * 2.0 * math.pi
    ky = torch.fft.fftfreq(n[1], d=dx[1], dtype=torch.float32) * 2.0 * math.pi
    kz = torch.fft.rfftfreq(n[2], d=dx[2], dtype=torch.float32) * 2.0 * math.pi
    k2 = kx.reshape(-1, 1, 1) * kx.reshape(-1, 1, 1) + ky.reshape(1, -1, 1) * ky.reshape(1, -1, 1) + kz.reshape(1, 1, -1) * kz.reshape(1, 1, -1)
    return -k2 * mobility, k2 * k2 * kappa_factor


def mu_double_well(c: torch.Tensor, A: float = 0.1) -> torch.Tensor:
    """d/dc [A*c^2*(c-1)^2] exactly as the reference's parser derives and evaluates it.

    Grammar/derivative rules: include/utils/MarlinExpressionParser.h:400-412,
    src/utils/MarlinExpressionParser.C:143-203 (differentiate), :50-141 (simplify);
    evaluation by aten ops src/utils/ParsedJITTensor.C:114-156.  Resulting tree:
        (A*(2*c)) * pow(c-1, 2)  +  (A*pow(c, 2)) * (2*(c-1))
    """
    cm1 = c - 1.0
    return (A * (2.0 * c)) * torch.pow(cm1, 2.0) + (A * torch.pow(c, 2.0)) * (2.0 * cm1)


def mu_pfhub(c: torch.Tensor, rho: float, ca: float, cb: float) -> torch.Tensor:
    """d/dc [rho*(c-ca)^2*(cb-c)^2] by the same mechanical rules (benchmarks/01_spinodal_decomposition/1a_solver.i:62-70).

    f = (rho * (c-ca)^2) * (cb-c)^2 (left-assoc product).  Product rule (f*g)' = f'*g + f*g':
      d[(rho*(c-ca)^2)] = rho * (2*(c-ca))            (power rule g*f^(g-1)*f', f' = 1 dropped, ^1 dropped)
      d[(cb-c)^2]       = 2*(cb-c) * (0-1) -> simplify folds (0-1) = -1:  (2*(cb-c))*-1
    """
    a = c - ca
    b = cb - c
    return (rho * (2.0 * a)) * torch.pow(b, 2.0) + (rho * torch.pow(a, 2.0)) * ((2.0 * b) * -1.0)


AB_BETA = [
    [1.0, 0.0, 0.0, 0.0, 0.0],
    [3.0 / 2.0, -1.0 / 2.0, 0.0, 0.0, 0.0],
    [23.0 / 12.0, -16.0 / 12.0, 5.0 / 12.0, 0.0, 0.0],
    [55.0 / 24.0, -59.0 / 24.0, 37.0 / 24.0, -9.0 / 24.0, 0.0],
    # AB5 row carries the reference's 190/720 (sic), src/tensor_solver/AdamsBashforthMoulton.C:72
    [190.0 / 720.0, -2774.0 / 720.0, 2616.0 / 720.0, -1274.0 / 720.0, 251.0 / 720.0],
]
AM_ALPHA = [
    [1.0, 0.0, 0.0, 0.0, 0.0],
    [0.5, 0.5, 0.0, 0.0, 0.0],
    [5.0 / 12.0, 8.0 / 12.0, -1.0 / 12.0, 0.0, 0.0],
    [9.0 / 24.0, 19.0 / 24.0, -5.0 / 24.0, 1.0 / 24.0, 0.0],
    [251.0 / 720.0, 646.0 / 720.0, -264.0 / 720.0, 106.0 / 720.0, -19.0 / 720.0],
]


@dataclass
class History:
    """TensorBuffer<T>::advanceState / getOldTensor (include/tensor_buffers/TensorBuffer.h:62-79,110-116)."""
    max_states: int
    old: List[torch.Tensor] = field(default_factory=list)

    def advance(self, u: torch.Tensor) -> None:
        if len(self.old) < self.max_states:
            self.old.append(None)  # resize(+1)
        if self.old:
            for i in range(len(self.old) - 1, 0, -1):
                self.old[i] = self.old[i - 1]
            self.old[0] = u


class CahnHilliardABM:
    """TensorSolver::computeBuffer + AdamsBashforthMoulton::substep for the single-variable CH system.

    src/tensor_solver/TensorSolver.C:93-109, src/tensor_solver/AdamsBashforthMoulton.C:60-101,
    compute group order src/tensor_computes/ComputeGroup.C:61-84 (mu, mubar, Mbarmubar, cbar),
    history rule src/problems/TensorProblem.C:451-472 (advanceState is a no-op while timeStep()<=1).
    `mu_fn` is the parsed chemical potential (see mu_double_well / mu_pfhub).
    """

    def __init__(self, dom: Domain, c0: torch.Tensor, M: float, kappa_factor: float,
                 mu_fn: Callable[[torch.Tensor], torch.Tensor], substeps: int, predictor_order: int = 2):
        self.dom = dom
        self.c = c0.clone()
        self.mu = torch.zeros_like(c0)
        self.Mbar = reciprocal_laplacian_factor(dom, M)
        self.Lbar = reciprocal_laplacian_square_factor(dom, kappa_factor)
        self.mu_fn = mu_fn
        self.substeps = substeps
        self.pred = predictor_order - 1                     # AdamsBashforthMoulton.C:48
        self.hist = History(max_states=self.pred)           # :55-56 (corrector_steps = 0 here)
        self.Nhat: Optional[torch.Tensor] = None
        self.cbar: Optional[torch.Tensor] = None
        self.time_step = 0
        self.order_log: List[int] = []
        self.dt_old: Optional[float] = None                 # FEProblem::dtOld(): the previous time step's dt
        self.dt_changed = False
        self._substep = 0

    def _advance_state(self) -> None:
        if self.time_step <= 1:                             # TensorProblem.C:455
            return
        if self.Nhat is not None:
            self.hist.advance(self.Nhat)

    def substep(self, sub_dt: float) -> None:
        # root compute group (ComputeGroup.C:61-84)
        self.mu = self.mu_fn(self.c)                        # ParsedCompute.C:216
        mubar = self.dom.fft(self.mu)                       # PerformFFT.C:37
        self.Nhat = self.Mbar * mubar                       # Mbarmubar = Mbar*mubar
        self.cbar = self.dom.fft(self.c)
        n_old = len(self.hist.old)
        # AdamsBashforthMoulton.C:75,88-91: "If dt changes between steps, we start at first order again"
        order = min(0 if (self._substep < self.pred and self.dt_changed) else n_old, self.pred)
        self.order_log.append(order)
        ubar = self.cbar + (sub_dt * AB_BETA[order][0]) * self.Nhat            # :94
        for i in range(order):
            ubar += (sub_dt * AB_BETA[order][i + 1]) * self.hist.old[i]        # :95-96
        ubar /= (1.0 - sub_dt * self.Lbar)                                     # :99
        self.c = self.dom.ifft(ubar)                                           # :101

    def step(self, dt: float) -> None:
        """One MOOSE time step: incrementStepOrReject -> advanceState, then TensorSolver::computeBuffer."""
        self.time_step += 1
        self.dt_changed = self.dt_old is not None and dt != self.dt_old     # _dt != _dt_old (TensorSolver.C:48)
        self.dt_old = dt
        self._advance_state()
        sub_dt = dt / self.substeps                         # TensorSolver.C:96
        for s in range(self.substeps):
            self._substep = s
            self.substep(sub_dt)
            if s < self.substeps - 1:                       # :104-105
                self._advance_state()


class CoupledPFMech:
    """test/tests/tensor_compute/coupled_pf_mech.i: Cahn-Hilliard with the elastic chemical potential of a homogeneous small-strain
    solid (eigenstrain e0*c), integrated by the legacy FFTSemiImplicit time integrator.  PARITY UNPINNED (the input is in no test
    spec and has no gold data).  Compute order = the dependency-resolved order of the [Solve] group: mu, mubar, cbar, qsmech
    (writes disp_*), mumechbar, mumech, Mbarmubar = Mbar*(mubar+mumechbar); then FFTSemiImplicit::computeBuffer
    (src/tensor_timeintegrators/FFTSemiImplicit.C:43-62) with history_size 1; history rule as CahnHilliardABM."""

    def __init__(self, dom: Domain, c0: torch.Tensor, M: float, kappa_factor: float, mu_fn, substeps: int, mu: float, lam: float,
                 e0: float):
        self.dom, self.c, self.mu_fn, self.substeps = dom, c0.clone(), mu_fn, substeps
        self.Mbar = reciprocal_laplacian_factor(dom, M)
        self.Lbar = reciprocal_laplacian_square_factor(dom, kappa_factor)
        self.lame_mu, self.lam, self.e0 = mu, lam, e0
        self.hist = History(max_states=1)
        self.Nhat = None
        self.disp = [torch.zeros_like(c0) for _ in range(3)]
        self.mumech = torch.zeros_like(c0)
        self.time_step = 0

    def _advance_state(self):
        if self.time_step <= 1:
            return
        if self.Nhat is not None:
            self.hist.advance(self.Nhat)

    def substep(self, sub_dt: float):
        mubar = self.dom.fft(self.mu_fn(self.c))
        cbar = self.dom.fft(self.c)
        self.disp = quasistatic_elasticity(self.dom, cbar, self.lame_mu, self.lam, self.e0)
        mumechbar = elastic_chemical_potential(self.dom, cbar, self.disp, self.lame_mu, self.lam, self.e0)
        self.mumech = self.dom.ifft(mumechbar)
        self.Nhat = self.Mbar * (mubar + mumechbar)
        if not self.hist.old:
            ubar = (cbar + sub_dt * self.Nhat) / (1.0 - sub_dt * self.Lbar)                                    # :51
        else:
            ubar = (cbar + sub_dt / 2.0 * (3.0 * self.Nhat - self.hist.old[0])) / (1.0 - sub_dt * self.Lbar)   # :58-60
        self.c = self.dom.ifft(ubar)

    def step(self, dt: float):
        self.time_step += 1
        self._advance_state()
        for s in range(self.substeps):
            self.substep(dt / self.substeps)
            if s < self.substeps - 1:
                self._advance_state()


def ch_substep_ops(c, Mbar, Lbar, Nhat_old, sub_dt, order, mu_fn, dom):
    """One bare substep (used for the CPU baseline timing and operator-level parity)."""
    mu = mu_fn(c)
    mubar = dom.fft(mu)
    Nhat = Mbar * mubar
    cbar = dom.fft(c)
    ubar = cbar + (sub_dt * AB_BETA[order][0]) * Nhat
    for i in range(order):
        ubar += (sub_dt * AB_BETA[order][i + 1]) * Nhat_old[i]
    ubar /= (1.0 - sub_dt * Lbar)
    return dom.ifft(ubar), Nhat, cbar, mu


# --------------------------------------------------------------------------------------
# de Geus mechanics
# --------------------------------------------------------------------------------------
def trans2(A2):
    return torch.einsum("...ij->...ji", A2)                 # src/utils/MarlinUtils.C:147-151


def ddot42(A4, B2):
    return torch.einsum("...ijkl,...lk->...ij", A4, B2)     # :153-157


def ddot44(A4, B4):
    return torch.einsum("...ijkl,...lkmn->...ijmn", A4, B4)  # :159-163


def dot22(A2, B2):
    return torch.einsum("...ij,...jk->...ik", A2, B2)       # :165-169


def dot24(A2, B4):
    return torch.einsum("...ij,...jkmn->...ikmn", A2, B4)   # :171-175


def dot42(A4, B2):
    return torch.einsum("...ijkl,...lm->...ijkm", A4, B2)   # :177-181


def dyad22(A2, B2):
    return torch.einsum("...ij,...kl->...ijkl", A2, B2)     # :183-187


def _unsqueeze0(t, ndim):
    for _ in range(ndim):
        t = t.unsqueeze(0)
    return t


class MechIdentities:
    """Identity tensors of FFTMechanics / HyperElasticIsotropic ctors (src/tensor_computes/FFTMechanics.C:49-55)."""

    def __init__(self, dim: int):
        ti = torch.eye(dim, dtype=F64)
        self.ti = ti
        self.I = _unsqueeze0(ti, dim)
        self.I4 = _unsqueeze0(torch.einsum("il,jk", ti, ti), dim)
        self.I4rt = _unsqueeze0(torch.einsum("ik,jl", ti, ti), dim)
        self.I4s = (self.I4 + self.I4rt) / 2.0
        self.II = dyad22(self.I, self.I)


def ghat4(dom: Domain) -> torch.Tensor:
    """Projection operator as the reference stores it (src/tensor_computes/FFTMechanics.C:74-84)."""
    dim = dom.dim
    ti = torch.eye(dim, dtype=F64)
    q = dom.k_grid()
    Q = dom.k_square().unsqueeze(-1).unsqueeze(-1)
    M = torch.where(Q == 0, 0.0, q.unsqueeze(-2) * q.unsqueeze(-1) / Q)
    M = M.unsqueeze(-3).unsqueeze(-1)
    delta_im = ti.unsqueeze(1).unsqueeze(1).expand(dim, dim, dim, dim)
    return (M * delta_im).to(C128)


def hyper_elastic_isotropic(dom: Domain, ids: MechIdentities, F, K, mu):
    """src/tensor_computes/HyperElasticIsotropic.C:42-52 -> (P, K4)."""
    vs = dom.value_shape([1, 1, 1, 1])
    C4 = K.reshape(vs) * ids.II + 2.0 * mu.reshape(vs) * (ids.I4s - 1.0 / 3.0 * ids.II)
    S = ddot42(C4, 0.5 * (dot22(trans2(F), F) - ids.I))
    P = dot22(F, S)
    K4 = dot24(S, ids.I4) + ddot44(ddot44(ids.I4rt, dot42(dot24(F, C4), trans2(F))), ids.I4rt)
    return P, K4


def conjugate_gradient_solve(A, b, x0=None, tol=1e-6, maxiter=0):
    """MooseTensor::conjugateGradientSolve with identity preconditioner (include/utils/MarlinUtils.h:55-131)."""
    x = x0.clone() if x0 is not None else torch.zeros_like(b)
    b_norm = torch.norm(b).item()
    if b_norm == 0.0:
        return x, 0, 0.0
    if not maxiter:
        maxiter = b.numel()
    r = b - A(x)
    z = r
    p = z.clone()
    rz_old = torch.sum(r * z).item()
    res_norm = float("nan")
    for k in range(maxiter):
        Ap = A(p)
        alpha = rz_old / torch.sum(p * Ap).item()
        x = x + alpha * p
        r = r - alpha * Ap
        res_norm = torch.norm(r).item()
        if res_norm <= tol * b_norm:
            return x, k + 1, res_norm
        z = r
        rz_new = torch.sum(r * z).item()
        beta = rz_new / rz_old
        p = z + beta * p
        rz_old = rz_new
    return x, maxiter, res_norm


@dataclass
class MechStats:
    newton_its: int = 0
    cg_its: List[int] = field(default_factory=list)


class FFTMechanicsOracle:
    """FFTMechanics::computeBuffer (src/tensor_computes/FFTMechanics.C:96-163) with HyperElasticIsotropic."""

    def __init__(self, dom: Domain, K, mu, l_tol, nl_rel_tol, nl_abs_tol, l_max_its=None, nl_max_its=100):
        self.dom = dom
        self.ids = MechIdentities(dom.dim)
        self.K, self.mu = K, mu
        self.Ghat4 = ghat4(dom)
        self.l_tol, self.nl_rel_tol, self.nl_abs_tol = l_tol, nl_rel_tol, nl_abs_tol
        self.l_max_its = l_max_its if l_max_its is not None else int(torch.tensor(dom.shape).prod().item())
        self.nl_max_its = nl_max_its
        self.r2_shape = dom.value_shape([dom.dim, dom.dim])
        self.P = None
        self.K4 = None

    def G(self, A2):
        return self.dom.ifft_batched(ddot42(self.Ghat4, self.dom.fft_batched(A2))).reshape(-1)

    def K_dF(self, dFm):
        return trans2(ddot42(self.K4, trans2(dFm.reshape(self.r2_shape))))

    def G_K_dF(self, dFm):
        return self.G(self.K_dF(dFm))

    def compute(self, F, applied: Optional[torch.Tensor]) -> Tuple[torch.Tensor, MechStats]:
        stats = MechStats()
        u = F
        self.P, self.K4 = hyper_elastic_isotropic(self.dom, self.ids, u, self.K, self.mu)
        if applied is not None:
            b = -self.G_K_dF(applied.expand(self.r2_shape))
            u = u + applied.expand(self.r2_shape)
        else:
            b = -self.G_K_dF(torch.zeros_like(F))
        Fn = torch.linalg.norm(u).item()
        iiter = 0
        dFm = torch.zeros_like(b)
        while True:
            dFm, its, _ = conjugate_gradient_solve(self.G_K_dF, b, dFm, self.l_tol, self.l_max_its)
            stats.cg_its.append(its)
            u = u + dFm.reshape(self.r2_shape)
            self.P, self.K4 = hyper_elastic_isotropic(self.dom, self.ids, u, self.K, self.mu)
            b = -self.G(self.P)
            anorm = torch.linalg.norm(dFm).item()
            rnorm = anorm / Fn
            if (rnorm < self.nl_rel_tol or anorm < self.nl_abs_tol) and iiter > 0:
                break
            iiter += 1
            if iiter > self.nl_max_its:
                raise RuntimeError("Exceeded the maximum number of nonlinear iterations without converging.")
        stats.newton_its = iiter + 1
        return u, stats


def small_strain_linear_elastic(dom: Domain, K, mu, E: torch.Tensor, l_tol: float, l_max_its: Optional[int] = None, closed_form: bool = False):
    """The 'small-strain linear-elastic RVE, Gamma-operator fixed point' of BASELINE configs[2].  PARITY UNPINNED: the reference has no
    small-strain solve (only FFTMechanics.C:96-163, finite strain); this is that routine's first linear system at F = I -- where
    HyperElasticIsotropic.C:42-52 gives S = 0 and K4 = C4 = K II + 2 mu (I4s - II/3) -- with sigma = C4 : eps in place of P(F):
    b = -G(C4 : E) (FFTMechanics.C:116-117), conjugateGradientSolve (MarlinUtils.h:55-123), eps = E + d_eps.
    Returns (eps, sigma, cg iterations).  closed_form: apply G through gamma_closed_form (sizes where Ghat4 does not fit)."""
    dim = dom.dim
    r2 = dom.value_shape([dim, dim])
    ident = torch.eye(dim, dtype=F64).expand(r2).contiguous()
    _, K4 = hyper_elastic_isotropic(dom, MechIdentities(dim), ident, K, mu)
    if closed_form:
        G = lambda A2: gamma_closed_form(dom, A2.reshape(r2)).reshape(-1)
    else:
        Gh = ghat4(dom)
        G = lambda A2: dom.ifft_batched(ddot42(Gh, dom.fft_batched(A2.reshape(r2)))).reshape(-1)
    K_d = lambda v: trans2(ddot42(K4, trans2(v.reshape(r2))))
    n_max = l_max_its if l_max_its is not None else int(torch.tensor(dom.shape).prod().item())
    b = -G(K_d(E.expand(r2)))
    x, its, _ = conjugate_gradient_solve(lambda v: G(K_d(v)), b, torch.zeros_like(b), l_tol, n_max)
    eps = E.expand(r2) + x.reshape(r2)
    return eps, K_d(eps), its


# batched transforms: trailing value dims are batch (DomainAction.C:859-863 applies
# rfftn over the leading `dim` axes of a [..., 3, 3] tensor)
def _fft_batched(self: Domain, t):
    axes = tuple(range(self.dim))
    if self.slab_c2c:                       # FFT_SLAB: full c2c transform (DomainAction.C:279-281, 869-938)
        return torch.fft.fftn(t, dim=axes)
    return torch.fft.rfftn(t, dim=axes)


def _ifft_batched(self: Domain, t):
    axes = tuple(range(self.dim))
    if self.slab_c2c:                       # ifftSlab returns torch::real (DomainAction.C:1016)
        return torch.real(torch.fft.ifftn(t, dim=axes))
    return torch.fft.irfftn(t, self.shape, dim=axes)


Domain.fft_batched = _fft_batched
Domain.ifft_batched = _ifft_batched


def macroscopic_shear(dom: Domain, F, t: float) -> torch.Tensor:
    """test/src/tensor_computes/MacroscopicShearTensor.C:31-41."""
    avg = dom.average(F)
    applied = torch.eye(dom.dim, dtype=F64)
    applied[0, 1] = applied[0, 1] + t
    return applied - avg


def compute_displacements(dom: Domain, F: torch.Tensor) -> torch.Tensor:
    """ComputeDisplacements::computeBuffer (src/tensor_computes/ComputeDisplacements.C:53-107) -> [(n+1)..., dim]."""
    dim = dom.dim
    Fbox = dom.average(F)
    Hbar = dom.fft_batched(F - Fbox)
    q = dom.k_grid() * (-1j)
    numer = torch.einsum("...ij,...j->...i", Hbar, q.to(torch.complex128))
    denom = dom.k_square().unsqueeze(-1)
    u_periodic_bar = torch.where(denom == 0, 0.0, numer / denom)
    X = torch.stack([a.expand(dom.shape) for a in dom.axis[:dim]], -1)            # DomainAction::updateXGrid (:1457-1477)
    u_aff = torch.einsum("ij,...j->...i", Fbox - torch.eye(dim, dtype=F64), X)
    u_periodic = dom.ifft_batched(u_periodic_bar)
    mode = {3: "trilinear", 2: "bilinear", 1: "linear"}[dim]
    return torch.nn.functional.interpolate((u_aff + u_periodic).movedim(-1, 0).unsqueeze(1), size=[n + 1 for n in dom.shape],
                                           mode=mode, align_corners=True).squeeze(1).movedim(0, -1)


def quasistatic_elasticity(dom: Domain, cbar: torch.Tensor, mu: float, lam: float, e0: float) -> List[torch.Tensor]:
    """FFTQuasistaticElasticity::computeBuffer (src/tensor_computes/FFTQuasistaticElasticity.C:46-104), 3-D.  PARITY UNPINNED
    (no gold data in the reference): statement by statement, incl. at::linalg_solve."""
    two_pi_i = torch.tensor(complex(0.0, 2.0 * math.pi), dtype=C128)
    ul = 2.0 * mu + lam
    kx, ky, kz = two_pi_i * dom.kaxis[0], two_pi_i * dom.kaxis[1], two_pi_i * dom.kaxis[2]
    Axx = ul * kx * kx + mu * ky * ky + mu * kz * kz
    s = Axx.shape
    Axy = ((lam + mu) * kx * ky).expand(s)
    Axz = ((lam + mu) * kx * kz).expand(s)
    Ayy = ul * ky * ky + mu * kx * kx + mu * kz * kz
    Ayz = ((lam + mu) * ky * kz).expand(s)
    Azz = ul * kz * kz + mu * kx * kx + mu * ky * ky
    Axx[0, 0, 0] = 1.0
    Ayy[0, 0, 0] = 1.0
    Azz[0, 0, 0] = 1.0
    e = 2.0 * e0 * cbar * (3.0 * lam + mu)
    e[0, 0, 0] = 0.0
    b = torch.stack([kx * e, ky * e, kz * e], -1)
    A = torch.stack([torch.stack([Axx, Axy, Axz], -1), torch.stack([Axy, Ayy, Ayz], -1), torch.stack([Axz, Ayz, Azz], -1)], -1)
    x = torch.linalg.solve(A, b)
    return [dom.ifft(x[..., i]) for i in range(3)]


def elastic_chemical_potential(dom: Domain, cbar: torch.Tensor, disp: Sequence[torch.Tensor], mu: float, lam: float,
                               e0: float) -> torch.Tensor:
    """FFTElasticChemicalPotential::computeBuffer (src/tensor_computes/FFTElasticChemicalPotential.C:47-61).  PARITY UNPINNED."""
    two_pi_i = torch.tensor(complex(0.0, 2.0 * math.pi), dtype=C128)
    kx, ky, kz = two_pi_i * dom.kaxis[0], two_pi_i * dom.kaxis[1], two_pi_i * dom.kaxis[2]
    ux, uy, uz = dom.fft(disp[0]), dom.fft(disp[1]), dom.fft(disp[2])
    return -e0 * (e0 * (9.0 * lam * cbar + mu * 6.0 * cbar) - (2.0 * mu + 3.0 * lam) * (kx * ux + ky * uy + kz * uz))


def von_mises_stress(stress: torch.Tensor, dim: int) -> torch.Tensor:
    """ComputeVonMisesStress::computeBuffer (src/tensor_computes/ComputeVonMisesStress.C:31-66)."""
    if dim == 3:
        xx, yy, zz = stress[..., 0, 0], stress[..., 1, 1], stress[..., 2, 2]
        xy, yz, zx = stress[..., 0, 1], stress[..., 1, 2], stress[..., 2, 0]
        term4 = 6 * (xy.pow(2) + yz.pow(2) + zx.pow(2))
        return torch.sqrt(0.5 * ((xx - yy).pow(2) + (yy - zz).pow(2) + (zz - xx).pow(2) + term4))
    xx, yy, xy = stress[..., 0, 0], stress[..., 1, 1], stress[..., 0, 1]
    return torch.sqrt(0.5 * ((xx - yy).pow(2) + 6 * xy.pow(2)))


def gamma_closed_form(dom: Domain, A2: torch.Tensor) -> torch.Tensor:
    """Closed form of G(A): out_ij = q_j (sum_k A^_ik q_k)/|q|^2, zero at q=0.

    Not a reference call sequence: used by tests to show the stored-Ghat4 form and the fused
    form agree (4e-14), which is what the HIP kernel implements.
    """
    Ah = dom.fft_batched(A2)
    q = dom.k_grid().to(C128)
    Q = dom.k_square()
    s = torch.einsum("...ik,...k->...i", Ah, q)
    invQ = torch.where(Q == 0, torch.zeros_like(Q), 1.0 / Q)
    out = torch.einsum("...i,...j->...ij", s, q) * invQ.unsqueeze(-1).unsqueeze(-1)
    return dom.ifft_batched(out)


# --------------------------------------------------------------------------------------
# General (multi-variable) Adams-Bashforth-Moulton solver with the Adams-Moulton corrector
# --------------------------------------------------------------------------------------
class SplitOperatorABM:
    """TensorSolver::computeBuffer + AdamsBashforthMoulton::substep for any number of variables, with the optional
    corrector (src/tensor_solver/AdamsBashforthMoulton.C:48-178, SplitOperatorBase.C:39-64, TensorSolver.C:93-109).

    `compute(state) -> None` is the root compute group: it reads the real-space buffers in `state` and (re)binds the
    reciprocal buffers; `variables` = list of (buffer, reciprocal_buffer, linear_reciprocal tensor | None,
    nonlinear_reciprocal).  History of the nonlinear reciprocal buffers follows TensorBuffer<T>::advanceState with the
    TensorProblem rule that nothing advances while timeStep() <= 1.
    """

    def __init__(self, dom: Domain, state: dict, compute: Callable[[dict], None], variables, substeps: int,
                 predictor_order: int = 2, corrector_order: int = 2, corrector_steps: int = 0):
        self.dom, self.state, self.compute, self.vars = dom, state, compute, variables
        self.substeps = substeps
        self.pred = predictor_order - 1
        self.corr = corrector_order - 1
        self.csteps = corrector_steps
        depth = max(self.pred, self.corr)                    # :55-56
        self.hist = {v[3]: History(max_states=depth) for v in variables}
        self.time_step = 0
        self.dt_old: Optional[float] = None
        self.dt_changed = False
        self._substep = 0

    def _advance_state(self):
        if self.time_step <= 1:
            return
        for name, h in self.hist.items():
            if name in self.state:
                h.advance(self.state[name])

    def substep(self, sub_dt: float):
        s = self.state
        self.compute(s)                                      # :63
        for (u, rb, L, N) in self.vars:
            old = self.hist[N].old
            order = min(0 if (self._substep < self.pred and self.dt_changed) else len(old), self.pred)    # :75,88-91
            ubar = s[rb] + (sub_dt * AB_BETA[order][0]) * s[N]
            for i in range(order):
                ubar += (sub_dt * AB_BETA[order][i + 1]) * old[i]
            if L is not None:
                ubar /= (1.0 - sub_dt * L)
            s[u] = self.dom.ifft(ubar)
        if self.csteps:                                      # :117-177
            ubar_n = [s[rb] for (_, rb, _, _) in self.vars]
            N_n = [s[N] for (_, _, _, N) in self.vars] if self.corr > 0 else None
            for _ in range(self.csteps):
                self.compute(s)
                for k, (u, rb, L, N) in enumerate(self.vars):
                    old = self.hist[N].old
                    order = min(1 if (self._substep < self.corr and self.dt_changed) else len(old) + 1, self.corr)    # :153-154
                    if order == 0:
                        continue
                    ubar = ubar_n[k] + (sub_dt * AM_ALPHA[order][0]) * s[N]
                    ubar += (sub_dt * AM_ALPHA[order][1]) * N_n[k]
                    for i in range(order - 1):
                        ubar += (sub_dt * AM_ALPHA[order][i + 2]) * old[i]
                    if L is not None:
                        ubar /= (1.0 - sub_dt * L)
                    s[u] = self.dom.ifft(ubar)

    def step(self, dt: float):
        self.time_step += 1
        self.dt_changed = self.dt_old is not None and dt != self.dt_old
        self.dt_old = dt
        self._advance_state()
        sub_dt = dt / self.substeps
        for k in range(self.substeps):
            self._substep = k
            self.substep(sub_dt)
            if k < self.substeps - 1:
                self._advance_state()


def brusselator_problem(n: int = 150, A: float = 1.0, B: float = 3.5):
    """test/tests/solvers/diagonal.i: 2-D n^2 on [0, 2 pi]^2, u0 = sin(x) sin(y), v0 = 0, Du = -1e-2 k^2, Dv = -1e-3 k^2."""
    dom = Domain(2, [n, n], [2.0 * math.pi, 2.0 * math.pi])
    state = {"u": (torch.sin(dom.axis[0]) * torch.sin(dom.axis[1])).expand(dom.shape).contiguous(),
             "v": torch.zeros(dom.shape, dtype=F64)}
    Du = reciprocal_laplacian_factor(dom, 1e-2)
    Dv = reciprocal_laplacian_factor(dom, 1e-3)

    def compute(s):
        s["u_bar"] = dom.fft(s["u"])
        s["v_bar"] = dom.fft(s["v"])
        u, v = s["u"], s["v"]
        s["source_u"] = (A - (B + 1.0) * u) + torch.pow(u, 2.0) * v           # 'A - (B+1)*u +u^2*v'
        s["source_u_bar"] = dom.fft(s["source_u"])
        s["source_v"] = B * u - torch.pow(u, 2.0) * v                          # 'B*u - u^2*v'
        s["source_v_bar"] = dom.fft(s["source_v"])

    variables = [("u", "u_bar", Du, "source_u_bar"), ("v", "v_bar", Dv, "source_v_bar")]
    return dom, state, compute, variables


class CoupledABM:
    """AdamsBashforthMoultonCoupled::substep (src/tensor_solver/AdamsBashforthMoultonCoupled.C:84-272) under
    TensorSolver::computeBuffer: Adams-Bashforth right-hand sides per variable (:118-138), the dense operator assembled by
    stacking the columns of each row and then the rows on a new LAST axis (:160-178; the solver therefore sees
    A[..., a, b] = delta_ab - dt * L_ba), the cast of the stacked complex right-hand side to the real dtype of L (:183,
    which keeps Re(rhs) only -- the gold files test/tests/solvers/gold/coupled_*.csv are reproduced only with this cast),
    at::linalg_solve (:187), inverse transforms (:190-192) and the optional Adams-Moulton corrector (:198-270), where an
    order-0 corrector re-solves with rhs = ubar_n.  `L[i][j]` = real tensor or None; `variables` as in SplitOperatorABM
    (their linear_reciprocal entry is the diagonal and is ignored here in favour of L)."""

    def __init__(self, dom: Domain, state: dict, compute: Callable[[dict], None], variables, L, substeps: int,
                 predictor_order: int = 2, corrector_order: int = 2, corrector_steps: int = 0,
                 real_rhs: bool = True, transposed: bool = True):
        self.dom, self.state, self.compute, self.vars, self.L = dom, state, compute, variables, L
        self.substeps = substeps
        self.pred, self.corr, self.csteps = predictor_order - 1, corrector_order - 1, corrector_steps
        self.real_rhs, self.transposed = real_rhs, transposed
        self.hist = {v[3]: History(max_states=max(self.pred, self.corr)) for v in variables}
        self.time_step = 0
        self.dt_old: Optional[float] = None
        self.dt_changed = False
        self._substep = 0

    _advance_state = SplitOperatorABM._advance_state
    step = SplitOperatorABM.step

    def solve(self, rhs, sub_dt):
        n = len(rhs)
        zeros = torch.zeros(self.dom.rshape, dtype=F64)
        rows = [torch.stack([self.L[i][j] if self.L[i][j] is not None else zeros for j in range(n)], -1) for i in range(n)]
        Lm = torch.stack(rows, -1)                            # [grid..., j, i]  (:176)
        if not self.transposed:
            Lm = Lm.transpose(-1, -2)
        A = torch.eye(n, dtype=F64) - sub_dt * Lm
        b = torch.stack(rhs, -1)
        if self.real_rhs:
            b = b.real.contiguous()                           # .to(base_dtype) of a complex tensor keeps the real part (:183)
        else:
            A = A.to(torch.complex128)
        return torch.unbind(torch.linalg.solve(A, b), -1)

    def _publish(self, sol):
        for (u, _, _, _), x in zip(self.vars, sol):
            self.state[u] = self.dom.ifft(x if x.is_complex() else x.to(torch.complex128))

    def substep(self, sub_dt: float):
        s = self.state
        self.compute(s)
        rhs = []
        for (u, rb, _, N) in self.vars:
            old = self.hist[N].old
            order = min(0 if (self._substep < self.pred and self.dt_changed) else len(old), self.pred)   # AdamsBashforthMoultonCoupled.C:110,134
            r = s[rb] + (sub_dt * AB_BETA[order][0]) * s[N]
            for i in range(order):
                r += (sub_dt * AB_BETA[order][i + 1]) * old[i]
            rhs.append(r)
        self._publish(self.solve(rhs, sub_dt))
        if self.csteps:
            ubar_n = [s[rb] for (_, rb, _, _) in self.vars]
            N_n = [s[N] for (_, _, _, N) in self.vars] if self.corr > 0 else None
            for _ in range(self.csteps):
                self.compute(s)
                rhs = []
                for k, (u, rb, _, N) in enumerate(self.vars):
                    old = self.hist[N].old
                    order = min(1 if (self._substep < self.corr and self.dt_changed) else len(old) + 1, self.corr)   # :222
                    if order == 0:
                        rhs.append(ubar_n[k])
                        continue
                    r = ubar_n[k] + (sub_dt * AM_ALPHA[order][0]) * s[N]
                    r += (sub_dt * AM_ALPHA[order][1]) * N_n[k]
                    for i in range(order - 1):
                        r += (sub_dt * AM_ALPHA[order][i + 2]) * old[i]
                    rhs.append(r)
                self._publish(self.solve(rhs, sub_dt))


def coupled_diffusion_problem(n: int = 150, nonlinear: bool = False):
    """test/tests/solvers/coupled.i (dense operator [[D1, D2], [D2, D1]], zero nonlinear terms, for
    AdamsBashforthMoultonCoupled) and nl_coupled.i (the same cross-diffusion written as nonlinear terms Du = D2*v_bar,
    Dv = D2*u_bar for the diagonal AdamsBashforthMoulton): 2-D n^2 on [0, 2 pi]^2, u0 = sin x sin y, v0 = cos x cos y."""
    dom = Domain(2, [n, n], [2.0 * math.pi, 2.0 * math.pi])
    state = {"u": (torch.sin(dom.axis[0]) * torch.sin(dom.axis[1])).expand(dom.shape).contiguous(),
             "v": (torch.cos(dom.axis[0]) * torch.cos(dom.axis[1])).expand(dom.shape).contiguous(),
             "zero": torch.zeros(dom.rshape, dtype=torch.complex128)}
    D1 = reciprocal_laplacian_factor(dom, 1e-2)
    D2 = reciprocal_laplacian_factor(dom, 1e-3)

    def compute(s):
        s["u_bar"] = dom.fft(s["u"])
        s["v_bar"] = dom.fft(s["v"])
        if nonlinear:
            s["Du"] = D2 * s["v_bar"]
            s["Dv"] = D2 * s["u_bar"]

    if nonlinear:
        variables = [("u", "u_bar", D1, "Du"), ("v", "v_bar", D1, "Dv")]
    else:
        variables = [("u", "u_bar", D1, "zero"), ("v", "v_bar", D1, "zero")]
    return dom, state, compute, variables, [[D1, D2], [D2, D1]]


class ETDRK4:
    """ETDRK4Solver::substep (src/tensor_solver/ETDRK4Solver.C:29-115) for the same variable tuples as SplitOperatorABM."""

    def __init__(self, dom: Domain, state: dict, compute: Callable[[dict], None], variables, substeps: int = 1):
        self.dom, self.state, self.compute, self.vars, self.substeps = dom, state, compute, variables, substeps
        self.sub_time = 0.0

    def substep(self, dt: float):
        s = self.state
        self.compute(s)

        def evaluate_nonlinear(stage):
            for (u, _, _, _), ub in zip(self.vars, stage):
                s[u] = self.dom.ifft(ub)
            self.compute(s)
            return [s[N] for (_, _, _, N) in self.vars]

        ubar_n = [s[rb] for (_, rb, _, _) in self.vars]
        N1 = [s[N] for (_, _, _, N) in self.vars]
        lin = [L if L is not None else torch.zeros_like(ub) for (_, _, L, _), ub in zip(self.vars, ubar_n)]
        E, Eh, p1, p2, p3, ub_b = [], [], [], [], [], []
        for i in range(len(self.vars)):
            Ldt = lin[i] * dt
            E.append(torch.exp(Ldt))
            Eh.append(torch.exp(Ldt / 2.0))
            denom = Ldt * Ldt * Ldt
            a1 = dt * (-4.0 - 3.0 * Ldt + E[i] * (4.0 - Ldt)) / denom
            a2 = dt * (2.0 + Ldt + E[i] * (-2.0 + Ldt)) / denom
            a3 = dt * (-4.0 - 3.0 * Ldt - Ldt * Ldt + E[i] * (4.0 - Ldt)) / denom
            zero = Ldt == 0.0
            if zero.any().item():
                dtt = torch.full_like(Ldt, dt)
                a1 = torch.where(zero, dtt, a1)
                a2 = torch.where(zero, dtt * dtt / 2.0, a2)
                a3 = torch.where(zero, dtt * dtt / 6.0, a3)
            p1.append(a1)
            p2.append(a2)
            p3.append(a3)
            ub_b.append(Eh[i] * ubar_n[i] + 0.5 * dt * N1[i])
        N2 = evaluate_nonlinear(ub_b)
        N3 = evaluate_nonlinear([Eh[i] * ubar_n[i] + 0.5 * dt * N2[i] for i in range(len(self.vars))])
        N4 = evaluate_nonlinear([E[i] * ubar_n[i] + dt * N3[i] for i in range(len(self.vars))])
        for i, (u, _, _, _) in enumerate(self.vars):
            ubar = E[i] * ubar_n[i] + p1[i] * N1[i] + 2.0 * p2[i] * (N2[i] + N3[i]) + p3[i] * N4[i]
            s[u] = self.dom.ifft(ubar)

    def step(self, dt: float):
        sub_dt = dt / self.substeps
        for _ in range(self.substeps):
            self.substep(sub_dt)
            self.sub_time += sub_dt


def dealiasing_tensor(dom: Domain, method: str, p: float = 16.0, alpha: float = 36.0) -> torch.Tensor:
    """DeAliasingTensor::computeBuffer (src/tensor_computes/DeAliasingTensor.C:37-65): SHARP 2/3-rule mask or the Hou-Li
    exponential filter on the reciprocal axes (unused axes are the 1-element tensor {0}: they never cut and add exp(0))."""
    ax = [dom.kaxis[d] if d < dom.dim else torch.zeros(1, dtype=F64) for d in range(3)]
    mx = [torch.max(torch.abs(a)).item() for a in ax]
    if method == "SHARP":
        return torch.where((torch.abs(ax[0]) > 2 * mx[0] / 3) | (torch.abs(ax[1]) > 2 * mx[1] / 3) | (torch.abs(ax[2]) > 2 * mx[2] / 3),
                           0.0, 1.0).reshape(dom.rshape)
    px = torch.pow(torch.abs(ax[0]) / (mx[0] if mx[0] else 1.0), p)
    py = torch.pow(torch.abs(ax[1]) / (mx[1] if mx[1] else 1.0), p)
    pz = torch.pow(torch.abs(ax[2]) / (mx[2] if mx[2] else 1.0), p)
    return torch.exp(-alpha * (px + py + pz)).reshape(dom.rshape)


def explicit_cahn_hilliard_substep(dom: Domain, c, Mbar, Mkappabarbar, smooth, sub_dt, A: float = 0.1):
    """one ForwardEulerSolver::substep (src/tensor_solver/ForwardEulerSolver.C:27-38) of test/tests/cahnhilliard/
    cahnhilliard_explicit_smooth.i: mu = f'(c); dc_dt_bar = smooth * (Mbar*mubar - Mkappabarbar*cbar); c = ifft(cbar + sub_dt *
    dc_dt_bar).  -> (c_new, mu)"""
    mu = mu_double_well(c, A)
    mubar, cbar = dom.fft(mu), dom.fft(c)
    dc_dt_bar = smooth * (Mbar * mubar - Mkappabarbar * cbar)
    return dom.ifft(cbar + sub_dt * dc_dt_bar), mu


# --------------------------------------------------------------------------------------
# Iterative solvers
# --------------------------------------------------------------------------------------
def swift_hohenberg_linear(dom: Domain, r: float, alpha: float) -> torch.Tensor:
    """SwiftHohenbergLinear::computeBuffer (src/tensor_computes/SwiftHohenbergLinear.C:35-39)."""
    k2 = dom.k_square()
    return r - alpha * alpha * (1.0 - k2) * (1.0 - k2)


class SecantSolver:
    """SecantSolver::substep under TensorSolver::computeBuffer (src/tensor_solver/SecantSolver.C:60-176): per variable a
    pointwise secant iteration in reciprocal space on R(u) = (N(u) + L u) dt + u_old - u, bootstrapped by a semi-implicit
    Euler step of length dt_epsilon; a failed solve restores the old solution.  No history.  `variables` as in
    SplitOperatorABM.  `iterations` / `converged` are what IterativeTensorSolverInterface exposes to the time stepper."""

    def __init__(self, dom: Domain, state: dict, compute: Callable[[dict], None], variables, substeps: int = 1,
                 max_iterations: int = 30, relative_tolerance: float = 1e-9, absolute_tolerance: float = 1e-9,
                 damping: float = 1.0, dt_epsilon: float = 1e-4):
        self.dom, self.state, self.compute, self.vars, self.substeps = dom, state, compute, variables, substeps
        self.max_it, self.rtol, self.atol, self.damping, self.eps = (max_iterations, relative_tolerance, absolute_tolerance,
                                                                     damping, dt_epsilon)
        self.iterations, self.converged = 0, True
        self.predictors: List[Tuple[str, float]] = []
        self.hist: dict = {}
        self.time_step = 0

    def add_predictor(self, buffer: str, scale: float = 1.0) -> None:
        """LinearTensorPredictor on a solver output (src/tensor_predictor/LinearTensorPredictor.C:19-46; history_size 2).  The
        reference has no test with a predictor: this part is NOT pinned by gold data."""
        self.predictors.append((buffer, scale))
        self.hist[buffer] = History(max_states=2)

    def _advance_state(self) -> None:                                 # TensorProblem.C:451-472
        if self.time_step <= 1:
            return
        for b, h in self.hist.items():
            h.advance(self.state[b])

    def _apply_predictors(self) -> None:                              # IterativeTensorSolverInterface.C:19-24
        for b, scale in self.predictors:
            old = self.hist[b].old
            if len(old) > 1:
                diff = old[0] - old[1]
                self.state[b] = self.state[b] + diff if scale == 1.0 else self.state[b] + diff * scale

    def substep(self, sub_dt: float):
        s, n = self.state, len(self.vars)
        self.compute(s)                                               # :73
        u_old, Rprev, uprev, R0 = [None] * n, [None] * n, [None] * n, [0.0] * n
        for i, (ub, rb, L, N) in enumerate(self.vars):
            u, Nn = s[rb], s[N]
            Rprev[i] = (Nn + L * u) * sub_dt if L is not None else Nn * sub_dt
            uprev[i] = u
            R0[i] = torch.norm(Rprev[i]).item()
            u_old[i] = u
            s[ub] = self.dom.ifft((u + self.eps * Nn) / (1.0 - self.eps * L) if L is not None else u + self.eps * Nn)
        self._apply_predictors()                                      # :100
        all_converged = False
        self.iterations = 0
        while self.iterations < self.max_it:                          # :112-165
            self.compute(s)
            all_converged = True
            for i, (ub, rb, L, N) in enumerate(self.vars):
                u, Nn = s[rb], s[N]
                R = (Nn + L * u) * sub_dt + u_old[i] - u if L is not None else Nn * sub_dt + u_old[i] - u
                dx, dy = u - uprev[i], R - Rprev[i]
                du = torch.where(dy != 0, -R * dx / dy, 0.0)
                uprev[i], Rprev[i] = u, R
                s[ub] = self.dom.ifft(u + du if self.damping == 1.0 else u + du * self.damping)
                Rn = torch.norm(R).item()
                if math.isnan(Rn):
                    all_converged = False
                    self.iterations = self.max_it
                    break
                all_converged = all_converged and (Rn < self.atol or Rn / R0[i] < self.rtol)
            if all_converged:
                self.converged = True
                break
            self.iterations += 1
        if not all_converged:
            for i, (ub, _, _, _) in enumerate(self.vars):
                s[ub] = self.dom.ifft(u_old[i])
            self.converged = False

    def step(self, dt: float):
        self.time_step += 1
        self._advance_state()
        for k in range(self.substeps):
            self.substep(dt / self.substeps)
            if k < self.substeps - 1:                                 # TensorSolver.C:104-105
                self._advance_state()


class BroydenSolver:
    """BroydenSolver::substep under TensorSolver::computeBuffer (src/tensor_solver/BroydenSolver.C:34-176): Broyden iteration per
    reciprocal grid point on the stacked residual of all variables with a persistent inverse-Jacobian approximation `M`
    [grid..., n, n] (complex, initial_jacobian_guess * I), the hard-wired half step (`u + sk * 0.5`, :128) and the
    |denom| > 1e-12 guard of the rank-one update (:158-161).  The reference has no regression test for this solver, so this
    restatement is NOT pinned by gold data (parity unpinned): it follows the source line by line."""

    def __init__(self, dom: Domain, state: dict, compute: Callable[[dict], None], variables, substeps: int = 1,
                 max_iterations: int = 5, relative_tolerance: float = 1e-9, absolute_tolerance: float = 1e-9,
                 initial_jacobian_guess: float = 1.0):
        self.dom, self.state, self.compute, self.vars, self.substeps = dom, state, compute, variables, substeps
        self.max_it, self.rtol, self.atol = max_iterations, relative_tolerance, absolute_tolerance
        n = len(variables)
        self.M = (torch.eye(n, dtype=torch.complex128) * initial_jacobian_guess).expand(list(dom.rshape) + [n, n])
        self.iterations, self.converged = 0, True

    def _stack(self):
        s = self.state
        u = torch.stack([s[rb] for (_, rb, _, _) in self.vars], -1)
        N = torch.stack([s[N_] for (_, _, _, N_) in self.vars], -1)
        L = torch.stack([Lv for (_, _, Lv, _) in self.vars], -1)
        return u, N, L

    def substep(self, sub_dt: float):
        s = self.state
        self.compute(s)
        u_old = torch.stack([s[rb] for (_, rb, _, _) in self.vars], -1)
        u, N, L = self._stack()
        R = (N + L * u) * sub_dt
        R0norm = torch.norm(R).item()
        self.iterations = 0
        while self.iterations < self.max_it:
            Rnorm = torch.norm(R).item()
            if math.isnan(Rnorm):
                raise RuntimeError("NAN!")
            if Rnorm < self.atol or Rnorm / R0norm < self.rtol:
                self.converged = True
                return
            sk = -torch.matmul(self.M, R.unsqueeze(-1))
            skT = sk.squeeze(-1).unsqueeze(-2)
            u_out = torch.unbind(u + sk.squeeze(-1) * 0.5, -1)
            for (ub, _, _, _), x in zip(self.vars, u_out):
                s[ub] = self.dom.ifft(x)
            self.compute(s)
            u, N, L = self._stack()
            Rnew = (N + L * u) * sub_dt + u_old - u
            yk = (Rnew - R).unsqueeze(-1)
            denom = torch.matmul(skT, yk)
            self.M = self.M + torch.where(torch.abs(denom) > 1e-12, torch.matmul((sk - torch.matmul(self.M, yk)), skT) / denom, 0.0)
            R = Rnew
            self.iterations += 1
        self.converged = False

    def step(self, dt: float):
        for _ in range(self.substeps):
            self.substep(dt / self.substeps)


class IterationAdaptiveDT:
    """TensorSolveIterationAdaptiveDT (src/timesteppers/TensorSolveIterationAdaptiveDT.C:66-88,162-175): the first step
    uses `dt`; afterwards dt_old grows by growth_factor when the solver's last iteration count is below min_iterations and
    shrinks by cutback_factor above max_iterations; MOOSE's Transient clamps to dtmax."""

    def __init__(self, dt: float, min_iterations: int, max_iterations: int, growth_factor: float = 2.0,
                 cutback_factor: float = 0.5, dtmax: float = 1e30):
        self.dt0, self.min_it, self.max_it, self.grow, self.cut, self.dtmax = (dt, min_iterations, max_iterations,
                                                                               growth_factor, cutback_factor, dtmax)
        self.dt_old = 0.0

    def next_dt(self, t_step: int, iterations: int) -> float:
        if t_step == 1:
            dt = self.dt0
        else:
            dt = self.dt_old
            if iterations < self.min_it:
                dt *= self.grow
            elif iterations > self.max_it:
                dt *= self.cut
        dt = min(dt, self.dtmax)
        self.dt_old = dt
        return dt
