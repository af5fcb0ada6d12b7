"""Native ParsedCompute on the GPU: fused hiprtc kernels vs the same op sequence in libTorch (oracle) and vs finite
differences (the strategy of the reference's unit/src/ParsedTensorTest.C)."""
import math

import pytest
import torch

from oracle import marlin_oracle as mo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from marlin_amd.api import Context
    return Context(2, [12, 10], [2 * math.pi, 3.0])


def _pc(ctx, *a, **k):
    from marlin_amd.api import ParsedCompute
    return ParsedCompute(ctx, *a, **k)


def test_chemical_potentials_bit_exact(ctx):
    """the two free energies of the benchmark configs, derived symbolically, == the oracle's op sequence bit for bit"""
    torch.manual_seed(0)
    c = torch.rand(4097, dtype=torch.float64)
    mu = _pc(ctx, "0.1*c^2*(c-1)^2", inputs=["c"], derivatives=["c"])(c.cuda()).cpu()
    assert torch.equal(mu, mo.mu_double_well(c, 0.1))
    mu = _pc(ctx, "rho_s*(c-c_alpha)^2*(c_beta-c)^2", inputs=["c"], constants={"rho_s": 5.0, "c_alpha": 0.3, "c_beta": 0.7},
             derivatives=["c"])(c.cuda()).cpu()
    assert torch.equal(mu, mo.mu_pfhub(c, 5.0, 0.3, 0.7))


def test_brusselator_sources_bit_exact(ctx):
    """test/tests/solvers/diagonal.i:58-77"""
    torch.manual_seed(1)
    u, v = torch.rand(1000, dtype=torch.float64) * 3, torch.rand(1000, dtype=torch.float64)
    A, B = 1.0, 3.5
    su = _pc(ctx, "A - (B+1)*u +u^2*v", inputs=["u", "v"], constants={"A": A, "B": B})(u.cuda(), v.cuda()).cpu()
    sv = _pc(ctx, "B*u - u^2*v", inputs=["u", "v"], constants={"A": A, "B": B})(u.cuda(), v.cuda()).cpu()
    assert torch.equal(su, (A - (B + 1.0) * u) + torch.pow(u, 2.0) * v)
    assert torch.equal(sv, B * u - torch.pow(u, 2.0) * v)


def test_reciprocal_product_complex(ctx):
    """Mbarmubar = Mbar*mubar (examples/cahn_hilliard/cahnhilliard2.i:86-91): real x complex"""
    torch.manual_seed(2)
    M = torch.rand(777, dtype=torch.float64) - 0.5
    z = torch.complex(torch.rand(777, dtype=torch.float64), torch.rand(777, dtype=torch.float64))
    p = _pc(ctx, "Mbar*mubar", inputs=["Mbar", "mubar"], complex_inputs=["mubar"])
    assert p.is_complex
    assert torch.equal(p(M.cuda(), z.cuda()).cpu(), M * z)
    q = _pc(ctx, "(a*b + a/b - a)*2", inputs=["a", "b"], complex_inputs=["a", "b"])
    w = torch.complex(torch.rand(777, dtype=torch.float64) + 0.5, torch.rand(777, dtype=torch.float64))
    ref = (z * w + z / w - z) * 2
    assert (q(z.cuda(), w.cuda()).cpu() - ref).abs().max().item() <= 1e-14


def test_extra_symbols_real_and_reciprocal(ctx):
    """x, y, kx, ky, k2, t, pi (ParsedCompute.C:139-161) on the context's grids (diagonal.i:20-27: sin(x)*sin(y))"""
    dom = mo.Domain(2, [12, 10], [2 * math.pi, 3.0])
    n = 120
    got = _pc(ctx, "sin(x)*sin(y) + t*pi", extra_symbols=True)(count=n, time=0.25).cpu().reshape(12, 10)
    ref = torch.sin(dom.axis[0]) * torch.sin(dom.axis[1]) + 0.25 * math.pi
    assert (got - ref).abs().max().item() <= 2e-15
    nk = 12 * 6
    got = _pc(ctx, "-k2*0.2 + kx - 2*ky", extra_symbols=True, reciprocal=True)(count=nk).cpu().reshape(12, 6)
    ref = -dom.k_square() * 0.2 + dom.kaxis[0] - 2 * dom.kaxis[1]
    assert torch.equal(got, ref)
    spec = _pc(ctx, "i*kx*a", inputs=["a"], complex_inputs=["a"], extra_symbols=True, reciprocal=True)
    a = torch.complex(torch.rand(12, 6, dtype=torch.float64), torch.rand(12, 6, dtype=torch.float64))
    assert (spec(a.cuda()).cpu() - 1j * dom.kaxis[0] * a).abs().max().item() <= 1e-14


FUNCS = ["sin(a)", "cos(a)", "tan(a)", "sinh(a)", "cosh(a)", "tanh(a)", "asin(a/2)", "acos(a/2)", "atan(a)", "exp(a)", "log(a)",
         "log10(a)", "log2(a)", "sqrt(a)", "abs(a-0.5)", "a^2.5", "a^b", "a^(0-2)", "a^0.5", "a^3", "atan2(a, b)", "hypot(a, b)",
         "min(a, b)*max(a, b)", "if(a > b, a*a, b)", "a % 0.3", "floor(a*3) + ceil(b*3) + round(a*7) + trunc(b*5)"]


@pytest.mark.parametrize("expr", FUNCS)
def test_functions_match_torch(ctx, expr):
    torch.manual_seed(3)
    a = torch.rand(513, dtype=torch.float64) + 0.1
    b = torch.rand(513, dtype=torch.float64) + 0.1
    env = {"a": a, "b": b, "sin": torch.sin, "cos": torch.cos, "tan": torch.tan, "sinh": torch.sinh, "cosh": torch.cosh,
           "tanh": torch.tanh, "asin": torch.asin, "acos": torch.acos, "atan": torch.atan, "exp": torch.exp, "log": torch.log,
           "log10": torch.log10, "log2": torch.log2, "sqrt": torch.sqrt, "abs": torch.abs, "atan2": torch.atan2,
           "hypot": torch.hypot, "min": torch.minimum, "max": torch.maximum, "floor": torch.floor, "ceil": torch.ceil,
           "round": torch.round, "trunc": torch.trunc}
    py = expr.replace("^", "**")
    if expr.startswith("if("):
        ref = torch.where(a > b, a * a, b)
    elif "%" in expr:
        ref = torch.remainder(a, 0.3)
    else:
        ref = eval(py, {"__builtins__": {}}, env)
    got = _pc(ctx, expr, inputs=["a", "b"])(a.cuda(), b.cuda()).cpu()
    assert ((got - ref).abs() / ref.abs().clamp_min(1.0)).max().item() <= 1e-14


@pytest.mark.parametrize("expr", ["a^3*b - sin(a*b)", "exp(-a^2)/(1+b^2)", "sqrt(a*a+b*b)*log(a+2)", "tanh(a)*atan(b) + a^b",
                                  "hypot(a, b) + atan2(a, b)"])
def test_derivative_vs_finite_difference(ctx, expr):
    torch.manual_seed(4)
    a = (torch.rand(257, dtype=torch.float64) + 0.2).cuda()
    b = (torch.rand(257, dtype=torch.float64) + 0.2).cuda()
    f = _pc(ctx, expr, inputs=["a", "b"])
    for var, (da, db) in (("a", (1.0, 0.0)), ("b", (0.0, 1.0))):
        h = 1e-6
        fd = (f(a + h * da, b + h * db) - f(a - h * da, b - h * db)) / (2 * h)
        d = _pc(ctx, expr, inputs=["a", "b"], derivatives=[var])(a, b)
        assert (d - fd).abs().max().item() <= 1e-7 * max(1.0, fd.abs().max().item())


@pytest.mark.parametrize("shape", [(64, 64, 64), (12, 10, 9), (128, 64, 64), (120, 60, 64), (64, 48, 150), (160, 40, 240)])
def test_parsed_free_energy_in_ch_substep(shape):
    """MRL_FE_PARSED: the user's free energy, differentiated symbolically and compiled INTO the forward z pass on
    fast-path shapes (hiprtc instance of k_z_fwd), must reproduce the built-in families bit for bit (same tree).  Round 5: also on
    the planned-unfused path (radix-30 / radix-20 lengths on any axis, ch_planned.hip) -- before, a parsed free energy sent those
    grids to the any-length path"""
    from marlin_amd.api import Context, ParsedCompute, ch_params, FE_PFHUB
    L = [3.0, 2.0, 2.5]
    ctx = Context(3, list(shape), L)
    planned_unfused = any(n in (60, 120, 150, 160, 240) for n in shape)
    ctx.set_profiling(True)
    torch.manual_seed(9)
    c0 = (torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44).cuda()
    cases = [(ch_params(), ParsedCompute(ctx, "0.1*c^2*(c-1)^2", inputs=["c"], derivatives=["c"])),
             (ch_params(family=FE_PFHUB, coef=(5.0, 0.3, 0.7), mobility=5.0, kappa=-10.0),
              ParsedCompute(ctx, "rho_s*(c-c_alpha)^2*(c_beta-c)^2", inputs=["c"],
                            constants={"rho_s": 5.0, "c_alpha": 0.3, "c_beta": 0.7}, derivatives=["c"]))]
    for builtin, parsed in cases:
        pp = ch_params(mobility=builtin.mobility, kappa=builtin.kappa, parsed=parsed)
        res = []
        for prm in (builtin, pp):
            c, N0, N1 = c0.clone(), ctx.empty_hist(), ctx.empty_hist()
            c1, c2, mu = torch.empty_like(c), torch.empty_like(c), torch.empty_like(c)
            ctx.ch_substep(prm, c, c1, N0, [], 0, 1e-3)
            ctx.ch_substep(prm, c1, c2, N1, [N0], 1, 1e-3, mu=mu)
            res.append((c2.cpu(), N1.cpu(), mu.cpu()))
        for a, b in zip(*res):
            assert torch.equal(a, b)
        if planned_unfused:   # both parameter sets ran the planned kernels: 2 + 2 substeps, none through the any-length transforms
            slots = {k["kernel"]: k["launches"] for k in ctx.get_profile() if k["launches"]}
            assert slots.get("chp_A_z_fwd", 0) >= 4 and "ch_kspace" not in slots, slots
        # the multi-substep call (run-time compiled k_z_inv_fwd with the generated chemical potential between two substeps)
        multi = []
        for prm in (builtin, pp):
            ring = [ctx.empty_hist(), ctx.empty_hist()]
            out, mu = torch.empty_like(c0), torch.empty_like(c0)
            ctx.ch_substeps(prm, c0.clone(), out, ring, 1, 0, 2, 4, True, 1e-3, mu=mu)
            multi.append((out.cpu(), mu.cpu()))
        for a, b in zip(*multi):
            if shape[2] in (150, 160, 180, 240):
                # the fused inverse + forward z kernel of the two-stage plans (fft_two_z.h): its run-time compiled instance and the
                # built-in one contract different multiply-adds in the radix-15 / radix-16 butterflies (measured 2e-16 per substep)
                assert (a - b).abs().max().item() <= 1e-14
            else:
                assert torch.equal(a, b)


def test_gradient_tensor_gold_gpu():
    """test/tests/typed_tensors/tests (gradient.i): the three components of GradientTensor through the HIP transforms
    (generic path: 20 x 10 x 5, odd r2c axis) and generated k-space kernels `cbar*i*k_d`, against the reference's gold file"""
    import numpy as np
    from marlin_amd.api import Context, ParsedCompute
    from tests.conftest import load_golden
    from tests.test_oracle_golden import _node_mode
    g = load_golden("typed_gradient_gold.npz")
    ctx = Context(3, [20, 10, 5], [1.0, 1.0, 1.0])
    c = ParsedCompute(ctx, "sin(x*8*pi)+cos(y*4*pi)+sin(z*2*pi)", extra_symbols=True)()
    assert np.abs(_node_mode(c.cpu()) - g["c.1"]).max() <= 1e-14
    cbar = ctx.fft(c)
    for k, nm in zip(("kx", "ky", "kz"), "xyz"):
        gbar = ParsedCompute(ctx, f"cbar*i*{k}", ["cbar"], complex_inputs=["cbar"], extra_symbols=True, reciprocal=True)(cbar)
        assert np.abs(_node_mode(ctx.ifft(gbar).cpu()) - g[f"grad_c_{nm}.1"]).max() <= 1e-12


def test_local_vars_derivative_gpu():
    """test/tests/parsed_tensor/tests (local_vars_derivative.i): d/da of `r:=sqrt(a^2+1); r^2` equals 2a"""
    from marlin_amd.api import Context, ParsedCompute
    ctx = Context(2, [20, 20], [2.0, 2.0])
    a = ParsedCompute(ctx, "x + 0.5*y", extra_symbols=True)()
    d = ParsedCompute(ctx, "r:=sqrt(a^2+1); r^2", ["a"], derivatives=["a"])(a)
    err = (d - 2 * a).abs()
    assert ctx.sum(err) / 400 * 4.0 <= 1e-13          # TensorIntegralPostprocessor of |df_da - 2a|; gold: 0


def test_histogram_gold_gpu():
    """test/tests/histogram/tests (test.i, CSVDiff): TensorHistogram of 0.1 x^2 + 0.2 y^2 + 0.3 z^2 on 10^3, 20 bins on [0, 1]
    -- bin edges torch.linspace(0, 1, 21) as the reference builds them -- and the histogramdd edge conventions"""
    from marlin_amd.api import Context, ParsedCompute
    from tests.conftest import load_golden
    g = load_golden("fft_gold.npz")["test_out_hist_0001"]
    ctx = Context(3, [10, 10, 10], [1.0, 1.0, 1.0])
    c = ParsedCompute(ctx, "0.1*x^2+0.2*y^2+0.3*z^2", extra_symbols=True)()
    edges = torch.linspace(0.0, 1.0, 21, dtype=torch.float64)
    counts = ctx.histogram(c, edges.tolist())
    assert counts == [int(v) for v in g[:, 1]] and sum(counts) == 1000
    # conventions: [e_i, e_i+1), last bin closed, outside values and NaN dropped -- against torch.histogramdd on the CPU
    torch.manual_seed(1)
    v = torch.cat([torch.rand(5000, dtype=torch.float64) * 1.4 - 0.2, edges, torch.tensor([float("nan"), -0.2, 1.0, 1.0000001])])
    want = torch.histogramdd(v[~torch.isnan(v)].reshape(-1, 1), [edges]).hist
    assert ctx.histogram(v.cuda(), edges.tolist()) == [int(x) for x in want]



@pytest.mark.parametrize("expr", ["x^2", "x^3", "sin(x)", "cos(x)", "exp(x)", "log(x)", "1/x", "sqrt(x)", "a := x^2; a * x"])
def test_reference_unit_test_second_derivatives(ctx, expr):
    """TEST(ParsedTensorTest, SecondDerivatives), unit/src/ParsedTensorTest.C:546-609: the symbolic second derivative against central
    differences of the compiled expression, x = linspace(0.1, 2.01, 11), h = 1e-4, relative 1e-3 -- the reference's numbers"""
    x = torch.linspace(0.1, 2.01, 11, dtype=torch.float64).cuda()
    h = 1e-4
    f = _pc(ctx, expr, inputs=["x"])
    d2 = _pc(ctx, expr, inputs=["x"], derivatives=["x", "x"])(x)
    fd = (f(x + h) - 2.0 * f(x) + f(x - h)) / (h * h)
    rel = ((d2 - fd).abs() / (fd.abs() + h)).max().item()
    assert rel < 1e-3, (expr, rel)


def test_reference_unit_test_constants(ctx):
    """TEST(ParsedTensorTest, Constants), unit/src/ParsedTensorTest.C:611-700: named constants in expressions, in let bindings and
    under differentiation (1e-12, the reference's fp64 epsilon)"""
    x = torch.linspace(0.1, 2.01, 11, dtype=torch.float64).cuda()
    K = {"pi": math.pi, "e": math.e}
    assert (_pc(ctx, "x * pi", inputs=["x"], constants={"pi": math.pi})(x) - x * math.pi).abs().max().item() <= 1e-12
    assert (_pc(ctx, "sin(x) + pi * e", inputs=["x"], constants=K)(x) - (torch.sin(x) + math.pi * math.e)).abs().max().item() <= 1e-12
    assert (_pc(ctx, "x * pi", inputs=["x"], constants={"pi": math.pi}, derivatives=["x"])(x) - math.pi).abs().max().item() <= 1e-12
    assert (_pc(ctx, "a := x * pi; a + a", inputs=["x"], constants={"pi": math.pi})(x) - 2.0 * x * math.pi).abs().max().item() <= 1e-12


# ---- TEST(ParsedTensorTest, Parse), unit/src/ParsedTensorTest.C:19-205, case by case ---------------------------------------------
# (expression, gold as the reference writes it in libTorch, derivative_check) -- x = linspace(0.1, 2.01, 11)[:, None],
# y = linspace(0.11, 3.02, 15)[None, :], n = max(x*y)*1.01, broadcast to the 11 x 15 grid (the library evaluates full-size arrays)
_PARSE_CASES = [
    ("hypot(x,y)", lambda x, y, n: torch.hypot(x, y), True),
    ("sqrt(x^2+y^2+n)", lambda x, y, n: torch.sqrt(x * x + y * y + n), True),
    ("sqrt(x*x+y*y+n)", lambda x, y, n: torch.sqrt(x * x + y * y + n), True),
    ("sqrt(x^2+y^2)", lambda x, y, n: torch.sqrt(x * x + y * y), True),
    ("sqrt(x*x+y*y)", lambda x, y, n: torch.sqrt(x * x + y * y), True),
    ("tan((x-y)/2)", lambda x, y, n: torch.tan((x - y) / 2.0), True),
    ("tanh(x-y)", lambda x, y, n: torch.tanh(x - y), True),
    ("cos(y)", lambda x, y, n: torch.cos(y) + 0 * x, True),
    ("sin(y)", lambda x, y, n: torch.sin(y) + 0 * x, True),
    ("cosh(y)", lambda x, y, n: torch.cosh(y) + 0 * x, True),
    ("sinh(y)", lambda x, y, n: torch.sinh(y) + 0 * x, True),
    ("atan(x + y)", lambda x, y, n: torch.atan(x + y), True),
    ("asin((x * y / 2) / n)", lambda x, y, n: torch.asin((x * y / 2.0) / n), True),
    ("acos((x * y / 2) / n)", lambda x, y, n: torch.acos((x * y / 2.0) / n), True),
    ("acosh(x+y+1)", lambda x, y, n: torch.acosh(x + y + 1), True),
    ("asinh(x-y)", lambda x, y, n: torch.asinh(x - y), True),
    ("atan2(x,y)", lambda x, y, n: torch.atan2(x, y), True),
    ("1/sqrt(x+y)", lambda x, y, n: 1.0 / torch.sqrt(x + y), True),
    ("sin(y)-cos(y)", lambda x, y, n: torch.sin(y) - torch.cos(y) + 0 * x, True),
    ("(x * y) / n", lambda x, y, n: (x * y) / n, True),
    ("y/x", lambda x, y, n: y / x, False),
    ("-x", lambda x, y, n: -x + 0 * y, True),
    ("rsqrt(x*y)", lambda x, y, n: 1.0 / torch.sqrt(x * y), True),
    ("exp(x*y)", lambda x, y, n: torch.exp(x * y), True),
    ("exp2(x*y)", lambda x, y, n: torch.pow(2.0, x * y), True),
    ("(x*y) % 1.5", lambda x, y, n: torch.remainder(x * y, 1.5), True),
    ("log(x)", lambda x, y, n: torch.log(x) + 0 * y, True),
    ("log10(x)", lambda x, y, n: torch.log10(x) + 0 * y, True),
    ("log2(x)", lambda x, y, n: torch.log2(x) + 0 * y, True),
    ("pow(y, x)", lambda x, y, n: torch.pow(y, x), True),
    ("abs(y-x)", lambda x, y, n: torch.abs(y - x), False),
    ("floor(x-y)", lambda x, y, n: torch.floor(x - y), False),
    ("ceil(x-y)", lambda x, y, n: torch.ceil(x - y), False),
    ("round(x-y)", lambda x, y, n: torch.round(x - y), False),
    ("trunc(x-y)", lambda x, y, n: torch.trunc(x - y), True),
    ("min(x^3,y^2)", lambda x, y, n: torch.minimum(x * x * x, y * y), True),
    ("max(x^2,sin(4*y))", lambda x, y, n: torch.maximum(x * x, torch.sin(y * 4.0)), False),
    ("pow(2, x)", lambda x, y, n: torch.pow(2, x) + 0 * y, True),
    ("pow(x, 1.0/3.0)", lambda x, y, n: torch.pow(x, 1.0 / 3.0) + 0 * y, True),
    ("if(x<1 | y>=2, x, y)", lambda x, y, n: torch.where(torch.logical_or(x < 1, y >= 2), x + 0 * y, y + 0 * x), True),
    ("if(x<=1 & y>2, x*x, 3*y)", lambda x, y, n: torch.where(torch.logical_and(x <= 1, y > 2), x * x + 0 * y, y * 3 + 0 * x), True),
    ("r2:=x^2+y^2; sqrt(r2)", lambda x, y, n: torch.sqrt(x * x + y * y), True),
]


def _parse_inputs():
    x = torch.linspace(0.1, 2.01, 11, dtype=torch.float64).unsqueeze(1)
    y = torch.linspace(0.11, 3.02, 15, dtype=torch.float64).unsqueeze(0)
    n = torch.max(x * y) * 1.01
    shape = (11, 15)
    return x.expand(shape).contiguous(), y.expand(shape).contiguous(), n.expand(shape).contiguous()


@pytest.mark.parametrize("expr,gold,derivative_check", _PARSE_CASES, ids=[c[0] for c in _PARSE_CASES])
def test_reference_unit_test_parse_section(ctx, expr, gold, derivative_check):
    """`check(expression, gold)` of TEST(ParsedTensorTest, Parse): the compiled expression == the libTorch gold to epsilon = 1e-12, and
    (where the reference checks it) the symbolic derivative w.r.t. x, y and n against a forward difference with the reference's
    step 1e-6 and its acceptance: relative error < 1e-5 or absolute difference < sqrt(epsilon)"""
    x, y, n = _parse_inputs()
    eps, eps_fd, rel_tol, abs_tol = 1e-12, 1e-6, 1e-5, 1e-6
    f = _pc(ctx, expr, inputs=["x", "y", "n"])
    xs = [t.cuda() for t in (x, y, n)]
    r = f(*xs)
    assert (r.cpu() - gold(x, y, n)).abs().max().item() <= eps
    if not derivative_check:
        return
    for k, var in enumerate("xyn"):
        pert = [t + eps_fd if i == k else t for i, t in enumerate(xs)]
        dr1 = (f(*pert) - r) / eps_fd
        dr2 = _pc(ctx, expr, inputs=["x", "y", "n"], derivatives=[var])(*xs)
        abs_diff = (dr2 - dr1).abs()
        rel = (abs_diff / (dr1.abs() + eps_fd)).max().item()
        assert rel < rel_tol or abs_diff.max().item() < abs_tol, (expr, var, rel, abs_diff.max().item())


@pytest.mark.parametrize("expr,var,derivative", [
    ("y/x", "x", "-y/x^2"), ("y/x", "y", "1/x"),
    ("max(x^2,sin(4*y))", "x", "if(x^2>=sin(4*y),2*x,0)"), ("max(x^2,sin(4*y))", "y", "if(x^2>=sin(4*y),0,4*cos(4*y))"),
    ("x2:=x^2; sinx2:=sin(x2); 4*sinx2", "x", "8*x*cos(x*x)"), ("a:=sin(x^2); a + 2*a + 3*a", "x", "12*x*cos(x^2)")])
def test_reference_unit_test_parse_section_check2(ctx, expr, var, derivative):
    """`check2(expression, dvar, derivative, compile)` of the same test: the symbolic derivative == the hand-written one,
    relative 1e-5 (the library always compiles: the reference's compile = true / false pair is one case here)"""
    x, y, n = (t.cuda() for t in _parse_inputs())
    r1 = _pc(ctx, expr, inputs=["x", "y", "n"], derivatives=[var])(x, y, n)
    r2 = _pc(ctx, derivative, inputs=["x", "y", "n"])(x, y, n)
    rel = ((r1 - r2).abs() / (r1.abs() + 1e-12)).max().item()
    assert rel <= 1e-5, (expr, var, rel)
