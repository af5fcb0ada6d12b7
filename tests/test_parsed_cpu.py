"""Expression front end of the native ParsedCompute (marlin_amd/csrc/expr.hip) on CPU: parse -> differentiate ->
simplify must produce the trees the reference's parser produces (include/utils/MarlinExpressionParser.h:383-427,
src/utils/MarlinExpressionParser.C:50-235), because the tree fixes the floating-point evaluation order."""
import pytest

from marlin_amd.api import MarlinHipError, ParsedCompute


def tree(expr, **kw):
    return ParsedCompute(None, expr, **kw).tree


def test_ch_chemical_potential_tree():
    """SURVEY A.3 (verified against mu.10 of the reference gold file): d/dc[0.1*c^2*(c-1)^2]"""
    t = tree("0.1*c^2*(c-1)^2", inputs=["c"], derivatives=["c"])
    assert t == "(((0.10000000000000001 * (2 * c)) * ((c - 1) ^ 2)) + ((0.10000000000000001 * (c ^ 2)) * (2 * (c - 1))))"


def test_pfhub_tree():
    t = tree("rho_s*(c-c_alpha)^2*(c_beta-c)^2", inputs=["c"], constants={"rho_s": 5, "c_alpha": 0.3, "c_beta": 0.7},
             derivatives=["c"])
    assert t == ("(((rho_s * (2 * (c - c_alpha))) * ((c_beta - c) ^ 2)) + "
                 "((rho_s * ((c - c_alpha) ^ 2)) * (-(2 * (c_beta - c)))))")


@pytest.mark.parametrize("expr,expected", [
    ("a+b*c", "(a + (b * c))"),                    # precedence
    ("a-b-c", "((a - b) - c)"),                    # left associative
    ("a/b*c", "((a / b) * c)"),
    ("a^b^c", "(a ^ (b ^ c))"),                    # right associative
    ("-a^2", "(-(a ^ 2))"),                        # unary binds weaker than ^
    ("2*3+a", "(6 + a)"),                          # literal folding
    ("a*1 + 0", "a"), ("0 - a", "(-a)"), ("a*0", "0"), ("a^1", "a"), ("a^0", "1"), ("a*-1", "(-a)"), ("0/a", "0"),
    ("sin(0) + a", "a"), ("min(2, 3)*a", "(2 * a)"),
    ("a < b & b <= c | !a", "(((a < b) & (b <= c)) | (!a))"),
    ("if(a > 0, b, c)", "if((a > 0), b, c)"),
    ("s := a + b; s * s", "((a + b) * (a + b))"),  # local variables
])
def test_grammar_and_simplification(expr, expected):
    assert tree(expr, inputs=["a", "b", "c"]) == expected


@pytest.mark.parametrize("expr,var,expected", [
    ("a*b", "a", "b"),                                        # 1*b + a*0
    ("a/b", "a", "(b / (b ^ 2))"),                            # (1*b - a*0)/b^2
    ("a^3", "a", "(3 * (a ^ 2))"),
    ("sin(a*b)", "a", "(cos((a * b)) * b)"),
    ("exp(a)", "a", "exp(a)"),
    ("log(a)", "a", "(1 / a)"),
    ("sqrt(a)", "a", "(1 / (2 * sqrt(a)))"),
    ("tanh(a)", "a", "(1 / (cosh(a) * cosh(a)))"),
    ("a^b", "a", "((a ^ b) * (b * (1 / a)))"),                # general power rule, b symbolic
    ("c", "a", "0"),
])
def test_derivative_rules(expr, var, expected):
    assert tree(expr, inputs=["a", "b", "c"], derivatives=[var]) == expected


def test_second_derivative_is_simplified_once_at_the_end():
    assert tree("a^3", inputs=["a"], derivatives=["a", "a"]) == "(3 * (2 * a))"


def test_named_constants_stay_symbolic():
    assert tree("A - (B+1)*u +u^2*v", inputs=["u", "v"], constants={"A": 1, "B": 3.5}) == "((A - ((B + 1) * u)) + ((u ^ 2) * v))"


def test_errors_follow_the_reference():
    with pytest.raises(MarlinHipError, match="Duplicate buffer name"):
        tree("a", inputs=["a", "a"])
    with pytest.raises(MarlinHipError, match="reserved name 'x'"):
        ParsedCompute(None, "x", inputs=["x"], extra_symbols=True)
    with pytest.raises(MarlinHipError, match="not listed in `inputs`"):
        tree("a*k", inputs=["a"], constants={"k": 2}, derivatives=["k"])
    with pytest.raises(MarlinHipError, match="Invalid function"):
        tree("a +* 2", inputs=["a"])
    with pytest.raises(MarlinHipError, match="Derivative not implemented"):
        tree("erf(a)", inputs=["a"], derivatives=["a"])


def test_complex_typing():
    p = ParsedCompute(None, "Mbar*mubar", inputs=["Mbar", "mubar"], complex_inputs=["mubar"])
    assert p.is_complex and "cscale(" in p.source
    assert not ParsedCompute(None, "Mbar*2", inputs=["Mbar"]).is_complex
    assert ParsedCompute(None, "i*kx*a", inputs=["a"], extra_symbols=True, reciprocal=True).is_complex
