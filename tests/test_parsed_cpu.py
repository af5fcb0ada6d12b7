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


# ---- unit/src/ParsedTensorTest.C, the sections not mirrored above -----------------------------------------
@pytest.mark.parametrize("expr,expected", [
    # TEST(ParsedTensorTest, Simplify) :411-545 -- constant folding, the algebraic identities, nesting (our printer writes 5 for the
    # reference's "5.000000": std::to_string vs shortest round-trip; the trees are the same)
    ("2 + 3", "5"), ("4 * 5", "20"), ("2 ^ 3", "8"), ("x * 0", "0"), ("x * 1", "x"), ("x + 0", "x"), ("x - 0", "x"), ("x / 1", "x"),
    ("x ^ 0", "1"), ("x ^ 1", "x"), ("sin(0)", "0"), ("(x + 0) * 1 + 0", "x"),
    # ... ErrorHandling :396-409: let bindings are simplified, function calls on constants are folded
    ("a := 2 + 3; a * x", "(5 * x)"), ("sqrt(4) + log(1) + exp(0)", "3"),
])
def test_reference_unit_test_simplify_section(expr, expected):
    assert tree(expr, inputs=["x", "y"]) == expected


@pytest.mark.parametrize("expr,expected", [
    # TEST(ParsedTensorTest, Substitute) :206-310.  The reference substitutes into let expressions and keeps the bindings in its
    # tree; this front end resolves a binding by substituting it into the body at parse time, so the SAME cases read as: every use of
    # a bound name is the bound expression (evaluation order inside it unchanged), a later binding sees the earlier ones, and a
    # bound name shadows an input of the same name
    ("a := x + 1; a * x", "((x + 1) * x)"),
    ("a := x; b := a + 1; b * x", "((x + 1) * x)"),
    ("r := x^2 + y^2; sqrt(r) + r", "(sqrt(((x ^ 2) + (y ^ 2))) + ((x ^ 2) + (y ^ 2)))"),
    ("y := 2*x; x + y", "(x + (2 * x))"),            # the binding shadows the input y
    ("s := sin(x); s + cos(x) * s", "(sin(x) + (cos(x) * sin(x)))"),
])
def test_reference_unit_test_substitute_section_through_let_bindings(expr, expected):
    assert tree(expr, inputs=["x", "y"]) == expected


@pytest.mark.parametrize("expr,column", [
    # TEST(ParsedTensorTest, ErrorHandling) :312-394: the same inputs are rejected at the same column ("Line 1:<column>" in the
    # reference's message; ours reports the 0-based position)
    ("x + ", 5), ("(x + y", 7), ("x + y)", 6), ("sin(x", 6), ("a := ; x + a", 6), ("x + * y", 5), ("", 1), ("1.2.3 + x", 4),
])
def test_reference_unit_test_error_handling_section(expr, column):
    with pytest.raises(MarlinHipError) as e:
        tree(expr, inputs=["x", "y"])
    assert e.value.code == -1 and "Invalid function" in e.value.message
    assert f"at position {column - 1} of" in e.value.message


def test_reference_unit_test_undefined_names_and_constants():
    """:356-394 (a name that is neither an input nor a constant), :397-402 (derivative with respect to a name the expression does not
    contain is 0), TEST(ParsedTensorTest, Constants) :663-676 (a constant is not a variable: d/dx[x + pi] = 1)"""
    with pytest.raises(MarlinHipError, match="unknown variable 'q'"):
        tree("x + q", inputs=["x"])
    assert tree("x + y", inputs=["x", "y", "z"], derivatives=["z"]) == "0"
    assert tree("x + pi", inputs=["x"], constants={"pi": 3.141592653589793}, derivatives=["x"]) == "1"
    assert tree("x * pi", inputs=["x"], constants={"pi": 3.141592653589793}, derivatives=["x"]) == "pi"
