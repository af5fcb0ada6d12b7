// Parsed pointwise expressions: the native counterpart of ParsedCompute + ParsedJITTensor
// (src/tensor_computes/ParsedCompute.C:50-265, src/utils/ParsedJITTensor.C:63-156) -- expression text ->
// AST -> symbolic derivatives -> simplification -> HIP source -> hiprtc -> one fused pointwise kernel.
// (The reference lowers the same tree to a torch-JIT graph of one aten op per node.)
//
// What is mirrored from the reference, because it fixes the floating-point evaluation ORDER:
//   grammar and associativity       include/utils/MarlinExpressionParser.h:383-427
//       (| &) < comparisons < (+ -) < (* / %) < unary (- !) < ^ (right-assoc.) ; f(args) ; a := e; locals
//   differentiation rules           src/utils/MarlinExpressionParser.C:143-203, 609-880
//       (f*g)' = f'*g + f*g' ; (f/g)' = (f'*g - f*g')/g^2 ; (f^c)' = (c * f^(c-1)) * f' ; chain rules per function
//   simplification rules            src/utils/MarlinExpressionParser.C:50-141, 250-268, 515-600
//       constant folding of literals, x+0, 0+x, x-0, 0-x -> -x, x*0, x*1, x*-1 -> -x, 0/x, x/1, x^0, x^1, 1^x
//       applied ONCE, after all requested derivatives (ParsedJITTensor::compile)
//   named constants stay symbolic   (they are graph inputs in the reference, ParsedCompute.C:128-132), only
//       literals fold; pow(x, 2|3|-2|0.5|-0.5|-1) lowers like ATen's pow_tensor_scalar (x*x, x*x*x, 1/(x*x), sqrt, ...)
// The evaluation itself is ours: one kernel, every temporary in a register, `-ffp-contract=off`.
#include <hip/hiprtc.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <sstream>

#include "mrl_internal.h"
#include "fft_pow2_launch.h"
#include "fft_two_z.h"

namespace mrl {
namespace ex {

struct Node;
typedef std::shared_ptr<const Node> P;

enum Kind { NUM, VAR, NEG, NOT, BIN, CMP, LOGIC, CALL };

struct Node {
  Kind kind;
  double value = 0.0;      // NUM
  std::string name;        // VAR, CALL; operator text for BIN / CMP / LOGIC
  std::vector<P> a;        // operands / arguments
};

static P num(double v) {
  auto n = std::make_shared<Node>();
  n->kind = NUM;
  n->value = v;
  return n;
}
static P var(const std::string &s) {
  auto n = std::make_shared<Node>();
  n->kind = VAR;
  n->name = s;
  return n;
}
static P un(Kind k, P x) {
  auto n = std::make_shared<Node>();
  n->kind = k;
  n->a = {std::move(x)};
  return n;
}
static P bin(Kind k, const std::string &op, P l, P r) {
  auto n = std::make_shared<Node>();
  n->kind = k;
  n->name = op;
  n->a = {std::move(l), std::move(r)};
  return n;
}
static P call(const std::string &f, std::vector<P> args) {
  auto n = std::make_shared<Node>();
  n->kind = CALL;
  n->name = f;
  n->a = std::move(args);
  return n;
}
static inline P add(P l, P r) { return bin(BIN, "+", l, r); }
static inline P sub(P l, P r) { return bin(BIN, "-", l, r); }
static inline P mul(P l, P r) { return bin(BIN, "*", l, r); }
static inline P dvd(P l, P r) { return bin(BIN, "/", l, r); }
static inline P pw(P l, P r) { return bin(BIN, "^", l, r); }

struct Error {
  std::string msg;
};

// ---- parser -------------------------------------------------------------------------------------
struct Parser {
  const std::string s;
  size_t i = 0;
  const std::set<std::string> &known;
  std::map<std::string, P> locals;

  Parser(const std::string &text, const std::set<std::string> &k) : s(text), known(k) {}
  void ws() {
    while (i < s.size() && std::isspace((unsigned char)s[i])) ++i;
  }
  bool eat(const char *tok) {
    ws();
    const size_t n = std::strlen(tok);
    if (s.compare(i, n, tok) == 0) {
      i += n;
      return true;
    }
    return false;
  }
  bool peek(const char *tok) {
    ws();
    return s.compare(i, std::strlen(tok), tok) == 0;
  }
  [[noreturn]] void fail(const std::string &m) { throw Error{m + " at position " + std::to_string(i) + " of '" + s + "'"}; }

  bool ident(std::string &out) {
    ws();
    size_t j = i;
    if (j < s.size() && (std::isalpha((unsigned char)s[j]) || s[j] == '_')) {
      while (j < s.size() && (std::isalnum((unsigned char)s[j]) || s[j] == '_')) ++j;
      out = s.substr(i, j - i);
      i = j;
      return true;
    }
    return false;
  }

  P statements() {
    for (;;) {
      const size_t save = i;
      std::string id;
      if (ident(id) && eat(":=")) {
        P e = logical();
        if (!eat(";")) fail("expected ';' after local variable definition");
        locals[id] = e;
        continue;
      }
      i = save;
      break;
    }
    P e = logical();
    ws();
    if (i != s.size()) fail("unexpected trailing input");
    return e;
  }
  P logical() {
    P l = comparison();
    for (;;) {
      if (eat("|"))
        l = bin(LOGIC, "|", l, comparison());
      else if (eat("&"))
        l = bin(LOGIC, "&", l, comparison());
      else
        return l;
    }
  }
  P comparison() {
    P l = additive();
    static const char *ops[] = {"<=", ">=", "==", "!=", "<", ">"};
    for (const char *op : ops)
      if (eat(op)) return bin(CMP, op, l, additive());
    return l;
  }
  P additive() {
    P l = multitive();
    for (;;) {
      if (eat("+"))
        l = add(l, multitive());
      else if (eat("-"))
        l = sub(l, multitive());
      else
        return l;
    }
  }
  P multitive() {
    P l = unary();
    for (;;) {
      if (eat("*"))
        l = mul(l, unary());
      else if (eat("/"))
        l = dvd(l, unary());
      else if (eat("%"))
        l = bin(BIN, "%", l, unary());
      else
        return l;
    }
  }
  P unary() {
    if (eat("-")) return un(NEG, unary());
    if (peek("!") && !peek("!=")) {
      eat("!");
      return un(NOT, unary());
    }
    return power();
  }
  P power() {
    P b = primary();
    if (eat("^")) return pw(b, power());
    return b;
  }
  P primary() {
    ws();
    if (eat("(")) {
      P e = logical();
      if (!eat(")")) fail("expected ')'");
      return e;
    }
    if (i < s.size() && std::isdigit((unsigned char)s[i])) {
      size_t j = i;
      while (j < s.size() && std::isdigit((unsigned char)s[j])) ++j;
      if (j < s.size() && s[j] == '.' && j + 1 < s.size() && std::isdigit((unsigned char)s[j + 1])) {
        ++j;
        while (j < s.size() && std::isdigit((unsigned char)s[j])) ++j;
      }
      if (j < s.size() && (s[j] == 'e' || s[j] == 'E')) {
        size_t k = j + 1;
        if (k < s.size() && (s[k] == '+' || s[k] == '-')) ++k;
        if (k < s.size() && std::isdigit((unsigned char)s[k])) {
          while (k < s.size() && std::isdigit((unsigned char)s[k])) ++k;
          j = k;
        }
      }
      const double v = std::strtod(s.substr(i, j - i).c_str(), nullptr);
      i = j;
      return num(v);
    }
    std::string id;
    if (ident(id)) {
      if (eat("(")) {
        std::vector<P> args;
        if (!eat(")")) {
          do args.push_back(logical());
          while (eat(","));
          if (!eat(")")) fail("expected ')' after function arguments");
        }
        return call(id, args);
      }
      auto it = locals.find(id);
      if (it != locals.end()) return it->second;
      if (!known.count(id)) fail("unknown variable '" + id + "'");
      return var(id);
    }
    fail("expected a number, a variable or '('");
  }
};

// ---- differentiation ------------------------------------------------------------------------------
static P diff(const P &e, const std::string &v) {
  switch (e->kind) {
    case NUM: return num(0.0);
    case VAR: return num(e->name == v ? 1.0 : 0.0);
    case NEG: return un(NEG, diff(e->a[0], v));
    case NOT:
    case CMP:
    case LOGIC: return num(0.0);
    case BIN: {
      const P &l = e->a[0], &r = e->a[1];
      const P dl = diff(l, v), dr = diff(r, v);
      const char op = e->name[0];
      if (op == '+') return add(dl, dr);
      if (op == '-') return sub(dl, dr);
      if (op == '*') return add(mul(dl, r), mul(l, dr));
      if (op == '/') return dvd(sub(mul(dl, r), mul(l, dr)), pw(r, num(2.0)));
      if (op == '^') {
        if (r->kind == NUM) return mul(mul(r, pw(l, num(r->value - 1.0))), dl);
        return mul(pw(l, r), add(mul(dr, call("log", {l})), mul(r, dvd(dl, l))));
      }
      return dl;  // '%': derivative w.r.t. the first operand
    }
    case CALL: {
      const std::string &f = e->name;
      if (e->a.empty()) return num(0.0);
      const P &x = e->a[0];
      const P dx = diff(x, v);
      auto sq = [](const P &t) { return mul(t, t); };
      if (f == "sin") return mul(call("cos", {x}), dx);
      if (f == "cos") return mul(un(NEG, call("sin", {x})), dx);
      if (f == "tan") return dvd(dx, sq(call("cos", {x})));
      if (f == "sinh") return mul(call("cosh", {x}), dx);
      if (f == "cosh") return mul(call("sinh", {x}), dx);
      if (f == "tanh") return dvd(dx, sq(call("cosh", {x})));
      if (f == "exp") return mul(call("exp", {x}), dx);
      if (f == "exp2") return mul(mul(call("exp2", {x}), call("log", {num(2.0)})), dx);
      if (f == "log") return dvd(dx, x);
      if (f == "log10") return dvd(dx, mul(x, call("log", {num(10.0)})));
      if (f == "log2") return dvd(dx, mul(x, call("log", {num(2.0)})));
      if (f == "sqrt") return dvd(dx, mul(num(2.0), call("sqrt", {x})));
      if (f == "rsqrt") return mul(un(NEG, dvd(call("rsqrt", {x}), mul(num(2.0), x))), dx);
      if (f == "asin") return dvd(dx, call("sqrt", {sub(num(1.0), sq(x))}));
      if (f == "acos") return un(NEG, dvd(dx, call("sqrt", {sub(num(1.0), sq(x))})));
      if (f == "atan") return dvd(dx, add(num(1.0), sq(x)));
      if (f == "asinh") return dvd(dx, call("sqrt", {add(sq(x), num(1.0))}));
      if (f == "acosh") return dvd(dx, call("sqrt", {sub(sq(x), num(1.0))}));
      if (f == "atanh") return dvd(dx, sub(num(1.0), sq(x)));
      if (f == "abs") return mul(dvd(x, call("abs", e->a)), dx);
      if (e->a.size() == 2) {
        const P &y = e->a[1];
        const P dy = diff(y, v);
        if (f == "hypot") return add(mul(dvd(x, e), dx), mul(dvd(y, e), dy));
        if (f == "atan2") return dvd(sub(mul(y, dx), mul(x, dy)), add(sq(y), sq(x)));  // atan2(y=arg0, x=arg1)
        if (f == "pow") return mul(e, add(mul(y, dvd(dx, x)), mul(call("log", {x}), dy)));
        if (f == "min") return call("if", {bin(CMP, "<", x, y), dx, dy});
        if (f == "max") return call("if", {bin(CMP, ">", x, y), dx, dy});
      }
      if (f == "if" && e->a.size() == 3) return call("if", {e->a[0], diff(e->a[1], v), diff(e->a[2], v)});
      if (f == "round" || f == "ceil" || f == "floor" || f == "trunc") return num(0.0);
      throw Error{"Derivative not implemented for function: " + f};
    }
  }
  return num(0.0);
}

// ---- simplification -------------------------------------------------------------------------------
static bool fold_call(const std::string &f, const std::vector<double> &c, double &out) {
  if (c.size() == 1) {
    const double x = c[0];
    static const std::map<std::string, double (*)(double)> fn = {
        {"sin", std::sin},   {"cos", std::cos},     {"tan", std::tan},     {"sinh", std::sinh},   {"cosh", std::cosh},
        {"tanh", std::tanh}, {"asin", std::asin},   {"acos", std::acos},   {"atan", std::atan},   {"asinh", std::asinh},
        {"acosh", std::acosh}, {"atanh", std::atanh}, {"exp", std::exp},   {"log", std::log},     {"log10", std::log10},
        {"log2", std::log2}, {"sqrt", std::sqrt},   {"abs", std::fabs},    {"ceil", std::ceil},   {"floor", std::floor},
        {"round", std::round}, {"trunc", std::trunc}};
    auto it = fn.find(f);
    if (it == fn.end()) return false;
    out = it->second(x);
    return true;
  }
  if (c.size() == 2) {
    if (f == "min") out = std::min(c[0], c[1]);
    else if (f == "max") out = std::max(c[0], c[1]);
    else if (f == "atan2") out = std::atan2(c[0], c[1]);
    else if (f == "hypot") out = std::hypot(c[0], c[1]);
    else if (f == "pow") out = std::pow(c[0], c[1]);
    else return false;
    return true;
  }
  if (c.size() == 3 && f == "if") {
    out = c[0] != 0.0 ? c[1] : c[2];
    return true;
  }
  return false;
}

static P simp(const P &e) {
  switch (e->kind) {
    case NUM:
    case VAR: return e;
    case NEG: {
      const P x = simp(e->a[0]);
      if (x->kind == NUM) return num(-x->value);
      return un(NEG, x);
    }
    case NOT: {
      const P x = simp(e->a[0]);
      if (x->kind == NUM) return num(x->value == 0.0 ? 1.0 : 0.0);
      return un(NOT, x);
    }
    case CMP: {
      const P l = simp(e->a[0]), r = simp(e->a[1]);
      if (l->kind == NUM && r->kind == NUM) {
        const double a = l->value, b = r->value;
        const std::string &o = e->name;
        const bool t = o == "<" ? a < b : o == ">" ? a > b : o == "<=" ? a <= b : o == ">=" ? a >= b : o == "==" ? a == b : a != b;
        return num(t ? 1.0 : 0.0);
      }
      return bin(CMP, e->name, l, r);
    }
    case LOGIC: {
      const P l = simp(e->a[0]), r = simp(e->a[1]);
      const bool lc = l->kind == NUM, rc = r->kind == NUM;
      const bool is_and = e->name == "&";
      if (lc && rc) {
        const bool a = l->value != 0.0, b = r->value != 0.0;
        return num((is_and ? (a && b) : (a || b)) ? 1.0 : 0.0);
      }
      if (is_and && ((lc && l->value == 0.0) || (rc && r->value == 0.0))) return num(0.0);
      if (!is_and && ((lc && l->value != 0.0) || (rc && r->value != 0.0))) return num(1.0);
      return bin(LOGIC, e->name, l, r);
    }
    case BIN: {
      const P l = simp(e->a[0]), r = simp(e->a[1]);
      const bool lc = l->kind == NUM, rc = r->kind == NUM;
      const char op = e->name[0];
      if (lc && rc) {
        const double a = l->value, b = r->value;
        switch (op) {
          case '+': return num(a + b);
          case '-': return num(a - b);
          case '*': return num(a * b);
          case '/': return num(a / b);
          case '^': return num(std::pow(a, b));
          default: return num(std::fmod(a, b));
        }
      }
      switch (op) {
        case '+':
          if (lc && l->value == 0.0) return r;
          if (rc && r->value == 0.0) return l;
          break;
        case '-':
          if (rc && r->value == 0.0) return l;
          if (lc && l->value == 0.0) return simp(un(NEG, r));
          break;
        case '*':
          if ((lc && l->value == 0.0) || (rc && r->value == 0.0)) return num(0.0);
          if (lc && l->value == 1.0) return r;
          if (rc && r->value == 1.0) return l;
          if (lc && l->value == -1.0) return simp(un(NEG, r));
          if (rc && r->value == -1.0) return simp(un(NEG, l));
          break;
        case '/':
          if (lc && l->value == 0.0) return num(0.0);
          if (rc && r->value == 1.0) return l;
          break;
        case '^':
          if (rc && r->value == 0.0) return num(1.0);
          if (rc && r->value == 1.0) return l;
          if (lc && l->value == 1.0) return num(1.0);
          break;
        default: break;
      }
      return bin(BIN, e->name, l, r);
    }
    case CALL: {
      std::vector<P> args;
      std::vector<double> cv;
      bool all = true;
      for (const P &x : e->a) {
        P sx = simp(x);
        if (sx->kind == NUM)
          cv.push_back(sx->value);
        else
          all = false;
        args.push_back(sx);
      }
      double folded;
      if (all && fold_call(e->name, cv, folded)) return num(folded);
      return call(e->name, args);
    }
  }
  return e;
}

static std::string fmt(double v) {
  char buf[64];
  if (v == std::floor(v) && std::fabs(v) < 1e15)
    snprintf(buf, sizeof(buf), "%.0f", v);
  else
    snprintf(buf, sizeof(buf), "%.17g", v);
  return buf;
}

static std::string str(const P &e) {
  switch (e->kind) {
    case NUM: return fmt(e->value);
    case VAR: return e->name;
    case NEG: return "(-" + str(e->a[0]) + ")";
    case NOT: return "(!" + str(e->a[0]) + ")";
    case BIN:
    case CMP:
    case LOGIC: return "(" + str(e->a[0]) + " " + e->name + " " + str(e->a[1]) + ")";
    case CALL: {
      std::string r = e->name + "(";
      for (size_t i = 0; i < e->a.size(); ++i) r += (i ? ", " : "") + str(e->a[i]);
      return r + ")";
    }
  }
  return "?";
}

// ---- code generation ------------------------------------------------------------------------------
struct Val {
  std::string code;
  bool cplx;
};

struct Gen {
  std::ostringstream body;
  int tmp = 0;
  std::map<std::string, Val> syms;  // variable / constant name -> value expression
  std::map<const Node *, Val> memo;

  static std::string lit(double v) {
    char buf[64];
    if (std::isinf(v)) return v > 0 ? "(1.0/0.0)" : "(-1.0/0.0)";
    if (std::isnan(v)) return "(0.0/0.0)";
    snprintf(buf, sizeof(buf), "%.17g", v);
    std::string s = buf;
    if (s.find_first_of(".eEn") == std::string::npos) s += ".0";
    return "(" + s + ")";
  }
  Val emit(const std::string &expr, bool c) {
    const std::string name = "t" + std::to_string(tmp++);
    body << "    const " << (c ? "c128 " : "double ") << name << " = " << expr << ";\n";
    return Val{name, c};
  }
  static std::string C(const Val &v) { return v.cplx ? v.code : "c128{" + v.code + ", 0.0}"; }

  Val gen(const P &e) {
    auto it = memo.find(e.get());
    if (it != memo.end()) return it->second;
    Val r = gen1(e);
    memo[e.get()] = r;
    return r;
  }
  Val gen1(const P &e) {
    switch (e->kind) {
      case NUM: return Val{lit(e->value), false};
      case VAR: {
        auto it = syms.find(e->name);
        if (it == syms.end()) throw Error{"unknown variable '" + e->name + "'"};
        return it->second;
      }
      case NEG: {
        const Val x = gen(e->a[0]);
        return emit(x.cplx ? "cneg(" + x.code + ")" : "-" + x.code, x.cplx);
      }
      case NOT: {
        const Val x = gen(e->a[0]);
        if (x.cplx) throw Error{"logical not of a complex value"};
        return emit("(" + x.code + " == 0.0) ? 1.0 : 0.0", false);
      }
      case CMP: {
        const Val l = gen(e->a[0]), r = gen(e->a[1]);
        if (l.cplx || r.cplx) throw Error{"comparison of complex values"};
        return emit("(" + l.code + " " + e->name + " " + r.code + ") ? 1.0 : 0.0", false);
      }
      case LOGIC: {
        const Val l = gen(e->a[0]), r = gen(e->a[1]);
        if (l.cplx || r.cplx) throw Error{"logical operation on complex values"};
        const char *op = e->name == "&" ? "&&" : "||";
        return emit("((" + l.code + " != 0.0) " + op + " (" + r.code + " != 0.0)) ? 1.0 : 0.0", false);
      }
      case BIN: {
        const char op = e->name[0];
        if (op == '^') return gen_pow(e->a[0], e->a[1]);
        const Val l = gen(e->a[0]), r = gen(e->a[1]);
        if (!l.cplx && !r.cplx) {
          if (op == '%') return emit("mrl_remainder(" + l.code + ", " + r.code + ")", false);
          return emit(l.code + " " + std::string(1, op) + " " + r.code, false);
        }
        // type promotion as ATen does it: the real operand becomes (re, 0); for * and / by a real this is a scaling
        switch (op) {
          case '+': return emit("cadd(" + C(l) + ", " + C(r) + ")", true);
          case '-': return emit("csub(" + C(l) + ", " + C(r) + ")", true);
          case '*':
            if (!l.cplx) return emit("cscale(" + r.code + ", " + l.code + ")", true);
            if (!r.cplx) return emit("cscale(" + l.code + ", " + r.code + ")", true);
            return emit("cmul(" + l.code + ", " + r.code + ")", true);
          case '/':
            if (!r.cplx) return emit("cdivr(" + l.code + ", " + r.code + ")", true);
            return emit("cdiv(" + C(l) + ", " + r.code + ")", true);
          default: throw Error{"operator % on complex values"};
        }
      }
      case CALL: return gen_call(e);
    }
    throw Error{"internal: bad node"};
  }
  Val gen_pow(const P &b, const P &x) {
    const Val base = gen(b);
    if (base.cplx) {
      if (x->kind == NUM && x->value == 2.0) return emit("cmul(" + base.code + ", " + base.code + ")", true);
      throw Error{"general powers of complex values are not supported"};
    }
    if (x->kind == NUM) {  // ATen pow(Tensor, Scalar) special cases
      const double p = x->value;
      const std::string &c = base.code;
      if (p == 2.0) return emit(c + " * " + c, false);
      if (p == 3.0) return emit("(" + c + " * " + c + ") * " + c, false);
      if (p == -2.0) return emit("1.0 / (" + c + " * " + c + ")", false);
      if (p == 0.5) return emit("sqrt(" + c + ")", false);
      if (p == -0.5) return emit("1.0 / sqrt(" + c + ")", false);
      if (p == -1.0) return emit("1.0 / " + c, false);
    }
    const Val ex = gen(x);
    if (ex.cplx) throw Error{"complex exponents are not supported"};
    return emit("pow(" + base.code + ", " + ex.code + ")", false);
  }
  Val gen_call(const P &e) {
    const std::string &f = e->name;
    std::vector<Val> a;
    for (const P &x : e->a) a.push_back(gen(x));
    static const std::set<std::string> one = {"sin",  "cos",  "tan",   "sinh",  "cosh", "tanh", "asin", "acos", "atan", "asinh",
                                              "acosh", "atanh", "exp",  "exp2",  "log",  "log10", "log2", "sqrt", "ceil", "floor",
                                              "trunc"};
    if (a.size() == 1 && !a[0].cplx) {
      if (one.count(f)) return emit(f + "(" + a[0].code + ")", false);
      if (f == "abs") return emit("fabs(" + a[0].code + ")", false);
      if (f == "rsqrt") return emit("1.0 / sqrt(" + a[0].code + ")", false);
      if (f == "round") return emit("rint(" + a[0].code + ")", false);  // torch.round: half to even
    }
    if (a.size() == 1 && a[0].cplx) {
      if (f == "exp") return emit("cexp_(" + a[0].code + ")", true);
      if (f == "abs") return emit("hypot(" + a[0].code + ".x, " + a[0].code + ".y)", false);
      if (f == "real") return emit(a[0].code + ".x", false);
      if (f == "imag") return emit(a[0].code + ".y", false);
      if (f == "conj") return emit("c128{" + a[0].code + ".x, -" + a[0].code + ".y}", true);
    }
    if (a.size() == 2 && !a[0].cplx && !a[1].cplx) {
      if (f == "min") return emit("fmin(" + a[0].code + ", " + a[1].code + ")", false);
      if (f == "max") return emit("fmax(" + a[0].code + ", " + a[1].code + ")", false);
      if (f == "atan2" || f == "hypot" || f == "pow") return emit(f + "(" + a[0].code + ", " + a[1].code + ")", false);
    }
    if (a.size() == 3 && f == "if" && !a[0].cplx) {
      if (a[1].cplx || a[2].cplx) return emit("(" + a[0].code + " != 0.0) ? " + C(a[1]) + " : " + C(a[2]), true);
      return emit("(" + a[0].code + " != 0.0) ? " + a[1].code + " : " + a[2].code, false);
    }
    throw Error{"Unknown or unsupported function: " + f};
  }
};

static const char *kPrelude = R"(
struct c128 { double x, y; };
__device__ __forceinline__ c128 cadd(c128 a, c128 b) { return c128{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c128 csub(c128 a, c128 b) { return c128{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c128 cneg(c128 a) { return c128{-a.x, -a.y}; }
__device__ __forceinline__ c128 cmul(c128 a, c128 b) { return c128{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ c128 cscale(c128 a, double s) { return c128{a.x * s, a.y * s}; }
__device__ __forceinline__ c128 cdivr(c128 a, double s) { return c128{a.x / s, a.y / s}; }
__device__ __forceinline__ c128 cdiv(c128 a, c128 b) {
  const double d = b.x * b.x + b.y * b.y;
  return c128{(a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d};
}
__device__ __forceinline__ c128 cexp_(c128 a) { const double m = exp(a.x); return c128{m * cos(a.y), m * sin(a.y)}; }
__device__ __forceinline__ double mrl_remainder(double a, double b) { const double r = fmod(a, b); return (r != 0.0 && ((r < 0.0) != (b < 0.0))) ? r + b : r; }
struct ExprArgs {
  const double *in[16];
  double *out;
  long long n;
  long long dims[3];
  const double *ax[3];   // by USER axis: x, y, z
  const double *kax[3];
  int map[3];            // internal axis (0..2) that carries user axis d
  double t;
  double consts[16];
};
)";

}  // namespace ex
}  // namespace mrl

using namespace mrl;
using namespace mrl::ex;

struct mrl_parsed {
  mrl_ctx *ctx = nullptr;
  P ast;
  std::vector<std::string> inputs;
  std::vector<int> input_cplx;
  std::vector<std::string> const_names;
  std::vector<double> const_values;
  bool extra = false;
  int space = 0;
  bool out_cplx = false;
  std::string text, source, err;
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;
  // k_z_fwd<N, CH> instances with this expression compiled in as the chemical potential (MRL_FE_PARSED)
  std::map<int, std::pair<hipModule_t, hipFunction_t>> zfwd;
};

struct ExprArgsHost {
  const double *in[16];
  double *out;
  long long n;
  long long dims[3];
  const double *ax[3];
  const double *kax[3];
  int map[3];
  double t;
  double consts[16];
};

static int build_source(mrl_parsed *p) {
  Gen g;
  std::ostringstream head;
  for (size_t i = 0; i < p->inputs.size(); ++i) {
    const std::string v = "v" + std::to_string(i);
    if (p->input_cplx[i])
      head << "    const c128 " << v << " = reinterpret_cast<const c128 *>(a.in[" << i << "])[e];\n";
    else
      head << "    const double " << v << " = a.in[" << i << "][e];\n";
    g.syms[p->inputs[i]] = Val{v, p->input_cplx[i] != 0};
  }
  for (size_t i = 0; i < p->const_names.size(); ++i) g.syms[p->const_names[i]] = Val{"a.consts[" + std::to_string(i) + "]", false};
  if (p->extra) {
    head << "    const long long r_ = e / a.dims[2];\n"
            "    const long long ii[3] = {r_ / a.dims[1], r_ % a.dims[1], e % a.dims[2]};\n"
            "    const double x = a.ax[0][ii[a.map[0]]], y = a.ax[1][ii[a.map[1]]], z = a.ax[2][ii[a.map[2]]];\n"
            "    const double kx = a.kax[0][ii[a.map[0]]], ky = a.kax[1][ii[a.map[1]]], kz = a.kax[2][ii[a.map[2]]];\n"
            "    const double k2 = kx * kx + ky * ky + kz * kz;\n";
    for (const char *s : {"x", "y", "z", "kx", "ky", "kz", "k2"}) g.syms[s] = Val{s, false};
    g.syms["t"] = Val{"a.t", false};
    g.syms["pi"] = Val{Gen::lit(M_PI), false};
    g.syms["e"] = Val{Gen::lit(std::exp(1.0)), false};
    g.syms["i"] = Val{"c128{0.0, 1.0}", true};
  }
  Val out;
  try {
    out = g.gen(p->ast);
  } catch (const Error &er) {
    p->err = er.msg;
    return MRL_ERR_INVALID;
  }
  p->out_cplx = out.cplx;
  std::ostringstream src;
  src << kPrelude << "extern \"C\" __global__ void __launch_bounds__(256) mrl_expr(ExprArgs a) {\n"
      << "  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < a.n; e += (long long)gridDim.x * 256) {\n"
      << head.str() << g.body.str();
  if (out.cplx)
    src << "    reinterpret_cast<c128 *>(a.out)[e] = " << out.code << ";\n";
  else
    src << "    a.out[e] = " << out.code << ";\n";
  src << "  }\n}\n";
  p->source = src.str();
  return MRL_OK;
}

static int compile_module(mrl_parsed *p) {
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, p->source.c_str(), "mrl_expr.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    p->err = "hiprtcCreateProgram failed";
    return MRL_ERR_HIP;
  }
  const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off"};
  const hiprtcResult rc = hiprtcCompileProgram(prog, 3, opts);
  if (rc != HIPRTC_SUCCESS) {
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    std::string log(ls, '\0');
    if (ls) hiprtcGetProgramLog(prog, &log[0]);
    p->err = "hiprtc compilation failed: " + log;
    hiprtcDestroyProgram(&prog);
    return MRL_ERR_HIP;
  }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  std::vector<char> code(cs);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  if (hipModuleLoadData(&p->module, code.data()) != hipSuccess ||
      hipModuleGetFunction(&p->fn, p->module, "mrl_expr") != hipSuccess) {
    p->err = "loading the compiled expression kernel failed";
    return MRL_ERR_HIP;
  }
  return MRL_OK;
}

// ---- the parsed expression as the chemical potential of the fused Cahn-Hilliard z pass ---------------------
static const char *kPow2Embed =
#include "pow2_embed.inc"
    ;

namespace mrl {

int parsed_check_mu(mrl_ctx *ctx, const mrl_parsed *p) {
  if (!p) return set_error(ctx, MRL_ERR_INVALID, "MRL_FE_PARSED: mrl_ch_params.parsed is null");
  if (p->ctx != ctx) return set_error(ctx, MRL_ERR_INVALID, "MRL_FE_PARSED: the expression belongs to another context");
  if (p->inputs.size() != 1 || p->input_cplx[0] || p->out_cplx || p->extra)
    return set_error(ctx, MRL_ERR_INVALID,
                     "MRL_FE_PARSED: the free energy must be a real expression of exactly one real input (no extra symbols)");
  if (p->const_values.size() > 8) return set_error(ctx, MRL_ERR_INVALID, "MRL_FE_PARSED: at most 8 named constants");
  return MRL_OK;
}

int parsed_eval1(mrl_parsed *p, const double *c, double *mu, long long n) {
  const double *in[1] = {c};
  return mrl_parsed_eval(p, in, mu, n, 0.0);
}

// the same tree as the stand-alone kernel, emitted as a device function of (c, named constants)
static int build_mu_function(mrl_parsed *p, std::string &out) {
  ex::Gen g;
  g.syms[p->inputs[0]] = ex::Val{"c_", false};
  for (size_t i = 0; i < p->const_names.size(); ++i) g.syms[p->const_names[i]] = ex::Val{"k_[" + std::to_string(i) + "]", false};
  ex::Val r;
  try {
    r = g.gen(p->ast);
  } catch (const ex::Error &er) {
    p->err = er.msg;
    return MRL_ERR_INVALID;
  }
  std::string body = g.body.str();
  // the kernel generator indents for a loop body and names the temporaries t<i>: reuse as is
  // contraction is switched off for the expression only: the transform around it keeps the AOT build's code generation,
  // so the run-time compiled pass is bit-identical to a built-in family with the same tree
  out = "namespace mrl { namespace p2 {\n__device__ double mrl_user_mu(double c_, const double *k_) {\n#pragma clang fp contract(off)\n" + body + "    return " + r.code +
        ";\n}\n} }\n";
  return MRL_OK;
}

// lines per workgroup / threads per workgroup of the z kernels (ZPlan<N>, fft_pow2.h)
// (ea: the fused inverse + forward kernel, ZPlanEA<N>)
// (experiment bit 1 << 29: the uniform 30- / 20-point plans where the two-stage plans of fft_two_z.h would run, as in ch_planned.hip)
static bool two_stage_z(const mrl_ctx *ctx, int N) { return p2::two_stage_z_len(N) && !(ctx->exp & (1 << 29)); }

static int plan_shape(mrl_ctx *ctx, int N, int *T, int *NT, size_t *lds, bool ea = false) {
  *T = 0;
  if (two_stage_z(ctx, N)) {
    switch (N) {
#define MRL_Z2(NN_) case NN_: *T = p2::ZPlan2<NN_>::LPB; *NT = p2::ZPlan2<NN_>::NT; *lds = p2::lds_two_z<NN_>(ea ? 2 : 1); break;
      MRL_Z2(120) MRL_Z2(150) MRL_Z2(160) MRL_Z2(180) MRL_Z2(240) MRL_Z2(300) MRL_Z2(320)
#undef MRL_Z2
    }
    return MRL_OK;
  }
  // the forward z pass exists for every planned length (the radix-30 / radix-20 lengths of ch_planned.hip included: their k_z_fwd is a
  // plain kernel with one array per thread); the fused inverse + forward pass only for the lengths of the fused family
  if (ea ? !pow2_ok(N) : !plain_ok(N)) return set_error(ctx, MRL_ERR_UNSUPPORTED, "parsed z pass: unsupported length %d", N);
  if (ea) {
    MRL_SWITCH_N(N, (*T = p2::ZPlanEA<NN>::T, *NT = p2::ZPlanEA<NN>::NT, *lds = p2::lds_line_ea<NN>()));
  } else {
    MRL_SWITCH_N_PLAIN(N, (*T = p2::ZPlan<NN>::T, *NT = p2::ZPlan<NN>::NT, *lds = p2::lds_line<NN>()));
  }
  if (*T == 0) return set_error(ctx, MRL_ERR_UNSUPPORTED, "parsed z pass: unsupported length %d", N);
  return MRL_OK;
}

// run-time compiled z passes with the generated chemical potential: mode 1 = k_z_fwd<N, 1, PARSED>, 2 = k_z_fwd<N, 2, PARSED>,
// 3 = k_z_inv_fwd<N, PARSED>, 4 = its MU_ONLY form; cached per (N, mode)
static int parsed_z_kernel(mrl_ctx *ctx, mrl_parsed *p, int N, int mode, hipFunction_t *fn_out) {
  if (mode < 1 || mode > 4) return set_error(ctx, MRL_ERR_INVALID, "parsed z pass: mode %d", mode);
  const bool two = two_stage_z(ctx, N);
  if (two && (mode == 2 || mode == 4)) return set_error(ctx, MRL_ERR_UNSUPPORTED, "parsed z pass: no carry-over form for length %d", N);
  const int key = N * 16 + mode + (two ? 8 : 0);
  auto it = p->zfwd.find(key);
  if (it == p->zfwd.end()) {
    std::string mu_fn;
    if (build_mu_function(p, mu_fn) != MRL_OK) return set_error(ctx, MRL_ERR_INVALID, "expression: %s", p->err.c_str());
    // device-only translation unit: vector types and the public enum values, the embedded kernel headers, the generated
    // chemical potential; the kernel is instantiated through a name expression
    std::string src =
        "typedef double2 cplx;\n"
        "#define MRL_FE_DOUBLE_WELL 0\n#define MRL_FE_PFHUB 1\n#define MRL_FE_PARSED 2\n"
        "namespace mrl { typedef ::cplx cplx; }\n";
    src += kPow2Embed;
    src += mu_fn;
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "mrl_z_fwd_parsed.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
      return set_error(ctx, MRL_ERR_HIP, "hiprtcCreateProgram failed");
    const std::string name = two ? (mode == 3 ? "mrl::p2::k_z_inv_fwd2<" + std::to_string(N) + ", 2>" : "mrl::p2::k_z_fwd2<" + std::to_string(N) + ", 1, 2>")
                             : mode == 3 ? "mrl::p2::k_z_inv_fwd<" + std::to_string(N) + ", 2, false>"
                             : mode == 4 ? "mrl::p2::k_z_inv_fwd<" + std::to_string(N) + ", 2, true>"
                                         : "mrl::p2::k_z_fwd<" + std::to_string(N) + ", " + std::to_string(mode) + ", 2>";
    hiprtcAddNameExpression(prog, name.c_str());
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    if (hiprtcCompileProgram(prog, 3, opts) != HIPRTC_SUCCESS) {
      size_t ls = 0;
      hiprtcGetProgramLogSize(prog, &ls);
      std::string log(ls, '\0');
      if (ls) hiprtcGetProgramLog(prog, &log[0]);
      hiprtcDestroyProgram(&prog);
      return set_error(ctx, MRL_ERR_HIP, "hiprtc compilation of the fused z pass failed: %s", log.c_str());
    }
    const char *lowered = nullptr;
    if (hiprtcGetLoweredName(prog, name.c_str(), &lowered) != HIPRTC_SUCCESS || !lowered) {
      hiprtcDestroyProgram(&prog);
      return set_error(ctx, MRL_ERR_HIP, "hiprtcGetLoweredName failed for %s", name.c_str());
    }
    const std::string mangled = lowered;
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess || hipModuleGetFunction(&fn, mod, mangled.c_str()) != hipSuccess)
      return set_error(ctx, MRL_ERR_HIP, "loading the run-time compiled z pass failed");
    it = p->zfwd.emplace(key, std::make_pair(mod, fn)).first;
  }
  *fn_out = it->second.second;
  return MRL_OK;
}

struct ChDevHost {  // = p2::ChDev
  int family;
  double c0, c1, c2;
  double k[8];
};
static ChDevHost parsed_chdev(const mrl_parsed *p) {
  ChDevHost chp{};
  chp.family = MRL_FE_PARSED;
  for (size_t i = 0; i < p->const_values.size(); ++i) chp.k[i] = p->const_values[i];
  return chp;
}

int parsed_z_fwd_launch(mrl_ctx *ctx, mrl_parsed *p, int N, int mode, const double *in, cplx *out0, cplx *out1, double *mu_out,
                        long long nlines, unsigned lay_lpp, unsigned lay_pad) {
  int T = 0, NT = 0;
  size_t lds = 0;  // twiddle table + the line tile of the z kernels (MapLine<N>)
  MRL_TRY(plan_shape(ctx, N, &T, &NT, &lds));
  if (mode != 1 && mode != 2) return set_error(ctx, MRL_ERR_INVALID, "parsed z pass: mode %d", mode);
  hipFunction_t fn;
  MRL_TRY(parsed_z_kernel(ctx, p, N, mode, &fn));
  // the argument block of k_z_fwd(const double*, cplx*, cplx*, double*, ChDev, long long, const cplx*, ZLay)
  ChDevHost chp = parsed_chdev(p);
  const cplx *tw = ctx->ax[2].d_tw;
  struct { unsigned lpp, pad; } zl = {lay_lpp, lay_pad};  // = p2::ZLay
  void *params[] = {&in, &out0, &out1, &mu_out, &chp, &nlines, &tw, &zl};
  const long long nb = (nlines + T - 1) / T;
  MRL_HIP(ctx, hipModuleLaunchKernel(fn, (unsigned)nb, 1, 1, (unsigned)NT, 1, 1, (unsigned)lds, ctx->stream, params, nullptr));
  return MRL_OK;
}

// k_z_inv_fwd<N, PARSED>(const cplx*, cplx*, cplx*, double*, ChDev, double, long long, const cplx*, ZLay); nlines = line pairs
int parsed_z_inv_fwd_launch(mrl_ctx *ctx, mrl_parsed *p, int N, const cplx *in, cplx *out0, cplx *out1, double *mu_out,
                            double scale, long long nlines, bool mu_only, unsigned lay_lpp, unsigned lay_pad) {
  int T = 0, NT = 0;
  size_t lds = 0;  // twiddle table + the line tile of the z kernels (MapLine<N>)
  MRL_TRY(plan_shape(ctx, N, &T, &NT, &lds, true));
  hipFunction_t fn;
  MRL_TRY(parsed_z_kernel(ctx, p, N, mu_only ? 4 : 3, &fn));
  ChDevHost chp = parsed_chdev(p);
  const cplx *tw = ctx->ax[2].d_tw;
  struct { unsigned lpp, pad; } zl = {lay_lpp, lay_pad};  // = p2::ZLay
  void *params[] = {&in, &out0, &out1, &mu_out, &chp, &scale, &nlines, &tw, &zl};
  const long long nb = (nlines + T - 1) / T;
  MRL_HIP(ctx, hipModuleLaunchKernel(fn, (unsigned)nb, 1, 1, (unsigned)NT, 1, 1, (unsigned)lds, ctx->stream, params, nullptr));
  return MRL_OK;
}

}  // namespace mrl

extern "C" {

int mrl_parsed_create(mrl_ctx *ctx, mrl_parsed **out, const char *expression, int n_inputs, const char *const *input_names,
                      const int *input_is_complex, int n_constants, const char *const *constant_names,
                      const double *constant_values, int n_derivatives, const char *const *derivatives, int extra_symbols,
                      int space) {
  if (!out || !expression || n_inputs < 0 || n_inputs > 16 || n_constants < 0 || n_constants > 16 ||
      (n_inputs && !input_names) || (n_constants && (!constant_names || !constant_values)) ||
      (n_derivatives && !derivatives))
    return set_error(ctx, MRL_ERR_INVALID, "mrl_parsed_create: bad argument");
  *out = nullptr;
  auto p = std::make_unique<mrl_parsed>();
  p->ctx = ctx;
  p->extra = extra_symbols != 0;
  p->space = space;
  static const char *reserved[] = {"i", "x", "kx", "y", "ky", "z", "kz", "k2", "t", "pi", "e"};
  std::set<std::string> known;
  auto is_reserved = [&](const std::string &n) {
    if (!p->extra) return false;
    for (const char *r : reserved)
      if (n == r) return true;
    return false;
  };
  for (int i = 0; i < n_inputs; ++i) {
    const std::string n = input_names[i];
    if (known.count(n)) return set_error(ctx, MRL_ERR_INVALID, "inputs: Duplicate buffer name.");
    if (is_reserved(n)) return set_error(ctx, MRL_ERR_INVALID, "inputs: Cannot use reserved name '%s' for coupled fields.", n.c_str());
    known.insert(n);
    p->inputs.push_back(n);
    p->input_cplx.push_back(input_is_complex ? input_is_complex[i] : 0);
  }
  for (int i = 0; i < n_constants; ++i) {
    const std::string n = constant_names[i];
    if (known.count(n)) return set_error(ctx, MRL_ERR_INVALID, "constant_names: Duplicate constant name.");
    if (is_reserved(n)) return set_error(ctx, MRL_ERR_INVALID, "constant_names: Cannot use reserved name '%s' for constant.", n.c_str());
    known.insert(n);
    p->const_names.push_back(n);
    p->const_values.push_back(constant_values[i]);
  }
  if (p->extra)
    for (const char *r : reserved) known.insert(r);
  try {
    Parser ps(expression, known);
    P ast = ps.statements();
    for (int i = 0; i < n_derivatives; ++i) {
      const std::string d = derivatives[i];
      if (std::find(p->inputs.begin(), p->inputs.end(), d) == p->inputs.end())
        return set_error(ctx, MRL_ERR_INVALID,
                         "derivatives: Derivative w.r.t `%s` was requested, but it is not listed in `inputs`.", d.c_str());
      ast = diff(ast, d);
    }
    p->ast = simp(ast);
  } catch (const Error &e) {
    return set_error(ctx, MRL_ERR_INVALID, "expression: Invalid function: %s", e.msg.c_str());
  }
  p->text = str(p->ast);
  if (build_source(p.get()) != MRL_OK) return set_error(ctx, MRL_ERR_INVALID, "expression: %s", p->err.c_str());
  if (ctx) {  // without a context: parse / differentiate / simplify / generate only (no GPU needed)
    const int rc = compile_module(p.get());
    if (rc != MRL_OK) return set_error(ctx, rc, "%s", p->err.c_str());
  }
  *out = p.release();
  return MRL_OK;
}

void mrl_parsed_destroy(mrl_parsed *p) {
  if (!p) return;
  if (p->module) (void)hipModuleUnload(p->module);
  for (auto &kv : p->zfwd)
    if (kv.second.first) (void)hipModuleUnload(kv.second.first);
  delete p;
}

int mrl_parsed_is_complex(const mrl_parsed *p) { return p && p->out_cplx ? 1 : 0; }
const char *mrl_parsed_string(const mrl_parsed *p) { return p ? p->text.c_str() : ""; }
const char *mrl_parsed_source(const mrl_parsed *p) { return p ? p->source.c_str() : ""; }

int mrl_parsed_eval(mrl_parsed *p, const double *const *d_inputs, double *d_out, int64_t count, double time) {
  if (!p || !p->ctx) return MRL_ERR_INVALID;
  mrl_ctx *ctx = p->ctx;
  if (!p->fn) return set_error(ctx, MRL_ERR_INVALID, "mrl_parsed_eval: expression was created without a context");
  if (!d_out || count < 0 || (!p->inputs.empty() && !d_inputs)) return set_error(ctx, MRL_ERR_INVALID, "mrl_parsed_eval: bad argument");
  ExprArgsHost a{};
  for (size_t i = 0; i < p->inputs.size(); ++i) {
    if (!d_inputs[i]) return set_error(ctx, MRL_ERR_INVALID, "mrl_parsed_eval: input '%s' is null", p->inputs[i].c_str());
    a.in[i] = d_inputs[i];
  }
  a.out = d_out;
  a.t = time;
  for (size_t i = 0; i < p->const_values.size(); ++i) a.consts[i] = p->const_values[i];
  if (p->extra) {
    const long long *dims = p->space == 1 ? ctx->nrec : ctx->nloc;
    for (int d = 0; d < 3; ++d) {
      a.dims[d] = dims[d];
      // user axis d lives on internal axis off+d; axes beyond `dim` map to an unused internal axis (value {0})
      const int ia = d < ctx->dim ? ctx->off + d : (ctx->off > 0 ? d - ctx->dim : d);
      a.map[d] = ia;
      a.ax[d] = ctx->d_x[ia];
      a.kax[d] = ctx->d_k[ia];
    }
    const long long total = dims[0] * dims[1] * dims[2];
    if (count != total)
      return set_error(ctx, MRL_ERR_INVALID, "mrl_parsed_eval: extra_symbols expressions evaluate on the whole %s grid (%lld points)",
                       p->space == 1 ? "reciprocal" : "real", total);
  }
  a.n = count;
  if (count == 0) return MRL_OK;
  long long nb = (count + 255) / 256;
  if (nb > 8192) nb = 8192;
  void *params[] = {&a};
  ProfScope ps(ctx, "parsed_compute");
  MRL_HIP(ctx, hipModuleLaunchKernel(p->fn, (unsigned)nb, 1, 1, 256, 1, 1, 0, ctx->stream, params, nullptr));
  return MRL_OK;
}

}  // extern "C"
