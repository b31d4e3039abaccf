"""A small interpreter for the WGSL subset the reference's shaders are written in.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): it exists so that the reference's own
shader TEXT (src/passes/shaders/*.wgsl, read in place from the reference checkout, never copied)
can be executed on this machine and its outputs committed as golden vectors
(tests/golden/make_wgsl_vectors.py -> tests/golden/wgsl_vectors.npz).  The C oracle and the HIP
kernels are then checked against vectors that come from the reference's code rather than from a
restatement of it.

What it implements: module-scope `const`, `struct`, resource `var` declarations, functions;
`let` / `var` / assignment (incl. compound and ++/--) / `if` / `for` / `while` / `break` /
`continue` / `return`; f32, i32, u32, bool, AbstractInt / AbstractFloat with WGSL's conversion
rank (abstract values are evaluated in Python int / float64 and converted on first contact with a
concrete type; `let` / `var` without a type concretise to i32 / f32), vecN, mat3x3, fixed arrays,
structs, `ptr<function, T>` (& and *), swizzles, and the builtins the shaders call.

Where WGSL leaves results implementation-defined the interpreter follows DESIGN.md "Pinned
arithmetic": no FMA contraction, dot summed left to right, normalize = v / sqrt(dot(v, v)),
mix(a, b, t) = a * (1 - t) + b * t, reflect(i, n) = i - 2 * dot(n, i) * n, bilinear sampling
x * W - 0.5 / floor / fp32 weights / horizontal-then-vertical, and transcendental functions
supplied by the caller (the pinned polynomial implementations).
"""
import math
import re

import numpy as np

F32 = np.float32


class I32(int):
    pass


class U32(int):
    pass


def _i32(v):
    v = int(v) & 0xFFFFFFFF
    return I32(v - (1 << 32) if v & 0x80000000 else v)


def _u32(v):
    return U32(int(v) & 0xFFFFFFFF)


def kind(x):
    t = type(x)
    if t is F32:
        return "f"
    if t is I32:
        return "i"
    if t is U32:
        return "u"
    if t is bool or t is np.bool_:
        return "b"
    if t is float:
        return "F"
    if t is int:
        return "I"
    raise TypeError(f"not a scalar: {x!r}")


def convert(x, k):
    """Value conversion between scalar kinds (constructors f32(), u32(), ... and concretisation)."""
    kx = kind(x)
    if kx == k:
        return x
    if k == "f":
        if kx == "b":
            return F32(1.0 if x else 0.0)
        with np.errstate(over="ignore"):
            return F32(float(x)) if kx in "Ff" else F32(int(x))       # int -> f32 rounds to nearest even
    if k == "F":
        return float(x)
    if k in "iu":
        if kx in "fF":
            v = float(x)
            if v != v:
                v = 0.0
            lo, hi = (-(1 << 31), (1 << 31) - 1) if k == "i" else (0, (1 << 32) - 1)
            v = max(lo, min(hi, math.trunc(v) if math.isfinite(v) else (hi if v > 0 else lo)))
            return I32(v) if k == "i" else U32(v)
        return _i32(int(x)) if k == "i" else _u32(int(x))
    if k == "I":
        return int(x)
    if k == "b":
        return bool(x)
    raise TypeError(k)


RANK = {"I": 0, "F": 1}


def unify(a, b):
    ka, kb = kind(a), kind(b)
    if ka == kb:
        return a, b, ka
    if ka in RANK and kb in RANK:
        return float(a), float(b), "F"
    if ka in RANK and kb not in RANK:
        if ka == "F" and kb in "iu":
            raise TypeError("AbstractFloat cannot convert to an integer type")
        return convert(a, kb), b, kb
    if kb in RANK and ka not in RANK:
        if kb == "F" and ka in "iu":
            raise TypeError("AbstractFloat cannot convert to an integer type")
        return a, convert(b, ka), ka
    raise TypeError(f"no common type for {ka} and {kb}")


class Vec:
    __slots__ = ("e",)

    def __init__(self, e):
        self.e = list(e)

    def __len__(self):
        return len(self.e)

    def __repr__(self):
        return f"Vec({self.e})"


class Mat:
    __slots__ = ("cols",)

    def __init__(self, cols):
        self.cols = cols


class Struct:
    __slots__ = ("name", "f")

    def __init__(self, name, fields):
        self.name = name
        self.f = fields


class Arr:
    __slots__ = ("e",)

    def __init__(self, e):
        self.e = e


class Ref:
    """ptr<function, T>: a cell (container, key) that can be read and written."""
    __slots__ = ("c", "k")

    def __init__(self, c, k):
        self.c, self.k = c, k

    def get(self):
        c = self.c
        return c.e[self.k] if isinstance(c, (Vec, Arr)) else (c.f[self.k] if isinstance(c, Struct) else c[self.k])

    def set(self, v):
        c = self.c
        if isinstance(c, (Vec, Arr)):
            c.e[self.k] = v
        elif isinstance(c, Struct):
            c.f[self.k] = v
        else:
            c[self.k] = v


def copyval(v):
    if isinstance(v, Vec):
        return Vec(v.e)
    if isinstance(v, Struct):
        return Struct(v.name, {k: copyval(x) for k, x in v.f.items()})
    if isinstance(v, Arr):
        return Arr([copyval(x) for x in v.e])
    if isinstance(v, Mat):
        return Mat([copyval(c) for c in v.cols])
    return v


# ------------------------------------------------------------------------------ scalar arithmetic

def _scalar_bin(op, a, b):
    if op in ("&&", "||"):
        return (bool(a) and bool(b)) if op == "&&" else (bool(a) or bool(b))
    if op in ("<<", ">>"):
        ka = kind(a)
        s = int(b)
        if ka in "ui":
            s %= 32      # WGSL "Bit Expressions": a concrete 32-bit operand is shifted by e2 modulo the bit width
        if ka == "u":
            return _u32(int(a) << s) if op == "<<" else _u32(int(a) >> s)
        if ka == "i":
            return _i32(int(a) << s) if op == "<<" else _i32(int(a) >> s)
        return (int(a) << s) if op == "<<" else (int(a) >> s)
    a, b, k = unify(a, b)
    if op in ("==", "!=", "<", "<=", ">", ">="):
        if k in "fF":
            x, y = float(a), float(b)
        else:
            x, y = a, b
        return {"==": x == y, "!=": x != y, "<": x < y, "<=": x <= y, ">": x > y, ">=": x >= y}[op]
    if k == "f":
        with np.errstate(all="ignore"):
            if op == "+":
                return F32(a + b)
            if op == "-":
                return F32(a - b)
            if op == "*":
                return F32(a * b)
            if op == "/":
                return F32(np.divide(a, b))
            if op == "%":
                return F32(np.fmod(a, b))
    if k == "F":
        if op == "/":
            return a / b if b != 0 else math.copysign(math.inf, a) * math.copysign(1.0, b) if a != 0 else math.nan
        return {"+": a + b, "-": a - b, "*": a * b, "%": math.fmod(a, b) if b else math.nan}[op]
    if k in "iuI":
        x, y = int(a), int(b)
        if op == "+":
            r = x + y
        elif op == "-":
            r = x - y
        elif op == "*":
            r = x * y
        elif op == "/":
            r = 0 if y == 0 else int(math.trunc(x / y)) if abs(x) < 1 << 52 else (abs(x) // abs(y)) * (1 if (x < 0) == (y < 0) else -1)
        elif op == "%":
            r = 0 if y == 0 else x - y * int(math.trunc(x / y))
        elif op == "&":
            r = x & y
        elif op == "|":
            r = x | y
        elif op == "^":
            r = x ^ y
        else:
            raise TypeError(op)
        return _i32(r) if k == "i" else (_u32(r) if k == "u" else r)
    if k == "b" and op in ("&", "|"):
        return (a and b) if op == "&" else (a or b)
    raise TypeError(f"operator {op} on kind {k}")


def binop(op, a, b):
    if isinstance(a, Mat) and isinstance(b, Vec) and op == "*":
        # column-major matrix times vector, summed left to right
        out = []
        for r in range(len(a.cols[0])):
            acc = None
            for c in range(len(a.cols)):
                term = _scalar_bin("*", a.cols[c].e[r], b.e[c])
                acc = term if acc is None else _scalar_bin("+", acc, term)
            out.append(acc)
        return Vec(out)
    if isinstance(a, Vec) and isinstance(b, Vec):
        if len(a) != len(b):
            raise TypeError("vector size mismatch")
        return Vec([_scalar_bin(op, x, y) for x, y in zip(a.e, b.e)])
    if isinstance(a, Vec):
        return Vec([_scalar_bin(op, x, b) for x in a.e])
    if isinstance(b, Vec):
        return Vec([_scalar_bin(op, a, y) for y in b.e])
    return _scalar_bin(op, a, b)


def unop(op, a):
    if isinstance(a, Vec):
        return Vec([unop(op, x) for x in a.e])
    if op == "!":
        return not bool(a)
    if op == "-":
        k = kind(a)
        if k == "f":
            return F32(-a)
        if k == "F":
            return -a
        if k == "i":
            return _i32(-int(a))
        if k == "I":
            return -int(a)
    raise TypeError(f"unary {op}")


# ------------------------------------------------------------------------------ lexer / parser

TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<num>0[xX][0-9a-fA-F]+[iu]?|(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?[fiuh]?)
  | (?P<id>[A-Za-z_][A-Za-z0-9_]*)
  | (?P<op>->|==|!=|<=|>=|&&|\|\||\+=|-=|\*=|/=|%=|\+\+|--|<<|>>|[-+*/%<>=!&|^~(){}\[\];:,.@])
""", re.X | re.S)

VEC_NAMES = {f"vec{n}{s}": (n, s) for n in (2, 3, 4) for s in "fiu"}
SCALAR_TYPES = {"f32": "f", "i32": "i", "u32": "u", "bool": "b"}


def tokenize(src):
    out, pos = [], 0
    while pos < len(src):
        m = TOKEN.match(src, pos)
        if not m:
            raise SyntaxError(f"bad character at {pos}: {src[pos:pos + 20]!r}")
        pos = m.end()
        if m.lastgroup == "ws":
            continue
        out.append((m.lastgroup, m.group(m.lastgroup)))
    out.append(("eof", ""))
    return out


class Parser:
    def __init__(self, src):
        self.t = tokenize(src)
        self.i = 0

    def peek(self, k=0):
        return self.t[self.i + k]

    def next(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def accept(self, val):
        if self.t[self.i][1] == val and self.t[self.i][0] in ("op", "id"):
            self.i += 1
            return True
        return False

    def expect(self, val):
        if not self.accept(val):
            raise SyntaxError(f"expected {val!r}, got {self.t[self.i]!r} (token {self.i})")

    def ident(self):
        k, v = self.next()
        if k != "id":
            raise SyntaxError(f"expected identifier, got {v!r}")
        return v

    def attributes(self):
        attrs = []
        while self.peek()[1] == "@":
            self.next()
            name = self.ident()
            args = []
            if self.accept("("):
                while not self.accept(")"):
                    args.append(self.next()[1])
                    self.accept(",")
            attrs.append((name, args))
        return attrs

    # ---- types
    def type_(self):
        name = self.ident()
        if name in SCALAR_TYPES:
            return ("scalar", SCALAR_TYPES[name])
        if name in VEC_NAMES:
            return ("vec",) + VEC_NAMES[name]
        if name in ("vec2", "vec3", "vec4"):
            self.expect("<")
            inner = self.type_()
            self.expect(">")
            return ("vec", int(name[3]), inner[1])
        if name.startswith("mat"):
            if self.accept("<"):
                self.type_()
                self.expect(">")
            return ("mat", name)
        if name == "array":
            self.expect("<")
            inner = self.type_()
            count = None
            if self.accept(","):
                count = self.expr(8)          # additive level: '>' closes the template list
            self.expect(">")
            return ("array", inner, count)
        if name == "ptr":
            self.expect("<")
            self.ident()
            self.expect(",")
            inner = self.type_()
            if self.accept(","):
                self.ident()
            self.expect(">")
            return ("ptr", inner)
        if name.startswith("texture_") or name.startswith("sampler"):
            if self.accept("<"):
                depth = 1
                while depth:
                    v = self.next()[1]
                    depth += (v == "<") - (v == ">")
            return ("resource", name)
        return ("struct", name)

    # ---- expressions (precedence climbing)
    PREC = [("||",), ("&&",), ("|",), ("^",), ("&",), ("==", "!="), ("<", "<=", ">", ">="), ("<<", ">>"), ("+", "-"),
            ("*", "/", "%")]

    def expr(self, level=0):
        if level == len(self.PREC):
            return self.unary()
        left = self.expr(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in self.PREC[level]:
            op = self.next()[1]
            right = self.expr(level + 1)
            left = ("binary", op, left, right)
        return left

    def unary(self):
        k, v = self.peek()
        if k == "op" and v in ("-", "!", "&", "*", "~"):
            self.next()
            e = self.unary()
            return {"-": ("unary", "-", e), "!": ("unary", "!", e), "&": ("addr", e), "*": ("deref", e), "~": ("unary", "~", e)}[v]
        return self.postfix(self.primary())

    def primary(self):
        k, v = self.next()
        if k == "num":
            return ("num", parse_number(v))
        if k == "op" and v == "(":
            e = self.expr()
            self.expect(")")
            return ("paren", e)
        if k == "id":
            if v in ("true", "false"):
                return ("num", v == "true")
            template = None
            if v in ("vec2", "vec3", "vec4", "array") and self.peek()[1] == "<":
                self.i -= 1
                template = self.type_()
                self.expect("(")
                return ("call", v, self.args(), template)
            if self.accept("("):
                return ("call", v, self.args(), None)
            return ("ident", v)
        raise SyntaxError(f"unexpected token {v!r}")

    def args(self):
        out = []
        while not self.accept(")"):
            out.append(self.expr())
            self.accept(",")
        return out

    def postfix(self, e):
        while True:
            if self.accept("."):
                e = ("member", e, self.ident())
            elif self.accept("["):
                idx = self.expr()
                self.expect("]")
                e = ("index", e, idx)
            else:
                return e

    # ---- statements
    def block(self):
        self.expect("{")
        out = []
        while not self.accept("}"):
            out.append(self.statement())
        return ("block", out)

    def simple_statement(self):
        """let / var / assignment / increment / call, without the trailing ';'."""
        if self.peek()[1] in ("let", "var", "const"):
            kw = self.next()[1]
            if self.accept("<"):
                while not self.accept(">"):
                    self.next()
            name = self.ident()
            ty = self.type_() if self.accept(":") else None
            init = self.expr() if self.accept("=") else None
            return (kw if kw != "const" else "let", name, ty, init)
        lhs = self.unary()
        k, v = self.peek()
        if v in ("=", "+=", "-=", "*=", "/=", "%="):
            self.next()
            return ("assign", lhs, v, self.expr())
        if v in ("++", "--"):
            self.next()
            return ("assign", lhs, "+=" if v == "++" else "-=", ("num", 1))
        return ("expr", lhs)

    def statement(self):
        k, v = self.peek()
        if v == "{":
            return self.block()
        if v == "if":
            self.next()
            cond = self.expr()
            then = self.block()
            other = None
            if self.accept("else"):
                other = self.statement() if self.peek()[1] == "if" else self.block()
            return ("if", cond, then, other)
        if v == "for":
            self.next()
            self.expect("(")
            init = None if self.peek()[1] == ";" else self.simple_statement()
            self.expect(";")
            cond = None if self.peek()[1] == ";" else self.expr()
            self.expect(";")
            update = None if self.peek()[1] == ")" else self.simple_statement()
            self.expect(")")
            return ("for", init, cond, update, self.block())
        if v == "while":
            self.next()
            cond = self.expr()
            return ("while", cond, self.block())
        if v == "return":
            self.next()
            e = None if self.peek()[1] == ";" else self.expr()
            self.expect(";")
            return ("return", e)
        if v in ("break", "continue"):
            self.next()
            self.expect(";")
            return (v,)
        s = self.simple_statement()
        self.expect(";")
        return s

    # ---- module
    def module(self):
        consts, structs, resources, funcs = [], {}, {}, {}
        while self.peek()[0] != "eof":
            attrs = self.attributes()
            v = self.peek()[1]
            if v == "const":
                self.next()
                name = self.ident()
                ty = self.type_() if self.accept(":") else None
                self.expect("=")
                consts.append((name, ty, self.expr()))
                self.expect(";")
            elif v == "struct":
                self.next()
                name = self.ident()
                self.expect("{")
                fields = []
                while not self.accept("}"):
                    self.attributes()
                    fname = self.ident()
                    self.expect(":")
                    fields.append((fname, self.type_()))
                    self.accept(",")
                self.accept(";")
                structs[name] = fields
            elif v == "var":
                self.next()
                if self.accept("<"):
                    while not self.accept(">"):
                        self.next()
                name = self.ident()
                self.expect(":")
                resources[name] = self.type_()
                self.expect(";")
            elif v == "fn":
                self.next()
                name = self.ident()
                self.expect("(")
                params = []
                while not self.accept(")"):
                    self.attributes()
                    pname = self.ident()
                    self.expect(":")
                    params.append((pname, self.type_()))
                    self.accept(",")
                ret = None
                if self.accept("->"):
                    self.attributes()
                    ret = self.type_()
                funcs[name] = (params, ret, self.block(), attrs)
            else:
                raise SyntaxError(f"unexpected module-scope token {v!r}")
        return consts, structs, resources, funcs


def parse_number(text):
    t = text
    if t.lower().startswith("0x"):
        suffix = t[-1] if t[-1] in "iu" else ""
        v = int(t[:-1] if suffix else t, 16)
        return _u32(v) if suffix == "u" else (_i32(v) if suffix == "i" else v)
    suffix = t[-1] if t[-1] in "fiuh" else ""
    body = t[:-1] if suffix else t
    is_float = any(c in body for c in ".eE")
    if suffix == "u":
        return _u32(int(body))
    if suffix == "i":
        return _i32(int(body))
    if suffix == "f":
        return F32(float(body))
    return float(body) if is_float else int(body)


# ------------------------------------------------------------------------------ interpreter

class _Return(Exception):
    def __init__(self, v):
        self.v = v


class _Break(Exception):
    pass


class _Continue(Exception):
    pass


SWZ = {"x": 0, "y": 1, "z": 2, "w": 3, "r": 0, "g": 1, "b": 2, "a": 3}


class Texture:
    """texels: (H, W, 4) float32; filter 'linear' | 'nearest'; address 'clamp' | 'repeat'."""

    def __init__(self, texels, filter="linear", address="clamp"):
        self.texels = np.ascontiguousarray(texels, np.float32)
        self.filter, self.address = filter, address
        self.stores = {}


class Interpreter:
    def __init__(self, src, math_fns, resources=None):
        self.consts_src, self.structs, self.resource_types, self.funcs = Parser(src).module()
        self.math = math_fns              # name -> callable on np.float32 scalars (sin cos tan log exp atan2 asin pow)
        self.res = dict(resources or {})  # resource name -> Texture | Arr | Struct | sampler description
        self.globals = {}
        for name, ty, e in self.consts_src:
            v = self.eval(e, [self.globals])
            if ty is not None:
                v = self.coerce(v, ty)
            self.globals[name] = v
        self.calls = 0

    # ---- types
    def zero(self, ty):
        k = ty[0]
        if k == "scalar":
            return {"f": F32(0), "i": I32(0), "u": U32(0), "b": False}[ty[1]]
        if k == "vec":
            return Vec([self.zero(("scalar", ty[2]))] * ty[1])
        if k == "array":
            n = int(self.eval(ty[2], [self.globals]))
            return Arr([self.zero(ty[1]) for _ in range(n)])
        if k == "struct":
            return Struct(ty[1], {f: self.zero(t) for f, t in self.structs[ty[1]]})
        raise TypeError(f"no zero value for {ty}")

    def coerce(self, v, ty):
        """Concretise abstract values towards a declared type."""
        k = ty[0]
        if k == "scalar":
            return convert(v, ty[1]) if kind(v) in "IF" or kind(v) == ty[1] else self._bad(v, ty)
        if k == "vec" and isinstance(v, Vec):
            return Vec([convert(x, ty[2]) if kind(x) in "IF" else x for x in v.e])
        return v

    @staticmethod
    def _bad(v, ty):
        raise TypeError(f"cannot initialise {ty} from {v!r}")

    @staticmethod
    def concretise(v):
        """let / var without a declared type: AbstractInt -> i32, AbstractFloat -> f32."""
        if isinstance(v, Vec):
            return Vec([Interpreter.concretise(x) for x in v.e])
        if isinstance(v, Arr):
            return Arr([Interpreter.concretise(x) for x in v.e])
        if type(v) is float:
            return F32(v)
        if type(v) is int:
            return _i32(v)
        return v

    # ---- expressions
    def lookup(self, name, env):
        for scope in reversed(env):
            if name in scope:
                return scope[name]
        if name in self.res:
            return self.res[name]
        raise NameError(name)

    def lvalue(self, e, env):
        t = e[0]
        if t == "ident":
            for scope in reversed(env):
                if e[1] in scope:
                    return Ref(scope, e[1])
            raise NameError(e[1])
        if t == "paren":
            return self.lvalue(e[1], env)
        if t == "deref":
            p = self.eval(e[1], env)
            return p
        if t == "member":
            base = self.lvalue(e[1], env).get()
            if isinstance(base, Struct):
                return Ref(base, e[2])
            if isinstance(base, Vec) and len(e[2]) == 1:
                return Ref(base, SWZ[e[2]])
            raise TypeError(f"cannot assign to member {e[2]}")
        if t == "index":
            base = self.lvalue(e[1], env).get()
            return Ref(base, int(self.eval(e[2], env)))
        raise TypeError(f"not an lvalue: {e}")

    def eval(self, e, env):
        t = e[0]
        if t == "num":
            return e[1]
        if t == "ident":
            return self.lookup(e[1], env)
        if t == "paren":
            return self.eval(e[1], env)
        if t == "binary":
            op = e[1]
            if op == "&&":
                return bool(self.eval(e[2], env)) and bool(self.eval(e[3], env))
            if op == "||":
                return bool(self.eval(e[2], env)) or bool(self.eval(e[3], env))
            return binop(op, self.eval(e[2], env), self.eval(e[3], env))
        if t == "unary":
            return unop(e[1], self.eval(e[2], env))
        if t == "member":
            base = self.eval(e[1], env)
            name = e[2]
            if isinstance(base, Struct):
                return base.f[name]
            if isinstance(base, Vec):
                if len(name) == 1:
                    return base.e[SWZ[name]]
                return Vec([base.e[SWZ[c]] for c in name])
            raise TypeError(f"member {name} of {base!r}")
        if t == "index":
            base = self.eval(e[1], env)
            idx = int(self.eval(e[2], env))
            if isinstance(base, (Vec, Arr)):
                if not 0 <= idx < len(base.e):
                    idx = min(max(idx, 0), len(base.e) - 1)      # WGSL: out-of-bounds access is clamped
                return base.e[idx]
            if isinstance(base, Mat):
                return base.cols[idx]
            raise TypeError("indexing a non-array")
        if t == "addr":
            inner = e[1]
            if inner[0] == "ident" and inner[1] in self.res:
                return self.res[inner[1]]                         # &storageBuffer (arrayLength)
            return self.lvalue(inner, env)
        if t == "deref":
            return self.eval(e[1], env).get()
        if t == "call":
            return self.call(e[1], [self.eval(a, env) for a in e[2]], e[3])
        raise TypeError(f"cannot evaluate {e}")

    # ---- calls
    def call(self, name, args, template=None):
        if name in self.funcs:
            return self.invoke(name, args)
        if name in self.structs:
            fields = self.structs[name]
            return Struct(name, {f: copyval(self.coerce(a, ty)) for (f, ty), a in zip(fields, args)})
        if name in VEC_NAMES or name in ("vec2", "vec3", "vec4"):
            n, k = VEC_NAMES[name] if name in VEC_NAMES else (template[1], template[2])
            flat = []
            for a in args:
                flat.extend(a.e if isinstance(a, Vec) else [a])
            if len(flat) == 1:
                flat = flat * n
            if len(flat) != n:
                raise TypeError(f"{name} from {len(flat)} components")
            return Vec([convert(x, k) for x in flat])
        if name in SCALAR_TYPES:
            return convert(args[0], SCALAR_TYPES[name])
        if name.startswith("mat3x3"):
            return Mat([Vec([convert(x, "f") for x in c.e]) for c in args])
        if name == "array":
            return Arr([copyval(a) for a in args])
        fn = getattr(self, "b_" + name, None)
        if fn is None:
            raise NameError(f"unknown function {name}")
        return fn(*args)

    def invoke(self, name, args):
        params, ret, body, _ = self.funcs[name]
        self.calls += 1
        scope = {}
        for (pname, pty), a in zip(params, args):
            scope[pname] = a if isinstance(a, (Ref, Texture)) or pty[0] == "resource" else copyval(self.coerce(a, pty))
        try:
            self.exec(body, [self.globals, scope])
        except _Return as r:
            return copyval(r.v)
        return None

    # ---- statements
    def exec(self, s, env):
        t = s[0]
        if t == "block":
            inner = env + [{}]
            for st in s[1]:
                self.exec(st, inner)
        elif t in ("let", "var"):
            if s[3] is None:
                v = self.zero(s[2])
            else:
                v = copyval(self.eval(s[3], env))
                v = self.coerce(v, s[2]) if s[2] is not None else self.concretise(v)
            env[-1][s[1]] = v
        elif t == "assign":
            ref = self.lvalue(s[1], env)
            v = self.eval(s[3], env)
            if s[2] != "=":
                v = binop(s[2][0], ref.get(), v)
            else:
                old = ref.get()
                if not isinstance(old, (Vec, Struct, Arr, Mat, Ref)) and kind(v) in "IF":
                    v = convert(v, kind(old))
                elif isinstance(old, Vec) and isinstance(v, Vec):
                    v = Vec([convert(x, kind(o)) if kind(x) in "IF" else x for x, o in zip(v.e, old.e)])
            ref.set(copyval(v))
        elif t == "expr":
            self.eval(s[1], env)
        elif t == "if":
            if bool(self.eval(s[1], env)):
                self.exec(s[2], env)
            elif s[3] is not None:
                self.exec(s[3], env)
        elif t == "for":
            inner = env + [{}]
            if s[1] is not None:
                self.exec(s[1], inner)
            while s[2] is None or bool(self.eval(s[2], inner)):
                try:
                    self.exec(s[4], inner)
                except _Break:
                    break
                except _Continue:
                    pass
                if s[3] is not None:
                    self.exec(s[3], inner)
        elif t == "while":
            while bool(self.eval(s[1], env)):
                try:
                    self.exec(s[2], env)
                except _Break:
                    break
                except _Continue:
                    pass
        elif t == "return":
            raise _Return(None if s[1] is None else self.eval(s[1], env))
        elif t == "break":
            raise _Break()
        elif t == "continue":
            raise _Continue()
        else:
            raise TypeError(f"statement {t}")

    # ---- builtins (pinned arithmetic)
    @staticmethod
    def _f(x):
        return convert(x, "f") if kind(x) in "IF" else x

    def _map(self, fn, *args):
        n = max((len(a) for a in args if isinstance(a, Vec)), default=0)
        if n == 0:
            return fn(*args)
        cols = [a.e if isinstance(a, Vec) else [a] * n for a in args]
        return Vec([fn(*xs) for xs in zip(*cols)])

    def b_dot(self, a, b):
        acc = None
        for x, y in zip(a.e, b.e):
            term = _scalar_bin("*", x, y)
            acc = term if acc is None else _scalar_bin("+", acc, term)
        return acc

    def b_cross(self, a, b):
        ax, ay, az = a.e
        bx, by, bz = b.e
        m, s = (lambda x, y: _scalar_bin("*", x, y)), (lambda x, y: _scalar_bin("-", x, y))
        return Vec([s(m(ay, bz), m(az, by)), s(m(az, bx), m(ax, bz)), s(m(ax, by), m(ay, bx))])

    def b_sqrt(self, x):
        def one(v):
            with np.errstate(all="ignore"):
                return F32(np.sqrt(self._f(v)))
        return self._map(one, x)

    def b_length(self, v):
        return self.b_sqrt(self.b_dot(v, v))

    def b_normalize(self, v):
        return binop("/", v, self.b_length(v))

    def b_abs(self, x):
        return self._map(lambda v: F32(abs(v)) if kind(v) == "f" else (abs(v) if kind(v) in "IF" else _i32(abs(int(v)))), x)

    def _minmax(self, a, b, pick_min):
        def one(x, y):
            x, y, k = unify(x, y)
            if k in "fF":
                fx, fy = float(x), float(y)
                if fx != fx:
                    return y
                if fy != fy:
                    return x
                return (x if fx < fy else y) if pick_min else (x if fx > fy else y)      # ties: second operand
            return (x if x < y else y) if pick_min else (x if x > y else y)
        return self._map(one, a, b)

    def b_min(self, a, b):
        return self._minmax(a, b, True)

    def b_max(self, a, b):
        return self._minmax(a, b, False)

    def b_clamp(self, x, lo, hi):
        return self.b_min(self.b_max(x, lo), hi)

    def b_mix(self, a, b, t):
        one = lambda x, y, w: _scalar_bin("+", _scalar_bin("*", x, _scalar_bin("-", F32(1.0), self._f(w))), _scalar_bin("*", y, self._f(w)))
        return self._map(one, a, b, t)

    def b_reflect(self, i, n):
        k = _scalar_bin("*", F32(2.0), self.b_dot(n, i))
        return binop("-", i, binop("*", n, k))

    def b_floor(self, x):
        return self._map(lambda v: F32(np.floor(self._f(v))), x)

    def b_round(self, x):
        return self._map(lambda v: F32(np.rint(self._f(v))), x)          # roundEven

    def b_sign(self, x):
        return self._map(lambda v: F32(np.sign(self._f(v))), x)

    def b_select(self, f, t, cond):
        if isinstance(cond, Vec):
            return Vec([tt if c else ff for ff, tt, c in zip(f.e, t.e, cond.e)])
        return t if cond else f

    def _math1(self, name, x):
        return self._map(lambda v: F32(self.math[name](self._f(v))), x)

    def b_sin(self, x):
        return self._math1("sin", x)

    def b_cos(self, x):
        return self._math1("cos", x)

    def b_tan(self, x):
        return self._math1("tan", x)

    def b_log(self, x):
        return self._math1("log", x)

    def b_exp(self, x):
        return self._math1("exp", x)

    def b_asin(self, x):
        return self._math1("asin", x)

    def b_atan2(self, y, x):
        return self._map(lambda a, b: F32(self.math["atan2"](self._f(a), self._f(b))), y, x)

    def b_pow(self, x, y):
        return self._map(lambda a, b: F32(self.math["pow"](self._f(a), self._f(b))), x, y)

    def b_arrayLength(self, buf):
        return U32(len(buf.e))

    # ---- textures
    def _texel(self, tex, x, y):
        return [F32(v) for v in tex.texels[y, x]]

    def _axis(self, coord, n, address):
        x = _scalar_bin("-", _scalar_bin("*", coord, F32(n)), F32(0.5))
        x0f = F32(np.floor(x))
        f = _scalar_bin("-", x, x0f)
        x0 = int(x0f) if math.isfinite(float(x0f)) and abs(float(x0f)) < 1e9 else 0
        if address == "repeat":
            return x0 % n, (x0 + 1) % n, f
        return min(max(x0, 0), n - 1), min(max(x0 + 1, 0), n - 1), f

    def _sample(self, tex, uv):
        h, w = tex.texels.shape[:2]
        u, v = self._f(uv.e[0]), self._f(uv.e[1])
        if tex.filter == "nearest":
            x = int(np.floor(_scalar_bin("*", u, F32(w))))
            y = int(np.floor(_scalar_bin("*", v, F32(h))))
            if tex.address == "repeat":
                x, y = x % w, y % h
            return Vec(self._texel(tex, min(max(x, 0), w - 1), min(max(y, 0), h - 1)))
        xa, xb, fx = self._axis(u, w, tex.address)
        ya, yb, fy = self._axis(v, h, tex.address)
        p00, p10 = self._texel(tex, xa, ya), self._texel(tex, xb, ya)
        p01, p11 = self._texel(tex, xa, yb), self._texel(tex, xb, yb)
        wx0, wy0 = _scalar_bin("-", F32(1.0), fx), _scalar_bin("-", F32(1.0), fy)
        out = []
        M, A = (lambda a, b: _scalar_bin("*", a, b)), (lambda a, b: _scalar_bin("+", a, b))
        for c in range(4):
            top = A(M(p00[c], wx0), M(p10[c], fx))
            bot = A(M(p01[c], wx0), M(p11[c], fx))
            out.append(A(M(top, wy0), M(bot, fy)))
        return Vec(out)

    def b_textureSampleLevel(self, tex, sampler, uv, level):
        return self._sample(tex, uv)

    def b_textureSample(self, tex, sampler, uv):
        return self._sample(tex, uv)

    def b_textureLoad(self, tex, coords, level):
        return Vec(self._texel(tex, int(coords.e[0]), int(coords.e[1])))

    def b_textureStore(self, tex, coords, value):
        tex.stores[(int(coords.e[0]), int(coords.e[1]))] = [float(self._f(x)) for x in value.e]


def struct_from_record(interp, name, rec):
    """numpy structured record (mi3pt_host.layout dtypes) -> Struct, by field name."""
    fields = {}
    for fname, fty in interp.structs[name]:
        v = rec[fname] if fname in rec.dtype.names else None
        if fty[0] == "vec":
            fields[fname] = Vec([convert(float(x) if fty[2] == "f" else int(x), fty[2]) for x in np.asarray(v).reshape(-1)[: fty[1]]])
        elif fty[0] == "scalar":
            fields[fname] = convert(float(v) if fty[1] == "f" else int(v), fty[1])
        elif fty[0] == "struct":
            sub = {k[len(fname) + 1:]: k for k in rec.dtype.names if k.startswith(fname + ".")}
            inner = {}
            for sname, sty in interp.structs[fty[1]]:
                raw = rec[sub[sname]]
                inner[sname] = (Vec([F32(x) for x in np.asarray(raw).reshape(-1)[: sty[1]]]) if sty[0] == "vec"
                                else convert(float(raw) if sty[1] == "f" else int(raw), sty[1]))
            fields[fname] = Struct(fty[1], inner)
        else:
            raise TypeError(fty)
    return Struct(name, fields)
