"""codegen: user-defined dynamics -> straight-line tape -> scalar-generic C++ functor -> gfx950 library.

The reference turns the CasADi local-system function into C source, compiles it with gcc and dlopens the result when
solver_settings.gen_code / load_lib are set (reference src/OptimalControlProblem.cpp:263-287,602-640).  CasADi is not
available here; the equivalent for this engine is:

1. trace(F, nx, nu): call the user's NumPy discrete map s_next = F(s, u) once on tracer objects; every arithmetic
   operation / NumPy ufunc lands on a tape in SSA form (common subexpressions merged, constants folded);
2. emit_functor(tape): the tape as a `template <class T> F(...)` functor in the form csrc/stage_models.hpp uses, so
   the evaluation kernels (csrc/stage_kernels.hpp) instantiate it with double (merit) and with one-direction dual
   numbers (Jacobian columns) exactly like the built-in zoo;
3. build_device_library(...): hipcc --offload-arch=gfx950 -shared, loaded by mpcqp_stage_create_user;
   build_host_library(...): the same functor compiled by g++ for CPU-side checks (tests, no GPU needed).

Supported inside F: + - * / ** (integer powers), unary -, np.sin cos tan exp log sqrt tanh square negative, indexing
s[..., i] / s[..., a:b], np.stack / np.concatenate along the last axis, Python and NumPy scalars as constants.
Data-dependent branches cannot be traced (the same restriction CasADi SX has)."""
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")

_UNARY = {"neg": "-", "sin": "sm_sin", "cos": "sm_cos", "tan": "sm_tan", "exp": "sm_exp", "log": "sm_log", "sqrt": "sm_sqrt", "tanh": "sm_tanh"}
_BINARY = {"add": "+", "sub": "-", "mul": "*", "div": "/"}
_NP_UNARY = {"neg": np.negative, "sin": np.sin, "cos": np.cos, "tan": np.tan, "exp": np.exp, "log": np.log, "sqrt": np.sqrt, "tanh": np.tanh}


class Tape:
    """SSA tape: nodes[i] = ("in", k) | ("const", value) | (unary, a) | (binary, a, b); outputs = node ids."""

    def __init__(self, n_in):
        self.n_in = n_in
        self.nodes = [("in", k) for k in range(n_in)]
        self._memo = {}
        self.outputs = []

    def const(self, v):
        return self._add(("const", float(v)))

    def _add(self, node):
        key = node if node[0] != "const" else ("const", np.float64(node[1]).tobytes())
        if key in self._memo:
            return self._memo[key]
        self.nodes.append(node)
        self._memo[key] = len(self.nodes) - 1
        return len(self.nodes) - 1

    def is_const(self, i):
        return self.nodes[i][0] == "const"

    def unary(self, op, a):
        if self.is_const(a):
            return self.const(_NP_UNARY[op](self.nodes[a][1]))
        return self._add((op, a))

    def binary(self, op, a, b):
        if self.is_const(a) and self.is_const(b):
            x, y = self.nodes[a][1], self.nodes[b][1]
            return self.const({"add": x + y, "sub": x - y, "mul": x * y, "div": x / y}[op])
        return self._add((op, a, b))

    # -- reference evaluation (NumPy, any dtype incl. complex): the checker for the generated code
    def evaluate(self, inputs):
        vals = [None] * len(self.nodes)
        for i, nd in enumerate(self.nodes):
            k = nd[0]
            if k == "in":
                vals[i] = inputs[nd[1]]
            elif k == "const":
                vals[i] = nd[1]
            elif k in _UNARY:
                vals[i] = _NP_UNARY[k](vals[nd[1]])
            else:
                a, b = vals[nd[1]], vals[nd[2]]
                vals[i] = a + b if k == "add" else a - b if k == "sub" else a * b if k == "mul" else a / b
        return [vals[o] for o in self.outputs]

    def live_nodes(self):
        """ids reachable from the outputs, in order"""
        need = set(); stack = list(self.outputs)
        while stack:
            i = stack.pop()
            if i in need:
                continue
            need.add(i)
            nd = self.nodes[i]
            if nd[0] in _UNARY:
                stack.append(nd[1])
            elif nd[0] in _BINARY:
                stack += [nd[1], nd[2]]
        return sorted(need)

    # -- simplifying constructors used by the symbolic differentiation (x + 0, x * 1, x * 0 ... fold away, so that a
    #    structurally zero derivative is the constant 0 and the Hessian's sparsity can be read off the tape)
    def _cv(self, i):
        return self.nodes[i][1] if self.is_const(i) else None

    def s_add(self, a, b):
        if self._cv(a) == 0.0: return b
        if self._cv(b) == 0.0: return a
        return self.binary("add", a, b)

    def s_sub(self, a, b):
        if self._cv(b) == 0.0: return a
        if self._cv(a) == 0.0: return self.s_neg(b)
        return self.binary("sub", a, b)

    def s_mul(self, a, b):
        for x, y in ((a, b), (b, a)):
            if self._cv(x) == 0.0: return self.const(0.0)
            if self._cv(x) == 1.0: return y
            if self._cv(x) == -1.0: return self.s_neg(y)
        return self.binary("mul", a, b)

    def s_div(self, a, b):
        if self._cv(a) == 0.0: return self.const(0.0)
        if self._cv(b) == 1.0: return a
        return self.binary("div", a, b)

    def s_neg(self, a):
        if self.nodes[a][0] == "neg": return self.nodes[a][1]
        return self.unary("neg", a)

    def partials(self, i):
        """[(argument, d node_i / d argument)] of node i as node ids (new nodes are appended)"""
        nd = self.nodes[i]; k = nd[0]
        one = self.const(1.0)
        if k in ("in", "const"): return []
        if k == "add": return [(nd[1], one), (nd[2], one)]
        if k == "sub": return [(nd[1], one), (nd[2], self.const(-1.0))]
        if k == "mul": return [(nd[1], nd[2]), (nd[2], nd[1])]
        if k == "div": return [(nd[1], self.s_div(one, nd[2])), (nd[2], self.s_neg(self.s_div(i, nd[2])))]
        a = nd[1]
        if k == "neg": return [(a, self.const(-1.0))]
        if k == "sin": return [(a, self.unary("cos", a))]
        if k == "cos": return [(a, self.s_neg(self.unary("sin", a)))]
        if k == "tan": return [(a, self.s_add(one, self.s_mul(i, i)))]
        if k == "exp": return [(a, i)]
        if k == "log": return [(a, self.s_div(one, a))]
        if k == "sqrt": return [(a, self.s_div(self.const(0.5), i))]
        if k == "tanh": return [(a, self.s_sub(one, self.s_mul(i, i)))]
        raise ValueError("no derivative rule for %s" % k)

    def clone(self):
        t = Tape(self.n_in)
        t.nodes = list(self.nodes); t._memo = dict(self._memo); t.outputs = list(self.outputs)
        for a in ("nx", "nu", "nr"):
            if hasattr(self, a): setattr(t, a, getattr(self, a))
        return t


def gradient_tape(tape):
    """reverse-mode sweep over the SSA tape of a scalar function: a new tape whose outputs are d out / d input_k, k < n_in"""
    g = tape.clone()
    (out,) = tape.outputs
    zero = g.const(0.0)
    bar = {out: g.const(1.0)}
    for i in sorted(tape.live_nodes(), reverse=True):
        if i not in bar or g.nodes[i][0] in ("in", "const"):
            continue
        for a, d in g.partials(i):
            bar[a] = g.s_add(bar.get(a, zero), g.s_mul(bar[i], d))
    g.outputs = [bar.get(k, zero) for k in range(tape.n_in)]
    return g


def tangent_outputs(tape, k):
    """forward-mode sweep in direction input k (symbolic): node ids of d outputs / d input_k, appended to a scratch clone"""
    t = tape.clone()
    zero, one = t.const(0.0), t.const(1.0)
    dot = {}
    for i in tape.live_nodes():
        nd = t.nodes[i]
        if nd[0] == "in":
            dot[i] = one if nd[1] == k else zero
        elif nd[0] == "const":
            dot[i] = zero
        else:
            acc = zero
            for a, d in t.partials(i):
                acc = t.s_add(acc, t.s_mul(dot[a], d))
            dot[i] = acc
    return t, [dot[o] for o in tape.outputs]


def hessian_mask(gtape):
    """structural sparsity [n_in, n_in] of the Hessian whose gradient tape is gtape (entries that are not identically zero)"""
    n = gtape.n_in
    mask = np.zeros((n, n), bool)
    for k in range(n):
        t, d = tangent_outputs(gtape, k)
        for r in range(n):
            mask[r, k] = not (t.is_const(d[r]) and t.nodes[d[r]][1] == 0.0)
    return mask | mask.T


class TS:
    """scalar tracer"""
    __array_priority__ = 1000

    def __init__(self, tape, idx):
        self.tape, self.idx = tape, idx

    @staticmethod
    def _lift(tape, v):
        if isinstance(v, TS):
            return v
        if isinstance(v, (int, float, np.integer, np.floating)):
            return TS(tape, tape.const(v))
        if isinstance(v, np.ndarray) and v.ndim == 0:
            return TS(tape, tape.const(float(v)))
        raise TypeError("cannot trace a value of type %s" % type(v).__name__)

    def _bin(self, op, other, swap=False):
        if isinstance(other, TV):
            return other._bin(op, self, swap=not swap)
        o = TS._lift(self.tape, other)
        a, b = (o, self) if swap else (self, o)
        return TS(self.tape, self.tape.binary(op, a.idx, b.idx))

    def __add__(self, o): return self._bin("add", o)
    def __radd__(self, o): return self._bin("add", o, True)
    def __sub__(self, o): return self._bin("sub", o)
    def __rsub__(self, o): return self._bin("sub", o, True)
    def __mul__(self, o): return self._bin("mul", o)
    def __rmul__(self, o): return self._bin("mul", o, True)
    def __truediv__(self, o): return self._bin("div", o)
    def __rtruediv__(self, o): return self._bin("div", o, True)
    def __neg__(self): return TS(self.tape, self.tape.unary("neg", self.idx))
    def __pos__(self): return self

    def __pow__(self, e):
        if isinstance(e, (int, np.integer)) or (isinstance(e, float) and e == int(e)):
            e = int(e)
            if e == 0:
                return TS(self.tape, self.tape.const(1.0))
            r = None; base = self; k = abs(e)
            while k:
                if k & 1:
                    r = base if r is None else r * base
                k >>= 1
                if k:
                    base = base * base
            return r if e > 0 else 1.0 / r
        if e == 0.5:
            return TS(self.tape, self.tape.unary("sqrt", self.idx))
        raise TypeError("only integer powers and ** 0.5 can be traced")

    def __bool__(self):
        raise TypeError("data-dependent branches cannot be traced (the dynamics must be straight-line code)")

    def _cmp(self, o):
        return self.__bool__()

    __lt__ = __le__ = __gt__ = __ge__ = _cmp

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        return _ufunc(self.tape, ufunc, method, inputs, kw)

    def __array_function__(self, func, types, args, kwargs):
        return _array_function(self.tape, func, args, kwargs)


class TV:
    """vector tracer standing for an array [..., k] (leading axes are the batch the device kernel supplies)"""
    __array_priority__ = 1000

    def __init__(self, tape, items):
        self.tape, self.items = tape, list(items)

    @property
    def shape(self):
        return (len(self.items),)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, key):
        if isinstance(key, tuple):
            key = [k for k in key if k is not Ellipsis]
            if len(key) != 1:
                raise IndexError("index the last axis only: s[..., i] or s[..., a:b]")
            key = key[0]
        if isinstance(key, slice):
            return TV(self.tape, self.items[key])
        if isinstance(key, (int, np.integer)):
            return self.items[int(key)]
        if isinstance(key, (list, np.ndarray)):
            return TV(self.tape, [self.items[int(i)] for i in key])
        raise IndexError("unsupported index %r" % (key,))

    def __iter__(self):
        return iter(self.items)

    def _bin(self, op, other, swap=False):
        if isinstance(other, TV):
            if len(other) != len(self):
                raise ValueError("shape mismatch %d vs %d" % (len(self), len(other)))
            others = other.items
        elif isinstance(other, np.ndarray) and other.ndim >= 1:
            if other.shape[-1] != len(self) or other.size != len(self):
                raise ValueError("constant array must have the vector's length")
            others = [float(v) for v in other.ravel()]
        else:
            others = [other] * len(self)
        out = []
        for a, b in zip(self.items, others):
            out.append(a._bin(op, b, swap))
        return TV(self.tape, out)

    def __add__(self, o): return self._bin("add", o)
    def __radd__(self, o): return self._bin("add", o, True)
    def __sub__(self, o): return self._bin("sub", o)
    def __rsub__(self, o): return self._bin("sub", o, True)
    def __mul__(self, o): return self._bin("mul", o)
    def __rmul__(self, o): return self._bin("mul", o, True)
    def __truediv__(self, o): return self._bin("div", o)
    def __rtruediv__(self, o): return self._bin("div", o, True)
    def __neg__(self): return TV(self.tape, [-a for a in self.items])
    def __pow__(self, e): return TV(self.tape, [a ** e for a in self.items])

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        return _ufunc(self.tape, ufunc, method, inputs, kw)

    def __array_function__(self, func, types, args, kwargs):
        return _array_function(self.tape, func, args, kwargs)


_UFUNC_BIN = {np.add: "add", np.subtract: "sub", np.multiply: "mul", np.true_divide: "div"}
_UFUNC_UN = {np.negative: "neg", np.sin: "sin", np.cos: "cos", np.tan: "tan", np.exp: "exp", np.log: "log", np.sqrt: "sqrt", np.tanh: "tanh"}


def _ufunc(tape, ufunc, method, inputs, kw):
    if method != "__call__" or kw.get("out") is not None:
        return NotImplemented
    if ufunc in _UFUNC_BIN:
        a, b = inputs
        op = _UFUNC_BIN[ufunc]
        if isinstance(a, (TS, TV)):
            return a._bin(op, b)
        return b._bin(op, a, True)
    if ufunc in _UFUNC_UN:
        (a,) = inputs
        op = _UFUNC_UN[ufunc]
        if isinstance(a, TV):
            return TV(tape, [TS(tape, tape.unary(op, x.idx)) for x in a.items])
        return TS(tape, tape.unary(op, a.idx))
    if ufunc is np.square:
        (a,) = inputs
        return a * a
    if ufunc is np.power:
        a, e = inputs
        return a ** e
    if ufunc is np.positive:
        return inputs[0]
    raise TypeError("NumPy function %s cannot be traced" % ufunc.__name__)


def _array_function(tape, func, args, kwargs):
    if func in (np.stack, np.concatenate):
        seq = args[0]
        axis = kwargs.get("axis", args[1] if len(args) > 1 else 0)
        if axis not in (-1,) and not (axis == 0 and all(isinstance(v, (TS, TV, int, float)) for v in seq)):
            raise TypeError("stack / concatenate along the last axis only")
        items = []
        for v in seq:
            if isinstance(v, TV):
                if func is np.stack:
                    raise TypeError("np.stack of vectors is not traceable; use np.concatenate")
                items += v.items
            else:
                items.append(TS._lift(tape, v))
        return TV(tape, items)
    raise TypeError("NumPy function %s cannot be traced" % func.__name__)


def _trace_fn(fn, nx, nu, n_out, what):
    tape = Tape(nx + nu)
    s = TV(tape, [TS(tape, i) for i in range(nx)])
    u = TV(tape, [TS(tape, nx + i) for i in range(nu)])
    out = fn(s, u)
    if isinstance(out, TS):
        out = TV(tape, [out])
    if isinstance(out, (list, tuple)):
        out = TV(tape, [TS._lift(tape, v) for v in out])
    if not isinstance(out, TV) or len(out) != n_out:
        raise ValueError("%s, %d components" % (what, n_out))
    tape.outputs = [TS._lift(tape, v).idx for v in out.items]
    tape.nx, tape.nu = nx, nu
    return tape


def trace_cost(lfun, nx, nu, nr):
    """stage cost l(s, u, r) -> scalar on tracers; returns (value tape, gradient tape) over inputs [s; u; r]"""
    tape = Tape(nx + nu + nr)
    mk = lambda a, b: TV(tape, [TS(tape, i) for i in range(a, b)])
    out = lfun(mk(0, nx), mk(nx, nx + nu), mk(nx + nu, nx + nu + nr))
    if isinstance(out, TV) and len(out) == 1:
        out = out.items[0]
    if isinstance(out, (int, float, np.integer, np.floating)):
        out = TS._lift(tape, out)
    if not isinstance(out, TS):
        raise ValueError("the stage cost must return one scalar")
    tape.outputs = [out.idx]
    tape.nx, tape.nu, tape.nr = nx, nu, nr
    return tape, gradient_tape(tape)


def _trace_link(kfun, nx, nu, nk):
    """kfun(s, u, s_next, u_next) -> [..., nk] on tracers over the inputs [s; u; s_next; u_next]"""
    tape = Tape(2 * (nx + nu))
    mk = lambda a, b: TV(tape, [TS(tape, i) for i in range(a, b)])
    f = nx + nu
    out = kfun(mk(0, nx), mk(nx, f), mk(f, f + nx), mk(f + nx, 2 * f))
    if isinstance(out, TS):
        out = TV(tape, [out])
    if isinstance(out, (list, tuple)):
        out = TV(tape, [TS._lift(tape, v) for v in out])
    if not isinstance(out, TV) or len(out) != nk:
        raise ValueError("kfun must return the link-constraint values, %d components" % nk)
    tape.outputs = [TS._lift(tape, v).idx for v in out.items]
    tape.nx, tape.nu = nx, nu
    tape.in_names = [("s", nx), ("u", nu), ("sn", nx), ("un", nu)]
    return tape


def trace(F, nx, nu, hfun=None, nh=0, h_lo=None, h_hi=None, lcost=None, lterm=None, kfun=None, nk=0, k_lo=None, k_hi=None):
    """Run F (and the optional per-stage path constraint hfun) once on tracers.  F(s, u) -> s_next with s [..., nx],
    u [..., nu] (the contract of models.StageOCP.F); hfun(s, u) -> [..., nh] with bounds h_lo <= hfun <= h_hi.
    lcost(s, u, r) -> scalar: a general stage cost summed over the frames (r = the reference parameter, size nx), replacing
    the diagonal tracking weights; lterm: the same for the last frame only (terminal cost).  Their gradients are derived
    on the tape (reverse mode); the kernels differentiate those once more with dual numbers for the exact Hessian."""
    tape = _trace_fn(F, nx, nu, nx, "F must return the next state")
    tape.nh = int(nh) if hfun is not None else 0
    tape.path = None
    if tape.nh:
        tape.path = _trace_fn(hfun, nx, nu, tape.nh, "hfun must return the path-constraint values")
        tape.h_lo = [float(v) for v in np.broadcast_to(np.asarray(h_lo, float), (tape.nh,))]
        tape.h_hi = [float(v) for v in np.broadcast_to(np.asarray(h_hi, float), (tape.nh,))]
    # link constraint k_lo <= kfun(s_k, u_k, s_{k+1}, u_{k+1}) <= k_hi between consecutive frames (rate limits and the like)
    tape.nk = int(nk) if kfun is not None else 0
    tape.link = None
    if tape.nk:
        tape.link = _trace_link(kfun, nx, nu, tape.nk)
        tape.k_lo = [float(v) for v in np.broadcast_to(np.asarray(k_lo, float), (tape.nk,))]
        tape.k_hi = [float(v) for v in np.broadcast_to(np.asarray(k_hi, float), (tape.nk,))]
    tape.cost = None
    if lcost is not None:
        L, G = trace_cost(lcost, nx, nu, nx)
        mask = hessian_mask(G)
        LT = GT = None
        if lterm is not None:
            LT, GT = trace_cost(lterm, nx, nu, nx)
            mask = mask | hessian_mask(GT)
        mask = mask | np.eye(mask.shape[0], dtype=bool)      # keep the diagonal in the pattern
        tape.cost = dict(L=L, G=G, LT=LT, GT=GT, mask=mask)
    elif lterm is not None:
        raise ValueError("a terminal cost needs a stage cost")
    return tape


# ------------------------------------------------------------------------------------------------- emission
def _lit(v):
    if v != v or v in (float("inf"), float("-inf")):
        raise ValueError("non-finite constant in the traced dynamics")
    return repr(float(v)) if "e" in repr(float(v)) or "." in repr(float(v)) else repr(float(v)) + ".0"


def _emit_body(tape):
    nx = tape.nx
    live = tape.live_nodes()
    ref = {}
    lines = []
    for i in live:
        nd = tape.nodes[i]
        if nd[0] == "in":
            nu = getattr(tape, "nu", tape.n_in - nx)
            if getattr(tape, "in_names", None):
                k = nd[1]
                for nm, cnt in tape.in_names:
                    if k < cnt:
                        ref[i] = "%s[%d]" % (nm, k); break
                    k -= cnt
            else:
                ref[i] = "s[%d]" % nd[1] if nd[1] < nx else "u[%d]" % (nd[1] - nx) if nd[1] < nx + nu else "r[%d]" % (nd[1] - nx - nu)
        elif nd[0] == "const":
            ref[i] = _lit(nd[1])
        elif nd[0] in _UNARY:
            f = _UNARY[nd[0]]
            expr = "-%s" % ref[nd[1]] if nd[0] == "neg" else "%s(%s)" % (f, ref[nd[1]])
            lines.append("    const T w%d = %s;" % (i, expr)); ref[i] = "w%d" % i
        else:
            lines.append("    const T w%d = %s %s %s;" % (i, ref[nd[1]], _BINARY[nd[0]], ref[nd[2]])); ref[i] = "w%d" % i
    for r, o in enumerate(tape.outputs):
        nd = tape.nodes[o]
        lines.append("    out[%d] = %s;" % (r, "T{} + %s" % ref[o] if nd[0] == "const" else ref[o]))
    return "\n".join(lines)


def _blit(v):
    return "INFINITY" if v == float("inf") else "-INFINITY" if v == float("-inf") else _lit(v)


def emit_functor(tape, name="SmUser"):
    """C++ source of the functor (same shape as the zoo's functors in csrc/stage_models.hpp)"""
    nh = getattr(tape, "nh", 0)
    nk = getattr(tape, "nk", 0)
    cost = getattr(tape, "cost", None)
    hbody = _emit_body(tape.path) if nh else ""
    kbody = _emit_body(tape.link) if nk else ""
    src = ("struct %s {\n  static constexpr int nx = %d, nu = %d, nh = %d, nk = %d, has_cost = %d, has_term = %d;\n"
           "  template <class T> SM_HD static void F(const double *, double, const T *s, const T *u, T *out) {\n%s\n  }\n"
           "  template <class T> SM_HD static void H(const T *s, const T *u, T *out) {\n%s\n  }\n"
           "  template <class T> SM_HD static void K(const T *s, const T *u, const T *sn, const T *un, T *out) {\n%s\n  }\n"
           % (name, tape.nx, tape.nu, nh, nk, 1 if cost else 0, 1 if cost and cost["LT"] is not None else 0, _emit_body(tape), hbody, kbody))
    if cost:
        # L: out[0] = l(s, u, r); LG: out[nx + nu + nx] = dl / d[s; u; r]; LT, LTG: the terminal frame's
        for fn, tp in (("L", cost["L"]), ("LG", cost["G"]), ("LT", cost["LT"] or cost["L"]), ("LTG", cost["GT"] or cost["G"])):
            src += "  template <class T> SM_HD static void %s(const T *s, const T *u, const T *r, T *out) {\n%s\n  }\n" % (fn, _emit_body(tp))
    src += "};\n"
    lo = ", ".join(_blit(v) for v in tape.h_lo) if nh else "0.0"
    hi = ", ".join(_blit(v) for v in tape.h_hi) if nh else "0.0"
    src += "static const double %s_h_lo[] = {%s};\nstatic const double %s_h_hi[] = {%s};\n" % (name, lo, name, hi)
    klo = ", ".join(_blit(v) for v in tape.k_lo) if nk else "0.0"
    khi = ", ".join(_blit(v) for v in tape.k_hi) if nk else "0.0"
    src += "static const double %s_k_lo[] = {%s};\nstatic const double %s_k_hi[] = {%s};\n" % (name, klo, name, khi)
    mk = ", ".join(str(int(v)) for v in cost["mask"].ravel()) if cost else "0"
    src += "static const unsigned char %s_cost_mask[] = {%s};\n" % (name, mk)
    return src


_DEVICE_TMPL = '''// generated by optimal_control_problem_amd/codegen.py -- do not edit
#include "stage_kernels.hpp"

%(functor)s
extern "C" {
int mpcqp_user_abi() { return STAGE_ABI_VERSION; }
void mpcqp_user_dims(int *nx, int *nu) { *nx = SmUser::nx; *nu = SmUser::nu; }
int mpcqp_user_nh() { return SmUser::nh; }
void mpcqp_user_path_bounds(double *lo, double *hi) { for (int i = 0; i < SmUser::nh; i++) { lo[i] = SmUser_h_lo[i]; hi[i] = SmUser_h_hi[i]; } }
int mpcqp_user_nk() { return SmUser::nk; }
void mpcqp_user_link_bounds(double *lo, double *hi) { for (int i = 0; i < SmUser::nk; i++) { lo[i] = SmUser_k_lo[i]; hi[i] = SmUser_k_hi[i]; } }
// general stage cost: 1 and the Hessian's structure over [s; u; r] (row-major (f + nx)^2 bytes), or 0 for diagonal tracking weights
int mpcqp_user_cost(unsigned char *mask) {
  if (!SmUser::has_cost) return 0;
  const int nl = 2 * SmUser::nx + SmUser::nu;
  for (int i = 0; i < nl * nl; i++) mask[i] = SmUser_cost_mask[i];
  return 1;
}
int mpcqp_user_eval(const StageDev *sd, int batch, const double *p, const double *x, const double *lbx, const double *ubx,
                    const double *lbg, const double *ubg, double *P, double *q, double *A, double *l, double *u, void *stream) {
  return (int)stage_launch_eval<SmUser>(*sd, batch, p, x, lbx, ubx, lbg, ubg, P, q, A, l, u, (hipStream_t)stream);
}
int mpcqp_user_merit(const StageDev *sd, int batch, const double *p, const double *x, double *f, double *gmax, void *stream) {
  return (int)stage_launch_merit<SmUser>(*sd, batch, p, x, f, gmax, (hipStream_t)stream);
}
}
'''

_HOST_TMPL = '''// generated by optimal_control_problem_amd/codegen.py -- host build of the same functor, for checks without a GPU
#include <cmath>
#include "stage_models.hpp"

%(functor)s
// val [1], grad [nl], hess [nl * nl] row-major over [s; u; r], nl = 2 nx + nu; term != 0: the terminal frame's cost
template <class M> static void host_cost(const double *s, const double *u, const double *r, int term, double *val, double *grad, double *hess) {
  if constexpr (M::has_cost != 0) {
    constexpr int nx = M::nx, nu = M::nu, nl = 2 * nx + nu;
    double v[1];
    if (term) M::template LT<double>(s, u, r, v); else M::template L<double>(s, u, r, v);
    val[0] = v[0];
    for (int c = 0; c < nl; c++) {
      Dual sd[nx], ud[nu], rd[nx], gd[nl];
      for (int i = 0; i < nx; i++) sd[i] = {s[i], i == c ? 1.0 : 0.0};
      for (int i = 0; i < nu; i++) ud[i] = {u[i], nx + i == c ? 1.0 : 0.0};
      for (int i = 0; i < nx; i++) rd[i] = {r[i], nx + nu + i == c ? 1.0 : 0.0};
      if (term) M::template LTG<Dual>(sd, ud, rd, gd); else M::template LG<Dual>(sd, ud, rd, gd);
      for (int i = 0; i < nl; i++) { hess[i * nl + c] = gd[i].d; grad[i] = gd[i].v; }
    }
  }
}
extern "C" {
void user_host_dims(int *nx, int *nu) { *nx = SmUser::nx; *nu = SmUser::nu; }
// out [nx], jac [nx * (nx + nu)] row-major: forward-mode duals, one direction per pass (what a device thread does)
void user_host_eval(const double *s, const double *u, double *out, double *jac) {
  constexpr int nx = SmUser::nx, nu = SmUser::nu, f = nx + nu;
  for (int c = 0; c < f; c++) {
    Dual sd[nx], ud[nu], od[nx];
    for (int i = 0; i < nx; i++) sd[i] = {s[i], i == c ? 1.0 : 0.0};
    for (int i = 0; i < nu; i++) ud[i] = {u[i], nx + i == c ? 1.0 : 0.0};
    SmUser::F<Dual>(nullptr, 0.0, sd, ud, od);
    for (int r = 0; r < nx; r++) { jac[r * f + c] = od[r].d; out[r] = od[r].v; }
  }
}
int user_host_nh() { return SmUser::nh; }
int user_host_nk() { return SmUser::nk; }
// out [nk], jac [nk * 2 (nx + nu)] row-major over [s; u; s_next; u_next]
void user_host_link(const double *s, const double *u, const double *sn, const double *un, double *out, double *jac) {
  constexpr int nx = SmUser::nx, nu = SmUser::nu, f = nx + nu, nk = SmUser::nk;
  if constexpr (nk > 0) {
    for (int c = 0; c < 2 * f; c++) {
      Dual sd[nx], ud[nu], snd[nx], und[nu], od[nk];
      for (int i = 0; i < nx; i++) { sd[i] = {s[i], i == c ? 1.0 : 0.0}; snd[i] = {sn[i], f + i == c ? 1.0 : 0.0}; }
      for (int i = 0; i < nu; i++) { ud[i] = {u[i], nx + i == c ? 1.0 : 0.0}; und[i] = {un[i], f + nx + i == c ? 1.0 : 0.0}; }
      SmUser::K<Dual>(sd, ud, snd, und, od);
      for (int r = 0; r < nk; r++) { jac[r * 2 * f + c] = od[r].d; out[r] = od[r].v; }
    }
  }
}
int user_host_has_cost() { return SmUser::has_cost; }
void user_host_cost(const double *s, const double *u, const double *r, int term, double *val, double *grad, double *hess) {
  host_cost<SmUser>(s, u, r, term, val, grad, hess);
}
// out [nh], jac [nh * (nx + nu)] row-major
void user_host_path(const double *s, const double *u, double *out, double *jac) {
  constexpr int nx = SmUser::nx, nu = SmUser::nu, f = nx + nu, nh = SmUser::nh;
  for (int c = 0; c < f; c++) {
    Dual sd[nx], ud[nu], od[nh > 0 ? nh : 1];
    for (int i = 0; i < nx; i++) sd[i] = {s[i], i == c ? 1.0 : 0.0};
    for (int i = 0; i < nu; i++) ud[i] = {u[i], nx + i == c ? 1.0 : 0.0};
    SmUser::H<Dual>(sd, ud, od);
    for (int r = 0; r < nh; r++) { jac[r * f + c] = od[r].d; out[r] = od[r].v; }
  }
}
}
'''


def _private_dir(d):
    """d exists, belongs to this user and nobody else may write to it (a library found there is dlopen'ed)"""
    st = os.stat(d)
    return st.st_uid == os.getuid() and not (st.st_mode & 0o022)


def cache_dir():
    """MPCQP_CACHE_DIR, else <package>/_gen, else a per-user 0700 directory under the system temp dir (read-only installs).
    Libraries in it are loaded without a rebuild, so a directory another user could have prepared is never used."""
    import tempfile
    for d in (os.environ.get("MPCQP_CACHE_DIR"), os.path.join(_HERE, "_gen"),
              os.path.join(tempfile.gettempdir(), "mpcqp_gen_%d" % os.getuid())):
        if not d:
            continue
        try:
            os.makedirs(d, mode=0o700, exist_ok=True)
            if os.access(d, os.W_OK) and _private_dir(d):
                return d
        except OSError:
            pass
    raise RuntimeError("no private writable directory for generated dynamics libraries (set MPCQP_CACHE_DIR)")


def _build(src_text, suffix, cmd_prefix):
    # the key covers everything the library is compiled from: the generated source, every header it includes and the command
    deps = "".join(open(os.path.join(CSRC, f)).read() for f in ("stage_kernels.hpp", "stage_models.hpp", "common.hpp"))
    deps += open(os.path.join(_HERE, "..", "include", "mpcqp.h")).read()
    key = hashlib.sha256((src_text + deps + " ".join(cmd_prefix)).encode()).hexdigest()[:20]
    base = os.path.join(cache_dir(), "user_%s_%s" % (suffix, key))
    so, src = base + ".so", base + (".hip" if suffix == "dev" else ".cpp")
    if not os.path.exists(so):
        with open(src, "w") as fh:
            fh.write(src_text)
        tmp = so + ".tmp%d" % os.getpid()
        subprocess.check_call(cmd_prefix + ["-I", CSRC, "-shared", "-fPIC", "-o", tmp, src])
        os.replace(tmp, so)
    return so


def build_device_library(tape):
    """gfx950 shared library for mpcqp_stage_create_user (hipcc cross-compiles without a GPU); cached by content"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return _build(_DEVICE_TMPL % {"functor": emit_functor(tape)}, "dev", [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950"])


def build_host_library(tape):
    return _build(_HOST_TMPL % {"functor": emit_functor(tape)}, "host", ["g++", "-O2", "-std=c++17", "-ffp-contract=off"])
