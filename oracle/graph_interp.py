#!/usr/bin/env python3
"""NumPy interpreter of the reference's TF-1.14 MetaGraphDef text proto (TEST INFRASTRUCTURE, build container only).

The reference's arithmetic for the rollout-collect + minibatch-update path lives in the graph file
    resources/ppo_cl/graphs/ppo_cpp_[4_5]_lr_0.0004_cr_0.1610_ent_0.0007.meta.txt   ("G")
that ppo2/session_creator.hpp:40 loads and ppo2/policies.hpp:33-77 / ppo2/ppo2.hpp:430-468 run through
tensorflow::Session::Run.  TensorFlow is not installable here, so this file EXECUTES G ITSELF: every node is
evaluated from its `op`, its `input` wiring (data and ^control edges) and its attrs exactly as the file states them.
Nothing here knows what PPO is -- no loss formula, no backward rule, no clip recipe, no tensor order is written down;
they all come out of G's 928 nodes (which tensors are multiplied, which reduction axes, which Select picks which
branch on a tie, in which order the 13 L2Loss terms are stacked, which ApplyAdam sees which gradient).

What IS restated from TensorFlow 1.14 (the pinned third-party dependency, G:1855-1856) are the semantics of the 55 op
*kernels* G uses, each a few lines in OPS below: elementwise ops, reductions, MatMul, shape algebra, StridedSlice,
DynamicStitch, BroadcastGradientArgs, TanhGrad (dy * (1 - y*y)), L2Loss (sum(x*x)/2), ApplyAdam (training_ops.cc:
alpha = lr*sqrt(1-b2p)/(1-b1p); m += (g-m)*(1-b1); v += (g*g-v)*(1-b2); var -= m*alpha/(sqrt(v)+eps)).
All float arithmetic is done in float32; reductions use NumPy's float32 pairwise sums (TF-CPU's Eigen reduction
order is unspecified), MatMul accumulates in float32 via float64 products rounded once per output
(`matmul_mode="f64round"`) or as a plain float32 k-ordered chain (`"f32chain"`): the two bracket any Eigen blocking.

RandomStandardNormal (G:5894) is the one node that cannot be reproduced (Philox with seed 0 = nondeterministic):
the caller injects the noise tensor, which is also how the product's parity mode feeds it.

Outputs are committed as tests/golden/g45_graph_run.npz by oracle/make_graph_golden.py; the oracle
(oracle/ppo_oracle.c) and the HIP path are both tested against them.  /root/reference is only read here.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from extract_fixtures import every, first, load_graph_nodes, tensor_from_attr, unquote  # noqa: E402

F32 = np.float32
DT = {"DT_FLOAT": np.float32, "DT_INT32": np.int32, "DT_INT64": np.int64, "DT_BOOL": np.bool_, "DT_DOUBLE": np.float64}


def _attrs(node):
    """attr blocks -> {key: python value} for the scalar kinds G uses (i, f, b, type, s, list(i))."""
    out = {}
    for a in every(node, "attr"):
        key = unquote(first(a, "key"))
        val = first(a, "value")
        if not isinstance(val, list):
            continue
        for k, v in val:
            if k == "i":
                out[key] = int(v)
            elif k == "f":
                out[key] = float(v)
            elif k == "b":
                out[key] = v == "true"
            elif k == "type":
                out[key] = v
            elif k == "s":
                out[key] = unquote(v)
            elif k == "list":
                out[key] = [int(x) for kk, x in v if kk == "i"] if isinstance(v, list) else []
    return out


class Ref:
    """Output of a VariableV2 node: a mutable slot (TF ref semantics)."""
    def __init__(self, store, name):
        self.store, self.name = store, name

    def get(self):
        return self.store[self.name]

    def set(self, v):
        self.store[self.name] = v


def val(x):
    return x.get() if isinstance(x, Ref) else x


def _f(x):
    return np.asarray(x, dtype=F32)


class GraphInterp:
    def __init__(self, path, matmul_mode="f64round"):
        self.nodes = load_graph_nodes(path)
        self.attrs = {n: _attrs(b) for n, b in self.nodes.items()}
        self.ops = {n: unquote(first(b, "op")) for n, b in self.nodes.items()}
        self.inputs = {n: [unquote(i) for i in every(b, "input")] for n, b in self.nodes.items()}
        self.vars = {}
        self.matmul_mode = matmul_mode
        self.executed = []           # node names in execution order of the last run (for the op census in the tests)

    # ---- public ---------------------------------------------------------------------------------------------------
    def run(self, fetches=(), feeds=None, targets=(), noise=None):
        """Session::Run(feeds, fetches, targets).  feeds: {"name:0" or "name": array}.  noise: {node name: array} for
        RandomStandardNormal nodes."""
        self._memo = {}
        self._feeds = {k.split(":")[0]: np.asarray(v) for k, v in (feeds or {}).items()}
        self._noise = noise or {}
        self.executed = []
        for t in targets:
            self._eval_node(t)
        return [np.array(val(self._tensor(f))) for f in fetches]

    def init(self):
        """Session::Run({}, {}, {"init"})  (session_creator.hpp:54)."""
        self.run(targets=["init"])

    # ---- evaluation ------------------------------------------------------------------------------------------------
    def _tensor(self, ref):
        name, idx = (ref.split(":") + ["0"])[:2] if ":" in ref else (ref, "0")
        out = self._eval_node(name)
        return out[int(idx)] if isinstance(out, tuple) else out

    def _eval_node(self, name):
        if name in self._memo:
            return self._memo[name]
        # iterative post-order walk (the graph is ~900 nodes deep in places; Python recursion is not needed)
        stack = [(name, False)]
        while stack:
            n, ready = stack.pop()
            if n in self._memo:
                continue
            if n in self._feeds:
                self._memo[n] = self._feeds[n]
                continue
            deps = [i.lstrip("^").split(":")[0] for i in self.inputs[n]]
            if not ready:
                stack.append((n, True))
                for d in reversed(deps):
                    if d not in self._memo:
                        stack.append((d, False))
                continue
            data = [self._lookup(i) for i in self.inputs[n] if not i.startswith("^")]
            op = self.ops[n]
            fn = getattr(self, "op_" + op, None)
            if fn is None:
                raise NotImplementedError("op %s (node %s)" % (op, n))
            self._memo[n] = fn(n, self.attrs[n], *data)
            self.executed.append(n)
        return self._memo[name]

    def _lookup(self, ref):
        name, _, idx = ref.partition(":")
        out = self._memo[name]
        return out[int(idx or 0)] if isinstance(out, tuple) else out

    # ---- op kernels (TF 1.14 semantics) ----------------------------------------------------------------------------
    def op_Const(self, n, a):
        return tensor_from_attr(self.nodes[n])

    def op_Placeholder(self, n, a):
        raise KeyError("placeholder %s was not fed" % n)

    def op_PlaceholderWithDefault(self, n, a, d):
        return val(d)

    def op_VariableV2(self, n, a):
        return Ref(self.vars, n)

    def op_Identity(self, n, a, x):
        return np.array(val(x))                      # a read: snapshot of the variable at execution time

    def op_Assign(self, n, a, ref, v):
        ref.set(np.array(val(v)))
        return ref

    def op_NoOp(self, n, a):
        return None

    def op_RandomStandardNormal(self, n, a, shape):
        if n not in self._noise:
            raise KeyError("RandomStandardNormal %s needs injected noise" % n)
        z = _f(self._noise[n])
        assert tuple(z.shape) == tuple(int(s) for s in shape), (z.shape, shape)
        return z

    # elementwise
    def op_Add(self, n, a, x, y): return val(x) + val(y)
    def op_Sub(self, n, a, x, y): return val(x) - val(y)
    def op_Mul(self, n, a, x, y): return val(x) * val(y)
    def op_RealDiv(self, n, a, x, y): return val(x) / val(y)
    def op_Neg(self, n, a, x): return -val(x)
    def op_Abs(self, n, a, x): return np.abs(val(x))
    def op_Square(self, n, a, x): return val(x) * val(x)
    def op_Sqrt(self, n, a, x): return np.sqrt(val(x))
    def op_Exp(self, n, a, x): return np.exp(val(x))
    def op_Tanh(self, n, a, x): return np.tanh(val(x))
    def op_TanhGrad(self, n, a, y, dy): return val(dy) * (F32(1) - val(y) * val(y))
    def op_Maximum(self, n, a, x, y): return np.maximum(val(x), val(y))
    def op_Minimum(self, n, a, x, y): return np.minimum(val(x), val(y))
    def op_GreaterEqual(self, n, a, x, y): return val(x) >= val(y)
    def op_LessEqual(self, n, a, x, y): return val(x) <= val(y)
    def op_Greater(self, n, a, x, y): return val(x) > val(y)
    def op_IsFinite(self, n, a, x): return np.isfinite(val(x))
    def op_Select(self, n, a, c, x, y): return np.where(val(c), val(x), val(y))
    def op_FloorDiv(self, n, a, x, y): return np.floor_divide(val(x), val(y))
    def op_FloorMod(self, n, a, x, y): return np.mod(val(x), val(y))
    def op_Cast(self, n, a, x): return np.asarray(val(x)).astype(DT[a["DstT"]])

    def op_AddN(self, n, a, *xs):
        acc = val(xs[0])
        for x in xs[1:]:
            acc = acc + val(x)                       # left to right, as AddN's kernel does
        return acc

    def op_L2Loss(self, n, a, x):
        x = val(x)
        return np.sum(x * x, dtype=F32) / F32(2)

    # reductions
    def _reduce(self, fn, a, x, axes):
        x = val(x)
        axes = tuple(int(i) for i in np.atleast_1d(val(axes)))
        return fn(x, axis=axes, keepdims=bool(a.get("keep_dims", False)), dtype=x.dtype)

    def op_Sum(self, n, a, x, axes): return self._reduce(np.sum, a, x, axes)
    def op_Prod(self, n, a, x, axes): return self._reduce(np.prod, a, x, axes)

    def op_Mean(self, n, a, x, axes):
        x = val(x)
        ax = tuple(int(i) for i in np.atleast_1d(val(axes)))
        cnt = int(np.prod([x.shape[i] for i in ax])) if ax else 1
        s = np.sum(x, axis=ax, keepdims=bool(a.get("keep_dims", False)), dtype=x.dtype)
        return (s / x.dtype.type(cnt)).astype(x.dtype)

    def op_MatMul(self, n, a, x, y):
        x, y = val(x), val(y)
        if a.get("transpose_a"):
            x = x.T
        if a.get("transpose_b"):
            y = y.T
        if self.matmul_mode == "f32chain":
            out = np.zeros((x.shape[0], y.shape[1]), F32)
            for k in range(x.shape[1]):              # k-ordered float32 multiply-add chain (two roundings per term)
                out = (out + (x[:, k:k + 1] * y[k:k + 1, :]).astype(F32)).astype(F32)
            return out
        return (x.astype(np.float64) @ y.astype(np.float64)).astype(F32)

    # shape algebra
    def op_Shape(self, n, a, x): return np.array(np.shape(val(x)), dtype=DT[a.get("out_type", "DT_INT32")])
    def op_ShapeN(self, n, a, *xs): return tuple(np.array(np.shape(val(x)), dtype=np.int32) for x in xs)
    def op_Reshape(self, n, a, x, s): return np.reshape(val(x), [int(i) for i in val(s)])
    def op_Fill(self, n, a, dims, v): return np.full([int(i) for i in val(dims)], val(v), dtype=np.asarray(val(v)).dtype)
    def op_Tile(self, n, a, x, m): return np.tile(val(x), [int(i) for i in val(m)])
    def op_Range(self, n, a, s, l, d): return np.arange(int(val(s)), int(val(l)), int(val(d)), dtype=np.int32)
    def op_Pack(self, n, a, *xs): return np.stack([np.asarray(val(x)) for x in xs], axis=a.get("axis", 0))
    def op_ConcatV2(self, n, a, *xs): return np.concatenate([np.asarray(val(x)) for x in xs[:-1]], axis=int(val(xs[-1])))
    def op_Slice(self, n, a, x, b, s):
        x = val(x)
        idx = tuple(slice(int(bi), None if int(si) < 0 else int(bi) + int(si)) for bi, si in zip(val(b), val(s)))
        return x[idx]

    def op_Split(self, n, a, axis, x):
        return tuple(np.split(val(x), a["num_split"], axis=int(val(axis))))

    def op_ConcatOffset(self, n, a, axis, *shapes):
        axis = int(val(axis)); off = 0; out = []
        for s in shapes:
            o = np.zeros(len(val(s)), np.int32); o[axis] = off; off += int(val(s)[axis]); out.append(o)
        return tuple(out)

    def op_BroadcastGradientArgs(self, n, a, s0, s1):
        s0, s1 = [int(i) for i in val(s0)], [int(i) for i in val(s1)]
        r = max(len(s0), len(s1))
        p0, p1 = [1] * (r - len(s0)) + s0, [1] * (r - len(s1)) + s1
        # BCast (tensorflow/core/util/bcast.h): an operand's gradient is summed over every axis where that operand has
        # extent 1 after left-padding -- including axes where both have extent 1 (a no-op sum; TF lists them too)
        for i in range(r):
            assert p0[i] == p1[i] or p0[i] == 1 or p1[i] == 1, (s0, s1)
        return np.array([i for i in range(r) if p0[i] == 1], np.int32), np.array([i for i in range(r) if p1[i] == 1], np.int32)

    def op_DynamicStitch(self, n, a, *xs):
        k = len(xs) // 2
        idx = [np.asarray(val(x)) for x in xs[:k]]; dat = [np.asarray(val(x)) for x in xs[k:]]
        size = max(int(i.max()) for i in idx if i.size) + 1
        out = np.zeros((size,) + dat[0].shape[idx[0].ndim:], dat[0].dtype)
        for i, d in zip(idx, dat):
            out[i.reshape(-1)] = d.reshape((-1,) + d.shape[i.ndim:])
        return out

    def _strided_index(self, a, shape, b, e, s):
        assert a.get("ellipsis_mask", 0) == 0 and a.get("new_axis_mask", 0) == 0
        idx = []
        for i in range(len(b)):
            if a.get("shrink_axis_mask", 0) >> i & 1:
                idx.append(int(b[i]))
                continue
            bi = None if a.get("begin_mask", 0) >> i & 1 else int(b[i])
            ei = None if a.get("end_mask", 0) >> i & 1 else int(e[i])
            idx.append(slice(bi, ei, int(s[i])))
        return tuple(idx)

    def op_StridedSlice(self, n, a, x, b, e, s):
        x = np.asarray(val(x))
        return x[self._strided_index(a, x.shape, val(b), val(e), val(s))]

    def op_StridedSliceGrad(self, n, a, shape, b, e, s, dy):
        out = np.zeros([int(i) for i in val(shape)], np.asarray(val(dy)).dtype)
        out[self._strided_index(a, out.shape, val(b), val(e), val(s))] = val(dy)
        return out

    # optimiser (tensorflow/core/kernels/training_ops.cc, ApplyAdam, use_nesterov = false)
    def op_ApplyAdam(self, n, a, var, m, v, b1p, b2p, lr, b1, b2, eps, g):
        assert not a.get("use_nesterov", False)
        b1p, b2p, lr, b1, b2, eps, g = [_f(val(x)) for x in (b1p, b2p, lr, b1, b2, eps, g)]
        one = F32(1)
        alpha = lr * np.sqrt(one - b2p) / (one - b1p)
        mv = m.get() + (g - m.get()) * (one - b1)
        vv = v.get() + (g * g - v.get()) * (one - b2)
        m.set(_f(mv)); v.set(_f(vv))
        var.set(_f(var.get() - (mv * alpha) / (np.sqrt(vv) + eps)))
        return var

    # never on a fetched path of the hot loop (summaries / saver): executing one is an error in this harness
    def _forbidden(self, n, *x):
        raise RuntimeError("node %s (%s) is outside the hot path" % (n, self.ops[n]))
    op_ScalarSummary = op_MergeSummary = op_SaveV2 = op_RestoreV2 = _forbidden


def default_graph_path(ref="/root/reference"):
    import glob
    g = sorted(glob.glob(os.path.join(ref, "resources", "ppo_cl", "graphs", "*.meta.txt")))
    if not g:
        raise FileNotFoundError("no graph under %s" % ref)
    return g[0]
