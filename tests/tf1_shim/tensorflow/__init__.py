"""A NumPy stand-in for the TensorFlow-1.3 symbols that /root/reference/boltzmann_machines uses.

TEST INFRASTRUCTURE ONLY.  Nothing under boltzmann_machines_amd/, bench.py's timed region or the C-ABI imports this;
`tests/golden/make_golden_from_reference.py` puts this directory in front of sys.path so that the UNMODIFIED reference
package (`/root/reference/boltzmann_machines`, Python 2 / TF 1.3 code that byte-compiles under Python 3.10) imports
it as `tensorflow`, builds its graphs with it and runs them.  The point: the golden fixtures of the parity tests
then come from the reference's own graph builders (rbm/base_rbm.py:244-531, dbm.py:233-769, layers.py) - their
control dependencies, loop structure, schedules, variable reads and writes - not from a second reading of them.

What is modelled (TF 1.x graph-mode semantics, as far as the reference relies on them)
  * a lazy graph of `Tensor` nodes with TF's name scopes (`scope/name_1` uniquification), collections,
    placeholders fed by name (`'input_data/X_batch:0'`), `Variable`s whose values live in the `Session`
    (`global_variables_initializer`, `Saver.save / restore`, `import_meta_graph`: in memory, keyed by path);
  * `Session.run(fetches, feed_dict)`: every node is evaluated at most once per run, fetches in list order, a node's
    control inputs before its data inputs.  A `Variable` read is NOT memoised: a consumer sees the value the variable
    holds when the consumer executes (TF-1 ref-variable behaviour), which is what gives `tf.control_dependencies` its
    meaning in dbm.py:521-523;
  * `tf.while_loop` with nested loop variables: the body is traced once, evaluated per iteration with a fresh memo;
  * float arithmetic in the dtype of the operands (float32 models compute in float32; Python scalars and NumPy
    operands are converted to the tensor operand's dtype, as `ops.convert_to_tensor(y, dtype=x.dtype)` does).
    `matmul` is NumPy's (BLAS) - the summation order of a float32 matmul is a backend detail, see DESIGN.md 5.

Random numbers.  TF draws from Philox4x32-10: key = seed, counter = (block, seed2) with element i of a float32
uniform = word i % 4 of block i / 4 (the reference's own known answer, rbm/tests/test_rbm.py:64-67, pins this).
  * an op WITH an op-level seed gets TF's literal stream: key = graph seed at creation (DEFAULT_GRAPH_SEED = 87654321
    without one), counter words 2, 3 = the op seed;
  * an op WITHOUT one gets (graph seed, id of the op in the graph) in real TF - unknowable without TF, so the stream
    of such an op is delegated to `set_rng_policy(fn)`: `fn(RandomSite)` returns `(key, word2, word3)`, or None to
    accept the literal stream of a seeded op.  The golden generator installs the engine's documented addressing
    (tests/golden/reference_rng_policy.py) there.  Without a policy an unseeded op raises.
"""
import builtins as _bi
import contextlib
import re

import numpy as np

from . import _philox

__version__ = '1.3.0-numpy-shim'

float32 = np.dtype('float32')
float64 = np.dtype('float64')
int32 = np.dtype('int32')
int64 = np.dtype('int64')
DEFAULT_GRAPH_SEED = 87654321


def _as_dtype(dt):
    if dt is None:
        return None
    if dt is _bi.bool or dt == 'bool':
        return np.dtype('bool')
    return np.dtype(dt)


# ----------------------------------------------------------------------------------------------- graph
class Frame(object):
    """the body of one tf.while_loop"""
    _n = 0

    def __init__(self, parent):
        Frame._n += 1
        self.id = Frame._n
        self.parent = parent


class Graph(object):
    def __init__(self):
        self.collections = {}
        self.seed = None
        self.nodes = []
        self.by_name = {}
        self.variables = []
        self._scope = ''
        self._used = {}
        self._ctrl = []
        self._frames = []

    # -- names ---------------------------------------------------------------------------------
    def unique_name(self, name, mark_as_used=True):
        full = self._scope + '/' + name if self._scope else name
        n = self._used.get(full, 0)
        if mark_as_used:
            self._used[full] = n + 1
        if n:
            base, i = full, n
            full = '%s_%d' % (base, i)
            while full in self._used:
                i += 1
                full = '%s_%d' % (base, i)
            if mark_as_used:
                self._used[full] = 1
        return full

    @contextlib.contextmanager
    def name_scope(self, name):
        old = self._scope
        self._scope = self.unique_name(name) if name else old
        try:
            yield self._scope + '/' if self._scope else ''
        finally:
            self._scope = old

    @contextlib.contextmanager
    def control_dependencies(self, ops):
        flat = [o for o in _flatten(ops) if o is not None]
        self._ctrl.append(flat)
        try:
            yield
        finally:
            self._ctrl.pop()

    @contextlib.contextmanager
    def as_default(self):
        global _default_graph
        old, _default_graph = _default_graph, self
        try:
            yield self
        finally:
            _default_graph = old

    def add_to_collection(self, name, value):
        self.collections.setdefault(name, []).append(value)

    def get_collection(self, name, scope=None):
        items = list(self.collections.get(name, []))
        if scope is not None:
            items = [x for x in items if hasattr(x, 'name') and re.match(scope, x.name)]
        return items

    def adopt(self, other):
        """import_meta_graph: this (fresh) graph takes over the nodes of a saved one; its own seed stays"""
        for a in ('collections', 'nodes', 'by_name', 'variables', '_used'):
            setattr(self, a, getattr(other, a))


_default_graph = Graph()
_default_session = []


def get_default_graph():
    return _default_graph


def reset_default_graph():
    global _default_graph
    _default_graph = Graph()


def set_random_seed(seed):
    _default_graph.seed = None if seed is None else int(seed)


def name_scope(name, default_name=None, values=None):
    return _default_graph.name_scope(name or default_name)


def control_dependencies(ops):
    return _default_graph.control_dependencies(ops)


def add_to_collection(name, value):
    _default_graph.add_to_collection(name, value)


def get_collection(name, scope=None):
    return _default_graph.get_collection(name, scope)


class GraphKeys(object):
    GLOBAL_VARIABLES = 'variables'
    SUMMARIES = 'summaries'


def _flatten(x):
    if isinstance(x, (list, tuple)):
        out = []
        for y in x:
            out.extend(_flatten(y))
        return out
    return [x]


def _pack_like(struct, flat):
    it = iter(flat)

    def rec(s):
        if isinstance(s, (list, tuple)):
            return [rec(y) for y in s]
        return next(it)
    return rec(struct)


# ----------------------------------------------------------------------------------------------- tensors
class TensorShape(object):
    def __init__(self, dims=None):
        self.dims = None if dims is None else list(dims)

    def as_list(self):
        return self.dims

    @property
    def ndims(self):
        return None if self.dims is None else len(self.dims)


class Tensor(object):
    _count = 0

    def __init__(self, op, inputs=(), attrs=None, name=None, dtype=None):
        g = _default_graph
        self.graph = g
        self.op = op
        self.inputs = list(inputs)
        self.attrs = attrs or {}
        self.dtype = _as_dtype(dtype)
        self.scope = g._scope
        self.op_name = g.unique_name(name or op)
        self.name = self.op_name + ':0'
        self.control_inputs = [c for lvl in g._ctrl for c in lvl]
        self.frame = g._frames[-1] if g._frames else None
        Tensor._count += 1
        self.creation_index = Tensor._count
        g.nodes.append(self)
        g.by_name[self.name] = self

    # -- python protocol -----------------------------------------------------------------------
    def __repr__(self):
        return '<shim.Tensor %s op=%s dtype=%s>' % (self.name, self.op, self.dtype)

    def __bool__(self):
        raise TypeError('using a tf.Tensor as a Python bool is not allowed (%r)' % self)

    def __iter__(self):
        raise TypeError('tf.Tensor %r is not iterable' % self)

    __hash__ = object.__hash__

    def get_shape(self):
        return TensorShape(self.attrs.get('static_shape'))

    def eval(self, feed_dict=None, session=None):
        return (session or _default_session[-1]).run(self, feed_dict=feed_dict)

    def __getitem__(self, idx):
        return Tensor('getitem', [self], {'idx': idx}, dtype=self.dtype)

    def __neg__(self):
        return _unary('neg', self)

    def __add__(self, o): return _binary('add', self, o)
    def __radd__(self, o): return _binary('add', o, self)
    def __sub__(self, o): return _binary('sub', self, o)
    def __rsub__(self, o): return _binary('sub', o, self)
    def __mul__(self, o): return _binary('mul', self, o)
    def __rmul__(self, o): return _binary('mul', o, self)
    def __truediv__(self, o): return _binary('div', self, o)
    def __rtruediv__(self, o): return _binary('div', o, self)
    __div__, __rdiv__ = __truediv__, __rtruediv__
    def __lt__(self, o): return _binary('less', self, o, out_dtype='bool')
    def __gt__(self, o): return _binary('greater', self, o, out_dtype='bool')
    def __le__(self, o): return _binary('less_equal', self, o, out_dtype='bool')
    def __ge__(self, o): return _binary('greater_equal', self, o, out_dtype='bool')


def _default_dtype(x):
    """dtype TF gives a Python / NumPy value: Python floats -> float32, Python ints -> int32, NumPy data keeps its own"""
    a = np.asarray(x)
    if isinstance(x, (np.ndarray, np.generic)):
        return a.dtype
    if a.dtype == np.float64:
        return float32
    if a.dtype == np.int64:
        return int32
    return a.dtype


def convert_to_tensor(x, dtype=None, name=None):
    dtype = _as_dtype(dtype)
    if isinstance(x, Tensor):
        return x
    if isinstance(x, (list, tuple)) and any(isinstance(y, Tensor) for y in _flatten(x)):
        parts = [convert_to_tensor(y, dtype) for y in x]
        return Tensor('stack', parts, dtype=parts[0].dtype)
    if dtype is None:
        dtype = _default_dtype(x)
    return Tensor('const', [], {'value': np.array(np.asarray(x), dtype=dtype)}, name=name or 'Const', dtype=dtype)


def _binary(op, a, b, out_dtype=None):
    if isinstance(a, Tensor) and not isinstance(b, Tensor):
        b = convert_to_tensor(b, a.dtype)
    elif isinstance(b, Tensor) and not isinstance(a, Tensor):
        a = convert_to_tensor(a, b.dtype)
    else:
        a, b = convert_to_tensor(a), convert_to_tensor(b)
    return Tensor(op, [a, b], dtype=out_dtype or a.dtype)


def _unary(op, a, dtype=None, **attrs):
    a = convert_to_tensor(a)
    return Tensor(op, [a], attrs, dtype=dtype or a.dtype)


class Variable(Tensor):
    def __init__(self, initial_value, dtype=None, name=None, trainable=True):
        dtype = _as_dtype(dtype)
        init = convert_to_tensor(initial_value, dtype if not isinstance(initial_value, Tensor) else None)
        Tensor.__init__(self, 'variable', [], {}, name=name or 'Variable', dtype=dtype or init.dtype)
        self.initial_value = init
        self.control_inputs = []
        self.frame = None
        self.graph.variables.append(self)
        self.graph.add_to_collection(GraphKeys.GLOBAL_VARIABLES, self)

    def assign(self, value):
        return assign(self, value)

    def assign_add(self, delta):
        return Tensor('assign_add', [convert_to_tensor(delta, self.dtype)], {'var': self}, dtype=self.dtype)

    def initialized_value(self):
        return self


def assign(ref, value, **kw):
    return Tensor('assign', [convert_to_tensor(value, ref.dtype)], {'var': ref}, dtype=ref.dtype)


def placeholder(dtype, shape=None, name=None):
    return Tensor('placeholder', [], {'static_shape': shape}, name=name or 'Placeholder', dtype=dtype)


def constant(value, dtype=None, shape=None, name=None):
    dtype = _as_dtype(dtype) or _default_dtype(value)
    a = np.array(np.asarray(value), dtype=dtype)
    if shape is not None:
        shp = [int(s) for s in shape]
        a = np.broadcast_to(a, shp).copy() if a.size == 1 else a.reshape(shp)
    return Tensor('const', [], {'value': a}, name=name or 'Const', dtype=dtype)


def identity(x, name=None):
    return Tensor('identity', [convert_to_tensor(x)], name=name or 'Identity', dtype=convert_to_tensor(x).dtype)


def cast(x, dtype, name=None):
    return Tensor('cast', [convert_to_tensor(x)], name=name or 'Cast', dtype=dtype)


def to_float(x):
    return cast(x, float32)


def to_int64(x):
    return cast(x, int64)


def _shape_arg(shape):
    """a shape given as a list of ints / floats / scalar tensors, or as one tensor: list of node inputs"""
    if isinstance(shape, Tensor):
        return [shape], True
    return [s if isinstance(s, Tensor) else int(s) for s in shape], False


def zeros(shape, dtype=float32, name=None):
    parts, whole = _shape_arg(shape)
    return Tensor('fill', [p for p in parts if isinstance(p, Tensor)], {'shape': parts, 'whole': whole, 'value': 0},
                  name=name or 'zeros', dtype=dtype)


def ones(shape, dtype=float32, name=None):
    parts, whole = _shape_arg(shape)
    return Tensor('fill', [p for p in parts if isinstance(p, Tensor)], {'shape': parts, 'whole': whole, 'value': 1},
                  name=name or 'ones', dtype=dtype)


def zeros_like(x, dtype=None):
    x = convert_to_tensor(x)
    return Tensor('like', [x], {'value': 0}, dtype=dtype or x.dtype)


def ones_like(x, dtype=None):
    x = convert_to_tensor(x)
    return Tensor('like', [x], {'value': 1}, dtype=dtype or x.dtype)


def shape(x):
    return Tensor('shape', [convert_to_tensor(x)], dtype=int32)


def range(*args):               # noqa: A001 - tf.range
    return Tensor('range', [convert_to_tensor(a) for a in args], dtype=int32)


def reshape(x, shp, name=None):
    parts, whole = _shape_arg(shp)
    x = convert_to_tensor(x)
    return Tensor('reshape', [x] + [p for p in parts if isinstance(p, Tensor)], {'shape': parts, 'whole': whole},
                  name=name or 'Reshape', dtype=x.dtype)


def transpose(x, perm=None):
    x = convert_to_tensor(x)
    return Tensor('transpose', [x], {'perm': perm}, dtype=x.dtype)


def expand_dims(x, axis):
    x = convert_to_tensor(x)
    return Tensor('expand_dims', [x], {'axis': axis}, dtype=x.dtype)


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    a, b = convert_to_tensor(a), convert_to_tensor(b)
    return Tensor('matmul', [a, b], {'ta': transpose_a, 'tb': transpose_b}, name=name or 'MatMul', dtype=a.dtype)


def einsum(eq, *xs):
    xs = [convert_to_tensor(x) for x in xs]
    return Tensor('einsum', xs, {'eq': eq}, dtype=xs[0].dtype)


def _reduce(kind):
    def f(x, axis=None, keep_dims=False, name=None):
        x = convert_to_tensor(x)
        return Tensor('reduce_' + kind, [x], {'axis': axis, 'keep': keep_dims}, dtype=x.dtype)
    return f


reduce_sum, reduce_mean, reduce_max, reduce_min = _reduce('sum'), _reduce('mean'), _reduce('max'), _reduce('min')


def add(a, b, name=None): return _binary('add', a, b)
def subtract(a, b, name=None): return _binary('sub', a, b)
def divide(a, b, name=None): return _binary('div', a, b)
def minimum(a, b): return _binary('minimum', a, b)
def maximum(a, b): return _binary('maximum', a, b)
def logical_and(a, b): return _binary('logical_and', a, b, out_dtype='bool')
def less(a, b): return _binary('less', a, b, out_dtype='bool')


def multiply(a, b, name=None):
    t = _binary('mul', a, b)
    return identity(t, name=name) if name else t


def square(x): return _unary('square', x)
def log(x): return _unary('log', x)
def exp(x): return _unary('exp', x)
def sqrt(x): return _unary('sqrt', x)
def floor(x): return _unary('floor', x)
def lgamma(x): return _unary('lgamma', x)
def log_sigmoid(x): return _unary('log_sigmoid', x)
def sigmoid(x): return _unary('sigmoid', x)


def clip_by_value(x, lo, hi):
    x = convert_to_tensor(x)
    return Tensor('clip', [x, convert_to_tensor(lo, x.dtype), convert_to_tensor(hi, x.dtype)], dtype=x.dtype)


def norm(x, ord='euclidean', axis=None):    # noqa: A002
    x = convert_to_tensor(x)
    return Tensor('norm', [x], {'ord': ord, 'axis': axis}, dtype=x.dtype)


def group(*ops, **kw):
    return Tensor('group', [o for o in _flatten(list(ops)) if o is not None], name=kw.get('name') or 'group_deps')


def no_op(name=None):
    return Tensor('group', [], name=name or 'NoOp')


class SparseTensor(object):
    def __init__(self, indices, values, dense_shape):
        self.indices, self.values, self.dense_shape = (convert_to_tensor(indices), convert_to_tensor(values),
                                                       convert_to_tensor(dense_shape))


def sparse_tensor_to_dense(sp, default_value=0):
    return Tensor('sparse_to_dense', [sp.indices, sp.values, sp.dense_shape], {'default': default_value},
                  dtype=sp.values.dtype)


def sparse_add(a, b):
    if isinstance(b, SparseTensor):
        return _binary('add', a, sparse_tensor_to_dense(b, 0))
    return _binary('add', sparse_tensor_to_dense(a, 0), b)


# ----------------------------------------------------------------------------------------------- random ops
class RandomSite(object):
    """what a policy gets to know about one execution of an unseeded random op"""

    def __init__(self, node, ctx):
        self.node = node
        self.kind = node.op                      # random_uniform | random_normal | multinomial
        self.scope = node.scope                  # full name-scope path at creation
        self.name = node.op_name
        self.creation_index = node.creation_index
        self.role = node.attrs.get('role')       # e.g. 'Bernoulli.sample', 'dropout', 'layer.init'
        self.graph_seed = ctx.session.graph.seed
        self.built_graph_seed = node.attrs.get('graph_seed')
        self.loop_iterations = [it for _, it in ctx.iter_stack]
        self.call = ctx.session.rng_calls        # runs of this session that executed a random op before this one
        self.initializer = ctx.initializing
        self.dtype = node.dtype


_rng_policy = [None]
_rng_trace = [None]


def set_rng_policy(fn):
    """fn(RandomSite) -> (key, counter_word_2, counter_word_3) for random ops without an op-level seed"""
    _rng_policy[0] = fn


def set_rng_trace(fn):
    """fn(RandomSite, stream, result) after every execution of a random op (diagnostics of the generator)"""
    _rng_trace[0] = fn


_compare_trace = [None]


def set_compare_trace(fn):
    """fn(scope, lhs, rhs) after every execution of a SCALAR `greater` (diagnostics of the generator: the loop condition
    `residual > mf_tol` of the mean-field while_loop, dbm.py:449-452)"""
    _compare_trace[0] = fn


def _k_greater(ctx, t, a, b):
    if _compare_trace[0] is not None and np.ndim(a) == 0 and np.ndim(b) == 0:
        _compare_trace[0](t.scope, float(a), float(b))
    return np.greater(a, b)


def _stream_of(node, ctx):
    seed = node.attrs.get('seed')
    site = RandomSite(node, ctx)
    if _rng_policy[0] is not None:
        r = _rng_policy[0](site)
        if r is not None:
            return r, site
    if seed is not None:                         # TF literal: (graph seed at creation or DEFAULT, op seed)
        g = node.attrs.get('graph_seed')
        return (DEFAULT_GRAPH_SEED if g is None else g, int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF), site
    raise RuntimeError('random op %s has no op-level seed: TF would derive its stream from the op id; install '
                       'a stream policy with tensorflow.set_rng_policy()' % node.name)


def random_uniform(shape, minval=0, maxval=None, dtype=float32, seed=None, name=None, role=None):
    parts, whole = _shape_arg(shape)
    dtype = _as_dtype(dtype)
    if maxval is None:
        maxval = 1
    ins = [p for p in parts if isinstance(p, Tensor)]
    lim = [convert_to_tensor(minval, dtype), convert_to_tensor(maxval, dtype)]
    return Tensor('random_uniform', ins + lim, {'shape': parts, 'whole': whole, 'seed': seed, 'nlim': 2,
                                                'graph_seed': _default_graph.seed, 'role': role},
                  name=name or 'random_uniform', dtype=dtype)


def random_normal(shape, mean=0.0, stddev=1.0, dtype=float32, seed=None, name=None, role=None):
    parts, whole = _shape_arg(shape)
    dtype = _as_dtype(dtype)
    ins = [p for p in parts if isinstance(p, Tensor)]
    lim = [convert_to_tensor(mean, dtype), convert_to_tensor(stddev, dtype)]
    return Tensor('random_normal', ins + lim, {'shape': parts, 'whole': whole, 'seed': seed, 'nlim': 2,
                                               'graph_seed': _default_graph.seed, 'role': role},
                  name=name or 'random_normal', dtype=dtype)


# ----------------------------------------------------------------------------------------------- while_loop
def while_loop(cond, body, loop_vars, shape_invariants=None, parallel_iterations=10, back_prop=True,
               swap_memory=False, name=None):
    g = _default_graph
    flat_init = [convert_to_tensor(x) for x in _flatten(loop_vars)]
    frame = Frame(g._frames[-1] if g._frames else None)
    with g.name_scope(name or 'while'):
        g._frames.append(frame)
        try:
            phs = [Tensor('loopvar', [], {'index': i}, name='loopvar', dtype=t.dtype) for i, t in enumerate(flat_init)]
            args = _pack_like(loop_vars, phs)
            cond_t = convert_to_tensor(cond(*args))
            out = body(*args)
            body_out = [convert_to_tensor(x) for x in _flatten(list(out) if isinstance(out, tuple) else out)]
        finally:
            g._frames.pop()
        if len(body_out) != len(flat_init):
            raise ValueError('while_loop: body returns %d tensors for %d loop variables' % (len(body_out), len(flat_init)))
        node = Tensor('while', flat_init, {'frame': frame, 'phs': phs, 'cond': cond_t, 'body': body_out}, name='while')
        outs = [Tensor('tuple_get', [node], {'i': i}, name='Exit', dtype=t.dtype) for i, t in enumerate(flat_init)]
    packed = _pack_like(loop_vars, outs)
    return tuple(packed) if isinstance(loop_vars, tuple) else packed


# ----------------------------------------------------------------------------------------------- evaluation
class _Ctx(object):
    def __init__(self, session, feeds, initializing=False):
        self.session = session
        self.feeds = feeds
        self.memo = {None: {}}
        self.iter_stack = []
        self.initializing = initializing
        self.used_rng = False

    def eval(self, t):
        if not isinstance(t, Tensor):
            return t
        if t in self.feeds:
            return self.feeds[t]
        if t.op == 'variable':
            try:
                return self.session.vars[t]
            except KeyError:
                raise RuntimeError('attempting to use uninitialized value %s' % t.name)
        memo = self.memo[t.frame.id if t.frame is not None else None]
        if t in memo:
            return memo[t]
        for c in t.control_inputs:
            self.eval(c)
        args = [self.eval(i) for i in t.inputs]
        val = _KERNELS[t.op](self, t, *args)
        memo[t] = val
        return val


def _np(x, dtype):
    return np.asarray(x, dtype=dtype)


def _resolve_shape(t, args):
    """shape attr with tensor entries replaced by their values; returns (shape list, remaining args)"""
    parts, whole = t.attrs['shape'], t.attrs['whole']
    n = sum(1 for p in parts if isinstance(p, Tensor))
    vals, rest = list(args[:n]), args[n:]
    if whole:
        return [int(v) for v in np.asarray(vals[0]).reshape(-1)], rest
    out = []
    for p in parts:
        out.append(int(vals.pop(0)) if isinstance(p, Tensor) else int(p))
    return out, rest


def _k_fill(ctx, t, *args):
    shp, _ = _resolve_shape(t, args)
    return np.full(shp, t.attrs['value'], dtype=t.dtype)


def _k_reshape(ctx, t, x, *args):
    shp, _ = _resolve_shape(t, args)
    return np.reshape(x, shp)


def _softplus(x):
    x = np.asarray(x)
    thr = x.dtype.type(np.log(np.finfo(x.dtype).eps) + 2.0)
    with np.errstate(over='ignore'):
        e = np.exp(x)
        return np.where(x > -thr, x, np.where(x < thr, e, np.log1p(e))).astype(x.dtype)


def _exp_eigen_f32(x0):
    """float32 exp as TensorFlow 1.3's CPU kernels compute it: Eigen 3.3 pexp<Packet4f> (Cephes expf; SSE build without FMA,
    so every pmadd is a rounded multiply and a rounded add).  Every line is ONE correctly rounded float32 operation of
    NumPy; the same sequence as oracle/bm_oracle.c orc_exp_eigen and csrc/bm_numerics.h exp_eigen (checked bit for bit by
    tests/test_oracle.py)."""
    f = np.float32
    x0 = np.asarray(x0, dtype=f)
    x = np.minimum(x0, f(88.3762626647950))
    x = np.maximum(x, f(-88.3762626647949))
    fx = x * f(1.44269504088896341)
    fx = fx + f(0.5)
    fx = np.floor(fx)
    tmp = fx * f(0.693359375)
    zz = fx * f(-2.12194440e-4)
    x = x - tmp
    x = x - zz
    z = x * x
    y = np.full_like(x, f(1.9875691500E-4))
    for c in (1.3981999507E-3, 8.3334519073E-3, 4.1665795894E-2, 1.6666665459E-1, 5.0000001201E-1):
        y = y * x
        y = y + f(c)
    y = y * z
    y = y + x
    y = y + f(1)
    p2 = ((fx.astype(np.int32) + np.int32(127)).astype(np.uint32) << np.uint32(23)).view(f)
    return np.maximum(y * p2, x0)


def _sigmoid(x):
    """tf.sigmoid = Eigen scalar_sigmoid_op: 1 / (1 + exp(-x)) in the operand's dtype.  float32 with Eigen's own exp (above)
    rather than NumPy's: the quantisation of this form decides how long the reference's mean-field loop runs
    (dbm.py:449-452), so the stand-in pins it operation by operation instead of inheriting libm's."""
    x = np.asarray(x)
    one = x.dtype.type(1)
    with np.errstate(over='ignore'):
        if x.dtype == np.float32:
            return (one / (one + _exp_eigen_f32(-x))).astype(x.dtype)
        return (one / (one + np.exp(-x))).astype(x.dtype)


def _k_matmul(ctx, t, a, b):
    if t.attrs['ta']:
        a = a.T
    if t.attrs['tb']:
        b = b.T
    return np.matmul(a, b).astype(t.dtype, copy=False)


def _k_reduce(fn):
    def k(ctx, t, x):
        axis = t.attrs['axis']
        if isinstance(axis, list):
            axis = tuple(axis)
        x = np.asarray(x)
        r = fn(x, axis=axis, keepdims=_bi.bool(t.attrs['keep']))
        return np.asarray(r, dtype=x.dtype)
    return k


def _k_norm(ctx, t, x):
    o, axis = t.attrs['ord'], t.attrs['axis']
    if o in (np.inf, 'inf'):
        return np.asarray(np.max(np.abs(x), axis=axis), dtype=x.dtype)
    if o in ('euclidean', 2):
        return np.sqrt(np.sum(x * x, axis=axis)).astype(x.dtype)
    raise NotImplementedError('tf.norm ord=%r' % (o,))


def _k_assign(ctx, t, value):
    var = t.attrs['var']
    v = np.array(value, dtype=var.dtype)
    ctx.session.vars[var] = v
    return v


def _k_assign_add(ctx, t, delta):
    var = t.attrs['var']
    v = np.asarray(ctx.session.vars[var] + delta, dtype=var.dtype)
    ctx.session.vars[var] = v
    return v


def _k_while(ctx, t, *init):
    a = t.attrs
    frame, phs, cond_t, body = a['frame'], a['phs'], a['cond'], a['body']
    vals = list(init)
    it = 0
    saved = ctx.memo.get(frame.id)
    while True:
        ctx.memo[frame.id] = dict(zip(phs, vals))
        ctx.iter_stack.append((t, it))
        try:
            if not _bi.bool(ctx.eval(cond_t)):
                break
            vals = [ctx.eval(x) for x in body]
        finally:
            ctx.iter_stack.pop()
        it += 1
    if saved is not None:
        ctx.memo[frame.id] = saved
    else:
        ctx.memo.pop(frame.id, None)
    return tuple(vals)


def _k_sparse_to_dense(ctx, t, idx, vals, shp):
    out = np.full([int(s) for s in shp], t.attrs['default'], dtype=vals.dtype)
    idx = np.asarray(idx).reshape(-1, len(shp))
    out[tuple(idx.T)] = vals
    return out


def _k_random(ctx, t, *args):
    shp, lim = _resolve_shape(t, args)
    n = int(np.prod(shp)) if shp else 1
    (key, w2, w3), site = _stream_of(t, ctx)
    ctx.used_rng = True
    a, b = lim
    if t.op == 'random_uniform':
        if t.dtype.kind == 'f':
            u = _philox.uniform(key, w2, w3, n, t.dtype)
            r = (u * (b - a) + a).astype(t.dtype) if (a != 0 or b != 1) else u
        else:                                    # integers: lo + word % range (TF UniformDistribution<int32>)
            w = _philox.words(key, w2, w3, n)
            r = (np.asarray(a, dtype=np.int64) + (w.astype(np.uint64) % np.uint64(int(b) - int(a))).astype(np.int64)
                 ).astype(t.dtype)
    else:
        z = _philox.normal(key, w2, w3, n, t.dtype)
        r = (z * b + a).astype(t.dtype)
    r = r.reshape(shp)
    if _rng_trace[0] is not None:
        _rng_trace[0](site, (key, w2, w3), r)
    return r


_KERNELS = {
    'const': lambda ctx, t: t.attrs['value'],
    'placeholder': lambda ctx, t: (_ for _ in ()).throw(RuntimeError('placeholder %s was not fed' % t.name)),
    'loopvar': lambda ctx, t: (_ for _ in ()).throw(RuntimeError('loop variable read outside its loop: %s' % t.name)),
    'identity': lambda ctx, t, x: x,
    'cast': lambda ctx, t, x: np.asarray(x).astype(t.dtype),
    'stack': lambda ctx, t, *xs: np.stack([np.asarray(x) for x in xs]),
    'getitem': lambda ctx, t, x: np.asarray(x)[t.attrs['idx']],
    'neg': lambda ctx, t, x: -x,
    'add': lambda ctx, t, a, b: np.asarray(a + b, dtype=t.dtype),
    'sub': lambda ctx, t, a, b: np.asarray(a - b, dtype=t.dtype),
    'mul': lambda ctx, t, a, b: np.asarray(a * b, dtype=t.dtype),
    'div': lambda ctx, t, a, b: np.asarray(a / b, dtype=t.dtype) if t.dtype.kind == 'f' else np.asarray(a // b, dtype=t.dtype),
    'less': lambda ctx, t, a, b: np.less(a, b),
    'greater': _k_greater,
    'less_equal': lambda ctx, t, a, b: np.less_equal(a, b),
    'greater_equal': lambda ctx, t, a, b: np.greater_equal(a, b),
    'logical_and': lambda ctx, t, a, b: np.logical_and(a, b),
    'minimum': lambda ctx, t, a, b: np.minimum(a, b).astype(t.dtype),
    'maximum': lambda ctx, t, a, b: np.maximum(a, b).astype(t.dtype),
    'square': lambda ctx, t, x: np.asarray(x * x, dtype=t.dtype),
    'log': lambda ctx, t, x: np.log(x).astype(t.dtype),
    'exp': lambda ctx, t, x: np.exp(x).astype(t.dtype),
    'sqrt': lambda ctx, t, x: np.sqrt(x).astype(t.dtype),
    'floor': lambda ctx, t, x: np.floor(x).astype(t.dtype),
    'lgamma': lambda ctx, t, x: _lgamma(x).astype(t.dtype),
    'sigmoid': lambda ctx, t, x: _sigmoid(x),
    'softplus': lambda ctx, t, x: _softplus(x),
    'log_sigmoid': lambda ctx, t, x: -_softplus(-np.asarray(x)),
    'softmax': lambda ctx, t, x: _softmax(x),
    'l2_loss': lambda ctx, t, x: np.asarray(np.sum(x * x) / x.dtype.type(2), dtype=t.dtype),
    'clip': lambda ctx, t, x, lo, hi: np.clip(x, lo, hi).astype(t.dtype),
    'fill': _k_fill,
    'like': lambda ctx, t, x: np.full(np.shape(x), t.attrs['value'], dtype=t.dtype),
    'shape': lambda ctx, t, x: np.asarray(np.shape(x), dtype=np.int32),
    'range': lambda ctx, t, *a: np.arange(*[int(v) for v in a], dtype=np.int32),
    'reshape': _k_reshape,
    'transpose': lambda ctx, t, x: np.transpose(x, t.attrs['perm']),
    'expand_dims': lambda ctx, t, x: np.expand_dims(x, t.attrs['axis']),
    'matmul': _k_matmul,
    'einsum': lambda ctx, t, *xs: np.einsum(t.attrs['eq'], *xs).astype(t.dtype),
    'reduce_sum': _k_reduce(np.sum),
    'reduce_mean': _k_reduce(np.mean),
    'reduce_max': _k_reduce(np.max),
    'reduce_min': _k_reduce(np.min),
    'norm': _k_norm,
    'group': lambda ctx, t, *xs: None,
    'assign': _k_assign,
    'assign_add': _k_assign_add,
    'while': _k_while,
    'tuple_get': lambda ctx, t, tup: tup[t.attrs['i']],
    'sparse_to_dense': _k_sparse_to_dense,
    'random_uniform': _k_random,
    'random_normal': _k_random,
    'summary': lambda ctx, t: b'',
    'init_all': lambda ctx, t: ctx.session._initialize_all(),
}


def register_kernel(op, fn):
    """lets tensorflow.contrib.distributions add its sampling kernels"""
    _KERNELS[op] = fn


def _lgamma(x):
    from scipy.special import gammaln
    return np.asarray(gammaln(np.asarray(x, dtype=np.float64)))


def _softmax(x):
    x = np.asarray(x)
    e = np.exp(x - np.max(x, axis=-1, keepdims=True))
    return (e / np.sum(e, axis=-1, keepdims=True)).astype(x.dtype)


# ----------------------------------------------------------------------------------------------- session
class ConfigProto(object):
    def __init__(self, *a, **kw):
        pass


class Session(object):
    def __init__(self, target='', graph=None, config=None):
        self.graph = graph or _default_graph
        self.vars = {}
        self.rng_calls = 0
        self.n_runs = 0

    def __enter__(self):
        _default_session.append(self)
        return self

    def __exit__(self, *exc):
        _default_session.pop()
        return False

    def close(self):
        pass

    def _feeds(self, feed_dict):
        feeds = {}
        for k, v in (feed_dict or {}).items():
            t = self.graph.by_name[k] if isinstance(k, str) else k
            feeds[t] = np.asarray(v, dtype=t.dtype)
        return feeds

    def run(self, fetches, feed_dict=None):
        ctx = _Ctx(self, self._feeds(feed_dict))
        self.n_runs += 1
        try:
            if isinstance(fetches, (list, tuple)):
                return [self._fetch(ctx, f) for f in fetches]
            return self._fetch(ctx, fetches)
        finally:
            if ctx.used_rng and not ctx.initializing:
                self.rng_calls += 1

    def _fetch(self, ctx, f):
        if isinstance(f, (list, tuple)):
            return [self._fetch(ctx, x) for x in f]
        if f is None:
            raise TypeError('Fetch argument None has invalid type')
        v = ctx.eval(f)
        return np.array(v) if isinstance(v, np.ndarray) else v

    def _initialize_all(self):
        ctx = _Ctx(self, {}, initializing=True)
        for var in self.graph.variables:
            self.vars[var] = np.array(ctx.eval(var.initial_value), dtype=var.dtype)
        return None


def global_variables_initializer():
    return Tensor('init_all', [], name='init')


# ----------------------------------------------------------------------------------------------- saver (in memory)
_CHECKPOINTS = {}
_META_GRAPHS = {}


class _Saver(object):
    def __init__(self, *a, **kw):
        pass

    def save(self, sess, save_path, global_step=None):
        import os
        path = save_path if global_step is None else '%s-%d' % (save_path, global_step)
        key = os.path.abspath(path)
        _CHECKPOINTS[key] = {v.name: np.array(val) for v, val in sess.vars.items()}
        _META_GRAPHS[key + '.meta'] = sess.graph
        d = os.path.dirname(key)
        if d and not os.path.isdir(d):
            os.makedirs(d)
        with open(key + '.shim-checkpoint', 'w') as f:       # a marker on disk, like the Saver's files
            f.write('in-memory checkpoint of the TF-1 shim\n')
        return path

    def restore(self, sess, save_path):
        import os
        data = _CHECKPOINTS[os.path.abspath(save_path)]
        for v in sess.graph.variables:
            sess.vars[v] = np.array(data[v.name])


class _Train(object):
    Saver = _Saver

    @staticmethod
    def import_meta_graph(path):
        import os
        _default_graph.adopt(_META_GRAPHS[os.path.abspath(path)])
        return _Saver()


train = _Train()


# ----------------------------------------------------------------------------------------------- nn / summary
class _NN(object):
    @staticmethod
    def sigmoid(x, name=None): return _unary('sigmoid', x)

    @staticmethod
    def softplus(x, name=None): return _unary('softplus', x)

    @staticmethod
    def softmax(x, name=None): return _unary('softmax', x)

    @staticmethod
    def l2_loss(x, name=None): return _unary('l2_loss', x)

    @staticmethod
    def dropout(x, keep_prob, noise_shape=None, seed=None, name=None):
        """tf.nn.dropout of TF 1.x: x / keep_prob * floor(keep_prob + U[0,1))"""
        x = convert_to_tensor(x)
        with name_scope(name or 'dropout'):
            u = random_uniform(shape(x), seed=seed, dtype=x.dtype, role='dropout')
            binary = _unary('floor', _binary('add', keep_prob, u))
            return _binary('mul', _binary('div', x, keep_prob), binary)


nn = _NN()


class _FileWriter(object):
    def __init__(self, logdir=None, graph=None, **kw):
        self.logdir = logdir
        self.events = []

    def add_summary(self, summary, global_step=None):
        self.events.append((global_step, summary))

    def flush(self): pass
    def close(self): pass


class _Summary(object):
    FileWriter = _FileWriter

    @staticmethod
    def _make(kind, name):
        t = Tensor('summary', [], {'kind': kind}, name=name)
        _default_graph.add_to_collection(GraphKeys.SUMMARIES, t)
        return t

    @staticmethod
    def histogram(name, values, **kw): return _Summary._make('histogram', name)

    @staticmethod
    def scalar(name, tensor, **kw): return _Summary._make('scalar', name)

    @staticmethod
    def image(name, tensor, max_outputs=3, **kw): return _Summary._make('image', name)

    @staticmethod
    def merge_all():
        return Tensor('summary', [], {'kind': 'merged'}, name='Merge/MergeSummary') \
            if _default_graph.get_collection(GraphKeys.SUMMARIES) else None


summary = _Summary()

bool = np.dtype('bool')         # noqa: A001 - tf.bool (the module uses builtins.bool as _bi.bool)
