"""Host-side logic above the C ABI: update rules, scaler, sharding arithmetic, synth."""
import numpy as np
import pytest

from scfgp_amd import synth
from scfgp_amd.optimizer import Optimizer as OPT, Shared, apply_updates
from scfgp_amd.scaler import Scaler
from scfgp_amd.sharded import shard_rows


class _G(object):
    def __init__(self):
        self.g = None

    def get_value(self, borrow=False):
        return self.g


def _run(algo, kw, theta0, gs, momentum=0.9):
    p = Shared(theta0); g = _G()
    up = OPT.apply_nesterov_momentum(getattr(OPT, algo)(p, g, **kw), momentum=momentum)
    out = []
    for gi in gs:
        g.g = gi
        apply_updates(up)
        out.append(p.get_value())
    return out


def test_adam_nesterov_three_steps():
    # Literal recurrences of SCFGP/Optimizer.py:314-330 followed by :88-96.  NOTE the
    # reference quirk reproduced here: apply_nesterov_momentum wraps
    # list(updates.keys())[0], and adam() inserts m_prev FIRST (:325), so the momentum
    # lands on the first-moment state, not on the parameter vector.
    rng = np.random.default_rng(1)
    th = rng.standard_normal(7); gs = [rng.standard_normal(7) for _ in range(3)]
    lr, b1, b2, eps, mom = 0.01, 0.9, 0.999, 1e-8, 0.9
    got = _run('adam', dict(learning_rate=lr, beta1=b1, beta2=b2, epsilon=eps), th, gs)
    m = np.zeros(7); v = np.zeros(7); vel = np.zeros(7); t = 0; x = th.copy()
    for k, g in enumerate(gs):
        t += 1
        a_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        m_t = b1 * m + (1 - b1) * g; v_t = b2 * v + (1 - b2) * g * g
        x = x - a_t * m_t / (np.sqrt(v_t) + eps)            # params: plain adam step
        vel_new = mom * vel + m_t - m                        # velocity of the m state
        m = mom * vel_new + m_t; vel = vel_new; v = v_t
        assert np.allclose(got[k], x, rtol=1e-13, atol=0)


def test_sgd_nesterov_three_steps():
    # sgd's only key is params (SCFGP/Optimizer.py:118), so here momentum acts on the vector
    rng = np.random.default_rng(2)
    th = rng.standard_normal(5); gs = [rng.standard_normal(5) for _ in range(3)]
    got = _run('sgd', dict(learning_rate=0.1), th, gs)
    vel = np.zeros(5); x = th.copy()
    for k, g in enumerate(gs):
        sg = x - 0.1 * g
        vel = 0.9 * vel + sg - x
        x = 0.9 * vel + sg
        assert np.allclose(got[k], x, rtol=1e-13, atol=0)


def _kat_kwargs(rule, kw):
    lr, r1, b2, eps = [float(v) for v in kw]
    return {'sgd': dict(learning_rate=lr), 'adagrad': dict(learning_rate=lr, epsilon=eps),
            'rmsprop': dict(learning_rate=lr, rho=r1, epsilon=eps), 'adadelta': dict(learning_rate=lr, rho=r1, epsilon=eps),
            'adam': dict(learning_rate=lr, beta1=r1, beta2=b2, epsilon=eps),
            'adamax': dict(learning_rate=lr, beta1=r1, beta2=b2, epsilon=eps)}[rule]


@pytest.mark.parametrize('wrapper', ['nesterov', 'plain', 'momentum'])
@pytest.mark.parametrize('rule', ['sgd', 'adagrad', 'rmsprop', 'adadelta', 'adam', 'adamax'])
def test_every_rule_follows_its_known_answer_vectors(rule, wrapper):
    """tests/golden/optimizer_kats.npz (tests/golden/make_optimizer_kats.py: the literal recurrences of SCFGP/Optimizer.py
    :27-382 on explicit state, independent of scfgp_amd/optimizer.py): three steps of every rule under
    apply_nesterov_momentum (what SCFGP.py:131 uses), without a wrapper and under apply_momentum -- parameter vector, the
    rule's state variables and the velocity after every step."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'optimizer_kats.npz'))
    p = Shared(z['theta0']); g = _G()
    up = getattr(OPT, rule)(p, g, **_kat_kwargs(rule, z[rule + '/kwargs']))
    plain_keys = list(up.keys())
    if wrapper == 'nesterov':
        up = OPT.apply_nesterov_momentum(up, momentum=float(z['momentum']))
    elif wrapper == 'momentum':
        up = OPT.apply_momentum(up, momentum=float(z['momentum']))
    vel = [k for k in up.keys() if k not in plain_keys]
    state = [k for k in plain_keys if k is not p and np.ndim(k.get_value(borrow=True)) == 1]      # accu | m, then delta_accu | v | u
    pre = '%s/%s/' % (rule, wrapper)
    for t, gi in enumerate(z['grads']):
        g.g = gi
        apply_updates(up)
        assert np.allclose(p.get_value(), z[pre + 'theta'][t + 1], rtol=1e-14, atol=0), (rule, wrapper, t)
        if state:
            assert np.allclose(state[0].get_value(), z[pre + 's1'][t + 1], rtol=1e-14, atol=0)
        if len(state) > 1:
            assert np.allclose(state[1].get_value(), z[pre + 's2'][t + 1], rtol=1e-14, atol=0)
        if vel:
            assert np.allclose(vel[0].get_value(), z[pre + 'vel'][t + 1], rtol=1e-14, atol=1e-300)


def test_optimizer_kats_file_is_what_its_generator_writes():
    import os
    from tests.golden import make_optimizer_kats as G
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'optimizer_kats.npz'))
    fresh = G.build()
    assert sorted(z.files) == sorted(fresh)
    for k in z.files:
        assert np.array_equal(z[k], fresh[k]), k
    # the vectors are not degenerate: every rule moves every coordinate with a non-zero gradient, wrappers differ
    for rule in G.RULES:
        th = z[rule + '/nesterov/theta']
        assert np.all(th[1][[0, 1, 3, 4, 6]] != th[0][[0, 1, 3, 4, 6]])
        assert not np.array_equal(z[rule + '/nesterov/theta'][3], z[rule + '/plain/theta'][3]) or rule != 'sgd'
        assert not np.array_equal(z[rule + '/nesterov/s1'][3], z[rule + '/plain/s1'][3]) or rule == 'sgd'


@pytest.mark.parametrize('algo,kw', [('sgd', {}), ('adagrad', {}), ('rmsprop', {}), ('adadelta', {}),
                                      ('adamax', {}), ('adam', {})])
def test_rules_run_and_descend(algo, kw):
    # minimise 0.5*|x|^2 : every bare rule must reduce the objective over a few dozen steps
    p = Shared(np.array([1.0, -2.0, 0.5])); g = _G()
    up = getattr(OPT, algo)(p, g, **kw)
    f0 = 0.5 * (p.get_value() ** 2).sum()
    for _ in range(60):
        g.g = p.get_value()
        apply_updates(up)
    assert 0.5 * (p.get_value() ** 2).sum() < f0


def test_nesterov_wraps_first_key_like_the_reference():
    # SCFGP/Optimizer.py:88: params = list(updates.keys())[0]
    p = Shared(np.zeros(3)); g = _G()
    base = OPT.adam(p, g)
    first = list(base.keys())[0]
    wrapped = OPT.apply_nesterov_momentum(base, momentum=0.9)
    assert first is not p                                  # adam inserts m_prev first
    assert len(wrapped) == len(base) + 1                   # one velocity state added
    assert wrapped[p] is base[p]                           # the parameter update itself is untouched


def test_defaults_match_reference():
    import inspect
    d = lambda f: {k: v.default for k, v in inspect.signature(f).parameters.items() if v.default is not inspect._empty}
    assert d(OPT.adam) == dict(learning_rate=0.01, beta1=0.9, beta2=0.99, epsilon=1e-8)     # Optimizer.py:279-283
    assert d(OPT.adamax) == dict(learning_rate=0.01, beta1=0.9, beta2=0.999, epsilon=1e-8)  # :334-338
    assert d(OPT.adadelta) == dict(learning_rate=0.01, rho=0.95, epsilon=1e-6)              # :216-219
    assert d(OPT.apply_nesterov_momentum) == dict(momentum=0.9)


def test_simultaneous_update_semantics():
    a = Shared(1.0); b = Shared(2.0)
    apply_updates({a: lambda: b.get_value() + 0, b: lambda: a.get_value() + 0})
    assert float(a.get_value()) == 2.0 and float(b.get_value()) == 1.0


@pytest.mark.parametrize('algo', Scaler.algos)
def test_scaler_round_trip(algo):
    rng = np.random.default_rng(3)
    X = np.exp(rng.standard_normal((200, 4))); X[:, 2] = 7.0                 # one constant column
    s = Scaler(algo); s.fit(X)
    t = s.forward_transform(X)
    assert t.shape == (200, 3)
    if algo != 'inv-normal':                                                # its backward is not an inverse in the reference
        back = s.backward_transform(t)
        assert np.allclose(back, X[:, [0, 1, 3]], rtol=1e-7, atol=1e-9)
    if algo in ('inv-normal', 'auto-inv-normal'):
        assert t.min() >= 0 and t.max() <= 1


def test_shard_rows_partition():
    for N, W in [(10, 3), (1000000, 8), (7, 8), (256, 2)]:
        blocks = [shard_rows(N, r, W) for r in range(W)]
        assert blocks[0][0] == 0 and blocks[-1][1] == N
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(W - 1))
        sizes = [hi - lo for lo, hi in blocks]
        assert max(sizes) - min(sizes) <= 1


def test_synth_is_counter_based():
    X = synth.make_X(123, 1000, 7)
    assert X.min() >= 0 and X.max() < 1 and abs(X.mean() - 0.5) < 0.02
    assert np.array_equal(synth.make_X(123, 100, 7, row0=400), X[400:500])   # any block, any rank
    z = synth.normal(5, 0, 200000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01


def test_residency_fingerprint_sees_every_element():
    """Writeable arrays are hashed completely at EVERY size (an in-place edit of one element of a 1e6 x 64 array
    must change the fingerprint); only read-only arrays, or arrays under a caller-maintained version token, are sampled, and
    two read-only arrays at unchanged addresses are not hashed at all."""
    from scfgp_amd import funcs
    rng = np.random.default_rng(0)
    X = rng.random((5000, 7)); y = rng.random((5000, 1))
    f0 = funcs._fingerprint(X, y)
    assert funcs._fingerprint(X, y) == f0
    X[4321, 6] = np.nextafter(X[4321, 6], 2.0)
    f1 = funcs._fingerprint(X, y)
    assert f1 != f0
    y[17, 0] += 1e-13
    assert funcs._fingerprint(X, y) != f1
    big = rng.random((1 << 24) + 12345)                        # above the old 2^24 sampling limit, threaded path
    k0 = funcs._content_key(big, False)
    assert k0[0] == 'full'
    big[5 * 64 + 7] += 1e-9                                    # an element no sample block of the old scheme contained
    assert funcs._content_key(big, False) != k0
    s0 = funcs._content_key(big, True)                         # trusted: sampled
    assert s0[0] == 'sampled'
    Xb = rng.random((70000, 3)); yb = rng.random((70000, 1))
    assert funcs._fingerprint(Xb, yb)[3][0] == 'full'
    assert funcs._fingerprint(Xb, yb, version=1)[3][0] == 'sampled'
    assert funcs._fingerprint(Xb, yb, version=1) != funcs._fingerprint(Xb, yb, version=2)
    Xb.flags.writeable = False; yb.flags.writeable = False
    assert funcs._fingerprint(Xb, yb)[3][0] == 'full'             # read-only alone buys no sampling: only a version token does


class _FakeEngine(object):
    def __init__(self):
        self.uploads = []

    def set_data(self, X, y, n_global=None):
        self.uploads.append((X.copy(), y.copy()))


def _funcs_without_a_gpu():
    """A CompiledFuncs whose engine records uploads: the residency logic of _sync_data runs without the library."""
    from scfgp_amd import funcs
    cf = funcs.CompiledFuncs.__new__(funcs.CompiledFuncs)
    cf.engine = _FakeEngine(); cf.allreduce = None; cf.n_global = None
    cf._resident = None; cf._resident_frozen = None
    return cf


def test_resident_rows_follow_what_the_caller_passes(monkeypatch):
    """ADVICE r03 (funcs.py _sync_data): (a) read-only arrays freed and re-created with the same shape land at the same
    address -- different contents must be uploaded; (b) a read-only VIEW of a writeable array hides nothing: an edit through
    the base is seen; (c) the same frozen objects are never hashed again; (d) writeable arrays are hashed on every call and
    an in-place edit is uploaded."""
    from scfgp_amd import funcs
    calls = []
    real = funcs._hash64
    monkeypatch.setattr(funcs, '_hash64', lambda buf: (calls.append(1), real(buf))[1])
    rng = np.random.default_rng(5)
    # (a) fresh frozen arrays per "fold", same shape: the allocator hands the freed block out again
    cf = _funcs_without_a_gpu()
    seen = set()
    for fold in range(6):
        X = rng.random((4096, 5)); y = rng.random((4096, 1))
        X.flags.writeable = False; y.flags.writeable = False
        seen.add(X.ctypes.data)
        cf._sync_data(X, y)
        assert len(cf.engine.uploads) == fold + 1 and np.array_equal(cf.engine.uploads[-1][0], X), fold
        # while the triple holds the fold's arrays their memory cannot be reused; once the caller drops its own references the
        # next fold's arrays are new objects and are hashed whatever address they get
        del X, y
    # (b) read-only view of a writeable base
    cf = _funcs_without_a_gpu()
    base = rng.random((3000, 4)); yb = rng.random((3000, 1))
    Xv = base.view(); Xv.flags.writeable = False
    yv = yb.view(); yv.flags.writeable = False
    assert not funcs._nobody_can_write(Xv)
    cf._sync_data(Xv, yv); cf._sync_data(Xv, yv)
    assert len(cf.engine.uploads) == 1
    base[1234, 2] += 1e-12                                      # through the base
    cf._sync_data(Xv, yv)
    assert len(cf.engine.uploads) == 2 and cf.engine.uploads[-1][0][1234, 2] == base[1234, 2]
    # (c) frozen owners: hashed once, then never
    cf = _funcs_without_a_gpu()
    X = np.array(rng.random((3000, 4))); y = np.array(rng.random((3000, 1)))
    X.flags.writeable = False; y.flags.writeable = False
    assert funcs._nobody_can_write(X) and funcs._nobody_can_write(np.frombuffer(b'12345678', dtype=np.float64))
    cf._sync_data(X, y)
    n0 = len(calls)
    for _ in range(5):
        cf._sync_data(X, y)
    assert len(calls) == n0 and len(cf.engine.uploads) == 1
    cf.set_data_version(7)                                      # a new token invalidates the shortcut
    cf._sync_data(X, y)
    assert len(calls) > n0
    # (d) writeable arrays: hashed every call; an edit is uploaded
    cf = _funcs_without_a_gpu()
    X = rng.random((3000, 4)); y = rng.random((3000, 1))
    cf._sync_data(X, y); n0 = len(calls)
    cf._sync_data(X, y)
    assert len(calls) == n0 + 2 and len(cf.engine.uploads) == 1
    y[2999, 0] = -y[2999, 0]
    cf._sync_data(X, y)
    assert len(cf.engine.uploads) == 2


def test_scaler_key_follows_contents_not_identity():
    from scfgp_amd.engine import HipEngine
    rng = np.random.default_rng(1)
    s = Scaler('auto-normal')
    s.fit(rng.random((300, 2)))
    k1 = HipEngine.scaler_key(s)
    assert HipEngine.scaler_key(s) == k1
    s.fit(rng.random((300, 2)) * 5 + 1)                        # same object, refitted in place
    assert HipEngine.scaler_key(s) != k1


@pytest.mark.parametrize('D,S,M', [(13, 8, 64), (32, 16, 256), (64, 32, 1024), (512, 64, 2048), (3, 2, 3), (8, 4, 60), (8, 2, 190)])
def test_gram_row_splits_tile_the_rows(D, S, M):
    """Row splits of the Gram products (scfgp_amd/csrc/kernels.h: RowSplits), uniform and tapered, checked by the
    library's host-only self-test: the splits tile [0, Np) in order on 256-row blocks."""
    from scfgp_amd import _lib
    lib = _lib.load()
    for N in (1, 257, 506, 3000, 100000, 125000, 1000000, 4000000):
        for dtype in (0, 1):
            for nsplit in (0, 1, 7, 16, 23, 48, 100, 1000):
                for taper in (0, 1, 2, 3):
                    assert lib.scfgp_selftest_row_splits(D, S, M, N, dtype, nsplit, taper) == 0, (N, dtype, nsplit, taper)
    assert lib.scfgp_selftest_row_splits(0, 1, 1, 5, 0, 0, 0) == -1


def test_output_arrays_are_reused_only_after_the_caller_let_go():
    """HipEngine hands out alpha / Li arrays from a small pool (a fresh 35.7 MB array per call costs ~1.4 ms of page faults at
    the headline shape); an array is handed out again only when nobody but the pool references it -- a view counts."""
    from scfgp_amd.engine import HipEngine

    class Shell(HipEngine):                                    # no context: only the pool logic
        def __init__(self):
            self._pool = {}; self.K = 40; self.P = 7; self.ctx = None

    e = Shell()
    a = e._fresh((40, 40)); b = e._fresh((40, 40))
    assert a is not b
    ida = id(a); a[...] = 1.0
    del a
    c = e._fresh((40, 40))
    assert id(c) == ida
    view = c[:3]; idc = id(c)
    del c
    d = e._fresh((40, 40))
    assert id(d) != idc and np.all(view == 1.0)
    cost, grad, alpha, Li = e._outputs(True)
    assert alpha.shape == (40, 1) and Li.shape == (40, 40) and grad.shape == (7,)
    assert e._outputs(False, factors=False)[2:] == (None, None)
    held = [e._fresh((40, 40)) for _ in range(6)]              # more than the pool keeps: still distinct arrays
    assert len({id(h) for h in held}) == 6
