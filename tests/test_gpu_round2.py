"""
GPU tests of the round-2 host-side fixes, all through the C ABI:
  * after index-list minibatches every "resident rows" entry point sees ALL rows again, with the n_global
    the data set was uploaded with (SCFGP/SCFGP.py:265 refits on all N rows);
  * the residency check hashes the arrays: an in-place edit of any element is seen (SCFGP/SCFGP.py:237: the
    reference re-reads its arguments on every call);
  * device scalers follow a refit of the SAME Scaler objects (SCFGP/SCFGP.py:155-156);
  * save()/load() carry the optimiser state (SCFGP/SCFGP.py:296-310, SCFGP/Optimizer.py:314-323,92-93);
  * scfgp_stream_fence orders the library's stream against another stream.
"""
import os

import numpy as np
import pytest

from oracle import scfgp_oracle as O
from tests.golden.make_oracle_kats import CASES, case_inputs

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def test_full_set_is_back_after_index_list_minibatches():
    from scfgp_amd.engine import HipEngine
    from scfgp_amd.funcs import CompiledFuncs
    name = 'kin8nm_like'
    N, D, S, M, T, seed = CASES[name]
    X, y, params, _ = case_inputs(name)
    c0, g0, a0, L0 = O.value_and_grad(X, y, params, S, M)
    eng = HipEngine(D, S, M); eng.set_params(params); eng.set_data(X, y)
    eng.eval_rows(np.arange(0, 150), True)
    # staged path (what CompiledFuncs.train_func / value_and_grad use)
    eng.pass1(); eng.factor(); eng.pass2(True); eng.adjoint(); eng.pass3()
    c, g, a, L = eng.finish(True)
    assert abs(float(c) - c0) < 1e-11 * abs(c0) and rel(g, g0) < 1e-8 and rel(a, a0) < 1e-8 and rel(L, L0) < 1e-9
    # a sharded upload keeps ITS n_global when the rows come back
    half = N // 2
    e2 = HipEngine(D, S, M); e2.set_params(params)
    e2.set_data(np.ascontiguousarray(X[:half]), np.ascontiguousarray(y[:half]), n_global=N)
    c_before = float(e2.eval(want_grad=False)[0])
    e2.eval_rows(np.arange(10, 60), False)
    assert float(e2.eval(want_grad=False)[0]) == c_before
    e2.eval_rows(np.arange(10, 60), False)
    e2.pass1(); e2.factor(); e2.pass2(False)
    assert float(e2.finish(False)[0]) == c_before
    # device optimiser: training after a minibatch runs on all rows
    cf = CompiledFuncs(D, S, M, params.copy(), 'adam', {'learning_rate': 0.01}, device_optimizer=True)
    cf._sync_params(); cf._sync_data(X, y)
    cf.engine.eval_rows(np.arange(0, 99), True)
    hist, _, _ = cf.train_iters(X, y, 1)
    assert abs(hist[0] - c0) < 1e-11 * abs(c0)
    for e in (eng, e2, cf.engine):
        e.close()


def test_facade_minibatch_training_ends_on_the_full_set():
    """SCFGP.optimize(nbatches>1): the final train_func(self.X, self.y) (SCFGP/SCFGP.py:265) is a fit of all N rows."""
    from scfgp_amd import SCFGP
    name = 'kin8nm_like'
    N, D, S, M, T, seed = CASES[name]
    X, y, _, _ = case_inputs(name)
    np.random.seed(11)
    model = SCFGP(sparsity=4, nfeats=10)
    model.set_data(X * 3 - 1, np.sin(X[:, :1] * 5) + 0.1 * y)
    model.optimize(None, None, max_iter=4, nbatches=3, batchsize=120)
    c0, a0, L0 = O.forward(model.X, model.y, model.params.get_value(), model.S, model.M)[:3]
    assert abs(model.evals['COST'][1][-1] - c0) < 1e-10 * abs(c0)
    assert rel(model.alpha, a0) < 1e-8 and rel(model.Li, L0) < 1e-9


def test_in_place_edit_of_any_element_is_seen():
    from scfgp_amd.funcs import CompiledFuncs
    name = 'kin8nm_like'
    N, D, S, M, T, seed = CASES[name]
    X, y, params, _ = case_inputs(name)
    cf = CompiledFuncs(D, S, M, params.copy())
    c0 = float(cf.train_func(X, y)[0])
    X[N // 2 + 1, D - 1] += 0.125                 # one element, no invalidate()
    c1 = float(cf.train_func(X, y)[0])
    c_ref = O.forward(X, y, params, S, M)[0]
    assert c1 != c0 and abs(c1 - c_ref) < 1e-10 * abs(c_ref)
    y[3, 0] -= 0.5
    c2 = float(cf.train_func(X, y)[0])
    c_ref = O.forward(X, y, params, S, M)[0]
    assert abs(c2 - c_ref) < 1e-10 * abs(c_ref)
    # under a caller-maintained version token the arrays are only sampled: the token is what announces an edit
    cf.invalidate()
    cf.set_data_version(1)
    c3 = float(cf.train_func(X, y)[0])
    X[7, 0] += 0.25
    cf.set_data_version(2)
    c4 = float(cf.train_func(X, y)[0])
    assert c4 != c3 and abs(c4 - O.forward(X, y, params, S, M)[0]) < 1e-10 * abs(c4)
    # read-only arrays at unchanged addresses: no hashing at all, same result
    Xf = X.copy(); yf = y.copy(); Xf.flags.writeable = False; yf.flags.writeable = False
    cf.set_data_version(None)
    c5 = float(cf.train_func(Xf, yf)[0]); c6 = float(cf.train_func(Xf, yf)[0])
    assert c5 == c4 and c6 == c5
    cf.engine.close()


def test_device_scalers_follow_a_refit_of_the_same_objects():
    from scfgp_amd import SCFGP
    rng = np.random.default_rng(8)
    np.random.seed(8)
    X1 = rng.uniform(-2, 2, (200, 3)); y1 = np.sin(X1[:, :1]) + 0.1 * rng.standard_normal((200, 1))
    X2 = rng.uniform(0, 9, (220, 3)); y2 = 4.0 + np.cos(X2[:, 1:2]) + 0.1 * rng.standard_normal((220, 1))
    Xt = rng.uniform(0, 9, (50, 3)); yt = 4.0 + np.cos(Xt[:, 1:2])
    model = SCFGP(sparsity=3, nfeats=8, device_scaler=True)
    model.set_data(X1, y1)
    model.optimize(None, None, max_iter=3)
    model.predict(Xt, yt)                         # registers the scalers fitted on (X1, y1)
    model.set_data(X2, y2)                        # refits the SAME Scaler objects in place
    mu_d, sd_d = model.predict(Xt, yt)
    dev = {k: model.evals[k][1][-1] for k in ('MAE', 'MSE', 'MNLP')}
    model.device_scaler = False
    mu_h, sd_h = model.predict(Xt, yt)
    assert np.allclose(mu_d, mu_h, rtol=1e-9, atol=1e-12) and np.allclose(sd_d, sd_h, rtol=1e-9, atol=1e-12)
    for k, v in dev.items():
        assert abs(model.evals[k][1][-1] - v) <= 1e-9 * max(1.0, abs(v)), k


@pytest.mark.parametrize('device_optimizer', [False, True])
def test_checkpoint_resumes_the_optimiser_trajectory(tmp_path, device_optimizer):
    """5 iterations, save, load into a fresh model, 5 more == 10 uninterrupted: bit for bit with the host rule,
    1e-12 with the device rule."""
    from scfgp_amd import SCFGP
    rng = np.random.default_rng(21)
    X = rng.uniform(-2, 2, (260, 4))
    y = np.sin(X[:, :1]) * X[:, 1:2] + 0.05 * rng.standard_normal((260, 1))
    algo = {'algo': 'adam', 'algo_params': {'learning_rate': 0.02, 'beta1': 0.9, 'beta2': 0.999, 'epsilon': 1e-8}}
    kw = dict(sparsity=3, nfeats=9, compat_noop_restore=True, device_optimizer=device_optimizer)

    def fresh():
        np.random.seed(77)
        m = SCFGP(**kw)
        m.set_data(X, y)
        return m

    full = fresh(); full.optimize(None, None, max_iter=10, algo=algo)
    first = fresh(); first.optimize(None, None, max_iter=5, algo=algo)
    path = os.path.join(str(tmp_path), 'ckpt.npz')
    first.save(path)
    with np.load(path) as z:
        assert 'opt_state_0' in z.files and str(z['opt_algo']) == 'adam'
    second = SCFGP(sparsity=1, nfeats=1, compat_noop_restore=True)
    second.set_data(X, y)
    second.load(path)
    assert second.device_optimizer == device_optimizer
    second.optimize(None, None, second.get_compiled_funcs(), max_iter=5, algo=algo)
    pa, pb = full.params.get_value(), second.params.get_value()
    if device_optimizer:
        assert rel(pb, pa) < 1e-12
    else:
        assert np.array_equal(pa, pb)
    ca, cb = full.evals['COST'][1], first.evals['COST'][1][:5] + second.evals['COST'][1][:5]
    assert np.allclose(ca[:10], cb, rtol=1e-12 if device_optimizer else 0, atol=0)


def test_stream_fence_orders_library_and_torch_streams():
    """Two shards on one GPU summed by torch on a SIDE stream: correct only because both sides are fenced."""
    import torch
    from scfgp_amd.engine import HipEngine
    from scfgp_amd.sharded import ShardedEvaluator, shard_rows
    name = 'c2_small_n'
    N, D, S, M, T, seed = CASES[name]
    X, y, params, _ = case_inputs(name)
    single = HipEngine(D, S, M); single.set_params(params); single.set_data(X, y)
    c0, g0, a0, L0 = single.eval(want_grad=True)
    engs = []
    for r in range(2):
        lo, hi = shard_rows(N, r, 2)
        e = HipEngine(D, S, M)                                 # private library stream each
        e.set_params(params); e.set_data(np.ascontiguousarray(X[lo:hi]), np.ascontiguousarray(y[lo:hi]), n_global=N)
        engs.append(e)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        peer = torch.cuda.current_stream().cuda_stream

        def allsum(stage):
            for e in engs: e.stream_fence(peer, 0)
            bufs = [e.exchange(stage) for e in engs]
            tot = bufs[0] + bufs[1]
            for b in bufs: b.copy_(tot)
            for e in engs: e.stream_fence(peer, 1)

        for e in engs: e.pass1()
        allsum(1)
        for e in engs: e.factor()
        for e in engs: e.pass2(True)
        allsum(2)
        for e in engs: e.adjoint()
        for e in engs: e.pass3()
        allsum(3)
        outs = [e.finish(True) for e in engs]
    for c, g, a, L in outs:
        assert abs(float(c) - float(c0)) < 1e-12 * abs(float(c0))
        assert rel(a, a0) < 1e-10 and rel(L, L0) < 1e-11 and rel(g, g0) < 1e-10
    assert ShardedEvaluator._peer_stream() is not False
    for e in engs + [single]:
        e.close()


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no torchrun environment starts two ranks (gloo rehearsal on this one GPU),
    prints n_gpus 2 and the same fp64 cost as one rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'TORCHELASTIC_RUN_ID')}
    base = [sys.executable, os.path.join(root, 'bench.py'), '--config', 'C2', '--rows', '20000', '--steps', '2', '--warmup', '1', '--no-cpu']
    one = subprocess.run(base, env=env, stdout=subprocess.PIPE, universal_newlines=True, timeout=600)
    two = subprocess.run(base + ['--gpus', '2', '--backend', 'gloo'], env=env, stdout=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert one.returncode == 0 and two.returncode == 0
    o1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith('{')][-1])
    o2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith('{')][-1])
    assert o1['n_gpus'] == 1 and o2['n_gpus'] == 2 and o2['config']['rows_per_gpu'] == 10000
    assert abs(o1['cost'] - o2['cost']) < 1e-11 * abs(o1['cost'])


@pytest.mark.parametrize('dtype,tol', [('f64', 1e-10), ('f32', 2e-4)])
def test_gram_row_splits_uniform_and_tapered_give_the_same_gram(dtype, tol):
    """The Gram products under 1, 3, 16 (XCD groups) and 40 row-split units, tapered or not, agree with the oracle."""
    from scfgp_amd.engine import HipEngine
    name = 'kin8nm_like'
    N, D, S, M, T, seed = CASES[name]
    X, y, params, _ = case_inputs(name)
    X = np.tile(X, (9, 1)); y = np.tile(y, (9, 1))               # 9000 rows: 36 row blocks
    Phi = O.feature_map(X, params, D, S, M)
    G0 = Phi.T @ Phi; g0 = Phi.T @ y.ravel()
    K = 2 * (S + M)
    c0, gr0, a0, L0 = O.value_and_grad(X, y, params, S, M)
    for nsplit, taper in ((0, 1), (1, 0), (3, 1), (16, 1), (16, 0), (40, 1), (40, 2), (48, 3)):
        eng = HipEngine(D, S, M, dtype=dtype); eng.set_params(params)
        eng.set_option('gram64', 0)                              # the fp32 product itself is under test
        eng.set_option('gram_nsplit', nsplit); eng.set_option('gram_taper', taper)
        eng.set_data(X, y)
        eng.pass1()
        Kp = eng.dims()['Kp']
        x1 = eng.debug_read('G', (Kp * Kp + Kp,))
        G = x1[:Kp * Kp].reshape(Kp, Kp)[:K, :K]; g = x1[Kp * Kp:Kp * Kp + K]
        assert rel(G, G0) < tol and rel(g, g0) < tol, (nsplit, taper, rel(G, G0), rel(g, g0))
        assert np.array_equal(G, G.T)
        eng.factor(); eng.pass2(True); eng.adjoint(); eng.pass3()
        c, gr, a, L = eng.finish(True)
        assert rel(gr, gr0) < max(tol, 1e-8) * 50
        eng.close()


@pytest.mark.parametrize('dtype,tol', [('f64', 1e-10), ('f32', 3e-6)])
@pytest.mark.parametrize('D,S,M,T', [(12, 9, 291, 5000), (30, 20, 44, 777), (20, 16, 1040, 40000)])
def test_predict_triangular_product_against_the_oracle(D, S, M, T, dtype, tol):
    """scfgp_predict forms sigma* from the triangular product Phi* Li^T (contraction of a column tile cut at its last
    column) and mu* from the tiles' own column bands: K = 600 (four 128-wide tiles, then 64-wide ones), 128 (one launch of
    64-wide tiles) and 2112 (the headline plan; two upload chunks) against the oracle on the same alpha / Li, T never a
    multiple of the row block."""
    from scfgp_amd.engine import HipEngine
    from scfgp_amd import synth
    seed = 0x5CF65000 + M
    K = 2 * (S + M)
    Xs = synth.make_X(seed, T, D)
    params = synth.make_params(seed + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
    rng = np.random.default_rng(seed)
    alpha = rng.standard_normal(K) / np.sqrt(K)
    Li = np.tril(rng.standard_normal((K, K))) / np.sqrt(K)
    eng = HipEngine(D, S, M, dtype=dtype); eng.set_params(params)
    mu, sd = eng.predict(Xs, alpha, Li)
    sel = np.r_[0:300, T // 2:T // 2 + 300, T - 300:T] if T > 2000 else np.arange(T)
    mu0, sd0 = O.predict(Xs[sel], alpha, Li, params, S, M)
    assert mu.shape == (T, 1) and sd.shape == (T,)
    assert rel(mu[sel], mu0) < tol and rel(sd[sel], sd0) < tol, (rel(mu[sel], mu0), rel(sd[sel], sd0))
    eng.close()


@pytest.mark.parametrize('dtype,ptol,ctol,gtol', [('f32', 2e-5, 2e-5, 3e-3), ('f64', 1e-11, 1e-10, 1e-8)])
@pytest.mark.parametrize('N,D,S,M', [(1500, 40, 16, 560), (3000, 12, 8, 184), (700, 24, 20, 300)])
def test_apply_tiles_fed_by_lds_dma_match(N, D, S, M, dtype, ptol, ctol, gtol):
    """Option apply_dma: the 128-wide tiles of the two apply products take both operands through LDS-DMA into a three-stage
    ring (no register staging), the 64-wide remainder and the mu slices as column bands; cost, gradient, alpha and the
    per-row adjoint scalars equal the register-staged path (same products, the 16 k of a stage summed in a permuted order)
    and the oracle at the mode's usual bounds.  K = 1152 (9 tiles), 384 (3), 640 (5).  fp32: 128- and 256-wide tiles;
    fp64: 128-wide tiles of 64 x 32 wave tiles (option value 2 means the same there)."""
    from scfgp_amd.engine import HipEngine
    from scfgp_amd import synth
    seed = 0x5CF66000 + N
    X = synth.make_X(seed, N, D)
    y = synth.normal(seed + 9, 0, N).reshape(-1, 1)
    params = synth.make_params(seed + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
    c0, g0, a0, _ = O.value_and_grad(X, y, params, S, M)
    out = {}
    for dma in (0, 1, 2):                                      # 2: 256-wide tiles (16 waves), a 128-wide one for an odd count
        eng = HipEngine(D, S, M, dtype=dtype); eng.set_params(params); eng.set_option('apply_dma', dma); eng.set_data(X, y)
        cost, grad, alpha, Li = eng.eval(want_grad=True)
        out[dma] = (float(cost), grad.copy(), alpha.copy(), eng.debug_read('p', (N,)).copy(), eng.debug_read('q', (N,)).copy())
        eng.close()
    for dma in (1, 2):
        assert abs(out[dma][0] - out[0][0]) < ptol * 1e-2 * abs(out[0][0])
        for k in (1, 2, 3, 4):
            assert rel(out[dma][k], out[0][k]) < ptol, (dma, k, rel(out[dma][k], out[0][k]))
        assert abs(out[dma][0] - c0) < ctol * abs(c0) and rel(out[dma][1], g0) < gtol and rel(out[dma][2], a0) < max(gtol / 3, 1e-9)


def _random_shapes(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        out.append((int(rng.integers(1, 3000)), int(rng.integers(1, 90)), int(rng.integers(1, 40)), int(rng.integers(1, 300))))
    return out


@pytest.mark.parametrize('dtype,ctol,gtol', [('f64', 1e-10, 1e-8), ('f32', 2e-5, 3e-3)])
def test_random_shapes_against_oracle(dtype, ctol, gtol):
    """24 seeded random (N, D, S, M): every tile-edge combination of the Gram job list (tall / wide / square / strip tiles,
    64- and 256-row granules), the apply column plans (128 + 64, 64 only) and the Cholesky step counts against the
    oracle's cost and gradient; fp64 tight, the fp32-storage modes at their usual bounds."""
    from scfgp_amd.engine import HipEngine
    for N, D, S, M in _random_shapes(24, 20261004):
        rng = np.random.default_rng(N * 131 + D * 17 + S * 3 + M)
        X = rng.random((N, D)); y = rng.standard_normal((N, 1))
        params = O.init_params(D, S, M, rng)
        params[0] = -0.4; params[1] = 0.1; params[2] = -0.6; params[3:3 + D * S] *= 0.6
        eng = HipEngine(D, S, M, dtype=dtype); eng.set_params(params); eng.set_data(X, y)
        cost, grad, alpha, Li = eng.eval(want_grad=True)
        c0, g0, a0, L0 = O.value_and_grad(X, y, params, S, M)
        assert abs(float(cost) - c0) < ctol * max(1.0, abs(c0)), (N, D, S, M, float(cost), c0)
        assert rel(grad, g0) < gtol, (N, D, S, M, rel(grad, g0))
        assert np.all(np.triu(Li, 1) == 0)
        eng.close()
