"""
CPU tests that pin the oracle: (1) against the reference's own artifact (Theano-computed
Li, alpha, COST), (2) against the committed oracle KATs, (3) gradient three-way agreement,
(4) the invariances of the objective.
"""
import os

import numpy as np
import pytest

from oracle import scfgp_oracle as O
from oracle import autograd_ref as AR
from tests.golden.make_oracle_kats import CASES, case_inputs

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b))


def test_oracle_reproduces_reference_artifact():
    z = np.load(os.path.join(GOLD, 'artifact_kat.npz'))
    S, M = int(z['S']), int(z['M'])
    cost, alpha, Li = O.forward(z['X'], z['y'], z['params'], S, M, gauss_hermite=True)
    assert rel(Li, z['Li']) < 1e-12
    assert rel(alpha, z['alpha']) < 1e-11
    assert abs(cost - float(z['cost'])) < 1e-12 * abs(float(z['cost']))
    # the closed form of the 30-point Gauss-Hermite term is exact
    cost_cf, _, _ = O.forward(z['X'], z['y'], z['params'], S, M, gauss_hermite=False)
    assert abs(cost_cf - cost) < 1e-13 * abs(cost)
    # pair identity: A[j,j] + A[J+j,J+j] = N s^2 + 2 lam   (cos^2 + sin^2 = 1)
    a, b = z['params'][0], z['params'][1]
    L = np.linalg.inv(z['Li']); A = L @ L.T
    J = S + M
    n_rec = (np.diag(A)[:J] + np.diag(A)[J:] - 2 * (np.exp(2 * a) + 1e-6)) / (np.exp(2 * b) * 2.0 / M)
    assert np.allclose(n_rec, 400.0, rtol=0, atol=1e-8)


@pytest.mark.parametrize('name', list(CASES))
def test_oracle_matches_committed_kats(name):
    z = np.load(os.path.join(GOLD, 'oracle_kats.npz'))
    N, D, S, M, T, seed = CASES[name]
    X, y, params, Xs = case_inputs(name)
    cost, grad, alpha, Li = O.value_and_grad(X, y, params, S, M, chunk=500)
    assert abs(cost - float(z[name + '/cost'])) < 1e-11 * abs(cost)
    assert rel(grad, z[name + '/grad']) < 1e-9
    assert rel(alpha, z[name + '/alpha']) < 1e-8
    if name + '/Li' in z.files:
        assert rel(Li, z[name + '/Li']) < 1e-9
    else:
        assert rel(Li[z[name + '/Li_rows']], z[name + '/Li_sample']) < 1e-9
        assert abs(np.linalg.norm(Li) - float(z[name + '/Li_fro'])) < 1e-9 * float(z[name + '/Li_fro'])
    mu, std = O.predict(Xs, alpha, Li, params, S, M)
    assert rel(mu, z[name + '/mu']) < 1e-8 and rel(std, z[name + '/std']) < 1e-9


def test_gradient_three_way():
    X, y, params, _ = case_inputs('tiny_257x5')
    N, D, S, M, T, seed = CASES['tiny_257x5']
    c1, g1, _, _ = O.value_and_grad(X, y, params, S, M, chunk=64)
    c2, g2, _, _ = AR.value_and_grad(X, y, params, S, M)
    assert abs(c1 - c2) < 1e-12 * abs(c2) and rel(g1, g2) < 1e-10
    idx = np.arange(0, len(params), max(1, len(params) // 12))
    fd = O.fd_grad(X, y, params, S, M, idx, h=1e-6)
    assert np.abs(fd - g1[idx]).max() < 1e-6 * max(1.0, np.abs(g1).max())


def test_invariances():
    name = 'kin8nm_like'
    N, D, S, M, T, seed = CASES[name]
    X, y, params, _ = case_inputs(name)
    c0, g0, al0, Li0 = O.value_and_grad(X, y, params, S, M)
    # phase invariance: cost and non-phase gradient unchanged, phase gradient ~ 0
    p2 = params.copy(); p2[-(S + M):] += np.linspace(0.1, 2.0, S + M)
    c1, g1, _, _ = O.value_and_grad(X, y, p2, S, M)
    assert abs(c1 - c0) < 1e-9 * abs(c0)
    assert rel(g1[:-(S + M)], g0[:-(S + M)]) < 1e-6
    assert np.abs(g0[-(S + M):]).max() < 1e-8 * np.abs(g0).max()
    # row permutation invariance
    perm = np.random.default_rng(0).permutation(N)
    c2, g2, al2, _ = O.value_and_grad(X[perm], y[perm], params, S, M)
    assert abs(c2 - c0) < 1e-10 * abs(c0) and rel(g2, g0) < 1e-7 and rel(al2, al0) < 1e-7
    # staged engine == monolithic
    e = O.OracleEngine(D, S, M); e.set_params(params); e.set_data(X, y)
    e.pass1(); e.factor(); e.pass2(True); e.adjoint(); e.pass3()
    c3, g3, al3, Li3 = e.finish(True)
    assert abs(c3 - c0) < 1e-12 * abs(c0) and rel(g3, g0) < 1e-10


def test_host_scaler_is_pinned_by_the_artifacts_fitted_dictionaries():
    """scfgp_amd/scaler.py (the comparator of every device-scaler test) fed with the FITTED dictionaries stored in the
    reference's artifact lands on the scaled rows that reproduce Theano's Li / alpha / COST: pins forward_transform of
    'auto-inv-normal' (X) and 'auto-normal' (y), SCFGP/Scaler.py:107-116; the backward transform (:126-135) must invert
    it, and fit() on the same raw rows must find the same min / max and nearly the same Box-Cox exponents (:39-97)."""
    from scfgp_amd.scaler import Scaler
    z = np.load(os.path.join(GOLD, 'artifact_kat.npz'))

    def scaler(tag):
        sc = Scaler(str(z[tag + '_algo']))
        sc.data = {k[3:]: ([int(c) for c in z[k]] if k.endswith('cols') else z[k]) for k in z.files
                   if k.startswith(tag + '_') and not k.endswith('algo')}
        return sc

    xs, ys = scaler('xs'), scaler('ys')
    assert xs.algo == 'auto-inv-normal' and ys.algo == 'auto-normal'
    X = xs.forward_transform(z['Xraw']); y = ys.forward_transform(z['yraw'])
    assert np.array_equal(X, z['X']) and np.array_equal(y, z['y'])
    # ... and these rows, through the oracle, are the artifact's Theano outputs (same assertion as above, end to end)
    cost, alpha, Li = O.forward(X, y, z['params'], int(z['S']), int(z['M']), True)
    assert rel(Li, z['Li']) < 1e-12 and abs(cost - float(z['cost'])) < 1e-13 * abs(float(z['cost']))
    assert rel(ys.backward_transform(y), z['yraw']) < 1e-13
    assert rel(xs.backward_transform(X), z['Xraw'][:, xs.data['cols']]) < 1e-8      # column B: exponent 5 of values near the minimum
    # the dictionaries are what fit() finds on these 400 rows: min / max exactly, exponents to the optimiser's tolerance
    for tag, raw, ref in (('X', z['Xraw'], xs), ('y', z['yraw'], ys)):
        f = Scaler(ref.algo); f.fit(raw)
        assert f.data['cols'] == ref.data['cols']
        assert np.array_equal(f.data['min'], ref.data['min']) and np.array_equal(f.data['max'], ref.data['max'])
        assert np.allclose(f.data['boxcox'], ref.data['boxcox'], rtol=2e-2), (tag, f.data['boxcox'], ref.data['boxcox'])


def test_row_chunked_literal_graph_equals_the_unchunked_one():
    """oracle/autograd_ref.py value_and_grad_chunked (what tools/cpu_full.py times at the full 1e6 rows) is the same function:
    cost, gradient, alpha and Li of the literal graph, the row tensors formed 700 rows at a time."""
    from oracle import autograd_ref as AR
    rng = np.random.default_rng(3)
    N, D, S, M = 3000, 7, 4, 30
    X = rng.random((N, D)); y = rng.standard_normal((N, 1))
    p = O.init_params(D, S, M, rng); p[:3] = [-0.5, 0.1, -0.7]
    c0, g0, a0, L0 = AR.value_and_grad(X, y, p, S, M)
    c1, g1, a1, L1 = AR.value_and_grad_chunked(X, y, p, S, M, chunk=700)
    assert abs(c1 - c0) < 1e-13 * abs(c0) and np.linalg.norm(g1 - g0) < 1e-12 * np.linalg.norm(g0)
    assert np.linalg.norm(a1 - a0) < 1e-10 * np.linalg.norm(a0) and np.linalg.norm(L1 - L0) < 1e-10 * np.linalg.norm(L0)


def test_oracle_predict_side_closes_the_artifacts_recorded_validation_metrics():
    """Row h of the coverage table: the predictive mean / band of SCFGP/SCFGP.py:138-148,278-293 on the artifact's 106
    validation rows against the MAE / NMAE / MSE / NMSE / MNLP / SCORE the reference itself recorded (tests/artifact_predict.py
    states the relations).  Here: the oracle; tests/test_gpu_round5.py runs the same checker over the HIP path."""
    from tests import artifact_predict as AP
    z = AP.load()
    S, M = int(z['S']), int(z['M'])
    pred = lambda Xs, alpha, Li: O.predict(Xs, alpha, Li, z['params'], S, M)
    mu_y, std_y = AP.host_tail(pred, z)
    got = AP.check(mu_y, std_y, z)
    assert got['closure'] < 1e-12 and abs(abs(got['e_changed_2016']) - 3.4613474221682736) < 1e-10
    # the checker has teeth: 1e-7 on ONE predictive mean, or 1e-4 relative on every sigma (towards larger), breaks it
    bad = mu_y.copy(); bad[17] += 1e-7
    with pytest.raises(AssertionError):
        AP.check(bad, std_y, z)
    with pytest.raises(AssertionError):
        AP.check(mu_y, std_y * (1 + 2e-3), z)
    # our own metrics over all 106 rows differ from the record exactly by the changed row's share
    m = AP.metrics(mu_y, std_y, z['yv_raw'])
    rec = z['val_metrics']
    assert abs(106 * (rec[0] - m[0]) - (abs(got['e_changed_2016']) - abs(got['e_changed_today']))) < 1e-11
    assert abs(106 * (rec[2] - m[2]) - (got['e_changed_2016'] ** 2 - got['e_changed_today'] ** 2)) < 1e-10
