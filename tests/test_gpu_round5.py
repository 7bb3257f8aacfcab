"""
Round-5 GPU tests.

Row h: the predictive mean / band of the HIP path on the artifact's 106 validation rows against the MAE / NMAE /
MSE / NMSE / MNLP / SCORE the REFERENCE recorded for them (tests/artifact_predict.py states the relations; the CPU tier runs
the same checker over the oracle) -- through pred_func, through scfgp_predict_y and through the facade.
"""
import numpy as np
import pytest

from tests import artifact_predict as AP

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


@pytest.mark.parametrize('dtype,tol', [('f64', 1e-10), ('f32', 2e-4)])
def test_pred_func_closes_the_artifacts_recorded_validation_metrics(dtype, tol):
    """pred_func (SCFGP/SCFGP.py:138-148) + the host tail of SCFGP.predict (:279-284)."""
    from scfgp_amd.funcs import CompiledFuncs
    z = AP.load()
    cf = CompiledFuncs(int(z['D']), int(z['S']), int(z['M']), z['params'].copy(), dtype=dtype)
    mu_y, std_y = AP.host_tail(cf.triple()[2], z)
    if dtype == 'f64':
        got = AP.check(mu_y, std_y, z, tol)
        print('pred_func vs the reference record:', got)
        assert abs(abs(got['e_changed_2016']) - 3.4613474221682736) < 1e-9
    else:
        # fp32 features: the means carry ~1e-6; the closure is then ~1e-5 and the MNLP bound (slack 1e-3) cannot be asserted
        rec = z['val_metrics']; ch = int(z['val_changed']); keep = np.arange(106) != ch
        e = (mu_y - z['yv_raw']).ravel()
        e_new = 106 * rec[0] - np.abs(e[keep]).sum()
        assert abs(106 * rec[2] - (e[keep] ** 2).sum() - e_new ** 2) < tol
    cf.engine.close()


def test_predict_y_closes_the_artifacts_recorded_validation_metrics():
    """scfgp_predict_y: X scaler, pred_func, y-scaler backward transform, band and the six metrics on the device."""
    from scfgp_amd.funcs import CompiledFuncs
    z = AP.load()
    xs, ys = AP.scalers(z)
    cf = CompiledFuncs(int(z['D']), int(z['S']), int(z['M']), z['params'].copy())
    mu_y, std_y, met = cf.pred_y(z['Xv_raw'], xs, ys, z['alpha'], z['Li'], z['yv_raw'])
    got = AP.check(mu_y, std_y, z)
    # the device's six numbers are the reference's formulas over the device's own rows ...
    m = AP.metrics(mu_y, std_y, z['yv_raw'])
    assert np.allclose([met[k] for k in AP.NAMES], m, rtol=1e-12, atol=0)
    # ... and differ from the reference's record by the changed row's share alone
    rec = dict(zip(AP.NAMES, z['val_metrics'].tolist()))
    assert abs(106 * (rec['MAE'] - met['MAE']) - (abs(got['e_changed_2016']) - abs(got['e_changed_today']))) < 1e-10
    assert abs(106 * (rec['MSE'] - met['MSE']) - (got['e_changed_2016'] ** 2 - got['e_changed_today'] ** 2)) < 1e-10
    assert abs(met['MSE'] / met['NMSE'] - rec['MSE'] / rec['NMSE']) < 1e-12 * np.var(z['yv_raw'])
    cf.engine.close()


@pytest.mark.parametrize('device_scaler', [False, True])
def test_facade_predict_on_the_artifacts_validation_rows(device_scaler):
    """SCFGP.predict(Xv, yv) of a model carrying the artifact's state: evals[...] as the reference would append them."""
    from scfgp_amd import SCFGP
    from scfgp_amd.funcs import Shared
    z = AP.load()
    model = SCFGP(sparsity=int(z['S']), nfeats=int(z['M']), device_scaler=device_scaler)
    model.X_scaler, model.y_scaler = AP.scalers(z)
    model.D = int(z['D']); model.params = Shared(z['params'].copy())
    model.build_hip_models('adam', {'learning_rate': 0.01})
    model.alpha, model.Li = z['alpha'], z['Li']
    mu_y, std_y = model.predict(z['Xv_raw'], z['yv_raw'])
    got = AP.check(mu_y, std_y, z)
    m = AP.metrics(mu_y, std_y, z['yv_raw'])
    assert np.allclose([model.evals[k][1][-1] for k in AP.NAMES], m, rtol=1e-12, atol=0)
    assert got['closure'] < 1e-10


# ---------------------------------------------------------------------------------------------------------------------------
# Ranks decide together (include/scfgp_hip.h): two contexts on one card play two ranks, their
# exchange buffers summed in process the way the all-reduce would.
# ---------------------------------------------------------------------------------------------------------------------------
def _two_ranks(dtype='f32', opts=(), name='kin8nm_like'):
    import torch
    from scfgp_amd.engine import HipEngine
    from scfgp_amd.sharded import shard_rows
    from tests.golden.make_oracle_kats import CASES, case_inputs
    N, D, S, M, T, seed = CASES[name]
    X, y, params, _ = case_inputs(name)
    stream = torch.cuda.current_stream().cuda_stream
    engs = []
    for r in range(2):
        lo, hi = shard_rows(N, r, 2)
        e = HipEngine(D, S, M, dtype, stream=stream)
        for k, v in opts:
            e.set_option(k, v)
        e.set_params(params); e.set_data(X[lo:hi], y[lo:hi], n_global=N)
        engs.append(e)
    return engs


def _allsum(engs, stage):
    bufs = [e.exchange(stage) for e in engs]
    assert len({b.numel() for b in bufs}) == 1
    tot = bufs[0].clone()
    for b in bufs[1:]:
        tot += b
    for b in bufs:
        b.copy_(tot)


def _sharded_eval(engs, want_grad=True, count=None):
    """the staged evaluation over in-process ranks; every rank must take every REDO decision alike"""
    for _ in range(6):
        if count is not None:
            count['pass1'] = count.get('pass1', 0) + 1
        for e in engs: e.pass1()
        _allsum(engs, 1)
        go = [e.factor() for e in engs]
        assert len(set(go)) == 1, go
        if go[0] is False:
            if count is not None:
                count['factor_redo'] = count.get('factor_redo', 0) + 1
            continue
        for e in engs: e.pass2(want_grad)
        _allsum(engs, 2)
        if want_grad:
            for e in engs: e.adjoint()
            for e in engs: e.pass3()
            _allsum(engs, 3)
        outs = [e.finish(want_grad) for e in engs]
        assert all(o is None for o in outs) or all(o is not None for o in outs)
        if outs[0] is not None:
            return outs
    raise AssertionError('did not settle')


@pytest.mark.parametrize('deny,agreed,npass1,nfredo', [(2, 1, 2, 0), (1, 0, 3, 1)])
def test_two_row_shards_commit_to_the_precision_level_both_can_reach(deny, agreed, npass1, nfredo):
    """fp32 mode, auto policy, thresholds 0: the summed matrix asks for level 2 on both ranks; rank 1's buffers of level `deny`
    and up are refused (option test_deny_level).  Both ranks must finish at the same level, with the refusal on record on both,
    after the same number of rounds, with the numbers of two contexts that were told to run that level from the start."""
    want2 = (('cond_threshold', 0), ('cond_threshold_w', 0))
    engs = _two_ranks('f32', want2)
    engs[1].set_option('test_deny_level', deny)
    n = {}
    outs = _sharded_eval(engs, True, n)
    assert n['pass1'] == npass1 and n.get('factor_redo', 0) == nfredo
    lv = [e.condition()['level'] for e in engs]
    assert lv == [agreed, agreed]
    assert 'refused' in engs[0].last_error() and 'refused' in engs[1].last_error()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)                              # replicated K x K stage on equal sums: bit-equal ranks
    ref = _two_ranks('f32', (('gram64', {0: 0, 1: 1}[agreed]),))
    routs = _sharded_eval(ref, True)
    for a, b in zip(outs[0], routs[0]):
        assert np.array_equal(a, b)
    n2 = {}
    outs2 = _sharded_eval(engs, True, n2)                        # settled: one round, the refused level is not tried again
    assert n2 == {'pass1': 1} and np.array_equal(outs2[0][1], outs[0][1])
    for e in engs + ref:
        e.close()


@pytest.mark.parametrize('stage,want_grad', [(1, True), (2, True), (3, True), (2, False)])
def test_a_failing_rank_marks_its_exchanges_and_every_rank_fails_together(stage, want_grad):
    """rank 1's sweep `stage` fails (option test_fail_stage): it calls scfgp_fail_stage for that and every later exchange, the
    sums still happen, rank 0 runs to its finish and gets SCFGP_EPEER there; both contexts then evaluate as if nothing happened"""
    from scfgp_amd.engine import PeerFailed
    engs = _two_ranks('f64')
    good = _sharded_eval(engs, want_grad)
    a, b = engs
    b.set_option('test_fail_stage', stage)
    failed = False

    def sweep(s, fn):
        nonlocal failed
        if failed:
            b.fail_stage(s, want_grad)
            return
        try:
            fn(b)
        except RuntimeError as ex:
            assert 'injected failure' in str(ex)
            failed = True
            b.fail_stage(s, want_grad)

    a.pass1(); sweep(1, lambda e: e.pass1()); _allsum(engs, 1)
    a.factor()
    if not failed: b.factor()
    a.pass2(want_grad); sweep(2, lambda e: e.pass2(want_grad)); _allsum(engs, 2)
    if want_grad:
        a.adjoint()
        if not failed: b.adjoint()
        a.pass3(); sweep(3, lambda e: e.pass3()); _allsum(engs, 3)
    assert failed
    with pytest.raises(PeerFailed):
        a.finish(want_grad)
    again = _sharded_eval(engs, want_grad)
    assert float(again[0][0]) == float(good[0][0]) and np.array_equal(again[1][2], good[1][2])
    for e in engs:
        e.close()


@pytest.mark.parametrize('dtype', ['f64', 'f32'])
def test_training_with_the_sums_inside_the_library_equals_the_plain_loop(dtype):
    """scfgp_train with a (one-rank) communicator attached: the iteration carries its three ncclAllReduce calls, eagerly and --
    option use_graph = 2 -- as a captured graph; parameters, cost history and optimiser state equal the plain context's bit for
    bit (the sum over one rank changes nothing).  A sweep that fails inside the loop still posts every sum it owes: the profile
    of the failed call shows them, and the context trains on afterwards."""
    from scfgp_amd.engine import HipEngine
    from tests.golden.make_oracle_kats import CASES, case_inputs
    name = 'kin8nm_like'
    N, D, S, M, T, seed = CASES[name]
    X, y, params, _ = case_inputs(name)

    def fresh(comm, graph=None):
        e = HipEngine(D, S, M, dtype); e.set_params(params); e.set_data(X, y, n_global=N)
        if comm:
            e.comm_init(1, 0, e.comm_unique_id())
        if graph is not None:
            e.set_option('use_graph', graph)
        e.opt_init('adam', learning_rate=0.02)
        return e

    plain = fresh(False)
    h0, a0, L0 = plain.train(4)
    p0 = plain.get_params()
    for graph in (None, 2):
        e = fresh(True, graph)
        h1, a1, L1 = e.train(4)
        assert np.array_equal(h1, h0) and np.array_equal(e.get_params(), p0) and np.array_equal(a1, a0) and np.array_equal(L1, L0)
        for w in range(4):
            assert np.array_equal(e.opt_state(w), plain.opt_state(w))
        e.close()
    e = fresh(True)
    e.set_option('test_fail_stage', 2)
    with pytest.raises(RuntimeError, match='injected failure'):
        e.train(3)
    e.set_params(params); e.opt_init('adam', learning_rate=0.02)
    h2, _, _ = e.train(4)
    assert np.array_equal(h2, h0)
    # the evaluation entry point: scfgp_eval posts the sums it owes too, then evaluates as before
    c0 = float(e.eval()[0])
    e.set_option('test_fail_stage', 3)
    e.set_profiling(True)
    with pytest.raises(RuntimeError, match='injected failure'):
        e.eval()
    assert [n for n, _ in e.timings() if n.startswith('exchange')] == ['exchange1', 'exchange2', 'exchange3']
    assert float(e.eval()[0]) == c0
    e.close(); plain.close()
    # a shard without its peers: the on-device loop would train on its own rows' sums -- refused
    lone = HipEngine(D, S, M, dtype); lone.set_params(params); lone.set_data(X[:N // 2], y[:N // 2], n_global=N)
    lone.opt_init('adam')
    with pytest.raises(ValueError, match='communicator'):
        lone.train(2)
    lone.close()


# ---------------------------------------------------------------------------------------------------------------------------
# Compute mode f16x3 (include/scfgp_hip.h: SCFGP_F16X3; scfgp_amd/csrc/apply_f16.hip, gram_f16.hip): a labelled SECONDARY mode -- fp32 mode
# whose two square apply products and two Gram products run as a three-term fp16 split.  It must pass fp32 mode's parity tier (|dcost| <= 1e-5
# max(1, |cost|), per-block gradient norms <= 1e-3, alpha / Li / mu* / sigma* as fp32 mode) against fp64 mode = the reference's arithmetic.
# ---------------------------------------------------------------------------------------------------------------------------
def _blocks(g, D, S, M):
    o = 3 + D * S
    return g[:3], g[3:o], g[o:o + M * S]


@pytest.mark.parametrize('f16_gram', [1, 0])
@pytest.mark.parametrize('N,D,S,M,abc', [(20000, 16, 16, 256, (-1.0, 0.0, -1.0)), (33000, 8, 32, 256, (-1.0, 0.0, -1.0)),
                                          (40000, 32, 16, 496, (-2.0, 0.5, -1.0)), (30000, 24, 20, 300, (-0.5, -0.5, -2.0))])
def test_f16x3_mode_passes_fp32_modes_parity_tier(N, D, S, M, abc, f16_gram):
    """K = 544, 576, 1024, 640: two or four 256-wide column tiles on the split (option apply_dma = 2 forces the 256-wide tiles below the
    size from which fp32 mode picks them itself) + an exact-fp32 remainder of 32 / 64 / 0 / 128 columns.  Against fp64 mode with fp32
    mode's tolerances, against fp32 mode itself (no worse than 4x its error where that error is measurable), and bit-equal repeats.
    f16_gram = 0 keeps the fp32 Gram (the mode's apply half alone); with it on, K / 256 rounded up = 3, 3, 4, 3 column blocks of the
    fp16 Gram's A side, the last one sticking out of Kp = 640, 640, 1024, 640 in three of the shapes, and row chunks of 1024 .. 3584 rows
    (the last one short; fewer rows per chunk than the 8192 of a large problem so that the chip has two rounds of jobs)."""
    from scfgp_amd import synth
    from scfgp_amd.engine import HipEngine
    seed = 0x5CF60A00 + M
    X = synth.make_X(seed, N, D); y = synth.normal(seed + 1, 0, N).reshape(-1, 1)
    params = synth.make_params(seed + 2, D, S, M, abc=abc)
    Xs = synth.make_X(seed + 3, 3000, D)
    outs = {}
    for dtype in ('f64', 'f32', 'f16x3'):
        e = HipEngine(D, S, M, dtype)
        if dtype != 'f64':
            e.set_option('gram64', 0); e.set_option('apply_dma', 2)      # no fp64 Gram in either: the comparison is about the split products
        if dtype == 'f16x3':
            e.set_option('f16_gram', f16_gram)
        e.set_params(params); e.set_data(X, y)
        e.set_profiling(True)
        a = e.eval()
        names = [n for n, _ in e.timings()]
        assert ('split_phi' in names) == (dtype == 'f16x3')              # the split path ran (and only there)
        assert ('split_v' in names) == (dtype == 'f16x3' and f16_gram == 1)
        b = e.eval()
        assert float(a[0]) == float(b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        mu, sd = e.predict(Xs, a[2], a[3])
        outs[dtype] = a + (mu, sd)
        e.close()
    c64, g64, a64, L64, mu64, sd64 = outs['f64']
    err = {}
    for dtype in ('f32', 'f16x3'):
        c, g, a, L, mu, sd = outs[dtype]
        err[dtype] = dict(cost=abs(float(c) - float(c64)) / max(1.0, abs(float(c64))), alpha=rel(a, a64), Li=rel(L, L64), mu=rel(mu, mu64),
                          sd=rel(sd, sd64), **{'g%d' % k: rel(u, v) for k, (u, v) in enumerate(zip(_blocks(g, D, S, M), _blocks(g64, D, S, M)))})
    print('\nf32  ', {k: '%.1e' % v for k, v in err['f32'].items()})
    print('f16x3', {k: '%.1e' % v for k, v in err['f16x3'].items()})
    e16, e32 = err['f16x3'], err['f32']
    assert e16['cost'] < 1e-5 and e16['g0'] < 1e-3 and e16['g1'] < 1e-3 and e16['g2'] < 1e-3
    if not f16_gram:       # alpha and Li come from pass 1 (the fp32 Gram, untouched then): identical to fp32 mode's
        assert np.array_equal(outs['f16x3'][2], outs['f32'][2]) and np.array_equal(outs['f16x3'][3], outs['f32'][3])
    for k in ('cost', 'g0', 'g1', 'g2', 'alpha', 'Li', 'mu', 'sd'):
        assert e16[k] <= 4 * e32[k] + 2e-9, (k, e16[k], e32[k])
    assert e16['alpha'] < 1e-3 and e16['Li'] < 1e-3 and e16['mu'] < 1e-3 and e16['sd'] < 1e-3


def test_f16x3_mode_at_the_headline_shape():
    """N = 1e6, D = 64, S = 32, M = 1024 (the library picks the 256-wide tiles itself): f16x3 against fp32 mode on the same rows --
    cost to 1e-7, gradient blocks to 2e-5 (fp32 mode itself is 2.5e-6 from fp64 mode there), alpha / Li to 1e-4 norm-wise, repeats
    bit-equal -- and the two apply stages at least twice as fast, the two Gram products at least 1.67 times."""
    import bench
    from scfgp_amd.engine import HipEngine
    N, D, S, M = bench.CONFIGS['H'][:4]
    e0 = HipEngine(D, S, M, 'f32')
    X, y, params = bench.build_problem(e0, N, D, S, M, 0, N, None)
    e0.set_params(params); e0.set_data(X, y)
    e0.eval(); e0.set_profiling(True)
    c0, g0, a0, L0 = e0.eval(); t0 = dict(e0.timings())
    e0.close()
    e1 = HipEngine(D, S, M, 'f16x3'); e1.set_params(params); e1.set_data(X, y)
    e1.eval(); e1.set_profiling(True)
    c1, g1, a1, L1 = e1.eval(); t1 = dict(e1.timings())
    c2, g2, _, _ = e1.eval()
    e1.close()
    assert float(c1) == float(c2) and np.array_equal(g1, g2)
    print('\ncost %.3e  alpha %.2e  Li %.2e  grad blocks %s' % (abs(float(c1) - float(c0)) / max(1.0, abs(float(c0))), rel(a1, a0), rel(L1, L0),
          ['%.1e' % rel(u, v) for u, v in zip(_blocks(g1, D, S, M), _blocks(g0, D, S, M))]))
    assert rel(a1, a0) < 1e-4 and rel(L1, L0) < 1e-4
    assert abs(float(c1) - float(c0)) < 1e-7 * max(1.0, abs(float(c0)))
    for u, v in zip(_blocks(g1, D, S, M), _blocks(g0, D, S, M)):
        assert rel(u, v) < 2e-5, rel(u, v)
    print('apply_v %.2f -> %.2f ms, apply_phibar %.2f -> %.2f ms, gram %.2f -> %.2f ms, gram_w %.2f -> %.2f ms, split_phi %.2f ms, split_v %.2f ms,'
          ' reduce_tiles %.2f -> %.2f ms' % (t0['apply_v'], t1['apply_v'], t0['apply_phibar'], t1['apply_phibar'], t0['gram'], t1['gram'],
                                            t0['gram_w'], t1['gram_w'], t1['split_phi'], t1['split_v'], t0['reduce_tiles'], t1['reduce_tiles']))
    assert t1['apply_v'] < 0.5 * t0['apply_v'] and t1['apply_phibar'] < 0.5 * t0['apply_phibar']
    assert t1['gram'] < 0.6 * t0['gram'] and t1['gram_w'] < 0.6 * t0['gram_w']     # 0.43-0.47 over the round's boxes


def test_f16x3_training_loop_under_the_graph_equals_its_evaluations_and_tracks_fp32_mode():
    """scfgp_train in f16x3 mode (K = 1024, 70000 rows: the library picks the 256-wide tiles itself, so the split products run inside the
    captured graph): the cost history equals eval + opt_step done one by one in the same mode, bit for bit, and stays within fp32
    mode's parity tier of fp32 mode's own history; repeated training from the same start is bit-equal."""
    from scfgp_amd import synth
    from scfgp_amd.engine import HipEngine
    N, D, S, M = 70000, 16, 16, 496
    seed = 0x5CF60B00
    X = synth.make_X(seed, N, D); y = synth.normal(seed + 1, 0, N).reshape(-1, 1)
    params = synth.make_params(seed + 2, D, S, M, abc=(-1.0, 0.0, -1.0))
    hist = {}
    for dtype in ('f32', 'f16x3'):
        e = HipEngine(D, S, M, dtype)
        e.set_option('gram64', 0)
        e.set_params(params); e.set_data(X, y)
        e.opt_init('adam', learning_rate=1e-3)
        h, a, L = e.train(4)
        hist[dtype] = h
        if dtype == 'f16x3':
            p_end = e.get_params()
            e.set_params(params); e.opt_init('adam', learning_rate=1e-3)
            h2, _, _ = e.train(4)
            assert np.array_equal(h2, h) and np.array_equal(e.get_params(), p_end)
            e.set_params(params); e.opt_init('adam', learning_rate=1e-3)
            e.set_profiling(True)
            steps = []
            for _ in range(4):
                c, g, _, _ = e.eval()
                steps.append(float(c)); e.opt_step(g)
            assert 'split_v' in [n for n, _ in e.timings()]
            assert np.array_equal(np.array(steps), h) and np.array_equal(e.get_params(), p_end)
        e.close()
    assert np.all(np.abs(hist['f16x3'] - hist['f32']) <= 1e-5 * np.maximum(1.0, np.abs(hist['f32']))), (hist['f16x3'], hist['f32'])
    assert hist['f32'][-1] < hist['f32'][0]


@pytest.mark.parametrize('level', [1, 2])
def test_f16x3_mode_under_the_precision_levels(level):
    """The precision levels of fp32 mode keep their own products in f16x3 mode: level 1 forms pass 1's Gram in fp64 (the split of Phi is
    then its own stage, the weighted Gram and both apply products stay on the fp16 pipe), level 2 runs pass 2 in factor form on fp32
    tiles (only Phibar = 2 Phi Abar + ... is left to the split).  Forced by zero thresholds; against fp64 mode no worse than 4x fp32
    mode at the same level, alpha and Li bit-equal to it at both levels (fp64 Gram in both modes), repeats bit-equal."""
    from scfgp_amd import synth
    from scfgp_amd.engine import HipEngine
    N, D, S, M = 33000, 8, 32, 256
    seed = 0x5CF60C00 + level
    X = synth.make_X(seed, N, D); y = synth.normal(seed + 1, 0, N).reshape(-1, 1)
    params = synth.make_params(seed + 2, D, S, M, abc=(-1.0, 0.0, -1.0))
    outs = {}
    for dtype in ('f64', 'f32', 'f16x3'):
        e = HipEngine(D, S, M, dtype)
        if dtype != 'f64':
            e.set_option('apply_dma', 2); e.set_option('cond_threshold', 0)
            if level == 2:
                e.set_option('cond_threshold_w', 0)
        e.set_params(params); e.set_data(X, y)
        e.eval()                                                   # settles the level (the first evaluation repeats itself)
        e.set_profiling(True)
        a = e.eval()
        names = [n for n, _ in e.timings()]
        if dtype != 'f64':
            assert int(e.condition()['level']) == level and 'gram64' in names
            assert ('apply_c' in names) == (level == 2)
        if dtype == 'f16x3':
            assert 'split_phi' in names and ('split_v' in names) == (level == 1)
        b = e.eval()
        assert float(a[0]) == float(b[0]) and np.array_equal(a[1], b[1])
        outs[dtype] = a
        e.close()
    c64, g64, a64, L64 = outs['f64']
    err = {}
    for dtype in ('f32', 'f16x3'):
        c, g, a, L = outs[dtype]
        err[dtype] = dict(cost=abs(float(c) - float(c64)) / max(1.0, abs(float(c64))), alpha=rel(a, a64), Li=rel(L, L64),
                          **{'g%d' % k: rel(u, v) for k, (u, v) in enumerate(zip(_blocks(g, D, S, M), _blocks(g64, D, S, M)))})
    print('\nf32  ', {k: '%.1e' % v for k, v in err['f32'].items()})
    print('f16x3', {k: '%.1e' % v for k, v in err['f16x3'].items()})
    assert np.array_equal(outs['f16x3'][2], outs['f32'][2]) and np.array_equal(outs['f16x3'][3], outs['f32'][3])
    for k in ('cost', 'g0', 'g1', 'g2'):
        assert err['f16x3'][k] <= 4 * err['f32'][k] + 2e-9, (k, err['f16x3'][k], err['f32'][k])
    assert err['f16x3']['cost'] < 1e-5 and max(err['f16x3'][k] for k in ('g0', 'g1', 'g2')) < 1e-3


def test_f16x3_mode_on_two_row_shards():
    """The exchange buffers do not know the compute mode: two in-process ranks in f16x3 mode (20000 rows each, K = 576, sums done by the
    test as an all-reduce would) against one f16x3 context on all 40000 rows -- equal up to the fp32 accumulation order (other chunk
    boundaries) -- and against fp64 mode within fp32 mode's tier; every rank returns the same numbers."""
    import torch
    from scfgp_amd import synth
    from scfgp_amd.engine import HipEngine
    from scfgp_amd.sharded import shard_rows
    N, D, S, M = 40000, 8, 32, 256
    seed = 0x5CF60D00
    X = synth.make_X(seed, N, D); y = synth.normal(seed + 1, 0, N).reshape(-1, 1)
    params = synth.make_params(seed + 2, D, S, M, abc=(-1.0, 0.0, -1.0))
    stream = torch.cuda.current_stream().cuda_stream
    ref = {}
    for dtype in ('f64', 'f16x3'):
        e = HipEngine(D, S, M, dtype, stream=stream)
        if dtype != 'f64':
            e.set_option('gram64', 0); e.set_option('apply_dma', 2)
        e.set_params(params); e.set_data(X, y)
        ref[dtype] = e.eval()
        e.close()
    engs = []
    for r in range(2):
        lo, hi = shard_rows(N, r, 2)
        e = HipEngine(D, S, M, 'f16x3', stream=stream)
        e.set_option('gram64', 0); e.set_option('apply_dma', 2)
        e.set_params(params); e.set_data(X[lo:hi], y[lo:hi], n_global=N)
        e.set_profiling(True)
        engs.append(e)
    outs = _sharded_eval(engs)
    assert 'split_v' in [n for n, _ in engs[1].timings()]
    (c0, g0, a0, L0), (c1, g1, a1, L1) = outs
    assert float(c0) == float(c1) and np.array_equal(g0, g1) and np.array_equal(a0, a1) and np.array_equal(L0, L1)
    c16, g16, a16, L16 = ref['f16x3']
    assert abs(float(c0) - float(c16)) < 1e-7 * max(1.0, abs(float(c16))) and rel(a0, a16) < 1e-4 and rel(L0, L16) < 1e-4
    c64, g64, a64, L64 = ref['f64']
    assert abs(float(c0) - float(c64)) < 1e-5 * max(1.0, abs(float(c64)))
    for u, v in zip(_blocks(g0, D, S, M), _blocks(g64, D, S, M)):
        assert rel(u, v) < 1e-3
    for e in engs: e.close()
