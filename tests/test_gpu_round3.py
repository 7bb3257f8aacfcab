"""
Round-3 GPU tests: the condition estimate of A = Phi^T Phi + lam I that the K x K stage reports, the automatic precision
escalation of the two Gram products in fp32 mode (include/scfgp_hip.h: scfgp_get_condition, option "gram64", SCFGP_REDO),
and fp32 mode against fp64 mode at the FULL size of the ill-conditioned BASELINE config C3 (and of C5).
The reference computes in float64 whatever the conditioning (SCFGP/SCFGP.py:95-96,104-110).
"""
import os
import numpy as np
import pytest

from oracle import scfgp_oracle as O
from scfgp_amd import synth

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


def grad_blocks(g, D, S, M):
    o = 3 + D * S
    return g[:3], g[3:o], g[o:o + M * S]


def _c3_like(N, D=8, S=32, M=256, seed=0x5CF63300):
    """kin8nm-like: few input dimensions, many frequencies => strongly correlated feature columns, A ill-conditioned"""
    X = synth.make_X(seed, N, D)
    params = synth.make_params(seed + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
    teacher = synth.make_params(seed + 0x0101, D, S, M, abc=(-1.0, 0.0, -1.0))
    Phi = O.feature_map(X, teacher, D, S, M)
    y = Phi @ synth.teacher_weights(seed + 0x0303, 2 * (S + M)) + 0.1 * synth.normal(seed + 0x0404, 0, N)
    y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    return X, y, params


def test_condition_estimate_is_a_lower_bound_and_drives_the_level():
    """cond_est = max L_ii^2 * max_j (A^-1)_jj <= cond_2(A) (numpy, oracle's A) and >= the diagonal ratio; in auto mode an
    ill-conditioned problem runs at level >= 1 and then alpha / Li equal fp64 mode's; with gram64 = 0 the same call
    returns rc 0 with alpha visibly off and says so through the estimate."""
    from scfgp_amd.engine import HipEngine
    N, D, S, M = 6000, 8, 32, 256
    X, y, params = _c3_like(N, D, S, M)
    K = 2 * (S + M)
    c0, g0, a0, L0 = O.value_and_grad(X, y, params, S, M)
    Phi = O.feature_map(X, params, D, S, M)
    w = np.linalg.eigvalsh(Phi.T @ Phi + (np.exp(2 * params[0]) + 1e-6) * np.eye(K))
    e64 = HipEngine(D, S, M, 'f64'); e64.set_params(params); e64.set_data(X, y)
    c64, g64, a64, L64 = e64.eval()
    cd64 = e64.condition()
    assert cd64['gram_fp64'] == 1.0 and cd64['level'] == 0.0
    assert cd64['cond_est'] <= w[-1] / w[0] * (1 + 1e-9)
    assert cd64['cond_est'] >= cd64['Lmax2'] / cd64['Lmin2'] * (1 - 1e-12)
    assert cd64['cond_est'] > 30, cd64                             # the case is meant to be ill-conditioned
    # auto (default): escalated, alpha and Li are the fp64 engine's
    e32 = HipEngine(D, S, M, 'f32'); e32.set_params(params); e32.set_data(X, y)
    c, g, a, L = e32.eval()
    cd = e32.condition()
    assert cd['level'] >= 1 and cd['gram_fp64'] == 1.0
    assert np.array_equal(a, a64) and np.array_equal(L, L64)
    assert abs(float(c) - float(c64)) < 1e-7 * max(1.0, abs(float(c64)))
    assert rel(a, a0) < 1e-7 and rel(L, L0) < 1e-8
    for u, v in zip(grad_blocks(g, D, S, M), grad_blocks(g64, D, S, M)):
        assert rel(u, v) < 1e-3
    c2, g2, a2, L2 = e32.eval()                                     # sticky: no repeat, same numbers
    assert float(c2) == float(c) and np.array_equal(g2, g) and np.array_equal(a2, a)
    # forward-only evaluation (train_func): level 1 is enough and is what it reports
    cf, _, af, _ = e32.eval(want_grad=False)
    assert np.array_equal(af, a64) and abs(float(cf) - float(c64)) < 1e-7 * max(1.0, abs(float(c64)))
    # plain fp32, on request: rc 0, alpha off by about err_per_cond * estimate, and the caller is told
    e32.set_option('gram64', 0)
    c3, g3, a3, L3 = e32.eval()
    cd0 = e32.condition()
    assert cd0['level'] == 0.0 and cd0['gram_fp64'] == 0.0
    err = rel(a3, a64)
    assert err > 1e-6 and 0.1 * cd0['alpha_err_fp32'] < err < 10 * cd0['alpha_err_fp32'], (err, cd0)
    assert abs(cd0['cond_est'] / cd64['cond_est'] - 1) < 1e-2
    # always-on levels
    for opt, lvl in ((1, 1), (3, 2)):
        e32.set_option('gram64', opt)
        c4, g4, a4, L4 = e32.eval()
        assert e32.condition()['level'] == lvl and np.array_equal(a4, a64)
    for u, v in zip(grad_blocks(g4, D, S, M), grad_blocks(g64, D, S, M)):      # level 2: the gradient as well
        assert rel(u, v) < 2e-4
    e32.close(); e64.close()


def test_well_conditioned_problem_stays_at_level_0():
    from scfgp_amd.engine import HipEngine
    N, D, S, M = 4000, 32, 16, 112
    seed = 0x5CF63400
    X = synth.make_X(seed, N, D)
    y = synth.normal(seed + 9, 0, N).reshape(-1, 1)
    params = synth.make_params(seed + 0x0202, D, S, M, abc=(1.0, 0.0, -1.0))       # lam = e^2: A is well conditioned
    eng = HipEngine(D, S, M, 'f32'); eng.set_params(params); eng.set_data(X, y)
    eng.set_profiling(True)
    c, g, a, L = eng.eval()
    cd = eng.condition()
    names = [n for n, _ in eng.timings()]
    assert cd['level'] == 0.0 and cd['gram_fp64'] == 0.0 and cd['cond_est'] < cd['threshold']
    assert 'gram' in names and 'gram64' not in names
    c0, g0, a0, L0 = O.value_and_grad(X, y, params, S, M)
    assert rel(a, a0) < 1e-5 and rel(L, L0) < 1e-5
    eng.close()


def test_staged_api_redo_and_two_shards_decide_alike():
    """scfgp_finish returns SCFGP_REDO (engine.finish() -> None) once when the level rises; two row shards on one GPU, summed
    by hand, take the same decision at the same time and end on the single-context result."""
    import torch
    from scfgp_amd.engine import HipEngine
    from scfgp_amd.sharded import ShardedEvaluator, shard_rows
    N, D, S, M = 5000, 8, 32, 256
    X, y, params = _c3_like(N, D, S, M, seed=0x5CF63500)
    single = HipEngine(D, S, M, 'f32'); single.set_params(params); single.set_data(X, y)
    single.pass1(); single.factor(); single.pass2(True); single.adjoint(); single.pass3()
    assert single.finish(True) is None                              # raised to the level its estimate asks for
    lvl = None
    for _ in range(2):
        single.pass1(); single.factor(); single.pass2(True); single.adjoint(); single.pass3()
        out = single.finish(True)
        if out is not None:
            break
    assert out is not None
    c0, g0, a0, L0 = out
    lvl = single.condition()['level']
    assert lvl >= 1
    stream = torch.cuda.current_stream().cuda_stream
    engs = []
    for r in range(2):
        lo, hi = shard_rows(N, r, 2)
        e = HipEngine(D, S, M, 'f32', stream=stream)
        e.set_params(params); e.set_data(np.ascontiguousarray(X[lo:hi]), np.ascontiguousarray(y[lo:hi]), n_global=N)
        engs.append(e)

    def allsum(stage):
        bufs = [e.exchange(stage) for e in engs]
        tot = bufs[0] + bufs[1]
        for b in bufs:
            b.copy_(tot)

    rounds = 0
    while True:
        rounds += 1
        for e in engs: e.pass1()
        allsum(1)
        for e in engs: e.factor()
        for e in engs: e.pass2(True)
        allsum(2)
        for e in engs: e.adjoint()
        for e in engs: e.pass3()
        allsum(3)
        outs = [e.finish(True) for e in engs]
        assert (outs[0] is None) == (outs[1] is None)
        if outs[0] is not None:
            break
        assert rounds < 4
    assert rounds >= 2
    for c, g, a, L in outs:
        assert abs(float(c) - float(c0)) < 1e-9 * max(1.0, abs(float(c0)))
        assert rel(a, a0) < 1e-9 and rel(L, L0) < 1e-10 and rel(g, g0) < 1e-4
    assert [e.condition()['level'] for e in engs] == [lvl, lvl]
    # the evaluator hides the repeat
    fresh = HipEngine(D, S, M, 'f32'); fresh.set_params(params); fresh.set_data(X, y)
    c1, g1, a1, L1 = ShardedEvaluator(fresh, None).eval(True)
    assert float(c1) == float(c0) and np.array_equal(a1, a0)
    for e in engs + [single, fresh]:
        e.close()


def test_device_training_probes_the_level_first():
    """scfgp_train cannot repeat an iteration (the update is applied on the device), so with the condition unknown it probes
    it once (pass 1 + factor) and runs the whole call at that level: the fp32 device-rule trajectory equals the fp32
    host-rule trajectory, whose evaluations escalate through SCFGP_REDO."""
    from scfgp_amd.funcs import CompiledFuncs
    N, D, S, M = 3000, 8, 16, 96
    X, y, params = _c3_like(N, D, S, M, seed=0x5CF63600)
    kw = {'learning_rate': 0.01, 'beta1': 0.9, 'beta2': 0.999, 'epsilon': 1e-8}
    host = CompiledFuncs(D, S, M, params.copy(), 'adam', kw, dtype='f32')
    dev = CompiledFuncs(D, S, M, params.copy(), 'adam', kw, dtype='f32', device_optimizer=True)
    n = 5
    costs_h = [float(host.train_iter_func(X, y)[0]) for _ in range(n)]
    assert host.engine.condition()['level'] >= 1
    hist, alpha, Li = dev.train_iters(X, y, n)
    assert dev.engine.condition()['level'] == host.engine.condition()['level']
    assert np.allclose(hist, costs_h, rtol=1e-7, atol=0)


@pytest.mark.parametrize('cfg', ['C3', 'C5'])
def test_fp32_mode_against_fp64_mode_at_full_size(cfg):
    """BASELINE configs C3 (D = 8: A ill-conditioned, estimate ~6e3, cond_2 ~2e6) and C5 at their full 1e6 rows: fp32 mode
    as shipped (auto precision level) against fp64 mode of the same library on the same rows (fp64 mode equals the oracle
    to 1e-12 wherever the oracle can run).  Measured in round 3 (profiles/r03_c3_owner.md), bounds 5-10x that:
      C3 level 2: cost 4e-9, grad (abc, l_F, r_F) see the bounds, alpha = Li = 0 (bit-equal), mu* 5e-7, sigma* 6e-10
      C5 level 0: cost 4e-11, grad 7e-11 / 2.3e-6 / 2.4e-6, alpha 4e-7, Li 6e-8, mu* 4e-7, sigma* 6e-12
    Plain fp32 at C3 (gram64 = 0) is asserted to REPORT its condition, not to be accurate: alpha 1.9e-3 there."""
    import bench
    from scfgp_amd.engine import HipEngine
    N, D, S, M = bench.CONFIGS[cfg][:4]
    e32 = HipEngine(D, S, M, 'f32')
    X, y, params = bench.build_problem(e32, N, D, S, M, 0, N, None)
    e64 = HipEngine(D, S, M, 'f64')
    for e in (e32, e64):
        e.set_params(params); e.set_data(X, y)
    out32 = e32.eval(); out64 = e64.eval()
    cd = e32.condition()
    par = bench._parity(out32, out64, e32, e64, D, S, M, cfg)
    print('\n%s fp32 (level %d, cond_est %.3g) vs fp64: ' % (cfg, cd['level'], cd['cond_est'])
          + ' '.join('%s %.2e' % (k, v) for k, v in par.items() if k != 'what'))
    if cfg == 'C3':
        assert cd['level'] == 2 and cd['cond_est'] > 1e3
        bounds = dict(cost=1e-7, grad_abc=1e-4, grad_lF=1e-4, grad_rF=5e-4, alpha=1e-9, Li=1e-9, mu=5e-6, std=1e-8)
    else:
        assert cd['level'] == 0 and cd['cond_est'] < 10
        bounds = dict(cost=5e-10, grad_abc=1e-9, grad_lF=3e-5, grad_rF=3e-5, alpha=5e-6, Li=1e-6, mu=5e-6, std=1e-10)
    for k, b in bounds.items():
        assert par[k] < b, (k, par[k], b)
    assert all(par[k] < 1e-5 for k in ('cost', 'alpha', 'Li', 'mu', 'std'))      # the north star's outputs at its tolerance
    if cfg == 'C3':
        e32.set_option('gram64', 0)
        o0 = e32.eval()
        cd0 = e32.condition()
        assert cd0['level'] == 0 and cd0['alpha_err_fp32'] > 1e-4 and rel(o0[2], out64[2]) > 1e-4
    e32.close(); e64.close()


@pytest.mark.parametrize('dtype,dma,ctol,gtol', [('f64', -1, 1e-10, 1e-8), ('f64', 1, 1e-10, 1e-8), ('f32', 0, 2e-5, 2e-3), ('f32', 2, 2e-5, 2e-3), ('f32', 1, 2e-5, 2e-3)])
def test_factor_form_of_pass2_matches_oracle(dtype, dma, ctol, gtol):
    """Option factor_form = 1: pass 2 as the reference writes it (SCFGP/SCFGP.py:112) -- C = Phi Li^T (triangular), v = rowsum(C^2),
    V = C Li (triangular), B W B = Li^T (C^T diag(q) C) Li, u = Li^T C^T p -- against the oracle: loader-staged tiles (f64, f32)
    and the LDS-DMA tiles (256- and 128-wide) with their partial k ranges; ragged K (64-wide remainder tile)."""
    from scfgp_amd.engine import HipEngine
    for N, D, S, M in ((2100, 8, 5, 155), (1300, 20, 24, 282)):        # K = 320 (2 x 128 + 64), 612 (4 x 128 + 64 + ragged)
        seed = 0x5CF63700 + N
        X = synth.make_X(seed, N, D)
        y = synth.normal(seed + 9, 0, N).reshape(-1, 1)
        params = synth.make_params(seed + 0x0202, D, S, M, abc=(-0.5, 0.0, -1.0))
        c0, g0, a0, L0 = O.value_and_grad(X, y, params, S, M)
        ref = None
        for ff in (0, 1):
            eng = HipEngine(D, S, M, dtype); eng.set_option('gram64', 0); eng.set_option('factor_form', ff)
            if dma >= 0:
                eng.set_option('apply_dma', dma)
            eng.set_params(params); eng.set_data(X, y)
            eng.set_profiling(True)
            c, g, a, L = eng.eval()
            names = [n for n, _ in eng.timings()]
            assert ('apply_c' in names and 'apply_vc' in names) == bool(ff) and ('apply_v' in names) == (not ff)
            assert abs(float(c) - c0) < ctol * max(1.0, abs(c0)), (ff, float(c), c0)
            for u, v in zip(grad_blocks(g, D, S, M), grad_blocks(g0, D, S, M)):
                assert rel(u, v) < gtol, (N, ff, rel(u, v))
            p = eng.debug_read('p', (N,)); V = eng.debug_read('V', (eng.dims()['Np'], eng.dims()['Kp']),
                                                               np.float64 if dtype == 'f64' else np.float32)
            if ref is None:
                ref = (p.copy(), V.copy())
            else:                                                     # same per-row adjoint scalars and the same V = Phi B
                assert rel(p, ref[0]) < (1e-9 if dtype == 'f64' else 1e-3)
                assert rel(V[:N], ref[1][:N]) < (1e-9 if dtype == 'f64' else 1e-3)
                assert np.all(V[N:] == 0) and np.all(V[:, 2 * (S + M):] == 0)
            c2, g2, _, _ = eng.eval()
            assert float(c2) == float(c) and np.array_equal(g2, g)
            eng.close()


def test_checkpoint_keeps_a_non_default_momentum_and_numpy_scalar_kwargs(tmp_path):
    """save() wrote opt_momentum but load() ignored it, and numpy-scalar keyword arguments broke json.dumps.
    A triple built with momentum 0.5 and learning_rate = np.float32(0.02): 3 iterations + save + load + 3 == 6."""
    import os
    from scfgp_amd import SCFGP
    rng = np.random.default_rng(31)
    X = rng.uniform(-2, 2, (200, 3))
    y = np.sin(X[:, :1]) + 0.3 * X[:, 1:2] + 0.05 * rng.standard_normal((200, 1))
    kw = {'learning_rate': np.float32(0.02), 'beta2': np.float64(0.999)}

    def fresh():
        np.random.seed(5)
        m = SCFGP(sparsity=3, nfeats=8)
        m.set_data(X, y)
        m.build_hip_models('adam', kw, momentum=0.5)
        return m

    def iters(m, n):
        for _ in range(n):
            c, a, L = m.train_iter_func(m.X, m.y)
        m.alpha, m.Li = a, L

    full = fresh(); iters(full, 6)
    first = fresh(); iters(first, 3)
    path = os.path.join(str(tmp_path), 'mom.npz')
    first.save(path)
    second = SCFGP(sparsity=1, nfeats=1); second.set_data(X, y); second.load(path)
    assert second._compiled.momentum == 0.5 and abs(second._compiled.algo_params['learning_rate'] - 0.02) < 1e-8
    iters(second, 3)
    assert np.array_equal(second.params.get_value(), full.params.get_value())
    default = fresh(); default.build_hip_models('adam', kw); iters(default, 6)
    assert not np.array_equal(default.params.get_value(), full.params.get_value())     # the momentum does matter


@pytest.mark.parametrize('dtype,gtol', [('f64', 1e-9), ('f32', 1e-3)])
def test_rank_s_backward_projection_matches_the_dense_one(dtype, gtol):
    """Option lowrank_bwd (automatic when D >> S): the reverse sweep of F = l_F r_F^T (SCFGP/SCFGP.py:83,100) through
    T~^T Zbar and X~^T (Zbar_L + Zbar_M r_F) instead of the dense X~^T Zbar.  Same gradient (all blocks, phases included) as
    the dense form and the oracle; odd J (scalar Zbar path), J a multiple of 4 (vector path), S + 1 > 64 (two tiles of T~)."""
    from scfgp_amd.engine import HipEngine
    for N, D, S, M in ((1500, 200, 8, 232), (1100, 150, 5, 200), (900, 300, 70, 250)):
        seed = 0x5CF63A00 + D
        X = synth.make_X(seed, N, D)
        y = synth.normal(seed + 9, 0, N).reshape(-1, 1)
        params = synth.make_params(seed + 0x0202, D, S, M, abc=(0.5, 0.0, -1.0))
        params[3:3 + D * S] *= 0.3                                 # |Z| stays moderate at D = 300
        c0, g0, a0, L0 = O.value_and_grad(X, y, params, S, M)
        outs = {}
        for lrb in (0, 1):
            eng = HipEngine(D, S, M, dtype); eng.set_option('gram64', 0); eng.set_option('lowrank_bwd', lrb)
            eng.set_params(params); eng.set_data(X, y)
            c, g, a, L = eng.eval()
            c2, g2, _, _ = eng.eval()
            assert float(c2) == float(c) and np.array_equal(g2, g)
            eng.opt_init('adam', learning_rate=0.005)
            hist, _, _ = eng.train(3)                              # the same pass 3 inside the captured training iteration
            outs[lrb] = (float(c), g, hist)
            assert abs(float(c) - c0) < (1e-10 if dtype == 'f64' else 2e-5) * max(1.0, abs(c0))
            for u, v in zip(grad_blocks(g, D, S, M), grad_blocks(g0, D, S, M)):
                assert rel(u, v) < gtol, (lrb, D, rel(u, v))
            J = S + M
            assert np.abs(g[-J:]).max() < (1e-9 if dtype == 'f64' else 1e-3) * np.abs(g).max()      # phases: zero gradient
            eng.close()
        assert outs[0][0] == outs[1][0]
        assert rel(outs[1][1], outs[0][1]) < (1e-11 if dtype == 'f64' else 1e-4)
        assert np.allclose(outs[1][2], outs[0][2], rtol=1e-10 if dtype == 'f64' else 1e-5, atol=0)


def test_level_drops_again_and_a_failed_fp32_cholesky_is_retried_in_fp64():
    """The automatic level is sticky with hysteresis: after the parameters move to a well-conditioned point the next
    evaluation still runs escalated, the one after it at level 0.  And when the fp32 Gram's error exceeds the jitter
    (lam = e^{2a} + 1e-6 tiny, duplicated frequencies: A numerically singular) the Cholesky of the fp32 Gram fails; the library
    retries on the fp64 Gram before it reports anything, and only fp64's own failure surfaces as LinAlgError."""
    from scfgp_amd.engine import HipEngine
    N, D, S, M = 5000, 8, 32, 256
    X, y, params = _c3_like(N, D, S, M, seed=0x5CF63B00)
    eng = HipEngine(D, S, M, 'f32'); eng.set_params(params); eng.set_data(X, y)
    eng.eval()
    assert eng.condition()['level'] >= 1
    easy = params.copy(); easy[0] = 1.5                            # lam = e^3
    eng.set_params(easy)
    c1, g1, a1, L1 = eng.eval()
    cd1 = eng.condition()
    assert cd1['level'] >= 1 and cd1['cond_est'] < cd1['threshold'] / 4       # ran escalated, asks for less
    c2, g2, a2, L2 = eng.eval()
    assert eng.condition()['level'] == 0
    assert abs(float(c2) - float(c1)) < 1e-6 * max(1.0, abs(float(c1))) and rel(a2, a1) < 1e-4
    eng.close()
    # numerically singular A: two identical blocks of frequencies and a tiny jitter
    N, D, S, M = 6000, 6, 4, 124
    seed = 0x5CF63C00
    X = synth.make_X(seed, N, D)
    y = synth.normal(seed + 9, 0, N).reshape(-1, 1)
    p = synth.make_params(seed + 0x0202, D, S, M, abc=(-5.5, 0.0, -1.0))      # lam = e^-11 + 1e-6 = 1.8e-5
    o = 3 + D * S
    rf = p[o:o + M * S].reshape(M, S); rf[M // 2:] = rf[:M // 2]              # F's columns come in identical pairs ...
    ph = p[o + M * S + S:]; ph[M // 2:] = ph[:M // 2]                         # ... with identical phases
    c0, g0, a0, L0 = O.value_and_grad(X, y, p, S, M)                          # float64 copes (lam I keeps A positive definite)
    e64 = HipEngine(D, S, M, 'f64'); e64.set_params(p); e64.set_data(X, y)
    c64 = float(e64.eval()[0]); e64.close()
    assert abs(c64 - c0) < 1e-6 * max(1.0, abs(c0))
    e32 = HipEngine(D, S, M, 'f32'); e32.set_params(p); e32.set_data(X, y)
    c, g, a, L = e32.eval()                                                   # whatever the fp32 Gram did, the answer is fp64's
    assert e32.condition()['level'] >= 1 and abs(float(c) - c64) < 1e-6 * max(1.0, abs(c64))
    e32.set_option('gram64', 0)                                               # plain fp32 on request: fails loudly or is visibly off
    try:
        cp = float(e32.eval()[0])
        assert e32.condition()['alpha_err_fp32'] > 1e-2 or abs(cp - c64) < 1e-2 * max(1.0, abs(c64))
    except np.linalg.LinAlgError:
        pass
    e32.close()


def test_one_rank_under_the_launcher_goes_through_rccl():
    """The driver's multi-GPU command line with one rank (`python -m torch.distributed.run --nproc-per-node 1 ... bench.py`):
    the rank creates an RCCL communicator and the three sums run as in-place `all_reduce` calls on torch tensors aliasing the
    library's exchange buffers, fenced against the library's stream -- the code path of N > 1 with the only thing this one-GPU
    box cannot supply, a second card.  Same fp64 cost as the plain single-process run, and the line carries the exchange
    stages."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'TORCHELASTIC_RUN_ID')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    args = ['--config', 'C2', '--rows', '20000', '--steps', '2', '--warmup', '1', '--no-cpu', '--no-secondary']
    one = subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + args, env=env, stdout=subprocess.PIPE,
                         universal_newlines=True, timeout=600)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '1'] + args
    rccl = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert one.returncode == 0 and rccl.returncode == 0
    o1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith('{')][-1])
    o2 = json.loads([ln for ln in rccl.stdout.splitlines() if ln.startswith('{')][-1])
    assert o2['n_gpus'] == 1 and o2['cost'] == o1['cost']          # a sum over one rank changes nothing, bit for bit
    st = o2['stages_ms']
    assert all(('exchange%d' % k) in st and st['exchange%d' % k] >= 0 for k in (1, 2, 3))
    assert 'exchange1' not in o1['stages_ms']


@pytest.mark.parametrize('dtype,opts,ctol,gtol', [
    ('f64', {'lowrank_bwd': 1}, 1e-10, 1e-8),
    ('f64', {'factor_form': 1}, 1e-10, 1e-8),
    ('f32', {'gram64': 3}, 2e-5, 3e-3),                              # level 2 forced: fp64 pass-1 Gram + factor form of pass 2
    ('f32', {'gram64': 0, 'lowrank_bwd': 1, 'factor_form': 1}, 2e-5, 3e-3),
])
def test_random_shapes_with_the_optional_paths_forced(dtype, opts, ctol, gtol):
    """The optional paths of this round (rank-S backward projection, factor form of pass 2, the escalated pass 1) forced on
    24 seeded random (N, D, S, M) -- ragged K, one-row problems, D < S and D >> S -- against the oracle's cost and gradient:
    every tile-edge combination of their kernels (triangular k ranges, T~ / U tile grids, fp64 Gram job list in an fp32 context)."""
    from scfgp_amd.engine import HipEngine
    rng0 = np.random.default_rng(20261005)
    for _ in range(24):
        N, D, S, M = (int(rng0.integers(1, 2500)), int(rng0.integers(1, 90)), int(rng0.integers(1, 40)), int(rng0.integers(1, 300)))
        rng = np.random.default_rng(N * 131 + D * 17 + S * 3 + M)
        X = rng.random((N, D)); y = rng.standard_normal((N, 1))
        params = O.init_params(D, S, M, rng)
        params[0] = -0.4; params[1] = 0.1; params[2] = -0.6; params[3:3 + D * S] *= 0.6
        eng = HipEngine(D, S, M, dtype)
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.set_params(params); eng.set_data(X, y)
        c0, g0, a0, L0 = O.value_and_grad(X, y, params, S, M)
        if not np.isfinite(c0):                                    # S = 1 or M = 1: a row's std is 0, -log 0 in the penalty (SCFGP.py:94,114-117)
            with pytest.raises(FloatingPointError):                # the engine reports it (SCFGP_ENONFINITE); the triple returns it
                eng.eval(want_grad=True)
            eng.close()
            continue
        cost, grad, alpha, Li = eng.eval(want_grad=True)
        assert abs(float(cost) - c0) < ctol * max(1.0, abs(c0)), (N, D, S, M, float(cost), c0)
        assert rel(grad, g0) < gtol, (N, D, S, M, rel(grad, g0))
        eng.close()


def test_non_finite_cost_comes_back_as_a_value_through_the_triple():
    """With S = 1 the penalty's `sig_l` (sum over rows of the std along a one-element axis, SCFGP/SCFGP.py:116) is 0 and
    kl's -log(sig) makes the cost +inf.  Theano's functions return that as a value, the reference's optimize loop reads a
    non-finite objective as "no improvement" (SCFGP/SCFGP.py:249-258) -- so do train_func / train_iter_func here, with alpha and
    Li as usual; the engine itself (the C ABI's SCFGP_ENONFINITE) raises."""
    from scfgp_amd.engine import HipEngine
    from scfgp_amd.funcs import CompiledFuncs
    N, D, S, M = 300, 6, 1, 20
    rng = np.random.default_rng(7)
    X = rng.random((N, D)); y = rng.standard_normal((N, 1))
    params = O.init_params(D, S, M, rng); params[:3] = (-0.5, 0.0, -1.0)
    c0, g0, a0, L0 = O.value_and_grad(X, y, params, S, M)
    assert np.isposinf(c0)
    cf = CompiledFuncs(D, S, M, params.copy())
    cost, alpha, Li = cf.train_func(X, y)
    assert np.isposinf(float(cost)) and rel(alpha, a0) < 1e-9 and rel(Li, L0) < 1e-9
    cost2, _, _ = cf.train_iter_func(X, y)                          # no exception either; the update then spreads the NaN gradient
    assert np.isposinf(float(cost2))
    eng = HipEngine(D, S, M); eng.set_params(params); eng.set_data(X, y)
    with pytest.raises(FloatingPointError):
        eng.eval()
    eng.close()


@pytest.mark.parametrize('dtype,ctol,gtol,ptol', [('f64', 1e-10, 1e-8, 1e-9), ('f32', 2e-5, 3e-3, 2e-4)])
@pytest.mark.parametrize('D,S,M', [(7, 3, 40), (40, 6, 150), (5, 12, 200)])
def test_one_context_through_a_random_sequence_of_calls(D, S, M, dtype, ctol, gtol, ptol):
    """Stale-state check: ONE context lives through 40 seeded random calls -- new rows (1 .. 3000, growing and shrinking), new
    parameters, evaluations with and without gradient, minibatch evaluations on index lists (duplicates, one row, all rows),
    predictions (1 .. 5000 rows), two device training iterations, and option changes between them (precision level, factor
    form, rank-S backward projection, LDS-DMA tiles, row splits) -- and every result is compared with the oracle on the
    inputs of that call.  Buffers that outlive a call (row buffers sized for an earlier, larger N; the factor-form and fp64
    feature buffers; slabs; the captured training graph) must never leak into a later one."""
    from scfgp_amd.engine import HipEngine
    rng = np.random.default_rng(D * 1000 + S * 10 + M + (0 if dtype == 'f64' else 7))
    eng = HipEngine(D, S, M, dtype)
    sizes = [1, 37, 255, 256, 257, 700, 1500, 3000]

    def new_params():
        p = O.init_params(D, S, M, rng)
        p[0] = -0.4 + 0.2 * rng.standard_normal(); p[1] = 0.1; p[2] = -0.6; p[3:3 + D * S] *= 0.6
        return p

    def new_data():
        n = int(rng.choice(sizes))
        return rng.random((n, D)), rng.standard_normal((n, 1))

    params = new_params(); X, y = new_data()
    eng.set_params(params); eng.set_data(X, y)
    alpha = Li = None
    options = [('gram64', (0, 2, 3)), ('factor_form', (-1, 1)), ('lowrank_bwd', (-1, 0, 1)), ('gram_nsplit', (0, 3)), ('gram_taper', (0, 1)),
               ('apply_dma', (-1, 0, 1, 2))]
    log = []
    for step in range(40):
        op = rng.choice(['data', 'params', 'eval', 'eval', 'rows', 'predict', 'option', 'train'])
        log.append(str(op))
        if op == 'data':
            X, y = new_data(); eng.set_data(X, y)
        elif op == 'params':
            params = new_params(); eng.set_params(params)
        elif op == 'option':
            name, values = options[int(rng.integers(len(options)))]
            v = int(rng.choice(values)); eng.set_option(name, v); log[-1] += ' %s=%d' % (name, v)
        elif op == 'eval':
            wg = bool(rng.integers(2))
            c, g, alpha, Li = eng.eval(want_grad=wg)
            c0, g0, a0, L0 = O.value_and_grad(X, y, params, S, M)
            assert abs(float(c) - c0) < ctol * max(1.0, abs(c0)), (step, log)
            assert rel(alpha, a0) < max(ptol, 0 if dtype == 'f64' else 1e-3), (step, log)
            if wg:
                assert rel(g, g0) < gtol, (step, log, rel(g, g0))
        elif op == 'rows':
            n = int(rng.choice([1, 10, max(1, X.shape[0] // 2), X.shape[0]]))
            idx = rng.integers(0, X.shape[0], n)
            c, g, alpha, Li = eng.eval_rows(idx, True)
            c0, g0, a0, L0 = O.value_and_grad(X[idx], y[idx], params, S, M)
            assert abs(float(c) - c0) < ctol * max(1.0, abs(c0)), (step, log)
            assert rel(g, g0) < gtol, (step, log, rel(g, g0))
        elif op == 'predict':
            if alpha is None:
                continue
            T = int(rng.choice([1, 100, 5000]))
            Xs = rng.random((T, D))
            mu, sd = eng.predict(Xs, alpha, Li)
            mu0, sd0 = O.predict(Xs, alpha, Li, params, S, M)
            assert mu.shape == (T, 1) and sd.shape == (T,)
            assert rel(mu, mu0) < ptol and rel(sd, sd0) < ptol, (step, log, rel(mu, mu0), rel(sd, sd0))
        elif op == 'train':
            eng.opt_init('adam', learning_rate=1e-3)
            hist, alpha, Li = eng.train(2)
            c0, g0, a0, L0 = O.value_and_grad(X, y, params, S, M)
            assert abs(hist[0] - c0) < ctol * max(1.0, abs(c0)), (step, log)       # first iteration: cost at the parameters it started from
            params = eng.get_params()
            c1, _, _, _ = O.value_and_grad(X, y, params, S, M)                     # ... and the vector the device moved to evaluates consistently
            c, _, alpha, Li = eng.eval(want_grad=False)
            assert abs(float(c) - c1) < ctol * max(1.0, abs(c1)), (step, log)
    eng.close()


@pytest.mark.parametrize('dtype,dma,ctol,gtol', [('f32', 1, 2e-5, 3e-3), ('f32', 2, 2e-5, 3e-3), ('f64', 1, 1e-10, 1e-8)])
def test_lds_dma_apply_tiles_on_random_widths(dtype, dma, ctol, gtol):
    """The LDS-DMA apply kernels (fp32: 128- and 256-wide tiles, fp64: 128-wide; the narrower remainder on the loader-staged
    kernel) forced on 12 seeded random shapes with K from ~130 to ~1500 -- ragged K, odd and even tile counts, row counts that
    are no multiple of 256 -- against the oracle, with and without the factor form (triangular k ranges)."""
    from scfgp_amd.engine import HipEngine
    rng0 = np.random.default_rng(20261006 + dma)
    for it in range(12):
        N, D, S, M = (int(rng0.integers(1, 4000)), int(rng0.integers(1, 40)), int(rng0.integers(2, 30)), int(rng0.integers(60, 720)))
        rng = np.random.default_rng(N * 131 + D * 17 + S * 3 + M)
        X = rng.random((N, D)); y = rng.standard_normal((N, 1))
        params = O.init_params(D, S, M, rng)
        params[0] = -0.4; params[1] = 0.1; params[2] = -0.6; params[3:3 + D * S] *= 0.6
        c0, g0, a0, L0 = O.value_and_grad(X, y, params, S, M)
        eng = HipEngine(D, S, M, dtype)
        eng.set_option('apply_dma', dma); eng.set_option('factor_form', it % 2)
        eng.set_params(params); eng.set_data(X, y)
        cost, grad, alpha, Li = eng.eval(want_grad=True)
        assert abs(float(cost) - c0) < ctol * max(1.0, abs(c0)), (N, D, S, M, float(cost), c0)
        assert rel(grad, g0) < gtol, (N, D, S, M, rel(grad, g0))
        eng.close()
