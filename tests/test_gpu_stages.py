"""
GPU parity, stage by stage: every intermediate of the HIP path (through the C ABI's
debug_read) against the CPU oracle's staged restatement on the same seeded inputs.
Tolerances are relative, norm-wise; fp64 mode is expected at ~1e-12.
"""
import numpy as np
import pytest

from oracle import scfgp_oracle as O
from tests.golden.make_oracle_kats import CASES, case_inputs

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


@pytest.mark.parametrize('name', ['tiny_257x5', 'kin8nm_like', 'c1_boston_shape'])
@pytest.mark.parametrize('dtype,tol', [('f64', 1e-9), ('f32', 2e-4)])
def test_stages_match_oracle(name, dtype, tol):
    from scfgp_amd.engine import HipEngine
    N, D, S, M, T, seed = CASES[name]
    X, y, params, Xs = case_inputs(name)
    J = S + M; K = 2 * J
    ora = O.OracleEngine(D, S, M); ora.set_params(params); ora.set_data(X, y)
    eng = HipEngine(D, S, M, dtype=dtype)
    eng.set_option('gram64', 0)          # the fp32 kernels themselves are under test: no escalation of the Gram products
    eng.set_params(params); eng.set_data(X, y)
    d = eng.dims(); Kp, Jp, Dp, Np = d['Kp'], d['Jp'], d['Dp'], d['Np']
    tdt = np.float64 if dtype == 'f64' else np.float32
    Dpp = -(-Dp // 128) * 128

    # unpack
    Fall = eng.debug_read('Fall', (Dp, Jp))
    a, b, c, l_F, r_F, F, l_FC, FC = O.unpack_params(params, D, S, M)
    assert rel(Fall[:D, :J], np.concatenate((l_F, F), 1)) < 1e-14
    assert rel(Fall[D, :J], np.concatenate((l_FC, FC), 1).ravel()) < 1e-13
    assert np.all(Fall[D + 1:] == 0) and np.all(Fall[:, J:] == 0)

    # sweep 1
    eng.pass1(); ora.pass1()
    Phi = eng.debug_read('Phi', (Np, Kp), tdt).astype(np.float64)
    assert rel(Phi[:N, :K], ora.Ph) < (1e-12 if dtype == 'f64' else 1e-6)
    assert np.all(Phi[N:] == 0) and np.all(Phi[:, K:] == 0)          # padding rows / columns are exact zeros
    x1 = eng.debug_read('G', (Kp * Kp + Kp + 8,))
    G = x1[:Kp * Kp].reshape(Kp, Kp)
    assert rel(G[:K, :K], ora.x1[:K * K].reshape(K, K)) < tol
    assert np.all(G[K:] == 0) and np.all(G[:, K:] == 0)
    assert rel(x1[Kp * Kp:Kp * Kp + K], ora.x1[K * K:K * K + K]) < tol
    assert abs(x1[Kp * Kp + Kp] - ora.x1[K * K + K]) < 1e-12 * abs(ora.x1[K * K + K])

    # K-stage 1
    eng.factor(); ora.factor()
    Li = eng.debug_read('Li', (Kp, Kp)); B = eng.debug_read('B', (Kp, Kp))
    vecs = eng.debug_read('vecs', (5, Kp))
    ctol = tol * 1e3          # conditioning of A enters here
    assert rel(Li[:K, :K], ora.Li) < ctol
    assert np.all(np.triu(Li, 1) == 0)
    assert rel(Li[K:, K:], np.eye(Kp - K)) < 1e-14
    assert rel(B[:K, :K], ora.B) < ctol
    assert rel(vecs[1, :K], ora.alpha) < ctol

    # sweep 2
    eng.pass2(True); ora.pass2(True)
    p = eng.debug_read('p', (Np,)); q = eng.debug_read('q', (Np,))
    assert rel(p[:N], ora.p) < ctol and rel(q[:N], ora.q) < ctol
    assert np.all(p[N:] == 0) and np.all(q[N:] == 0)
    Phi2 = eng.debug_read('Phi', (Np, Kp), tdt).astype(np.float64)
    assert np.array_equal(Phi2, Phi)                                   # the sweeps never write Phi
    x2 = eng.debug_read('W', (Kp * Kp + Kp + 8,))
    W = x2[:Kp * Kp].reshape(Kp, Kp)
    assert rel(W[:K, :K], ora.x2[:K * K].reshape(K, K)) < ctol
    assert rel(x2[Kp * Kp:Kp * Kp + K], ora.x2[K * K:K * K + K]) < ctol
    assert rel(x2[Kp * Kp + Kp:Kp * Kp + Kp + 2], ora.x2[K * K + K:K * K + K + 2]) < ctol

    # K-stage 2
    eng.adjoint(); ora.adjoint()
    Abar = eng.debug_read('Abar', (Kp, Kp))
    assert rel(Abar[:K, :K], ora.Abar) < ctol

    # sweep 3
    eng.pass3(); ora.pass3()
    x3 = eng.debug_read('XZ', (Dpp * Jp + 8,))
    XZ = x3[:Dpp * Jp].reshape(Dpp, Jp)
    assert rel(XZ[:D, :J], ora.x3[:D * J].reshape(D, J)) < ctol
    # bbar = sum Phibar o Phi: the oracle forms the literal row sum (its exchange 3 carries it); the library's closed form
    # 2 tr(Abar G) + ut^T Phi^T y + 2 sum q v + sum p mu needs neither matrix and appears in d_scalars after finish
    bbar_o = ora.x3[D * J + J]
    assert np.all(XZ[Dp:] == 0) and np.all(XZ[:, J:] == 0)

    cost, grad, alpha, Li_h = eng.finish(True)
    c_o, g_o, al_o, Li_o = ora.finish(True)
    print('\n%s %s: cost %.3e grad %.3e alpha %.3e Li %.3e' % (
        name, dtype, abs(cost - c_o) / abs(c_o), rel(grad, g_o), rel(alpha, al_o), rel(Li_h, Li_o)))
    assert abs(cost - c_o) < (1e-10 if dtype == 'f64' else 1e-5) * max(1.0, abs(c_o))
    assert rel(grad, g_o) < ctol
    assert rel(alpha, al_o) < ctol and rel(Li_h, Li_o) < ctol
    sc = eng.debug_read('scalars', (32,))
    assert abs(sc[4] - bbar_o) < (1e-10 if dtype == 'f64' else ctol) * max(1.0, abs(bbar_o)), (sc[4], bbar_o)     # R_BBAR
    assert abs(grad[1] - g_o[1]) < (1e-10 if dtype == 'f64' else ctol) * max(1.0, abs(g_o[1]))
    eng.close()


# (D, S, M) -> live rows of the phase contraction (S + 1 through the rank-S form when that is narrower, else D + 1) -> kernel
FMAP_SHAPES = [
    (8, 20, 40, '9 live rows: register kernel, 3 k-steps'),
    (13, 40, 27, '14: register kernel, 4 k-steps; odd J (scalar stores of the sine half)'),
    (40, 17, 130, '18 (rank-S form): register kernel, 5 k-steps; odd J'),
    (60, 30, 70, '31 (rank-S form): register kernel, 8 k-steps'),
    (64, 32, 96, '33 (rank-S form): register kernel, 9 k-steps (the headline depth)'),
    (90, 60, 33, '61 (rank-S form): LDS kernel, segmented k-tile stream; odd J'),
    (70, 80, 50, '71 live rows of X~ itself: LDS kernel, five k-tiles'),
]


@pytest.mark.parametrize('D,S,M,what', FMAP_SHAPES)
@pytest.mark.parametrize('dtype,tol', [('f64', 1e-12), ('f32', 1e-6)])
def test_feature_map_kernel_variants(D, S, M, what, dtype, tol):
    """Every dispatch of the feature map (register kernel at each instantiated depth, LDS kernel, rank-S and direct
    projection, even and odd J, N not a multiple of the 128-row block) against the oracle's Phi; padding stays zero."""
    from scfgp_amd.engine import HipEngine
    from scfgp_amd import synth
    N = 1000 + D                                               # never a multiple of 128
    seed = 0x5CF64000 + 131 * D + S
    X = synth.make_X(seed, N, D)
    y = synth.normal(seed + 9, 0, N).reshape(-1, 1)
    params = synth.make_params(seed + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
    J = S + M; K = 2 * J
    eng = HipEngine(D, S, M, dtype=dtype); eng.set_params(params); eng.set_data(X, y)
    eng.pass1()
    d = eng.dims(); Kp, Np = d['Kp'], d['Np']
    Phi = eng.debug_read('Phi', (Np, Kp), np.float64 if dtype == 'f64' else np.float32).astype(np.float64)
    Phi0 = O.feature_map(X, params, D, S, M)
    assert rel(Phi[:N, :K], Phi0) < tol, (what, rel(Phi[:N, :K], Phi0))
    assert np.all(Phi[N:] == 0) and np.all(Phi[:, K:] == 0)
    eng.close()
