"""
Diagnostic script (not a pytest test): prints the relative error of every stage's
intermediates against the CPU oracle without stopping at the first mismatch, then
per-stage timings on a mid-size problem.  Usage on the GPU box:
    python tests/gpu_diag.py [case ...]
"""
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import scfgp_oracle as O                                  # noqa: E402
from scfgp_amd import synth                                           # noqa: E402
from scfgp_amd.engine import HipEngine                                # noqa: E402
from tests.golden.make_oracle_kats import CASES, case_inputs          # noqa: E402


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def stages(name, dtype):
    N, D, S, M, T, seed = CASES[name]
    X, y, params, Xs = case_inputs(name)
    J = S + M; K = 2 * J
    ora = O.OracleEngine(D, S, M); ora.set_params(params); ora.set_data(X, y)
    eng = HipEngine(D, S, M, dtype=dtype)
    eng.set_params(params); eng.set_data(X, y)
    d = eng.dims(); Kp, Jp, Dp, Np = d['Kp'], d['Jp'], d['Dp'], d['Np']
    tdt = np.float32 if dtype == 'f32' else np.float64
    Dpp = -(-Dp // 128) * 128
    out = []
    a, b, c, l_F, r_F, F, l_FC, FC = O.unpack_params(params, D, S, M)
    Fall = eng.debug_read('Fall', (Dp, Jp))
    out.append(('Fall', rel(Fall[:D, :J], np.concatenate((l_F, F), 1))))
    out.append(('offs', rel(Fall[D, :J], np.concatenate((l_FC, FC), 1).ravel())))
    eng.pass1(); ora.pass1()
    Phi = eng.debug_read('Phi', (Np, Kp), tdt).astype(np.float64)
    out.append(('Phi', rel(Phi[:N, :K], ora.Ph)))
    out.append(('Phi_pad', float(np.abs(Phi[N:]).max() if Np > N else 0) + float(np.abs(Phi[:, K:]).max())))
    x1 = eng.debug_read('G', (Kp * Kp + Kp + 8,))
    G = x1[:Kp * Kp].reshape(Kp, Kp)
    out.append(('G', rel(G[:K, :K], ora.x1[:K * K].reshape(K, K))))
    out.append(('g', rel(x1[Kp * Kp:Kp * Kp + K], ora.x1[K * K:K * K + K])))
    out.append(('yy', abs(x1[Kp * Kp + Kp] - ora.x1[K * K + K]) / abs(ora.x1[K * K + K])))
    eng.factor(); ora.factor()
    Li = eng.debug_read('Li', (Kp, Kp)); B = eng.debug_read('B', (Kp, Kp)); vecs = eng.debug_read('vecs', (5, Kp))
    out.append(('Li', rel(Li[:K, :K], ora.Li)))
    out.append(('Li_upper', float(np.abs(np.triu(Li, 1)).max())))
    out.append(('B', rel(B[:K, :K], ora.B)))
    out.append(('alpha', rel(vecs[1, :K], ora.alpha)))
    eng.pass2(True); ora.pass2(True)
    p = eng.debug_read('p', (Np,)); q = eng.debug_read('q', (Np,))
    out.append(('p', rel(p[:N], ora.p))); out.append(('q', rel(q[:N], ora.q)))
    x2 = eng.debug_read('W', (Kp * Kp + Kp + 8,))
    out.append(('W', rel(x2[:Kp * Kp].reshape(Kp, Kp)[:K, :K], ora.x2[:K * K].reshape(K, K))))
    out.append(('h', rel(x2[Kp * Kp:Kp * Kp + K], ora.x2[K * K:K * K + K])))
    out.append(('T2kb', rel(x2[Kp * Kp + Kp:Kp * Kp + Kp + 2], ora.x2[K * K + K:K * K + K + 2])))
    eng.adjoint(); ora.adjoint()
    Abar = eng.debug_read('Abar', (Kp, Kp))
    out.append(('Abar', rel(Abar[:K, :K], ora.Abar)))
    out.append(('ut', rel(vecs[3, :K] * 0 + eng.debug_read('vecs', (5, Kp))[3, :K], ora.ut)))
    eng.pass3(); ora.pass3()
    x3 = eng.debug_read('XZ', (Dpp * Jp + 8,))
    XZ = x3[:Dpp * Jp].reshape(Dpp, Jp)
    out.append(('XZ', rel(XZ[:D, :J], ora.x3[:D * J].reshape(D, J))))
    out.append(('colsumZ', float(np.abs(XZ[D, :J] - ora.x3[D * J:D * J + J]).max())))
    cost, grad, alpha, Li_h = eng.finish(True)
    c_o, g_o, al_o, Li_o = ora.finish(True)
    out.append(('cost', abs(cost - c_o) / abs(c_o)))
    out.append(('grad', rel(grad, g_o)))
    out.append(('grad_abc', rel(grad[:3], g_o[:3])))
    out.append(('alpha_h', rel(alpha, al_o))); out.append(('Li_h', rel(Li_h, Li_o)))
    mu, sd = eng.predict(Xs, al_o, Li_o)
    mu_o, sd_o = O.predict(Xs, al_o, Li_o, params, S, M)
    out.append(('pred_mu', rel(mu, mu_o))); out.append(('pred_sd', rel(sd, sd_o)))
    # whole-call path must agree with the staged path
    c2, g2, a2, L2 = eng.eval(want_grad=True)
    out.append(('eval_vs_staged', abs(c2 - cost) + rel(g2, grad)))
    c3, _, a3, L3 = eng.eval(want_grad=False)
    out.append(('fwd_only', abs(c3 - cost) / abs(cost)))
    eng.close()
    print('%-16s %s  ' % (name, dtype) + '  '.join('%s=%.1e' % kv for kv in out), flush=True)


def timing(N, D, S, M, dtype, reps=3):
    seed = 0x5CF60077
    X = synth.make_X(seed, N, D)
    params = synth.make_params(seed + 2, D, S, M, abc=(-1.0, 0.0, -1.0))
    y = synth.normal(seed + 5, 0, N).reshape(-1, 1)
    eng = HipEngine(D, S, M, dtype=dtype)
    eng.set_params(params)
    t0 = time.time(); eng.set_data(X, y); t_up = time.time() - t0
    eng.eval(want_grad=True)
    eng.set_profiling(True)
    ts = []
    for _ in range(reps):
        t0 = time.time(); cost, g, a, L = eng.eval(want_grad=True); ts.append(time.time() - t0)
    tm = eng.timings()
    K = 2 * (S + M)
    falg = 10.0 * N * K * K + 4.0 * N * D * (S + M)
    print('TIMING N=%d D=%d S=%d M=%d %s: upload %.2fs eval %s ms  F_alg/t = %.1f TF/s cost=%.6f' % (
        N, D, S, M, dtype, t_up, ['%.1f' % (t * 1e3) for t in ts], falg / min(ts) / 1e12, cost), flush=True)
    print('   ' + '  '.join('%s=%.2f' % kv for kv in tm), flush=True)
    eng.close()


if __name__ == '__main__':
    names = sys.argv[1:] or ['tiny_257x5', 'kin8nm_like', 'c1_boston_shape', 'artifact_shape', 'c2_small_n']
    for nm in names:
        for dt in ('f64', 'f32'):
            try:
                stages(nm, dt)
            except Exception:
                print('FAILED', nm, dt); traceback.print_exc(); sys.stdout.flush()
    for cfg in [(100000, 32, 16, 256, 'f64'), (100000, 32, 16, 256, 'f32'),
                (131072, 64, 32, 1024, 'f32'), (131072, 64, 32, 1024, 'f64')]:
        try:
            timing(*cfg)
        except Exception:
            print('FAILED timing', cfg); traceback.print_exc(); sys.stdout.flush()
