"""Which fp32 stage owns the fp32-mode error of an ill-conditioned config?

For each config: fp32 mode and fp64 mode on the same rows, then
  * relative error of the fp32 Gram G and side vector Phi^T y against the fp64 ones,
  * condition of A = G + lam I (eigvalsh in numpy) next to the free estimates the K x K stage could report,
  * the HYBRID: fp32 engine whose exchange buffer 1 (G, Phi^T y, y^T y) is overwritten with the fp64 engine's before
    the factor stage -- i.e. exactly what a pass-1 Gram in fp64 from fp64 features would give -- and its parity block.
One JSON line per config on stdout.   python tools/c3_owner.py C3 H C5 [--rows N]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from scfgp_amd.engine import HipEngine                         # noqa: E402


def fenced_copy(dst_eng, src_eng, stage):
    import torch
    for e in (dst_eng, src_eng):
        e.stream_fence(None, 0)
    d, s = dst_eng.exchange(stage), src_eng.exchange(stage)
    d.copy_(s)
    dst_eng.stream_fence(None, 1)
    torch.cuda.synchronize()


def run(cfg, rows=None, chunk=None):
    import torch
    N, D, S, M, _dt, _ = bench.CONFIGS[cfg]
    N = rows or N
    J = S + M; K = 2 * J
    e32 = HipEngine(D, S, M, 'f32')
    e32.set_option('gram64', 0)                                # the plain fp32 products are the subject: no precision escalation, no REDO from finish()
    X, y, params = bench.build_problem(e32, N, D, S, M, 0, N, None)
    e64 = HipEngine(D, S, M, 'f64')
    for e in (e32, e64):
        e.set_params(params); e.set_data(X, y)
    if chunk:
        e32.set_option('gram_chunk', chunk)
    out32 = e32.eval(); out64 = e64.eval()
    res = {"cfg": cfg, "N": N, "K": K,
           "f32": bench._parity(out32, out64, e32, e64, D, S, M, "fp32 vs fp64")}
    # Gram of both modes
    e32.pass1(); e64.pass1()
    e32.stream_fence(None, 0); e64.stream_fence(None, 0)
    torch.cuda.synchronize()
    x32 = e32.exchange(1).cpu().numpy(); x64 = e64.exchange(1).cpu().numpy()
    d = e64.dims(); Kp = d['Kp']
    npk = x64.size - Kp - 8
    res["G_rel"] = bench.rel(x32[:npk], x64[:npk]); res["g_rel"] = bench.rel(x32[npk:npk + K], x64[npk:npk + K])
    res["G_maxabs_rel"] = float(np.abs(x32[:npk] - x64[:npk]).max() / np.abs(x64[:npk]).max())
    G = e64.debug_read('G', (Kp * Kp + Kp,))[:Kp * Kp].reshape(Kp, Kp)[:K, :K]
    lam = np.exp(2 * params[0]) + 1e-6
    w = np.linalg.eigvalsh(G + lam * np.eye(K))
    Li = out64[3]
    Ld2 = 1.0 / np.diag(Li) ** 2
    Bd = (Li * Li).sum(0)
    res["cond2"] = float(w[-1] / w[0]); res["lam_min"] = float(w[0]); res["lam_max"] = float(w[-1])
    res["est_Lratio"] = float(Ld2.max() / Ld2.min()); res["est_LmaxBmax"] = float(Ld2.max() * Bd.max())
    res["est_trA_Bmax"] = float((np.trace(G) + K * lam) * Bd.max())
    # hybrid: fp32 sweeps 2/3 behind the fp64 Gram
    e64.pass1(); e32.pass1()
    fenced_copy(e32, e64, 1)
    e32.factor(); e32.pass2(True); e32.adjoint(); e32.pass3()
    outh = e32.finish(True)
    e64.factor(); e64.pass2(True); e64.adjoint(); e64.pass3(); e64.finish(True)
    res["hybrid"] = bench._parity(outh, out64, e32, e64, D, S, M, "fp32 sweeps 2/3 behind the fp64 Gram, vs fp64")
    # hybrid 2: additionally exchange buffer 2 (B W B, u, T2, kbar) from the fp64 engine: what is left is pass 3 in fp32
    # (Phibar from the fp32 Phi, V, p, q and X~^T Zbar)
    e64.pass1(); e32.pass1()
    fenced_copy(e32, e64, 1)
    e32.factor(); e32.pass2(True); e64.factor(); e64.pass2(True)
    fenced_copy(e32, e64, 2)
    e32.adjoint(); e32.pass3(); outh2 = e32.finish(True)
    e64.adjoint(); e64.pass3(); e64.finish(True)
    res["hybrid2"] = bench._parity(outh2, out64, e32, e64, D, S, M, "fp32 pass 3 behind fp64 exchanges 1 and 2, vs fp64")
    e32.close(); e64.close()
    return res


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('cfgs', nargs='*', default=['C3'])
    ap.add_argument('--rows', type=int, default=None)
    ap.add_argument('--chunk', type=int, default=None)
    a = ap.parse_args()
    for cfg in a.cfgs:
        print(json.dumps(run(cfg, a.rows if cfg not in ('C1', 'C2') else None, a.chunk)), flush=True)
