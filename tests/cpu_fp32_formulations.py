"""CPU emulation (numpy, float32 BLAS per 4096-row chunk, float64 across chunks) of sweeps 2 and 3 of fp32 mode behind an
exact (fp64) Gram, in two formulations:
  V-form: V = Phi B,      v = rowsum(Phi o V), BWB = V^T diag(q) V,            Phibar = 2 Phi Abar + 2 q o V + ...
  C-form: C = Phi Li^T,   v = rowsum(C^2),     BWB = Li^T (C^T diag(q) C) Li,  Phibar = 2 Phi Abar + 2 q o (C Li) + ...
(the reference's own graph is the C-form: SCFGP/SCFGP.py:112).  Prints per-block gradient errors against float64.
    python tests/cpu_fp32_formulations.py [N] [D] [S] [M]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import scfgp_oracle as O          # noqa: E402
from scfgp_amd import synth                   # noqa: E402

f32 = np.float32


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def main(N=20000, D=8, S=32, M=1024, chunk=4096, seed=0x5CF600FF):
    J = S + M; K = 2 * J
    X = synth.make_X(seed, N, D)
    teacher = synth.make_params(seed + 0x0101, D, S, M, abc=(-1.0, 0.0, -1.0))
    params = synth.make_params(seed + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
    y = O.feature_map(X, teacher, D, S, M) @ synth.teacher_weights(seed + 0x0303, K) + 0.1 * synth.normal(seed + 0x0404, 0, N)
    y = (y - y.mean()) / y.std()
    a, b, c, l_F, r_F, F, l_FC, FC = O.unpack_params(params, D, S, M)
    lam = np.exp(2 * a) + 1e-6; kappa = np.log1p(np.exp(c)); em2a = np.exp(-2 * a)
    Phi = O.feature_map(X, params, D, S, M)
    G = Phi.T @ Phi; g = Phi.T @ y; yy = y @ y
    A = G + lam * np.eye(K)
    L = np.linalg.cholesky(A); Li = np.linalg.solve(L, np.eye(K)); B = Li.T @ Li
    alpha = B @ g
    print('N %d K %d  cond_est %.3g  cond2 %.3g' % (N, K, (np.diag(L) ** 2).max() * np.diag(B).max(), np.linalg.cond(A)))
    chunks = [(lo, min(N, lo + chunk)) for lo in range(0, N, chunk)]
    Phi32 = Phi.astype(f32); P64 = Phi32.astype(np.float64); X32 = X.astype(f32)

    def finish(v, mu, tn_gram, side_p, second_term, name):
        """v, mu: per-row moments; tn_gram(q) -> BWB (K,K) fp64; side_p(p) -> u; second_term(q) -> fp32 accumulators holding
        q o (Phi B) (N,K) to which Phi Abar is added in fp32."""
        d = kappa * (v + 1); r = mu - y
        e = 1 / d - (r * r + v) / d ** 2; q = 1 / d + kappa * e; p = 2 * r / d
        kbar = (e * (v + 1)).sum()
        BWB = tn_gram(q); u = side_p(p)
        Abar = B - BWB - 0.5 * (np.outer(u, alpha) + np.outer(alpha, u)) + em2a * np.outer(alpha, alpha)
        T3 = em2a * (yy - g @ alpha)
        abar = 2 * np.exp(2 * a) * np.trace(Abar) - 2 * T3 + 2 * (N - M)
        cbar = kbar / (1 + np.exp(-c)); ut = u - 2 * em2a * alpha
        acc = second_term(q)                                                  # fp32 (N,K)
        if acc is None:
            Pb = 2.0 * (Phi @ Abar) + 2 * q[:, None] * (Phi @ B)
        else:
            acc = acc + Phi32 @ Abar.astype(f32)                              # fp32 accumulate on top
            Pb = 2.0 * acc.astype(np.float64)
        Pb = Pb + np.outer(p, alpha) + np.outer(y, ut)
        if acc is None:
            bbar = (Pb * Phi).sum()
            Zb = Phi[:, :J] * Pb[:, J:] - Phi[:, J:] * Pb[:, :J]
            XZ = X.T @ Zb; colsum = Zb.sum(0)
        else:
            bbar = (Pb * P64).sum()
            Pb32 = Pb.astype(f32)
            Zb = Phi32[:, :J] * Pb32[:, J:] - Phi32[:, J:] * Pb32[:, :J]      # fp32
            XZ = sum((X32[lo:hi].T @ Zb[lo:hi]).astype(np.float64) for lo, hi in chunks)
            colsum = Zb.astype(np.float64).sum(0)
        grad = O._epilogue(params, D, S, M, dict(XZ=XZ, colsum=colsum, bbar=bbar), abar, cbar, N)
        return grad

    t0 = time.time()
    V = Phi @ B
    gref = finish((V * Phi).sum(1), Phi @ alpha, lambda q: V.T @ (q[:, None] * V), lambda p: V.T @ p, lambda q: None, 'fp64')
    print('fp64 reference %.0f s' % (time.time() - t0))
    o = 3 + D * S
    blocks = lambda gr: ' '.join('%s %.2e' % (nm, rel(u, v)) for nm, u, v in
                                 zip(('abc', 'l_F', 'r_F'), (gr[:3], gr[3:o], gr[o:o + M * S]), (gref[:3], gref[3:o], gref[o:o + M * S])))
    mu32 = P64 @ alpha
    # ---- V-form, fp32
    V32 = Phi32 @ B.astype(f32)
    V64 = V32.astype(np.float64)
    vV = (P64 * V64).sum(1)
    gramV = lambda q: sum((V32[lo:hi].T @ (q[lo:hi, None].astype(f32) * V32[lo:hi])).astype(np.float64) for lo, hi in chunks)
    gramV64 = lambda q: V64.T @ (q[:, None] * V64)
    print('V-form  v rel %.2e' % rel(vV, (V * Phi).sum(1)))
    print('V-form fp32 gram_w      :', blocks(finish(vV, mu32, gramV, lambda p: V64.T @ p, lambda q: q[:, None].astype(f32) * V32, 'V')))
    print('V-form fp64 gram_w (lvl2):', blocks(finish(vV, mu32, gramV64, lambda p: V64.T @ p, lambda q: q[:, None].astype(f32) * V32, 'V2')))
    # ---- C-form, fp32
    C32 = Phi32 @ Li.T.astype(f32)
    C64 = C32.astype(np.float64)
    vC = (C64 ** 2).sum(1)
    print('C-form  v rel %.2e' % rel(vC, (V * Phi).sum(1)))
    gramC = lambda q: Li.T @ sum((C32[lo:hi].T @ (q[lo:hi, None].astype(f32) * C32[lo:hi])).astype(np.float64) for lo, hi in chunks) @ Li
    gramC64 = lambda q: Li.T @ (C64.T @ (q[:, None] * C64)) @ Li
    sideC = lambda p: Li.T @ (C64.T @ p)
    second = lambda q: q[:, None].astype(f32) * (C32 @ Li.astype(f32))
    print('C-form fp32 gram_w      :', blocks(finish(vC, mu32, gramC, sideC, second, 'C')))
    print('C-form fp64 gram_w      :', blocks(finish(vC, mu32, gramC64, sideC, second, 'C2')))
    # C-form sweep 2, exact sweep 3 second term (isolates the q o (C Li) product)
    print('C-form fp32 gram_w, V-form second term:', blocks(finish(vC, mu32, gramC, sideC, lambda q: q[:, None].astype(f32) * V32, 'C3')))


if __name__ == '__main__':
    main(*[int(v) for v in sys.argv[1:5]])
