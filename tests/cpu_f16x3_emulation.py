"""
CPU emulation (not a test; it uses the oracle, hence under tests/): what a THREE-TERM fp16 split of the N-sized products would
cost in accuracy.  x = (h + l) 2^-e with h = fp16(x 2^e), l = fp16(x 2^e - h) and e chosen so that max |x| 2^e lies in
[2^14, 2^15) -- a power-of-two scale per operand matrix, which keeps l in fp16's normal range for every entry within 2^-18 of the
largest -- represents x to ~23 bits; a product is h.h + h.l + l.h in fp32 accumulators (products of two fp16 values are exact in
fp32: 11 + 11 bits), the l.l term (2^-22 relative) dropped.  Same operand bytes as fp32 (two fp16 planes), 3/16 of the fp32 MFMA
time at the fp16 matrix rate.  Compare with the six-term bf16 split (profiles/r02_tuning.md): 6 B per element, 6/16 of the time.
    python tests/cpu_f16x3_emulation.py > profiles/r05_f16x3_emulation.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import scfgp_oracle as O                                  # noqa: E402
from scfgp_amd import synth                                           # noqa: E402


def split16(x):
    e = 14 - int(np.floor(np.log2(np.abs(x).max())))
    xs = x * 2.0 ** e
    h = xs.astype(np.float16)
    l = (xs - h.astype(np.float64)).astype(np.float16)
    return h.astype(np.float32), l.astype(np.float32), e


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def main():
    for (N, D, S, M, abc) in ((20000, 16, 16, 256, (-1.0, 0.0, -1.0)), (20000, 8, 32, 256, (-1.0, 0.0, -1.0)), (30000, 32, 16, 496, (-2.0, 0.5, -1.0))):
        seed = 0x5CF60777 + M
        X = synth.make_X(seed, N, D); y = synth.normal(seed + 1, 0, N).reshape(-1, 1)
        p = synth.make_params(seed + 2, D, S, M, abc=abc)
        Phi = O.feature_map(X, p, D, S, M); K = Phi.shape[1]
        lam = np.exp(2 * p[0]) + 1e-6
        A = Phi.T @ Phi + lam * np.eye(K); B = np.linalg.inv(A)
        ph, pl, ea = split16(Phi); bh, bl, eb = split16(B)
        print('N=%d D=%d S=%d M=%d K=%d  cond_2(A) = %.3g' % (N, D, S, M, K, np.linalg.cond(A)))
        print('  representation (relative, norm-wise): Phi fp32 %.2e / f16x2 %.2e    B fp32 %.2e / f16x2 %.2e' % (
            rel(Phi.astype(np.float32).astype(np.float64), Phi), rel((ph.astype(np.float64) + pl) * 2.0 ** -ea, Phi),
            rel(B.astype(np.float32).astype(np.float64), B), rel((bh.astype(np.float64) + bl) * 2.0 ** -eb, B)))
        V64 = Phi @ B
        V32 = (Phi.astype(np.float32) @ B.astype(np.float32)).astype(np.float64)
        V16 = ((ph @ bh) + (ph @ bl) + (pl @ bh)).astype(np.float64) * 2.0 ** -(ea + eb)
        print('  V = Phi B:            fp32 %.2e    f16x3 %.2e' % (rel(V32, V64), rel(V16, V64)))
        G64 = Phi.T @ Phi; G32 = np.zeros((K, K)); G16 = np.zeros((K, K))
        for c in range(0, N, 4096):                                  # fp32 accumulators flushed to fp64 every 4096 rows, as the library does
            P32 = Phi[c:c + 4096].astype(np.float32); G32 += (P32.T @ P32).astype(np.float64)
            h, l = ph[c:c + 4096], pl[c:c + 4096]
            G16 += ((h.T @ h) + (h.T @ l) + (l.T @ h)).astype(np.float64) * 2.0 ** -(2 * ea)
        print('  G = Phi^T Phi:        fp32 %.2e    f16x3 %.2e' % (rel(G32, G64), rel(G16, G64)))
        g = Phi.T @ y; al0 = np.linalg.solve(A, g)
        for nm, G in (('fp32 ', G32), ('f16x3', G16)):
            al = np.linalg.solve(G + lam * np.eye(K), g)
            print('  alpha from that G:    %s %.2e' % (nm, rel(al, al0)))


if __name__ == '__main__':
    main()
