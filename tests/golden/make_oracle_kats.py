"""
Regenerates tests/golden/oracle_kats.npz: known-answer vectors from the float64
CPU oracle (literal graph + torch autograd of the literal graph) on seeded
synthetic inputs.  Inputs are NOT stored -- `case_inputs()` rebuilds them from
the seed -- only expected outputs are.

These pin what the reference's artifact cannot: the gradient (SCFGP/SCFGP.py:129)
and the predictive sigma (SCFGP/SCFGP.py:144).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from scfgp_amd import synth   # noqa: E402

# name -> (N, D, S, M, T_predict, seed)
CASES = {
    'tiny_257x5':      (257, 5, 3, 7, 33, 0x5CF60011),
    'artifact_shape':  (400, 13, 20, 50, 106, 0x5CF60012),
    'kin8nm_like':     (1000, 8, 4, 16, 100, 0x5CF60013),
    'c1_boston_shape': (506, 13, 8, 64, 64, 0x5CF60014),
    'c2_small_n':      (2048, 32, 16, 256, 128, 0x5CF60015),
}


def case_inputs(name):
    """(X, y, params, Xs) for a named case -- pure function of the seed."""
    from oracle import scfgp_oracle as O
    N, D, S, M, T, seed = CASES[name]
    X = synth.make_X(seed, N, D)
    params = synth.make_params(seed + 0x0202, D, S, M)
    params[0] = -1.0 + 0.3 * params[0]; params[1] = 0.3 * params[1]; params[2] = -1.0 + 0.3 * params[2]
    teacher = synth.make_params(seed + 0x0101, D, S, M, abc=(-1.0, 0.0, -1.0))
    f = O.feature_map(X, teacher, D, S, M) @ synth.teacher_weights(seed + 0x0303, 2 * (S + M))
    y = synth.finish_targets(seed + 0x0404, f)
    Xs = synth.make_X(seed + 0x0505, T, D)
    return X, y.reshape(-1, 1), params, Xs


def main():
    from oracle import scfgp_oracle as O, autograd_ref as AR
    out = {}
    for name, (N, D, S, M, T, seed) in CASES.items():
        X, y, params, Xs = case_inputs(name)
        cost, alpha, Li = O.forward(X, y, params, S, M, True)
        c2, grad, al2, Li2 = AR.value_and_grad(X, y, params, S, M)
        c3, g3, al3, Li3 = O.value_and_grad(X, y, params, S, M, chunk=300)
        mu, std = O.predict(Xs, alpha, Li, params, S, M)
        rel = lambda u, v: np.linalg.norm(u - v) / np.linalg.norm(v)
        print('%-16s cost %+.12e  |autograd-literal| %.1e  |3sweep-autograd| grad %.1e cost %.1e' % (
            name, cost, abs(c2 - cost) / abs(cost), rel(g3, grad), abs(c3 - cost) / abs(cost)))
        K = 2 * (S + M)
        out[name + '/cost'] = cost
        out[name + '/grad'] = grad
        out[name + '/alpha'] = alpha
        out[name + '/mu'] = mu
        out[name + '/std'] = std
        if K <= 160:
            out[name + '/Li'] = Li
        else:                                   # keep the fixture small: sampled rows + norms
            rows = np.r_[0:4, K // 2:K // 2 + 4, K - 4:K]
            out[name + '/Li_rows'] = rows
            out[name + '/Li_sample'] = Li[rows]
            out[name + '/Li_fro'] = np.linalg.norm(Li)
            out[name + '/Li_diag'] = np.diagonal(Li).copy()
    np.savez_compressed(os.path.join(HERE, 'oracle_kats.npz'), **out)


if __name__ == '__main__':
    main()
