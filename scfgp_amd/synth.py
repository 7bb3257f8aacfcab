"""
Counter-based synthetic inputs for benchmarks and parity tests.

u(i) = (splitmix64(seed XOR i) >> 11) * 2**-53 is a pure function of (seed, i), so
any row range of X can be produced independently on any rank without shipping
data, and CPU/GPU legs of a comparison see identical bits.

X ~ U(0,1) mirrors what the reference's default X scaler ('auto-inv-normal',
SCFGP/Scaler.py:116; default at SCFGP/SCFGP.py:34) emits; parameters are drawn
like SCFGP.init_params (SCFGP/SCFGP.py:65-71).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays."""
    with np.errstate(over='ignore'):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform(seed, start, count):
    """u(start) .. u(start+count-1) in [0,1), float64."""
    i = np.arange(start, start + count, dtype=np.uint64)
    return (splitmix64(np.uint64(seed) ^ i) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def normal(seed, start, count):
    """Box-Muller on two independent uniform streams (seed, seed+1)."""
    u1 = uniform(seed, start, count)
    u2 = uniform(seed + 1, start, count)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2 * np.pi * u2)


def make_X(seed, N, D, row0=0):
    """Rows [row0, row0+N) of the synthetic design matrix, X[n,d] = u(n*D+d)."""
    return uniform(seed, row0 * D, N * D).reshape(N, D)


def make_params(seed, D, S, M, abc=None):
    """Flat hyper-parameter vector drawn like SCFGP.init_params (SCFGP/SCFGP.py:65-71);
    `abc` fixes (a,b,c) (benchmarks use (-1,0,-1) to keep A well conditioned)."""
    o = 0
    head = normal(seed, o, 3) if abc is None else np.asarray(abc, np.float64); o += 3
    l_f = normal(seed, o, D * S); o += D * S
    r_f = uniform(seed + 7, o, M * S); o += M * S
    l_p = 2 * np.pi * uniform(seed + 7, o, S); o += S
    p = 2 * np.pi * uniform(seed + 7, o, M)
    return np.concatenate([head, l_f, r_f, l_p, p])


def teacher_weights(seed, K):
    return normal(seed, 0, K)


def finish_targets(seed, f, row0=0, mean=None, std=None):
    """y = standardise(f + 0.1*eps): `f` is the teacher's noiseless response for rows
    [row0, row0+len(f)).  With mean/std None they are taken from this block."""
    y = np.asarray(f, np.float64).ravel() + 0.1 * normal(seed, row0, len(f))
    mean = y.mean() if mean is None else mean
    std = y.std() if std is None else std
    return (y - mean) / std
