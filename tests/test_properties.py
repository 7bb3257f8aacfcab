"""Property tests (hypothesis) of the structure the GPU path relies on: every N-dependent quantity of
the objective is additive over rows, so any partition of the rows into shards, summed at the three
exchange points, reproduces the monolithic evaluation."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import scfgp_oracle as O
from scfgp_amd.sharded import shard_rows


@settings(max_examples=12, deadline=None)
@given(N=st.integers(12, 90), D=st.integers(1, 6), S=st.integers(2, 5), M=st.integers(2, 9),
       world=st.integers(1, 5), seed=st.integers(0, 10 ** 6))
def test_row_sharding_is_exact(N, D, S, M, world, seed):
    rng = np.random.default_rng(seed)
    X = rng.random((N, D)); y = rng.standard_normal((N, 1))
    params = O.init_params(D, S, M, rng)
    params[0] = -0.3; params[2] = -0.5
    c0, g0, a0, L0 = O.value_and_grad(X, y, params, S, M)
    engs = []
    for r in range(world):
        lo, hi = shard_rows(N, r, world)
        e = O.OracleEngine(D, S, M); e.set_params(params); e.set_data(X[lo:hi], y[lo:hi], n_global=N)
        engs.append(e)

    def allsum(stage):
        tot = sum(e.exchange(stage) for e in engs)
        for e in engs:
            e.exchange(stage)[...] = tot

    for e in engs: e.pass1()
    allsum(1)
    for e in engs: e.factor()
    for e in engs: e.pass2(True)
    allsum(2)
    for e in engs: e.adjoint()
    for e in engs: e.pass3()
    allsum(3)
    for e in engs:
        c, g, a, L = e.finish(True)
        assert abs(c - c0) <= 1e-9 * abs(c0)
        assert np.linalg.norm(g - g0) <= 1e-7 * np.linalg.norm(g0)
        assert np.linalg.norm(a - a0) <= 1e-7 * np.linalg.norm(a0)


@settings(max_examples=25, deadline=None)
@given(N=st.integers(1, 10 ** 7), world=st.integers(1, 64))
def test_shard_rows_is_a_partition(N, world):
    blocks = [shard_rows(N, r, world) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == N
    assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
    sizes = [hi - lo for lo, hi in blocks]
    assert max(sizes) - min(sizes) <= 1
