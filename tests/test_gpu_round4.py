"""
Round-4 GPU tests: the BASELINE configs that are DEFINED by their sharding, at full size on one card.

C4 (N = 4e6, D = 64, S = 32, M = 1024, fp32, "row-sharded across 2/4/8 GPUs") and C5 (N = 1e6, D = 512, S = 64, M = 2048,
fp32, "8 x MI355X") split the row sums of SCFGP/SCFGP.py:104 (Phi^T Phi), :108 (Phi^T y) and :126 (y^T y, the expected-NLL
sum) -- and of the reverse sweep of :129 -- over ranks.  Here the 8 ranks are 8 contexts on ONE GPU with 1/8 of the rows
each and `n_global` = N, their exchange buffers summed the way the all-reduce would; the result must be the single-context
one.  C4 is also checked against fp64 mode (the reference's arithmetic, SCFGP/SCFGP.py:95-96) at its full 4e6 rows:
Phi there has 8.7e9 elements, more than 2^32.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


def grad_blocks(g, D, S, M):
    o = 3 + D * S
    return dict(grad_abc=g[:3], grad_lF=g[3:o], grad_rF=g[o:o + M * S])


def _single(dtype, D, S, M, params, X, y, Xs):
    """One context on all rows: (cost, grad, alpha, Li), predictions on Xs, condition record, ms of the second evaluation."""
    import time
    from scfgp_amd.engine import HipEngine
    e = HipEngine(D, S, M, dtype)
    e.set_params(params); e.set_data(X, y)
    e.eval()
    t0 = time.perf_counter()
    out = e.eval()
    ms = (time.perf_counter() - t0) * 1e3
    mu, sd = e.predict(Xs, out[2], out[3])
    cd = e.condition()
    e.close()
    return out, (mu, sd), cd, ms


def _sharded(R, D, S, M, params, X, y, want_grad=True):
    """R fp32 contexts on one card, rows [lo_r, hi_r) each, n_global = N; exchange buffers summed in rank order."""
    import torch
    from scfgp_amd.engine import HipEngine
    from scfgp_amd.sharded import shard_rows
    N = X.shape[0]
    stream = torch.cuda.current_stream().cuda_stream
    engs = []
    for r in range(R):
        lo, hi = shard_rows(N, r, R)
        e = HipEngine(D, S, M, 'f32', stream=stream)
        e.set_params(params); e.set_data(X[lo:hi], y[lo:hi], n_global=N)
        engs.append(e)
    sizes = {}

    def allsum(stage):
        bufs = [e.exchange(stage) for e in engs]
        sizes[stage] = bufs[0].numel() * 8
        tot = bufs[0].clone()
        for b in bufs[1:]:
            tot += b
        for b in bufs:
            b.copy_(tot)

    outs = None
    for _ in range(3):                                          # at most two precision escalations; all ranks decide alike
        for e in engs: e.pass1()
        allsum(1)
        for e in engs: e.factor()
        for e in engs: e.pass2(want_grad)
        allsum(2)
        if want_grad:
            for e in engs: e.adjoint()
            for e in engs: e.pass3()
            allsum(3)
        outs = [e.finish(want_grad) for e in engs]
        assert all(o is None for o in outs) or all(o is not None for o in outs)
        if outs[0] is not None:
            break
    levels = [e.condition()['level'] for e in engs]
    for e in engs:
        e.close()
    return outs, sizes, levels


@pytest.mark.parametrize('cfg', ['C4', 'C5'])
def test_eight_row_shards_on_one_card_equal_the_single_context_at_full_size(cfg):
    """C4 and C5 as their 8 ranks would run them: 500 000- / 125 000-row shards, n_global != N, 20 MB / 71 MB exchange buffers.
    fp32 mode flushes its fp32 accumulators to fp64 every 4096 rows OF A ROW SPLIT, so another row partition rounds the Gram
    differently at the 6e-8 level: the bound on alpha / Li / the gradient is the fp32-vs-fp64 one, not bit equality (cost is
    insensitive: 1e-9).  Every rank must come back with the same numbers bit for bit (replicated K x K stage on equal sums).
    C4 additionally: fp32 mode against fp64 mode at 4e6 rows, the north star's outputs within 1e-5."""
    import bench
    from scfgp_amd import synth
    from scfgp_amd.engine import HipEngine
    N, D, S, M = bench.CONFIGS[cfg][:4]
    e0 = HipEngine(D, S, M, 'f32')
    X, y, params = bench.build_problem(e0, N, D, S, M, 0, N, None)
    e0.close()
    Xs = synth.make_X(bench.SEED + 0x0909, 4096, D)
    out32, pred32, cd32, ms32 = _single('f32', D, S, M, params, X, y, Xs)
    assert cd32['level'] == 0 and cd32['cond_est'] < 10
    if cfg == 'C4':
        out64, pred64, _, ms64 = _single('f64', D, S, M, params, X, y, Xs)
        par = dict(cost=abs(float(out32[0]) - float(out64[0])) / abs(float(out64[0])), alpha=rel(out32[2], out64[2]),
                   Li=rel(out32[3], out64[3]), mu=rel(pred32[0], pred64[0]), std=rel(pred32[1], pred64[1]))
        par.update({k: rel(v, w) for (k, v), w in zip(grad_blocks(out32[1], D, S, M).items(), grad_blocks(out64[1], D, S, M).values())})
        print('\nC4 fp32 (%.0f ms) vs fp64 (%.0f ms) at %d rows: ' % (ms32, ms64, N) + ' '.join('%s %.2e' % kv for kv in par.items()))
        bounds = dict(cost=5e-10, grad_abc=1e-9, grad_lF=3e-5, grad_rF=3e-5, alpha=5e-6, Li=1e-6, mu=5e-6, std=1e-10)   # the H / C5 bounds
        for k, b in bounds.items():
            assert par[k] < b, (k, par[k], b)
        assert all(par[k] < 1e-5 for k in ('cost', 'alpha', 'Li', 'mu', 'std'))
    outs, sizes, levels = _sharded(8, D, S, M, params, X, y)
    assert levels == [0] * 8
    K = 2 * (S + M)
    assert sizes[1] == sizes[2] and sizes[1] > 8 * K * (K + 1) // 2          # packed lower tiles + vector + scalars
    c0, g0, a0, L0 = out32
    c, g, a, L = outs[0]
    for o in outs[1:]:                                          # replicated K x K stage on the same sums: ranks agree bit for bit
        assert float(o[0]) == float(c) and np.array_equal(o[1], g) and np.array_equal(o[2], a) and np.array_equal(o[3], L)
    sh = dict(cost=abs(float(c) - float(c0)) / abs(float(c0)), alpha=rel(a, a0), Li=rel(L, L0))
    sh.update({k: rel(v, w) for (k, v), w in zip(grad_blocks(g, D, S, M).items(), grad_blocks(g0, D, S, M).values())})
    print('\n%s as 8 shards of %d rows vs one context (exchange buffers %.1f / %.1f MB): ' % (cfg, N // 8, sizes[1] / 1e6, sizes[3] / 1e6)
          + ' '.join('%s %.2e' % kv for kv in sh.items()))
    assert sh['cost'] < 1e-9 and sh['alpha'] < 5e-6 and sh['Li'] < 1e-6
    assert sh['grad_abc'] < 1e-8 and sh['grad_lF'] < 3e-5 and sh['grad_rF'] < 3e-5
    # forward only (train_func): exchange 2 shrinks to the 8 scalars
    outs_f, sizes_f, _ = _sharded(8, D, S, M, params, X, y, want_grad=False)
    assert sizes_f[2] == 64 and abs(float(outs_f[0][0]) - float(c)) <= 1e-15 * abs(float(c))
    assert np.array_equal(outs_f[0][2], a)


def test_eight_fp64_row_shards_equal_the_single_context_at_c2_size_with_uneven_rows():
    """The same exchange in the reference's arithmetic, where a row partition may not change anything above rounding of the
    fp64 sums: N = 100 003 rows (C2's shape; shards of 12 500 / 12 501 rows), 8 contexts, cost 1e-13, alpha / Li / grad 1e-10."""
    import torch
    import bench
    from scfgp_amd.engine import HipEngine
    from scfgp_amd.sharded import shard_rows
    N, D, S, M = bench.CONFIGS['C2'][:4]
    N += 3
    e0 = HipEngine(D, S, M, 'f64')
    X, y, params = bench.build_problem(e0, N, D, S, M, 0, N, None)
    e0.set_params(params); e0.set_data(X, y)
    c0, g0, a0, L0 = e0.eval()
    e0.close()
    stream = torch.cuda.current_stream().cuda_stream
    engs = []
    for r in range(8):
        lo, hi = shard_rows(N, r, 8)
        e = HipEngine(D, S, M, 'f64', stream=stream)
        e.set_params(params); e.set_data(X[lo:hi], y[lo:hi], n_global=N)
        engs.append(e)
    assert sorted(set(e.N for e in engs)) == [12500, 12501]

    def allsum(stage):
        bufs = [e.exchange(stage) for e in engs]
        tot = bufs[0].clone()
        for b in bufs[1:]:
            tot += b
        for b in bufs:
            b.copy_(tot)

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
        assert abs(float(c) - float(c0)) < 1e-13 * abs(float(c0))
        assert rel(a, a0) < 1e-10 and rel(L, L0) < 1e-10 and rel(g, g0) < 1e-10
        e.close()


def test_model_optimize_hashes_its_own_arrays_once(monkeypatch):
    """SCFGP.set_data freezes the arrays it creates (SCFGP/SCFGP.py:161-162 stores self.X / self.y; :237 passes them to
    train_iter_func on every iteration), so the triple's residency check hashes them on the FIRST call only: every later
    iteration of optimize goes straight to the evaluation.  An array the caller edits in place is still re-uploaded when it is
    passed to the triple directly (writeable arrays are hashed on every call)."""
    from scfgp_amd import SCFGP, funcs
    calls = []
    real = funcs._hash64
    monkeypatch.setattr(funcs, '_hash64', lambda buf: (calls.append(memoryview(buf).nbytes), real(buf))[1])
    rng = np.random.default_rng(11)
    np.random.seed(11)
    X = rng.uniform(-2, 2, (20000, 5))                          # 800 KB: above the 64K-element threshold of the sampled path
    y = np.sin(X[:, :1]) + 0.3 * X[:, 1:2] + 0.05 * rng.standard_normal((20000, 1))
    model = SCFGP(sparsity=4, nfeats=20)
    model.set_data(X, y)
    assert not model.X.flags.writeable and not model.y.flags.writeable and model.X.base is None
    model.optimize(max_iter=12, max_cvrg=100)
    big = [n for n in calls if n >= model.y.nbytes]             # hashes of X or y (everything smaller is scaler keys etc.)
    assert len(model.evals['COST'][1]) == 13                    # 12 iterations + the closing refit (SCFGP/SCFGP.py:265)
    assert len(big) == 2, big                                   # X and y, once, on the first train_iter_func call
    # the reference's own pattern of a caller-owned writeable array handed to the triple: edits are seen
    cf = model._compiled
    Xw = np.array(model.X); yw = np.array(model.y)
    c1 = float(cf.train_func(Xw, yw)[0])
    n0 = len(calls)
    assert float(cf.train_func(Xw, yw)[0]) == c1 and len(calls) == n0 + 2
    yw[123, 0] += 0.5
    assert float(cf.train_func(Xw, yw)[0]) != c1


@pytest.mark.parametrize('dtype', ['f64', 'f32'])
def test_sums_inside_the_library_with_a_one_rank_communicator(dtype):
    """scfgp_comm_unique_id / scfgp_comm_init (include/scfgp_hip.h): with a communicator attached the staged calls end in their
    own ncclAllReduce on the library's stream.  One rank is all a one-GPU box can supply (RCCL refuses two ranks on one device):
    the sum over one rank changes nothing, so eval(), the staged calls, eval_rows() and predict() return the plain context's
    numbers bit for bit, the profile shows the three exchanges, scfgp_train runs with them inside, and the communicator can be dropped."""
    from scfgp_amd.engine import HipEngine
    from tests.golden.make_oracle_kats import CASES, case_inputs
    name = 'kin8nm_like'
    N, D, S, M, T, seed = CASES[name]
    X, y, params, Xs = case_inputs(name)
    plain = HipEngine(D, S, M, dtype); plain.set_params(params); plain.set_data(X, y)
    c0, g0, a0, L0 = plain.eval()
    eng = HipEngine(D, S, M, dtype); eng.set_params(params); eng.set_data(X, y, n_global=N)
    uid = eng.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    eng.comm_init(1, 0, uid)
    with pytest.raises(ValueError):
        eng.comm_init(1, 0, uid)                                # one communicator per context
    eng.set_profiling(True)
    c, g, a, L = eng.eval()
    names = [n for n, _ in eng.timings()]
    assert [n for n in names if n.startswith('exchange')] == ['exchange1', 'exchange2', 'exchange3']
    assert float(c) == float(c0) and np.array_equal(g, g0) and np.array_equal(a, a0) and np.array_equal(L, L0)
    eng.pass1(); eng.factor(); eng.pass2(False)
    cf, _, af, _ = eng.finish(False)
    assert float(cf) == float(plain.eval(want_grad=False)[0]) and np.array_equal(af, a0)
    idx = np.arange(0, N, 3)
    cr, gr, _, _ = eng.eval_rows(idx)
    cr0, gr0, _, _ = plain.eval_rows(idx)
    assert float(cr) == float(cr0) and np.array_equal(gr, gr0)
    eng.opt_init('adam'); plain.opt_init('adam')               # round 5: the on-device loop carries the sums too
    h1, _, _ = eng.train(2); h0, _, _ = plain.train(2)
    assert np.array_equal(h1, h0) and np.array_equal(eng.get_params(), plain.get_params())
    eng.set_params(params); plain.set_params(params)
    eng.comm_destroy()
    eng.set_profiling(True)
    c2, _, _, _ = eng.eval()
    assert float(c2) == float(c0) and not [n for n, _ in eng.timings() if n.startswith('exchange')]
    eng.close(); plain.close()


def test_one_rank_under_the_launcher_with_the_sums_inside_the_library():
    """bench.py --native-rccl under the driver's launcher line with one rank: the 128-byte id travels over torch.distributed,
    the library joins the communicator and issues the three ncclAllReduce calls itself; same cost as the plain run, bit for bit,
    and the line says which path summed."""
    import json
    import os
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
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '1', '--native-rccl'] + args
    nat = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert one.returncode == 0 and nat.returncode == 0
    o1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith('{')][-1])
    o2 = json.loads([ln for ln in nat.stdout.splitlines() if ln.startswith('{')][-1])
    assert o2['n_gpus'] == 1 and o2['cost'] == o1['cost']
    assert 'issued by the library' in o2['config']['parallelism']
    st = o2['stages_ms']
    assert all(('exchange%d' % k) in st and st['exchange%d' % k] >= 0 for k in (1, 2, 3))


@pytest.mark.parametrize('S,M', [(32, 288), (32, 320), (16, 432), (32, 1024)])
def test_fp32_gram_tile_kinds_repeat_bit_for_bit_and_agree_with_fp64(S, M):
    """The fp32 Gram's job list by column count: K = 640 (five 128-blocks: an UNPAIRED last block row of off-diagonal 128 x 128
    tiles), 704 (the same + a 64-column strip), 896 (seven blocks: three tall pairs, unpaired row, no strip), 2112 (the headline
    list: tall, wide strip, diagonal blocks, strip).  All of them run on the hand-scheduled LDS-DMA loop whose `vmcnt` / `lgkmcnt`
    waits are counted by hand: a wrong count shows as results that differ between repeats (it did once, at 6e-10), a wrong tile
    map as a disagreement with fp64 mode (SCFGP/SCFGP.py:104-105,108: A = Phi^T Phi + ..., Phi^T y; backward of :111-113)."""
    from scfgp_amd import synth
    from scfgp_amd.engine import HipEngine
    N, D = 40000, 16
    X = synth.make_X(0x5CF60777 + M, N, D)
    y = synth.normal(0x5CF60778 + M, 0, N).reshape(-1, 1)
    params = synth.make_params(0x5CF60779 + M, D, S, M, abc=(-1.0, 0.0, -1.0))
    outs = {}
    for dtype in ('f32', 'f64'):
        e = HipEngine(D, S, M, dtype)
        if dtype == 'f32':
            e.set_option('gram64', 0)                               # stay in fp32 whatever the condition estimate says
        e.set_params(params); e.set_data(X, y)
        a = e.eval(want_grad=True)
        b = e.eval(want_grad=True)
        assert float(a[0]) == float(b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
        outs[dtype] = a
        e.close()
    (c32, g32, a32, L32), (c64, g64, a64, L64) = outs['f32'], outs['f64']
    # fp32 Gram without the precision levels: loose bounds (a misplaced tile is an O(1) error), the repeats above are exact
    assert abs(float(c32) - float(c64)) < 1e-5 * max(1.0, abs(float(c64)))
    assert rel(a32, a64) < 5e-3 and rel(L32, L64) < 1e-3 and rel(g32, g64) < 2e-2
