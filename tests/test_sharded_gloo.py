"""
Row sharding over 2 processes (gloo, CPU): the ShardedEvaluator driving per-rank engines
with three sums over ranks must reproduce the single-process result.  The compute engine
here is the CPU oracle's staged restatement (tests may use the oracle); on the GPU box the
same driver runs HipEngine shards over RCCL (tests/test_gpu_parity.py covers the HIP side).
"""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import scfgp_oracle as O
    from scfgp_amd.sharded import ShardedEvaluator, shard_rows, torch_allreduce
    from tests.golden.make_oracle_kats import CASES, case_inputs
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    name = 'kin8nm_like'
    N, D, S, M, T, seed = CASES[name]
    X, y, params, _ = case_inputs(name)
    lo, hi = shard_rows(N, rank, world)
    eng = O.OracleEngine(D, S, M)
    eng.set_params(params); eng.set_data(X[lo:hi], y[lo:hi], n_global=N)
    ev = ShardedEvaluator(eng, torch_allreduce(), time_exchanges=True)
    cost, grad, alpha, Li = ev.eval(True)
    ex = ev.exchange_ms()
    assert sorted(ex) == [1, 2, 3] and all(v >= 0 for v in ex.values()) and ev.exchange_ms() == {}
    cost_f, _, _, _ = ev.eval(False)
    assert sorted(ev.exchange_ms()) == [1, 2]                      # forward only: two sums

    class RedoOnce(object):
        """An engine whose first finish() asks for the stages again, like HipEngine after SCFGP_REDO (the library raised its
        precision level): every rank sees it at the same evaluation, so the ranks repeat the three sums together."""
        def __init__(self, inner):
            self.inner, self.calls = inner, 0

        def __getattr__(self, name):
            return getattr(self.inner, name)

        def finish(self, want_grad=True):
            self.calls += 1
            out = self.inner.finish(want_grad)
            return None if self.calls == 1 else out

    redo = RedoOnce(eng)
    c2, g2, _, _ = ShardedEvaluator(redo, torch_allreduce()).eval(True)
    assert redo.calls == 2 and float(c2) == float(cost) and np.array_equal(g2, grad)

    # ---- ranks decide together (include/scfgp_hip.h; the library's settle_level / scfgp_fail_stage, rehearsed on the oracle) ----
    class Counting(object):
        def __init__(self, inner):
            self.inner, self.n = inner, dict(pass1=0, factor_redo=0, finish_redo=0)

        def __getattr__(self, name):
            return getattr(self.inner, name)

        def pass1(self):
            self.n['pass1'] += 1; return self.inner.pass1()

        def factor(self):
            r = self.inner.factor(); self.n['factor_redo'] += r is False; return r

        def finish(self, want_grad=True):
            out = self.inner.finish(want_grad); self.n['finish_redo'] += out is None; return out

    def gather(obj):
        box = [None] * world
        dist.all_gather_object(box, obj)
        return box

    # (1) the summed matrix asks for level 2 on every rank; the LAST rank cannot allocate level `deny` and up
    for deny, agreed, npass1, nfredo in ((2, 1, 2, 0), (1, 0, 3, 1)):
        e = O.OracleEngine(D, S, M); e.set_params(params); e.set_data(X[lo:hi], y[lo:hi], n_global=N)
        e.want_level = 2
        if rank == world - 1:
            e.deny_level = deny
        ce = Counting(e)
        c3, g3, a3, _ = ShardedEvaluator(ce, torch_allreduce()).eval(True)
        rec = gather((e.level, e.ran_level, e.denied, ce.n['pass1'], ce.n['factor_redo'], ce.n['finish_redo'], float(c3)))
        assert all(r == rec[0] for r in rec), rec                # same level, same refusal on record, same number of rounds, same cost
        assert rec[0][:3] == (agreed, agreed, agreed + 1) and rec[0][3] == npass1 and rec[0][4] == nfredo and rec[0][5] == 1
        assert float(c3) == float(cost) and np.array_equal(g3, grad)
        c4, _, _, _ = ShardedEvaluator(ce, torch_allreduce()).eval(True)       # settled: one round, no further attempt at the refused level
        assert ce.n['pass1'] == npass1 + 1 and float(c4) == float(cost)

    # (2) a sweep fails on the last rank: it raises its own error, every other rank PeerFailed, nobody hangs, all recover
    for fail_at, grad_wanted in ((1, True), (2, True), (3, True), (2, False)):
        e = O.OracleEngine(D, S, M); e.set_params(params); e.set_data(X[lo:hi], y[lo:hi], n_global=N)
        if rank == world - 1:
            e.fail_at = fail_at
        ev2 = ShardedEvaluator(e, torch_allreduce())
        try:
            ev2.eval(grad_wanted)
            what = 'returned'
        except O.PeerFailed:
            what = 'peer'
        except RuntimeError as ex:
            what = 'own' if 'injected' in str(ex) else repr(ex)
        assert gather(what) == ['peer'] * (world - 1) + ['own']
        c5 = ev2.eval(grad_wanted)[0]
        assert float(c5) == float(cost)

    # (3) training on row shards: every rank applies the same host rule to the same summed gradient -- the vectors stay
    #     bit-equal without a broadcast (what scfgp_train does on the device with a communicator attached)
    from scfgp_amd.optimizer import Optimizer as OPT, Shared, apply_updates

    class _G(object):
        def __init__(self, n): self.g = np.zeros(n)
        def get_value(self, borrow=False): return self.g

    def train(engine, evaluator, iters=4):
        th = Shared(params.copy()); gs = _G(len(params))
        ups = OPT.apply_nesterov_momentum(OPT.adam(th, gs, learning_rate=0.01), momentum=0.9)
        hist = []
        for _ in range(iters):
            engine.set_params(th.get_value())
            c, g, _, _ = evaluator.eval(True)
            hist.append(float(c)); gs.g = g
            apply_updates(ups)
        return np.array(hist), th.get_value().copy()

    e = O.OracleEngine(D, S, M); e.set_data(X[lo:hi], y[lo:hi], n_global=N)
    hist, theta = train(e, ShardedEvaluator(e, torch_allreduce()))
    got = gather((hist.tobytes(), theta.tobytes()))
    assert all(r == got[0] for r in got)
    if rank == 0:
        es = O.OracleEngine(D, S, M); es.set_data(X, y)
        hist1, theta1 = train(es, ShardedEvaluator(es, None))
        # against one process on all rows: the phase block is left out -- its gradient is identically zero, adam turns
        # its 1e-16 rounding noise into steps of +-learning_rate, and the cost does not depend on it
        nph = S + M
        assert np.allclose(hist, hist1, rtol=1e-9, atol=0), (hist, hist1)
        assert np.linalg.norm(theta[:-nph] - theta1[:-nph]) < 1e-8 * np.linalg.norm(theta1[:-nph])
    if rank == 0:
        q.put((float(cost), grad, alpha, float(cost_f)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_equals_single():
    from oracle import scfgp_oracle as O
    from tests.golden.make_oracle_kats import CASES, case_inputs
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    cost, grad, alpha, cost_f = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    N, D, S, M, T, seed = CASES['kin8nm_like']
    X, y, params, _ = case_inputs('kin8nm_like')
    c0, g0, a0, _ = O.value_and_grad(X, y, params, S, M)
    assert abs(cost - c0) < 1e-12 * abs(c0) and abs(cost_f - c0) < 1e-12 * abs(c0)
    assert np.linalg.norm(grad - g0) < 1e-9 * np.linalg.norm(g0)
    assert np.linalg.norm(alpha - a0) < 1e-9 * np.linalg.norm(a0)
