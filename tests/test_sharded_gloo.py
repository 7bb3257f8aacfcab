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
