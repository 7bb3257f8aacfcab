"""
Two processes sharing the one GPU of the test box, each with its own HIP context and half of the
rows, driven by ShardedEvaluator over torch.distributed.  RCCL refuses two ranks on one device,
so the collective backend here is gloo (device buffers staged through the host inside
torch_allreduce); everything else -- per-rank contexts, exchange-buffer aliasing, n_global,
three sums per evaluation -- is the production multi-GPU flow of bench.py --gpus N.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from scfgp_amd.funcs import CompiledFuncs
    from scfgp_amd.sharded import shard_rows, torch_allreduce
    from tests.golden.make_oracle_kats import CASES, case_inputs
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    name = 'c2_small_n'
    N, D, S, M, T, seed = CASES[name]
    X, y, params, _ = case_inputs(name)
    lo, hi = shard_rows(N, rank, world)
    cf = CompiledFuncs(D, S, M, params.copy(), stream=torch.cuda.current_stream().cuda_stream,
                       allreduce=torch_allreduce())
    Xl, yl = np.ascontiguousarray(X[lo:hi]), np.ascontiguousarray(y[lo:hi])
    cost, grad, alpha, Li = cf.value_and_grad(Xl, yl)          # n_global discovered by an all-reduce
    c_f, a_f, L_f = cf.train_func(Xl, yl)
    c_i, _, _ = cf.train_iter_func(Xl, yl)                     # every rank applies the same update
    p_after = cf.params.get_value()
    if rank == 0:
        q.put((float(cost), grad, alpha, float(c_f), float(c_i), p_after))
    else:
        q.put(('p1', p_after))
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_sharded_matches_oracle():
    from oracle import scfgp_oracle as O
    from tests.golden.make_oracle_kats import CASES, case_inputs
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300), q.get(timeout=300)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0 = [g for g in got if g[0] != 'p1'][0]
    r1 = [g for g in got if g[0] == 'p1'][0]
    cost, grad, alpha, c_f, c_i, p_after = r0
    N, D, S, M, T, seed = CASES['c2_small_n']
    X, y, params, _ = case_inputs('c2_small_n')
    c0, g0, a0, _ = O.value_and_grad(X, y, params, S, M)
    assert abs(cost - c0) < 1e-11 * abs(c0) and abs(c_f - c0) < 1e-11 * abs(c0) and abs(c_i - c0) < 1e-11 * abs(c0)
    assert np.linalg.norm(grad - g0) < 1e-9 * np.linalg.norm(g0)
    assert np.linalg.norm(alpha - a0) < 1e-9 * np.linalg.norm(a0)
    assert np.array_equal(p_after, r1[1]) and not np.array_equal(p_after, params)
