#!/usr/bin/env python
"""
bench.py -- NLML+grad evaluations/sec of the SCFGP hot path on MI355X.

Default workload (BASELINE.json `metric`, config "H"): N=1e6, D=64, rank S=32, M=1024 (K=2112), fp32
compute mode, synthetic data; one "step" = one train_iter_func-equivalent evaluation with resident data:
cost + full gradient + alpha + Li back on the host.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C1|C2|C3|C4|C5|H] [--dtype f32|f64] [--rows N]

--gpus N > 1 without a torchrun environment: this process (which touches no GPU) starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD, relays rank 0's JSON line
and exits with the child's code.  Under torchrun (the driver's launch) every rank runs main() directly:
one rank per GPU over RCCL, the N rows sharded over the ranks (strong scaling -- total work fixed, as
the metric is quoted at fixed N), three all-reduces per evaluation.  Rank 0 prints ONE JSON line.

Before the timed loop a ~100 ms BOX PROBE (scfgp_box_probe: register-only fp32 MFMA loop, sustained clock, streaming copy)
is printed as `secondary.box`, so lines from different leases of the pool can be normalised.  With more than one rank the
three sums over ranks are bracketed by events on the stream they run on: `stages_ms.exchange1..3`, `comm_share`, and
`stages_ms_max_over_ranks` (one small all-reduce after the loop).

In the default single-GPU run at config H the same process then also runs the fp64 engine (the
reference's arithmetic, SCFGP/SCFGP.py:95-96,138) on the same rows: `secondary.f64` and the
fp32-vs-fp64 `parity_at_size` block, and times the CPU restatements on a bounded sample (`cpu_baseline`).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x5CF600FF
PEAK_TFLOPS = {'f32': 157.3, 'f64': 78.6, 'f16x3': 157.3}     # dense MFMA peaks, MI355X_MICROARCH.md / datasheet ('f16x3': a secondary mode, priced as fp32-equivalent work)
HBM_PEAK_GBS = 8000.0

# BASELINE.json `configs` (C1..C5) and the headline metric (H): N, D, S (rank), M, compute dtype
CONFIGS = {
    'C1': (506, 13, 8, 64, 'f64', "Boston Housing shape (N=506, D=13), rank=8, M=64, fp64 -- plumbing/correctness"),
    'C2': (100000, 32, 16, 256, 'f64', "Synthetic N=100k, D=32, rank=16, M=256, fp64, 1xMI355X"),
    'C3': (1000000, 8, 32, 1024, 'f32', "kin8nm-shaped synthetic N=1e6, D=8, rank=32, M=1024, fp32 -- HBM-bound feature map"),
    'C4': (4000000, 64, 32, 1024, 'f32', "Synthetic N=4e6, D=64, rank=32, M=1024, fp32, row-sharded with RCCL all-reduce"),
    'C5': (1000000, 512, 64, 2048, 'f32', "High-dim synthetic N=1e6, D=512, rank=64, M=2048, fp32 -- MFMA-bound X.theta and Gram"),
    'H': (1000000, 64, 32, 1024, 'f32', "headline metric: N=1e6, D=64, rank=32, M=1024, fp32"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='H', choices=sorted(CONFIGS))
    ap.add_argument('--dtype', default=None)
    ap.add_argument('--rows', type=int, default=None)
    ap.add_argument('--D', type=int, default=None)
    ap.add_argument('--S', type=int, default=None)
    ap.add_argument('--M', type=int, default=None)
    ap.add_argument('--cpu-rows', type=int, default=100000)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the fp64 leg and the parity block')
    ap.add_argument('--triple', action='store_true', help='with a custom shape (--rows): still time the triple and the on-device loop (secondary.through_triple)')
    ap.add_argument('--backend', default='nccl', help="'gloo' rehearses the multi-rank flow on one GPU")
    ap.add_argument('--native-rccl', action='store_true',
                    help='the three sums by the library itself (scfgp_comm_init: ncclAllReduce on its own stream) instead of torch.distributed')
    ap.add_argument('--opt', action='append', default=[], metavar='NAME=VALUE',
                    help='scfgp_set_option on the benchmarked engine (tuning runs only; the bench line records it)')
    a = ap.parse_args(argv)
    N, D, S, M, dt, label = CONFIGS[a.config]
    a.custom = any(v is not None for v in (a.rows, a.D, a.S, a.M))
    a.rows = a.rows or N; a.D = a.D or D; a.S = a.S or S; a.M = a.M or M; a.dtype = a.dtype or dt
    a.label = label if not a.custom else "custom shape derived from " + a.config
    return a


def spawn_ranks(a, argv):
    """--gpus N > 1 outside torchrun: start the N ranks as a child process (this parent has made no GPU call)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, universal_newlines=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + '\n')
    if p.returncode == 0 and line is not None:
        out = json.loads(line)
        assert out['n_gpus'] == a.gpus, 'asked for %d ranks, the job ran %d' % (a.gpus, out['n_gpus'])
        print(line)
    return p.returncode if p.returncode else (0 if line else 1)


def build_problem(eng, N, D, S, M, lo, hi, allreduce):
    """Rows [lo,hi) of the synthetic problem; teacher response computed by the HIP predict path."""
    from scfgp_amd import synth
    K = 2 * (S + M)
    X = synth.make_X(SEED, hi - lo, D, row0=lo)
    teacher = synth.make_params(SEED + 0x0101, D, S, M, abc=(-1.0, 0.0, -1.0))
    eng.set_params(teacher)
    f, _ = eng.predict(X, synth.teacher_weights(SEED + 0x0303, K), np.eye(K))
    y = f.ravel() + 0.1 * synth.normal(SEED + 0x0404, lo, hi - lo)
    mom = np.array([y.sum(), (y * y).sum(), float(len(y))])
    if allreduce is not None:
        allreduce(mom)
    mean = mom[0] / mom[2]; std = np.sqrt(mom[1] / mom[2] - mean * mean)
    y = ((y - mean) / std).reshape(-1, 1)
    params = synth.make_params(SEED + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
    return X, y, params


def _blas_threads():
    try:
        import threadpoolctl
        pools = threadpoolctl.threadpool_info()
        mine = [p['num_threads'] for p in pools if 'numpy' in p.get('filepath', '')] or [p['num_threads'] for p in pools]
        return int(max(mine or [os.cpu_count()]))
    except Exception:
        return int(os.cpu_count())


def _host_topology():
    """(sockets, physical cores per socket, logical CPUs) of this host from /proc/cpuinfo; (1, cpu_count, cpu_count) when unreadable."""
    try:
        phys, cores = set(), set()
        pid = cid = None
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('physical id'):
                pid = ln.split(':')[1].strip()
            elif ln.startswith('core id'):
                cid = ln.split(':')[1].strip()
            elif not ln.strip() and pid is not None:
                phys.add(pid); cores.add((pid, cid)); pid = cid = None
        ns = max(len(phys), 1)
        return ns, max(len(cores) // ns, 1), int(os.cpu_count())
    except Exception:
        n = int(os.cpu_count())
        return 1, n, n


def cpu_baseline(X, y, params, S, M, N_full, budget_rows):
    """The reference's CPU path cannot run (Theano is absent), so its stand-ins are timed on the host cores of
    this box on bounded samples of the same workload (the first rows), scaled linearly in rows:
      primary    torch-CPU float64 autograd of the LITERAL graph of SCFGP/SCFGP.py:92-129 (oracle/autograd_ref.py: forward +
                 reverse sweep as TT.grad would execute it, GH-30 tensor included) on ALL torch threads of the box
      one_socket the same on the physical cores of one socket, threads_8 on 8 threads (this container's core count)
      also       the numpy 3-sweep oracle (oracle/scfgp_oracle.py::value_and_grad), BLAS-threaded GEMMs
    The linear scaling of the sample is checked once per round at the full 1e6 rows (tools/cpu_full.py ->
    profiles/r04_cpu_full.json)."""
    import torch
    from oracle import autograd_ref as AR
    from oracle import scfgp_oracle as O
    all_threads = int(torch.get_num_threads())
    sockets, per_socket, logical = _host_topology()

    def timed(threads, rows):
        n = min(rows, X.shape[0])
        Xs, ys = np.ascontiguousarray(X[:n]), np.ascontiguousarray(y[:n])
        torch.set_num_threads(threads)
        t0 = time.time()
        AR.value_and_grad(Xs, ys, params, S, M)
        dt = time.time() - t0
        torch.set_num_threads(all_threads)
        return {"value": n / float(N_full) / dt, "unit": "evals/s", "cores": int(threads), "kind": "port",
                "sample": "first %d of %d rows: %.1f s, scaled linearly in rows (the K^3 stage is not scaled down)" % (n, N_full, dt)}

    out = timed(all_threads, budget_rows)
    out["sample"] = ("torch-CPU float64 autograd of the literal reference graph (oracle/autograd_ref.py, stand-in for Theano's compiled "
                     "TT.grad, SCFGP/SCFGP.py:129) on the " + out["sample"])
    out["host"] = {"sockets": sockets, "physical_cores_per_socket": per_socket, "logical_cpus": logical}
    # smaller samples for the smaller thread counts, so that each stays ~10-20 s
    s1 = min(per_socket, all_threads)
    out["one_socket"] = timed(s1, max(budget_rows * s1 // max(all_threads, 1), budget_rows // 4))
    out["threads_8"] = timed(min(8, all_threads), max(budget_rows // 5, 1000))
    n = min(budget_rows, X.shape[0])
    t0 = time.time()
    O.value_and_grad(np.ascontiguousarray(X[:n]), np.ascontiguousarray(y[:n]), params, S, M, chunk=n)   # one chunk: the oracle's fastest setting
    dt_np = time.time() - t0
    out["also"] = {"value": n / float(N_full) / dt_np, "unit": "evals/s", "cores": _blas_threads(), "kind": "port",
                   "sample": "numpy 3-sweep oracle (oracle/scfgp_oracle.py value_and_grad; GEMMs on the BLAS threads in `cores`, "
                             "element-wise numpy on one) on the first %d rows: %.1f s" % (n, dt_np)}
    full = os.path.join(ROOT, 'profiles', 'r04_cpu_full.json')
    if os.path.exists(full):
        out["full_size_check"] = json.load(open(full))
    return out


def box_probe(device):
    """What this device delivers: fp32 MFMA TFLOP/s of a register-only loop, the clock it held, streaming copy GB/s."""
    import ctypes as C
    from scfgp_amd import _lib
    out = (C.c_double * 7)()
    rc = _lib.load().scfgp_box_probe(int(device), out, 7)
    if rc != 0:
        return {"error": rc}
    return {"mfma_f32_TFLOPs": out[0], "mfma_f32_frac_of_peak": out[0] / PEAK_TFLOPS['f32'], "mfma_clock_GHz": out[1],
            "mfma_f32_TFLOPs_at_1_2_8_waves_per_simd": [out[3], out[4], out[5]],
            "copy_GBs": out[2], "copy_frac_of_8TBs": out[2] / HBM_PEAK_GBS, "read_GBs": out[6], "read_frac_of_8TBs": out[6] / HBM_PEAK_GBS,
            "what": "scfgp_box_probe before the engine is created: register-only v_mfma_f32_16x16x4_f32 loop on random operands "
                    "(no memory traffic), shader clock held during it, 1 GiB -> 1 GiB streaming copy (read + write), read-only stream of the same 2 GiB"}


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


def _timed_leg(eng, steps, warmup):
    """median wall time per evaluation, median stage times and the outputs of the last evaluation"""
    from scfgp_amd.sharded import ShardedEvaluator
    ev = ShardedEvaluator(eng, None)
    for _ in range(warmup):
        ev.eval(True)
    eng.set_profiling(True)
    per_kernel, times, out = {}, [], None
    for _ in range(steps):
        t0 = time.perf_counter()
        out = ev.eval(True)
        times.append(time.perf_counter() - t0)
        for name, ms in eng.timings():
            per_kernel.setdefault(name, []).append(ms)
    eng.set_profiling(False)
    return float(np.median(times)) * 1e3, {k: float(np.median(v)) for k, v in per_kernel.items()}, out


def _parity(out, ref, eng, eng_ref, D, S, M, what):
    from scfgp_amd import synth
    c, g, a, L = out
    c64, g64, a64, L64 = ref
    o = 3 + D * S
    Xs = synth.make_X(SEED + 0x0909, 4096, D)
    mu64, sd64 = eng_ref.predict(Xs, a64, L64)
    mu, sd = eng.predict(Xs, a, L)
    return {"what": what, "cost": abs(float(c) - float(c64)) / abs(float(c64)),
            "grad_abc": rel(g[:3], g64[:3]), "grad_lF": rel(g[3:o], g64[3:o]), "grad_rF": rel(g[o:o + M * S], g64[o:o + M * S]),
            "alpha": rel(a, a64), "Li": rel(L, L64), "mu": rel(mu, mu64), "std": rel(sd, sd64)}


def through_triple(X, y, params, D, S, M, local, n=3):
    """One train_iter_func call of the reference-shaped triple (scfgp_amd/funcs.py: residency check of X and y, evaluation,
    host update rule, parameter upload) -- what a user of SCFGP.optimize pays per iteration on top of `value`'s bare
    evaluation.  `read_only_arrays`: frozen arrays that own their memory, which is what SCFGP.set_data hands to the triple
    (model.py freezes its copies): checked by identity, nothing hashed.  `writeable_arrays`: a caller's own arrays passed to the
    triple directly: every byte hashed on every call (the reference re-reads its arguments on every call, SCFGP/SCFGP.py:237)."""
    from scfgp_amd.funcs import CompiledFuncs
    cf = CompiledFuncs(D, S, M, params.copy(), 'adam', {'learning_rate': 1e-3}, dtype='f32', device=local)
    res = {}
    Xf = np.array(X); yf = np.array(y)                          # the model's own copies (SCFGP.set_data)
    Xf.flags.writeable = False; yf.flags.writeable = False
    for tag, (Xc, yc) in (('writeable_arrays', (X, y)), ('read_only_arrays', (Xf, yf))):
        cf.train_iter_func(Xc, yc)                                  # upload + first touch
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            cf.train_iter_func(Xc, yc)
            ts.append(time.perf_counter() - t0)
        res[tag] = {"ms_per_call": float(np.median(ts)) * 1e3}
    cf.engine.close()
    res["note"] = ("CompiledFuncs.train_iter_func (host adam + Nesterov), median of %d calls; never `value`.  read_only_arrays is the "
                   "drop-in path (SCFGP.optimize on the arrays SCFGP.set_data froze); writeable_arrays pays a full hash per call" % n)
    # the on-device loop (scfgp_train: evaluation + update rule, no host between iterations): the captured graph, and -- what a
    # rank of a row-sharded job runs -- eager launches with the three sums inside the library (a one-rank communicator here)
    from scfgp_amd.engine import HipEngine
    iters = 8
    for tag, comm in (('device_loop', False), ('device_loop_sums_inside', True)):
        try:
            e = HipEngine(D, S, M, dtype='f32', device=local)
            e.set_params(params); e.set_data(X, y, n_global=X.shape[0])
            if comm:
                e.comm_init(1, 0, e.comm_unique_id())
            e.opt_init('adam', learning_rate=1e-3)
            e.train(2, want_factors=False)
            t0 = time.perf_counter()
            e.train(iters, want_factors=True)
            res[tag] = {"ms_per_iteration": (time.perf_counter() - t0) * 1e3 / iters, "iterations_per_call": iters}
            e.close()
        except Exception as ex:                                      # e.g. no librccl on the box
            res[tag] = {"error": repr(ex)}
    return res


def f64_leg_and_parity(X, y, params, D, S, M, local, f32_out, f32_eng, steps=5, warmup=2):
    """The reference's arithmetic is float64: run the fp64 engine on the SAME resident rows, time it, and report how far
    the fp32-mode outputs are from it at this size (relative, norm-wise; gradient per block)."""
    from scfgp_amd.engine import HipEngine
    N = X.shape[0]; J = S + M; K = 2 * J
    e64 = HipEngine(D, S, M, dtype='f64', device=local)
    e64.set_params(params); e64.set_data(X, y, n_global=N)
    ms, stages, ref = _timed_leg(e64, steps, warmup)
    ap_ms = float(np.median([stages.get('apply_v', 0), stages.get('apply_phibar', 0)]))
    ach = 2.0 * N * K * K / (ap_ms * 1e-3) / 1e12 if ap_ms > 0 else 0.0
    sec = {"evals_per_s": 1e3 / ms, "ms_per_step": ms, "steps": steps, "statistic": "median",
           "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS['f64'], "unit": "TFLOP/s",
                        "frac": ach / PEAK_TFLOPS['f64'], "kernel": "apply_dma_kernel<double> + 64-wide apply_kernel remainder (fp64 MFMA 16x16x4)",
                        "avg_launch_ms": ap_ms},
           "stages_ms": stages, "cost": float(ref[0])}
    base = "mode of this library on the same %d rows (fp64 mode equals the oracle to 1e-12 wherever the oracle is run: " \
           "tests/test_gpu_parity.py); relative, norm-wise" % N
    parity = _parity(f32_out, ref, f32_eng, e64, D, S, M, "fp32 mode (precision level %d) vs fp64 " % int(f32_eng.condition()['level']) + base)
    plain = None
    if f32_eng.condition()['level'] > 0:
        # the headline engine escalated its Gram products (ill-conditioned A): the same rows in PLAIN fp32 mode (option gram64 = 0),
        # its rate and how far its outputs are from fp64 mode -- both modes in one line
        ep = HipEngine(D, S, M, dtype='f32', device=local)
        ep.set_option('gram64', 0)
        ep.set_params(params); ep.set_data(X, y, n_global=N)
        msp, stp, outp = _timed_leg(ep, steps, warmup)
        cdp = ep.condition()
        plain = {"what": "fp32 mode with the precision escalation switched off (option gram64 = 0): every product in fp32 MFMA",
                 "evals_per_s": 1e3 / msp, "ms_per_step": msp, "steps": steps, "statistic": "median", "stages_ms": stp,
                 "cost": float(outp[0]), "cond_est": cdp['cond_est'], "alpha_err_predicted": cdp['alpha_err_fp32'],
                 "parity_at_size": _parity(outp, ref, ep, e64, D, S, M, "plain fp32 (gram64 = 0) vs fp64 " + base)}
        ep.close()
    # SECONDARY mode f16x3 (include/scfgp_hip.h: SCFGP_F16X3): fp32 mode whose two square apply products and two Gram products run as a
    # three-term fp16 split on the fp16 matrix pipe.  Same rows, its own parity block against fp64 mode; never `value`, never `dtype: f32`
    f16 = None
    try:
        e16 = HipEngine(D, S, M, dtype='f16x3', device=local)
        e16.set_params(params); e16.set_data(X, y, n_global=N)
        ms16, st16, out16 = _timed_leg(e16, steps, warmup)
        present = lambda names: float(np.median([st16[k] for k in names if st16.get(k, 0) > 0] or [0.0]))   # precision levels 1 / 2 replace stages
        lvl16 = int(e16.condition()['level'])                      # level 1: pass 1's Gram is fp64; level 2: factor form, only Phibar's product is split
        ap16, gr16 = present(('apply_v', 'apply_phibar')), present({0: ('gram', 'gram_w'), 1: ('gram_w',)}.get(lvl16, ()))
        n256 = (K // 128) // 2                                     # the 256-wide tiles run the split; the ragged remainder stays exact fp32
        fl16 = 3.0 * (2.0 * N * (256.0 * n256) * K)                 # three v_mfma_f32_16x16x32_f16 per tile and 32 k: 3 x the product's flops
        nts = -(-K // 128)                                           # the Gram multiplies the lower 128 x 128 tiles (diagonal ones whole)
        flg = 3.0 * 2.0 * N * 16384.0 * (nts * (nts + 1) // 2)
        tf = lambda fl, ms: fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        f16 = {"what": "fp32 mode with V = Phi B, Phibar = 2 Phi Abar + ..., G = Phi^T Phi and V^T diag(q) V as a three-term fp16 split (h.h + l.h + "
                       "h.l, fp32 accumulators, power-of-two operand scales; scfgp_amd/csrc/apply_f16.hip, gram_f16.hip); feature map, row statistics, "
                       "K x K stage, backward projection and everything else exact fp32 mode",
               "evals_per_s": 1e3 / ms16, "ms_per_step": ms16, "steps": steps, "statistic": "median", "stages_ms": st16,
               "cost": float(out16[0]), "precision_level": lvl16,
               "roofline": {"bound": "mfma", "achieved": tf(fl16, ap16), "peak": 2500.0,
                            "unit": "TFLOP/s (fp16 MFMA, executed: three instructions per output tile and 32 k)",
                            "frac": tf(fl16, ap16) / 2500.0, "avg_launch_ms": ap16,
                            "fp32_equivalent_TFLOPs": tf(2.0 * N * K * K, ap16),
                            "kernel": "apply_f16_kernel (256-wide tiles) + the exact-fp32 64-wide remainder of the same product",
                            "note": "the peak is the datasheet's at 2.4 GHz; under this kernel the clock settles at 1.5-1.6 GHz with the matrix pipe "
                                    "0.74-0.76 busy (busy x clock = 0.48-0.51 of the peak), under the Gram's at 1.7 GHz with 0.56 "
                                    "(profiles/r05_f16x3_pmc.txt): the fp16 pipe at this rate is power-bound"},
               "roofline_gram": None if gr16 <= 0 else {"bound": "mfma", "achieved": tf(flg, gr16), "peak": 2500.0,
                                 "unit": "TFLOP/s (fp16 MFMA, executed: three instructions per output tile and 32 rows over the lower 128 x 128 "
                                         "tiles of the Kp x Kp result, diagonal tiles whole)",
                                 "frac": tf(flg, gr16) / 2500.0, "avg_launch_ms": gr16, "fp32_equivalent_TFLOPs": tf(1.0 * N * K * K, gr16),
                                 "kernel": "gram_f16_kernel"},
               "parity_at_size": _parity(out16, ref, e16, e64, D, S, M, "f16x3 mode vs fp64 " + base)}
        e16.close()
    except Exception as ex:
        f16 = {"error": repr(ex)}
    e64.close()
    return sec, parity, plain, f16


def main(a):
    import torch
    from scfgp_amd.engine import HipEngine
    from scfgp_amd.sharded import ShardedEvaluator, attach_native_comm, shard_rows, torch_allreduce
    # stdout carries exactly ONE line, rank 0's JSON: anything libraries print to fd 1 meanwhile (RCCL's version banner
    # at communicator creation, for one) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0)) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    allreduce = None
    use_dist = world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ     # under torchrun even one rank goes through RCCL
    if use_dist:
        import torch.distributed as dist
        if a.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(a.backend)
        allreduce = torch_allreduce()
    N, D, S, M = a.rows, a.D, a.S, a.M
    J = S + M; K = 2 * J
    lo, hi = shard_rows(N, rank, world)
    # the probe runs before the engine exists (its own stream, its 2 GiB freed again); building the problem below takes seconds
    # of host work, so the chip's clock and thermal state have settled again when the warm-up starts
    box = box_probe(local) if rank == 0 else None
    eng = HipEngine(D, S, M, dtype=a.dtype, device=local, stream=torch.cuda.current_stream().cuda_stream)
    for kv in a.opt:
        eng.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    X, y, params = build_problem(eng, N, D, S, M, lo, hi, allreduce)
    eng.set_params(params)
    eng.set_data(X, y, n_global=N)
    native = bool(a.native_rccl or os.environ.get('SCFGP_NATIVE_RCCL') == '1') and use_dist and a.backend == 'nccl'
    if native:
        attach_native_comm(eng)                                # from here on the stage calls sum over the ranks themselves
    ev = ShardedEvaluator(eng, None if native else allreduce, time_exchanges=use_dist and not native)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        ev.eval(True)
    eng.set_profiling(True)
    per_kernel, step_s = {}, []
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        t1 = time.perf_counter()
        cost, grad, alpha, Li = ev.eval(True)                  # synchronous: results are on the host on return
        step_s.append(time.perf_counter() - t1)
        for name, ms in eng.timings():
            per_kernel.setdefault(name, []).append(ms)
        for st_, ms in ev.exchange_ms().items():
            per_kernel.setdefault('exchange%d' % st_, []).append(ms)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt, float(np.median(step_s))], device='cuda' if a.backend == 'nccl' else 'cpu', dtype=torch.float64)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt, med_s = float(tmax[0].item()), float(tmax[1].item())
    # slowest rank per stage: the same code path on every rank gives the same stage names in the same order
    stage_names = sorted(per_kernel)
    stage_max = None
    if use_dist and world > 1:
        smax = torch.tensor([float(np.median(per_kernel[k])) for k in stage_names], device=tmax.device, dtype=torch.float64)
        dist.all_reduce(smax, op=dist.ReduceOp.MAX)
        stage_max = dict(zip(stage_names, [float(v) for v in smax.tolist()]))
    cond = eng.condition()

    if rank == 0:
        ms_step = dt / a.steps * 1e3
        med = lambda k: float(np.median(per_kernel.get(k, [0])))
        # dominant kernel: the N x K x K "apply" product (Phi.B and Phi.Abar use the same kernel);
        # algorithmic flops per launch = 2 * rows_on_this_rank * K^2   
        ap_ms = float(np.median(per_kernel.get('apply_v', [0]) + per_kernel.get('apply_phibar', [0])))
        flops = 2.0 * (hi - lo) * K * K
        ach = flops / (ap_ms * 1e-3) / 1e12 if ap_ms > 0 else 0.0
        peak = PEAK_TFLOPS[a.dtype]
        falg = 10.0 * N * K * K + 4.0 * N * D * J
        traffic, traffic_src, pmc = None, None, {}
        tp = next((f for f in (os.path.join(ROOT, 'profiles', 'r%02d_pmc_traffic.json' % r) for r in (5, 4, 3, 2)) if os.path.exists(f)), '')
        if os.path.exists(tp) and not a.custom and world == 1:
            # HBM-side bytes per launch of the same kernels on the same workload, from rocprofv3 PMC passes
            # (FETCH_SIZE x2 + WRITE_SIZE, one counter per pass): they cannot be collected inside this process
            pmc = json.load(open(tp)).get(a.config + '_' + a.dtype, {})
            if 'apply_kernel_mean_GB_per_launch' in pmc:
                traffic = pmc['apply_kernel_mean_GB_per_launch'] * 1e9
                traffic_src = os.path.relpath(tp, ROOT)
        out = {
            "metric": "NLML+grad evals/sec", "value": a.steps / dt, "unit": "evals/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "ms_per_step_median": med_s * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "%s: %s -- N=%d D=%d S=%d M=%d (K=%d), resident rows, cost+grad+alpha+Li to host"
                                   % (a.config, a.label, N, D, S, M, K),
                       "rows_per_gpu": hi - lo, "parallelism": "row-sharded dp%d, 3 all-reduces/eval%s" % (
                           world, " (ncclAllReduce issued by the library, scfgp_comm_init)" if native else " (torch.distributed)"),
                       "F_alg_per_eval": falg, "F_alg_TFLOPs": falg / (dt / a.steps) / 1e12, "options": a.opt},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "apply product (Phi.B / Phi.Abar): one N x K x K product = 2*N*K^2 flops, issued as one launch for the "
                                   "full column tiles (at this size: apply_dma_kernel, fp32 256x256 tiles / fp64 256x128 tiles, for both products; "
                                   "otherwise apply_kernel, 256x128 tiles) plus a 256x64-tile launch of the same kernel for the ragged remainder "
                                   "(and, fp32, 128-wide tiles for the row blocks of a thin last round); "
                                   "avg_launch_ms is the median hipEvent time of that group (rocprof: sum of its kernels)",
                         "avg_launch_ms": ap_ms},
            "stages_ms": {k: float(np.median(v)) for k, v in per_kernel.items()},
            "stages_statistic": "median over the timed steps",
            "cost": float(cost),
            # conditioning of A and the precision level the evaluations ran at (include/scfgp_hip.h: scfgp_get_condition)
            "condition": {"cond_est": cond['cond_est'], "precision_level": int(cond['level']), "gram_fp64": bool(cond['gram_fp64']),
                          "alpha_err_if_plain_fp32": cond['alpha_err_fp32'], "thresholds": [cond['threshold'], cond['threshold_w']]},
        }
        if world > 1:
            ex = sum(float(np.median(per_kernel.get('exchange%d' % i, [0.0]))) for i in (1, 2, 3))
            out["comm_share"] = ex / ms_step if ms_step > 0 else 0.0
            out["comm_note"] = ("exchange1..3 in stages_ms: stream time of each sum over ranks, from the end of this rank's sweep to the "
                                "summed buffer (transfer + wait for the slowest rank); comm_share = their sum / ms_per_step, rank 0")
            out["stages_ms_max_over_ranks"] = stage_max
        # the two other figures the north star asks for: the feature map against the HBM roof (it writes Phi once:
        # rows x K x element size, SURVEY 8(d) kernel K3) and the Gram build against the MFMA peak (executed flops:
        # lower triangle in 64-column blocks, so about 1.06 x N K^2 rather than the algorithmic 2 N K^2)
        esz = 8 if a.dtype == 'f64' else 4
        fm_ms, gr_ms, xz_ms = med('featuremap'), med('gram'), med('xtz')
        b64 = -(-K // 64); nf = b64 // 2
        gram_exec = 2.0 * (hi - lo) * ((nf * (nf + 1) // 2) * 128 * 128 + (b64 % 2) * (nf + 1) * 64 * 128)
        out["secondary"] = {
            "featuremap": {"bound": "hbm", "achieved": (hi - lo) * K * esz / (fm_ms * 1e-3) / 1e9 if fm_ms > 0 else 0.0,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": fm_ms,
                           "rocprof_WRITE_SIZE_GBs": pmc.get('featuremap_WRITE_SIZE_GBs'),
                           "note": "Phi written once / time of project+featuremap kernels (hipEvents); rocprof_WRITE_SIZE_GBs is the "
                                   "same quantity from the rocprofv3 WRITE_SIZE counter (the traffic_source file of `roofline`)"},
            "gram": {"bound": "mfma", "achieved": gram_exec / (gr_ms * 1e-3) / 1e12 if gr_ms > 0 else 0.0, "peak": peak,
                     "unit": "TFLOP/s (executed)", "avg_launch_ms": gr_ms,
                     "MFMA_BUSY": pmc.get('gram_MFMA_BUSY')},
            "xtz": {"bound": "mfma", "achieved": 2.0 * (hi - lo) * (D + 1) * J / (xz_ms * 1e-3) / 1e12 if xz_ms > 0 else 0.0, "peak": peak,
                    "unit": "TFLOP/s", "avg_launch_ms": xz_ms, "share_of_step": xz_ms / ms_step if ms_step > 0 else 0.0},
        }
        for v in out["secondary"].values():
            v["frac"] = v["achieved"] / v["peak"]
        out["secondary"]["box"] = box
        if world == 1 and not a.no_secondary:
            # the same evaluation when the caller hands over HOST arrays every call (what the reference's Theano functions
            # receive, SCFGP/SCFGP.py:237): upload of X, y over PCIe from pageable memory + packing + evaluation.  Never `value`.
            ts = []
            for _ in range(3):
                t1 = time.perf_counter()
                eng.eval(X, y, want_grad=True)
                ts.append(time.perf_counter() - t1)
            out["secondary"]["pcie_inclusive"] = {"evals_per_s": 1.0 / float(np.median(ts)), "ms_per_step": float(np.median(ts)) * 1e3,
                                                  "host_bytes_per_call": int(X.nbytes + y.nbytes),
                                                  "note": "scfgp_eval with host X, y on every call (median of 3); resident-data rate is `value`"}
        # the fp64 leg and the fp32-vs-fp64 parity block: at the headline shape and at the two other 1e6-row fp32 configs (C3's
        # D = 8 makes A ill-conditioned: that is where fp32 products cost the most accuracy, and the line says so)
        if world == 1 and a.config in ('H', 'C3', 'C5') and a.dtype == 'f32' and not a.custom and not a.no_secondary:
            out["secondary"]["f64"], out["parity_at_size"], plain, f16 = f64_leg_and_parity(
                X, y, params, D, S, M, local, (cost, grad, alpha, Li), eng)
            out["secondary"]["f16x3"] = f16
            if plain is not None:
                out["secondary"]["plain_fp32"] = plain
        if world == 1 and not a.no_secondary and a.dtype == 'f32' and (not a.custom or a.triple):
            out["secondary"]["through_triple"] = through_triple(X, y, params, D, S, M, local)
        if not a.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(X, y, params, S, M, N, a.cpu_rows)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == '__main__':
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    if 'WORLD_SIZE' in os.environ and int(os.environ['WORLD_SIZE']) != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%s; the launcher decides the rank count\n'
                         % (args.gpus, os.environ['WORLD_SIZE']))
        sys.exit(2)
    main(args)
