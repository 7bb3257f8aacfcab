"""One evaluation of the CPU stand-in at the FULL headline size (N = 1e6, D = 64, S = 32, M = 1024), row-chunked so that the
N x K tensors fit in host memory (oracle/autograd_ref.py: value_and_grad_chunked -- the literal graph, every derivative autograd's),
and the same code on the first 1e5 rows: checks once that bench.py's `cpu_baseline` (1e5-row sample scaled x10) is a fair
stand-in.  Writes one JSON object to stdout (commit it as profiles/rNN_cpu_full.json).   python tools/cpu_full.py [rows]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from oracle import autograd_ref as AR
from scfgp_amd import synth

N, D, S, M = bench.CONFIGS['H'][:4]
if len(sys.argv) > 1:
    N = int(sys.argv[1])
X = synth.make_X(bench.SEED, N, D)
y = synth.normal(bench.SEED + 9, 0, N).reshape(-1, 1)
params = synth.make_params(bench.SEED + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
res = {"what": "oracle/autograd_ref.py value_and_grad_chunked (literal graph of SCFGP/SCFGP.py:92-129, float64, torch CPU), chunk 65536 rows",
       "threads": int(torch.get_num_threads()), "N": N, "D": D, "S": S, "M": M}
n = min(100000, N)
t0 = time.time(); AR.value_and_grad(X[:n], y[:n], params, S, M); dt_s = time.time() - t0
t0 = time.time(); c_s = AR.value_and_grad_chunked(X[:n], y[:n], params, S, M)[0]; dt_sc = time.time() - t0
t0 = time.time(); c, g, a, L = AR.value_and_grad_chunked(X, y, params, S, M); dt = time.time() - t0
res.update({"sample_rows": n, "sample_unchunked_s": dt_s, "sample_chunked_s": dt_sc, "full_chunked_s": dt, "cost": c,
            "evals_per_s_full": 1.0 / dt, "evals_per_s_from_sample_scaled": n / float(N) / dt_s,
            "full_over_scaled_sample": dt / (dt_s * N / n)})
print(json.dumps(res))
