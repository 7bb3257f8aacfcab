"""
HipEngine: thin, typed Python face of one scfgp_ctx (one GPU, one shard of rows).

It adds nothing numerically -- every method is one call through the C ABI of
libscfgp_hip.so (include/scfgp_hip.h) with numpy buffers allocated for the caller,
and turns error codes into the exceptions the reference raises
(numpy.linalg.LinAlgError for a failed Cholesky, SCFGP/SCFGP.py:106).
"""
import ctypes as C
import sys

import numpy as np

from . import _lib
from ._lib import SCFGP_F16X3, SCFGP_F32, SCFGP_F64, dptr

# 'f16x3': fp32 mode whose four N-sized products (apply and Gram) run as a three-term fp16 split (include/scfgp_hip.h: SCFGP_F16X3) -- a
# labelled secondary mode, chosen explicitly, never a default
_DTYPES = {'f64': SCFGP_F64, 'float64': SCFGP_F64, 'f32': SCFGP_F32, 'float32': SCFGP_F32, 'f16x3': SCFGP_F16X3,
           SCFGP_F64: SCFGP_F64, SCFGP_F32: SCFGP_F32, SCFGP_F16X3: SCFGP_F16X3}


def num_params(D, S, M):
    """Length of the flat hyper-parameter vector (SCFGP/SCFGP.py:72)."""
    return 3 + D * S + M * S + S + M


class PeerFailed(RuntimeError):
    """Row shards: another rank failed in this evaluation (SCFGP_EPEER); no rank's results are valid."""


class HipEngine(object):

    def __init__(self, D, S, M, dtype='f64', device=0, stream=None):
        self.lib = _lib.load()
        self.D, self.S, self.M = int(D), int(S), int(M)
        self.J = self.S + self.M
        self.K = 2 * self.J
        self.P = num_params(D, S, M)
        self.dtype = _DTYPES[dtype]
        self.device = int(device)
        self.ctx = C.c_void_p()
        self._pool = {}
        rc = self.lib.scfgp_create(C.byref(self.ctx), self.D, self.S, self.M, self.dtype, int(device),
                                   C.c_void_p(stream) if stream else None)
        if rc != 0:
            # a failed create still hands back the half-built context (for its error message): it must be destroyed
            msg = self.lib.scfgp_last_error(self.ctx).decode() if self.ctx else ''
            self.close()
            raise RuntimeError('scfgp_create failed (%d): %s' % (rc, msg))
        self.N = 0
        self.n_global = 0

    _pool = None
    nonfinite = 'raise'            # 'return': a NaN / Inf cost comes back as a value (what the reference's functions do)

    def close(self):
        if getattr(self, 'ctx', None):
            self.lib.scfgp_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- error mapping ----------------------------------------------------------------
    def _check(self, rc, what):
        if rc == 0:
            return
        msg = self.lib.scfgp_last_error(self.ctx).decode()
        if rc == -3:
            raise np.linalg.LinAlgError('%s: %s' % (what, msg))
        if rc == -4:
            # the outputs have been delivered; the reference's Theano functions do not trap a NaN / Inf cost (SCFGP/SCFGP.py:249-258
            # treats it as "no improvement"), so the callable triple asks for it back (funcs.py), a direct caller gets the error
            if self.nonfinite == 'return':
                return
            raise FloatingPointError('%s: %s' % (what, msg))
        if rc == -1:
            raise ValueError('%s: %s' % (what, msg))
        if rc == _lib.SCFGP_EPEER:
            raise PeerFailed('%s: %s' % (what, msg))
        raise RuntimeError('%s failed (%d, %s): %s' % (what, rc, _lib.ERRORS.get(rc, '?'), msg))

    # -- state --------------------------------------------------------------------------
    def set_params(self, params):
        p = np.ascontiguousarray(params, dtype=np.float64).ravel()
        if p.size != self.P:
            raise ValueError('expected %d parameters, got %d' % (self.P, p.size))
        self._check(self.lib.scfgp_set_params(self.ctx, dptr(p), self.P), 'set_params')

    def get_params(self):
        p = np.empty(self.P)
        self._check(self.lib.scfgp_get_params(self.ctx, dptr(p), self.P), 'get_params')
        return p

    @staticmethod
    def _check_xy(X, y, D):
        # Theano's dmatrix inputs reject anything but 2-d float64 (SCFGP/SCFGP.py:95)
        if not isinstance(X, np.ndarray) or X.dtype != np.float64 or X.ndim != 2:
            raise TypeError('X must be a 2-d float64 ndarray')
        if X.shape[1] != D:
            raise ValueError('X has %d columns, expected %d' % (X.shape[1], D))
        if y is not None:
            if not isinstance(y, np.ndarray) or y.dtype != np.float64 or y.ndim != 2 or y.shape != (X.shape[0], 1):
                raise TypeError('y must be a float64 ndarray of shape (N,1)')

    def set_data(self, X, y, n_global=None):
        self._check_xy(X, y, self.D)
        X = np.ascontiguousarray(X); y = np.ascontiguousarray(y)
        N = X.shape[0]
        self.N = N
        self.n_global = int(n_global) if n_global else N
        self._check(self.lib.scfgp_set_data(self.ctx, dptr(X), dptr(y), N, self.n_global), 'set_data')

    # -- whole evaluations ------------------------------------------------------------------
    def _fresh(self, shape):
        """An output array nobody else holds.  The reference's functions return freshly allocated arrays on every call; a
        fresh K x K float64 array (35.7 MB at K = 2112) costs the host ~1.4 ms per call in page faults and their release
        (measured against reused buffers, tests/gpu_wall.py).  So arrays handed out earlier are kept in a small pool and one
        is used again only once the caller has dropped every reference to it (views hold one through .base): from the
        caller's side it cannot be told from a new array."""
        pool = self._pool.setdefault(shape, [])
        for i in range(len(pool)):
            if sys.getrefcount(pool[i]) == 2:                  # the pool's reference + getrefcount's argument
                return pool[i]
        a = np.empty(shape)
        if len(pool) < 4:
            pool.append(a)
        return a

    def _outputs(self, want_grad, factors=True):
        cost = np.zeros(1)
        grad = np.empty(self.P) if want_grad else None
        alpha = self._fresh((self.K, 1)) if factors else None
        Li = self._fresh((self.K, self.K)) if factors else None
        return cost, grad, alpha, Li

    def eval(self, X=None, y=None, want_grad=True):
        """(cost 0-d, grad|None, alpha (K,1), Li (K,K)); X=None evaluates the resident rows."""
        cost, grad, alpha, Li = self._outputs(want_grad)
        if X is not None:
            self._check_xy(X, y, self.D)
            X = np.ascontiguousarray(X); y = np.ascontiguousarray(y)
            self.N = X.shape[0]; self.n_global = self.N
        rc = self.lib.scfgp_eval(self.ctx, dptr(X), dptr(y), 0 if X is None else X.shape[0], int(bool(want_grad)),
                                 dptr(cost), dptr(grad), dptr(alpha), dptr(Li))
        self._check(rc, 'eval')
        return cost.reshape(()), grad, alpha, Li

    def eval_rows(self, idx, want_grad=True):
        """Evaluation on the rows `idx` of the resident data set (device-side gather; per-batch N)."""
        idx = np.ascontiguousarray(idx, dtype=np.int64).ravel()
        cost, grad, alpha, Li = self._outputs(want_grad)
        rc = self.lib.scfgp_eval_rows(self.ctx, idx.ctypes.data_as(C.POINTER(C.c_int64)), idx.size, int(bool(want_grad)),
                                      dptr(cost), dptr(grad), dptr(alpha), dptr(Li))
        self._check(rc, 'eval_rows')
        return cost.reshape(()), grad, alpha, Li

    def predict(self, Xs, alpha, Li):
        """pred_func (SCFGP/SCFGP.py:138-148): returns mu (T,1), std (T,)."""
        self._check_xy(Xs, None, self.D)
        Xs = np.ascontiguousarray(Xs)
        alpha = np.ascontiguousarray(alpha, dtype=np.float64).reshape(-1)
        Li = np.ascontiguousarray(Li, dtype=np.float64)
        if alpha.size != self.K or Li.shape != (self.K, self.K):
            raise ValueError('alpha/Li have the wrong shape for K=%d' % self.K)
        T = Xs.shape[0]
        mu = np.empty((T, 1)); sd = np.empty(T)
        self._check(self.lib.scfgp_predict(self.ctx, dptr(Xs), T, dptr(alpha), dptr(Li), dptr(mu), dptr(sd)), 'predict')
        return mu, sd

    @staticmethod
    def scaler_key(scaler):
        """Contents of a fitted Scaler (it is refitted IN PLACE by SCFGP.set_data, so identity says nothing)."""
        d = scaler.data
        return (scaler.algo,) + tuple((k, np.asarray(d[k], dtype=np.float64).tobytes()) for k in sorted(d) if np.size(d[k]))

    SCALER_MODES = {'min-max': 1, 'normal': 2, 'inv-normal': 3, 'auto-normal': 4, 'auto-inv-normal': 5}

    def set_x_scaler(self, scaler):
        """Register a fitted scfgp_amd.scaler.Scaler so predict_raw can transform inputs on the device."""
        d = scaler.data
        arr = lambda k: np.ascontiguousarray(d[k], dtype=np.float64) if k in d and np.ndim(d[k]) else None
        bufs = [arr(k) for k in ('min', 'max', 'boxcox', 'mu', 'std')]
        for b in bufs:
            if b is not None and b.size != self.D:
                raise ValueError('scaler was fitted on %d columns, engine has D=%d' % (b.size, self.D))
        self._check(self.lib.scfgp_set_x_scaler(self.ctx, self.SCALER_MODES[scaler.algo], *[dptr(b) for b in bufs]),
                    'set_x_scaler')
        self._xcols = list(d['cols'])

    def predict_raw(self, Xs_raw, alpha, Li):
        """pred_func on UNSCALED inputs: column selection here, the element-wise transform on the GPU."""
        Xs = np.ascontiguousarray(np.asarray(Xs_raw, dtype=np.float64)[:, self._xcols])
        alpha = np.ascontiguousarray(alpha, dtype=np.float64).reshape(-1)
        Li = np.ascontiguousarray(Li, dtype=np.float64)
        T = Xs.shape[0]
        mu = np.empty((T, 1)); sd = np.empty(T)
        self._check(self.lib.scfgp_predict_raw(self.ctx, dptr(Xs), T, dptr(alpha), dptr(Li), dptr(mu), dptr(sd)), 'predict_raw')
        return mu, sd

    METRICS = ('MAE', 'NMAE', 'MSE', 'NMSE', 'MNLP', 'SCORE')

    def set_y_scaler(self, scaler):
        """Register the fitted single-column target scaler for predict_y."""
        d = scaler.data
        val = lambda k: float(np.asarray(d[k]).reshape(-1)[0]) if k in d and np.size(d[k]) else 0.0
        if len(d['cols']) != 1:
            raise ValueError('the y scaler must have exactly one column')
        self._check(self.lib.scfgp_set_y_scaler(self.ctx, self.SCALER_MODES[scaler.algo],
                                                *[val(k) for k in ('min', 'max', 'boxcox', 'mu', 'std')]), 'set_y_scaler')

    def predict_y(self, Xs_raw, alpha, Li, ys=None):
        """SCFGP.predict (SCFGP/SCFGP.py:278-294) entirely on the device: X scaling, pred_func, y-scaler backward
        transform of the mean and of the +-1 std band and, with raw targets ys, the six metrics.
        Returns mu_y (T,1), std_y (T,1), metrics dict or None."""
        Xs = np.asarray(Xs_raw, dtype=np.float64)
        if getattr(self, '_xcols', None) is not None:
            Xs = Xs[:, self._xcols]
        Xs = np.ascontiguousarray(Xs)
        alpha = np.ascontiguousarray(alpha, dtype=np.float64).reshape(-1)
        Li = np.ascontiguousarray(Li, dtype=np.float64)
        T = Xs.shape[0]
        mu = np.empty((T, 1)); sd = np.empty((T, 1)); met = np.empty(6)
        if ys is not None:
            ys = np.ascontiguousarray(ys, dtype=np.float64).reshape(-1)
            if ys.size != T:
                raise ValueError('ys has %d entries for %d test rows' % (ys.size, T))
        self._check(self.lib.scfgp_predict_y(self.ctx, dptr(Xs), T, dptr(alpha), dptr(Li), dptr(ys), dptr(mu), dptr(sd),
                                             dptr(met) if ys is not None else None), 'predict_y')
        return mu, sd, (dict(zip(self.METRICS, met.tolist())) if ys is not None else None)

    # -- staged evaluation (row-sharded data parallelism) --------------------------------------
    def pass1(self):
        self._check(self.lib.scfgp_pass1(self.ctx), 'pass1')

    def factor(self):
        """False when the library asks for the stages again from pass1 (SCFGP_REDO: the ranks have just agreed on a lower
        precision level than some of them ran pass 1 at -- every rank gets the same answer); True otherwise."""
        rc = self.lib.scfgp_factor(self.ctx)
        self._early = None
        if rc == _lib.SCFGP_REDO:
            return False
        self._check(rc, 'factor')
        return True

    def fail_stage(self, stage, want_grad=True):
        """This rank cannot compute sweep `stage` (1..3) of the evaluation in progress: mark its exchange buffer failed and, with
        a communicator attached, run the sum -- so that the peers reach their finish() and raise PeerFailed there instead of
        waiting in a collective (include/scfgp_hip.h, "ranks decide together")."""
        self._check(self.lib.scfgp_fail_stage(self.ctx, int(stage), int(bool(want_grad))), 'fail_stage')

    def fetch_factors(self):
        """Fetch alpha / Li as soon as the factor stage is done (overlaps the sweeps already queued); finish() returns them."""
        alpha = self._fresh((self.K, 1)); Li = self._fresh((self.K, self.K))
        self._check(self.lib.scfgp_fetch_factors(self.ctx, dptr(alpha), dptr(Li)), 'fetch_factors')
        self._early = (alpha, Li)

    def pass2(self, want_grad=True):
        self._check(self.lib.scfgp_pass2(self.ctx, int(bool(want_grad))), 'pass2')

    def adjoint(self):
        self._check(self.lib.scfgp_adjoint(self.ctx), 'adjoint')

    def pass3(self):
        self._check(self.lib.scfgp_pass3(self.ctx), 'pass3')

    def finish(self, want_grad=True):
        """(cost, grad, alpha, Li) -- or None when the library asks for the stages to be run again at the precision level
        it has just raised (SCFGP_REDO, include/scfgp_hip.h: the condition estimate of this evaluation was too high
        for the level it ran at)."""
        early = getattr(self, '_early', None)
        cost, grad, alpha, Li = self._outputs(want_grad, factors=early is None)
        if early is not None:
            alpha, Li = early
            rc = self.lib.scfgp_finish(self.ctx, int(bool(want_grad)), dptr(cost), dptr(grad), None, None)
        else:
            rc = self.lib.scfgp_finish(self.ctx, int(bool(want_grad)), dptr(cost), dptr(grad), dptr(alpha), dptr(Li))
        self._early = None
        if rc == _lib.SCFGP_REDO:
            return None
        self._check(rc, 'finish')
        return cost.reshape(()), grad, alpha, Li

    def stream_fence(self, peer_stream, direction):
        """Order the library's stream against `peer_stream` (a raw hipStream_t handle, 0/None = legacy default
        stream): direction 0 = peer waits for the library (before a collective), 1 = the library waits for peer."""
        self._check(self.lib.scfgp_stream_fence(self.ctx, C.c_void_p(peer_stream) if peer_stream else None, int(direction)),
                    'stream_fence')

    def exchange_ptr(self, stage):
        """(device pointer, number of float64) of exchange buffer `stage` (1..3)."""
        p = C.c_void_p(); n = C.c_int64()
        self._check(self.lib.scfgp_exchange(self.ctx, int(stage), C.byref(p), C.byref(n)), 'exchange')
        return p.value, n.value

    def exchange(self, stage):
        """The exchange buffer as a torch CUDA tensor aliasing the library's device memory."""
        import torch
        ptr, n = self.exchange_ptr(stage)

        class _Alias(object):
            __cuda_array_interface__ = {'shape': (n,), 'typestr': '<f8', 'data': (ptr, False), 'version': 2}
        # name the engine's own GPU: a bare 'cuda' is torch's CURRENT device and as_tensor would silently copy
        return torch.as_tensor(_Alias(), device=torch.device('cuda', self.device))

    # -- the sums inside the library (RCCL) -----------------------------------------------------------
    def comm_unique_id(self):
        """128 bytes identifying a new communicator (rank 0 creates them and hands them to the other ranks)."""
        buf = C.create_string_buffer(128)
        rc = self.lib.scfgp_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError('scfgp_comm_unique_id failed (%d): is librccl.so loadable?' % rc)
        return buf.raw

    def comm_init(self, nranks, rank, unique_id):
        """Join the communicator (collective over the ranks): from now on pass1 / pass2 / pass3 -- and with them eval() and
        eval_rows() -- end in their ncclAllReduce on the library's stream; no ShardedEvaluator, no fences."""
        if len(unique_id) != 128:
            raise ValueError('the unique id is 128 bytes')
        self._check(self.lib.scfgp_comm_init(self.ctx, int(nranks), int(rank), C.c_char_p(bytes(unique_id))), 'comm_init')

    def comm_destroy(self):
        self._check(self.lib.scfgp_comm_destroy(self.ctx), 'comm_destroy')

    # -- on-device optimiser ------------------------------------------------------------------------
    ALGOS = {'sgd': 0, 'adagrad': 1, 'rmsprop': 2, 'adadelta': 3, 'adam': 4, 'adamax': 5}

    def opt_init(self, algo, learning_rate=0.01, beta1=0.9, beta2=0.999, epsilon=1e-8, momentum=0.9):
        """Device-side update rule (beta1 doubles as rho for rmsprop/adadelta); momentum<0: no Nesterov."""
        h = np.array([learning_rate, beta1, beta2, epsilon], dtype=np.float64)
        self._check(self.lib.scfgp_opt_init(self.ctx, self.ALGOS[algo], dptr(h), 4, float(momentum)), 'opt_init')

    def opt_state(self, which, value=None):
        buf = np.empty(1 if which == 3 else self.P) if value is None else np.ascontiguousarray(value, dtype=np.float64).ravel()
        self._check(self.lib.scfgp_opt_state(self.ctx, 0 if value is None else 1, int(which), dptr(buf)), 'opt_state')
        return buf

    def opt_step(self, grad):
        """One step of the device rule with the given gradient (no evaluation); returns the updated parameter vector."""
        g = np.ascontiguousarray(grad, dtype=np.float64).ravel()
        self._check(self.lib.scfgp_opt_step(self.ctx, dptr(g), g.size), 'opt_step')
        return self.get_params()

    def train(self, n_iters, want_factors=True):
        """n_iters x (NLML+grad evaluation + update) on the resident rows without host round trips.
        Returns (cost history (n,), alpha, Li of the last evaluation)."""
        hist = np.empty(int(n_iters))
        alpha = np.empty((self.K, 1)) if want_factors else None
        Li = np.empty((self.K, self.K)) if want_factors else None
        self._check(self.lib.scfgp_train(self.ctx, int(n_iters), dptr(hist), dptr(alpha), dptr(Li)), 'train')
        return hist, alpha, Li

    # -- introspection ----------------------------------------------------------------------------
    def dims(self):
        out = (C.c_int64 * 7)()
        self._check(self.lib.scfgp_get_dims(self.ctx, out, 7), 'get_dims')
        return dict(zip(('K', 'Kp', 'Jp', 'Dp', 'Np', 'P', 'tile'), [int(v) for v in out]))

    CONDITION = ('cond_est', 'level', 'gram_fp64', 'alpha_err_fp32', 'threshold', 'threshold_w', 'Lmin2', 'Lmax2', 'Bmax')

    def condition(self):
        """Condition estimate of A and the precision level of the last finished evaluation (scfgp_get_condition)."""
        out = np.zeros(len(self.CONDITION))
        self._check(self.lib.scfgp_get_condition(self.ctx, dptr(out), out.size), 'get_condition')
        return dict(zip(self.CONDITION, out.tolist()))

    def last_error(self):
        """Message of the last failure -- or refusal (a precision level whose buffers could not be had) -- on this context."""
        return self.lib.scfgp_last_error(self.ctx).decode()

    def set_profiling(self, on=True):
        self._check(self.lib.scfgp_set_profiling(self.ctx, int(bool(on))), 'set_profiling')

    def timings(self):
        """[(stage name, milliseconds)] of the last evaluation (profiling must be on)."""
        n = 64
        ms = (C.c_double * n)(); names = (C.c_char_p * n)()
        k = self.lib.scfgp_get_timings(self.ctx, ms, names, n)
        return [(names[i].decode(), ms[i]) for i in range(max(k, 0))]

    def set_option(self, name, value):
        self._check(self.lib.scfgp_set_option(self.ctx, name.encode(), int(value)), 'set_option')

    def debug_read(self, name, shape, dtype=np.float64):
        out = np.empty(shape, dtype=dtype)
        n = self.lib.scfgp_debug_read(self.ctx, name.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes)
        if n < 0:
            self._check(int(n), 'debug_read')
        return out
