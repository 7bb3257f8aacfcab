"""
SCFGP model facade: the reference's public class (SCFGP/SCFGP.py:21) with its
constructor, attributes and methods -- set_data, optimize, predict, save, load,
get_compiled_funcs, minibatches, message -- plus the README-advertised fit()
(README.md:47-52).  Every number comes from the HIP library through the function
triple of scfgp_amd.funcs; this file is bookkeeping around it.

Deliberate departures from the reference:
  * parameters live in an explicit `Shared` vector, so "restore the best iterate" and
    the pull-back perturbation (SCFGP/SCFGP.py:256,263-264) act on the vector the
    compiled functions really use.  The reference re-binds self.params to new symbolic
    objects there, which makes those lines no-ops; `compat_noop_restore=True`
    reproduces that observable behaviour.
  * np.inf instead of np.Infinity (:224), guarded max_iter//10 (:242), a working
    rmsprop.
  * save()/load() use a portable .npz of arrays (the reference pickles compiled Theano
    functions, :296-310, which cannot be loaded without Theano).
"""
import string
import sys
import time

import numpy as np
import numpy.random as npr

from .funcs import CompiledFuncs
from .optimizer import Optimizer as OPT, Shared
from .scaler import Scaler


# optimize(**args) defaults, SCFGP/SCFGP.py:185-202
_OPT_DEFAULTS = dict(obj='COST', algo={'algo': None}, nbatches=1, batchsize=150, cvrg_tol=1e-4,
                     max_cvrg=18, max_iter=500)
_ADAM_DEFAULTS = dict(learning_rate=0.01, beta1=0.9, beta2=0.999, epsilon=1e-8)
_METRICS = (("SCORE", "Model Selection Score"), ("COST", "Hyperparameter Selection Cost"),
            ("MAE", "Mean Absolute Error"), ("NMAE", "Normalized Mean Absolute Error"),
            ("MSE", "Mean Square Error"), ("NMSE", "Normalized Mean Square Error"),
            ("MNLP", "Mean Negative Log Probability"), ("TIME(s)", "Training Time"))


class SCFGP(object):

    """Sparsely Correlated Fourier Features Based Gaussian Process (MI355X build)"""

    ID, NAME, verbose = "", "", True
    X_scaler, y_scaler = [None] * 2
    M, N, D = -1, -1, -1
    X, y, hyper, Li, alpha, train_func, pred_func = [None] * 7

    def __init__(self, sparsity=20, nfeats=18, evals=None,
                 X_scaling_method='auto-inv-normal', y_scaling_method='auto-normal', verbose=False,
                 dtype='f64', device=0, compat_noop_restore=False, device_optimizer=False, device_scaler=False):
        self.S = sparsity
        self.M = nfeats
        self.X_scaler = Scaler(X_scaling_method)
        self.y_scaler = Scaler(y_scaling_method)
        self.evals = {k: [title, []] for k, title in _METRICS} if evals is None else evals
        self.verbose = verbose
        self.dtype, self.device = dtype, device
        self.compat_noop_restore = compat_noop_restore
        self.device_optimizer = device_optimizer
        self.device_scaler = device_scaler          # predict(): X scaling inside the GPU packing kernel
        self.generate_ID()

    def message(self, *arg):
        if self.verbose:
            print(" ".join(map(str, arg)))
            sys.stdout.flush()

    def generate_ID(self):
        self.ID = ''.join(chr(npr.choice([ord(c) for c in (string.ascii_uppercase + string.digits)]))
                          for _ in range(5))
        self.NAME = "SCFGP (Sparsity=%d, Fourier Features=%d)" % (self.S, self.M)

    def init_params(self):
        """Same draw order and distributions as SCFGP/SCFGP.py:64-72."""
        a = npr.randn(1)
        b = npr.randn(1)
        c = npr.randn(1)
        l_f = npr.randn(self.D * self.S)
        r_f = npr.rand(self.M * self.S)
        l_p = 2 * np.pi * npr.rand(self.S)
        p = 2 * np.pi * npr.rand(self.M)
        self.params = Shared(np.concatenate([a, b, c, l_f, r_f, l_p, p]))

    # -- "compilation" ------------------------------------------------------------------------
    def build_hip_models(self, algo, algo_params, momentum=0.9):
        """Counterpart of build_theano_models (SCFGP/SCFGP.py:92-148): creates the GPU context
        and optimiser state bound to self.params; no symbolic build, no C compile.  momentum: the Nesterov
        momentum the reference hard-codes at SCFGP/SCFGP.py:131 (negative: none)."""
        self._compiled = CompiledFuncs(self.D, self.S, self.M, self.params, algo, algo_params, momentum=momentum,
                                       dtype=self.dtype, device=self.device,
                                       device_optimizer=self.device_optimizer)
        self.train_func, self.train_iter_func, self.pred_func = self._compiled.triple()

    build_theano_models = build_hip_models          # drop-in name

    def get_compiled_funcs(self):
        return self.train_func, self.train_iter_func, self.pred_func

    # -- data -----------------------------------------------------------------------------------
    def set_data(self, X, y):
        """X: (N,D) inputs, y: (N,1) targets in original units (SCFGP/SCFGP.py:153-170)."""
        self.message("-" * 60, "\nNormalizing SCFGP training data...")
        self.X_scaler.fit(X)
        self.y_scaler.fit(y)
        # the model's own copies, frozen: the triple's residency check then never hashes them again (funcs.py: _sync_data)
        self.X = np.array(self.X_scaler.forward_transform(X), dtype=np.float64, order='C')
        self.y = np.array(self.y_scaler.forward_transform(y), dtype=np.float64, order='C')
        self.X.flags.writeable = False
        self.y.flags.writeable = False
        self.message("done.")
        self.N, self.D = self.X.shape
        if 'train_func' not in self.__dict__.keys():
            self.message("-" * 60, "\nInitializing SCFGP hyperparameters...")
            self.init_params()
            self.message("done.")
        else:
            cost, self.alpha, self.Li = self.train_func(self.X, self.y)

    def minibatch_indices(self, n, batchsize, shuffle=True):
        """Row indices of successive minibatches (same shuffling and truncation as SCFGP.py:172-182)."""
        inds = np.arange(n)
        if shuffle:
            np.random.shuffle(inds)
        for start_ind in range(0, n - batchsize + 1, batchsize):
            yield inds[start_ind:start_ind + batchsize]

    def minibatches(self, X, y, batchsize, shuffle=True):
        assert len(X) == len(y)
        for batch in self.minibatch_indices(len(X), batchsize, shuffle):
            yield np.ascontiguousarray(X[batch]), np.ascontiguousarray(y[batch])

    # -- training ---------------------------------------------------------------------------------
    def optimize(self, Xv=None, yv=None, funcs=None, visualizer=None, **args):
        """Training driver, SCFGP/SCFGP.py:184-276 (same keyword arguments and defaults)."""
        opt = dict(_OPT_DEFAULTS)
        opt.update({k: v for k, v in args.items() if k in opt})
        obj = str(opt['obj']).upper()
        obj = obj if obj in self.evals else 'COST'
        algo, nbatches, batchsize = opt['algo'], opt['nbatches'], opt['batchsize']
        cvrg_tol, max_cvrg, max_iter = opt['cvrg_tol'], opt['max_cvrg'], opt['max_iter']
        if algo.get('algo') not in OPT.algos:
            algo = {'algo': 'adam', 'algo_params': dict(_ADAM_DEFAULTS)}
        for metric in self.evals.keys():
            self.evals[metric][1] = []
        if funcs is None:
            self.message("-" * 50, "\nCreating SCFGP HIP context...")
            self.build_hip_models(algo['algo'], algo.get('algo_params', {}))
            self.message("done.")
        else:
            self.train_func, self.train_iter_func, self.pred_func = funcs
        # the vector the triple really trains (a reused triple owns its own: Appendix B)
        owner = getattr(self.train_iter_func, '__self__', None)
        live = owner.params if isinstance(owner, CompiledFuncs) else self.params
        animate = None
        if visualizer is not None:
            visualizer.model = self
            animate = visualizer.train_with_plot()
        if Xv is None or yv is None:
            obj = 'COST'
            for k in ('MAE', 'NMAE', 'MSE', 'NMSE', 'MNLP', 'SCORE'):
                self.evals[k][1].append(0)
        self.min_obj_ind = 0
        train_start_time = time.time()
        min_obj_val, argmin_params, cvrg_iter = np.inf, live.get_value(), 0
        for iter in range(max_iter):
            if nbatches > 1:
                cost_sum, params_list, batch_count = 0, [], 0
                for batch in self.minibatch_indices(len(self.X), batchsize):
                    params_list.append(live.get_value())
                    if isinstance(owner, CompiledFuncs):      # gather the batch from the resident rows on the GPU
                        cost, self.alpha, self.Li = owner.train_iter_rows(self.X, self.y, batch)
                    else:
                        cost, self.alpha, self.Li = self.train_iter_func(np.ascontiguousarray(self.X[batch]),
                                                                         np.ascontiguousarray(self.y[batch]))
                    cost_sum += cost; batch_count += 1
                    if batch_count == nbatches:
                        break
                if not self.compat_noop_restore:
                    live.set_value(np.median(np.array(params_list), axis=0))       # :234
                self.evals['COST'][1].append(np.double(cost_sum / batch_count))
            else:
                cost, self.alpha, self.Li = self.train_iter_func(self.X, self.y)
                self.evals['COST'][1].append(cost)
            self.evals['TIME(s)'][1].append(time.time() - train_start_time)
            if Xv is not None and yv is not None:
                self.predict(Xv, yv)
            if iter % max(max_iter // 10, 1) == 1:
                self.message("-" * 17, "VALIDATION ITERATION", iter, "-" * 17)
                self._print_current_evals()
            if animate is not None:
                animate(iter)
            obj_val = self.evals[obj][1][-1]
            if obj_val < min_obj_val:
                if min_obj_val - obj_val < cvrg_tol:
                    cvrg_iter += 1
                else:
                    cvrg_iter = 0
                min_obj_val = obj_val
                self.min_obj_ind = len(self.evals['COST'][1]) - 1
                argmin_params = live.get_value()
            else:
                cvrg_iter += 1
            if iter > 30 and cvrg_iter > max_cvrg:
                break
            elif cvrg_iter > max_cvrg * 0.5 and not self.compat_noop_restore:
                randp = np.random.rand() * cvrg_iter / max_cvrg * 0.5
                live.set_value((1 - randp) * live.get_value() + randp * argmin_params)   # :263
        if not self.compat_noop_restore:
            live.set_value(argmin_params)                                                 # :264
        self.params = live
        cost, self.alpha, self.Li = self.train_func(self.X, self.y)
        self.evals['COST'][1].append(np.double(cost))
        self.evals['TIME(s)'][1].append(time.time() - train_start_time)
        if Xv is not None and yv is not None:
            self.predict(Xv, yv)
        self.min_obj_ind = len(self.evals['COST'][1]) - 1
        disp = self.verbose
        self.verbose = True
        self.message("-" * 19, "OPTIMIZATION RESULT", "-" * 20)
        self._print_current_evals()
        self.message("-" * 60)
        self.verbose = disp

    def fit(self, X, y, Xv=None, yv=None, funcs=None, visualizer=None, **opt):
        """README.md:47-52 / experiments/kin8nm/kin8nm.py:55,58: set_data + optimize."""
        self.set_data(X, y)
        self.optimize(Xv, yv, funcs, visualizer, **opt)
        return self

    # -- prediction ---------------------------------------------------------------------------------
    def predict(self, Xs, ys=None):
        """SCFGP/SCFGP.py:278-294."""
        owner = getattr(self.pred_func, '__self__', None)
        if self.device_scaler and isinstance(owner, CompiledFuncs):
            # scaling, pred_func, back-transform and metrics in one device call
            mu_y, std_y, met = owner.pred_y(Xs, self.X_scaler, self.y_scaler, self.alpha, self.Li, ys)
            if met is not None:
                for k in met:
                    self.evals[k][1].append(met[k])
            return mu_y, std_y
        else:
            self.Xs = np.ascontiguousarray(self.X_scaler.forward_transform(Xs), dtype=np.float64)
            mu_f, std_f = self.pred_func(self.Xs, self.alpha, self.Li)
        mu_y = self.y_scaler.backward_transform(mu_f)
        up_bnd_y = self.y_scaler.backward_transform(mu_f + std_f[:, None])
        dn_bnd_y = self.y_scaler.backward_transform(mu_f - std_f[:, None])
        std_y = 0.5 * (up_bnd_y - dn_bnd_y)
        if ys is not None:
            err = mu_y - ys
            mae, mse = np.mean(np.abs(err)), np.mean(err ** 2.)
            mnlp = 0.5 * np.mean((err / std_y) ** 2 + np.log(2 * np.pi * std_y ** 2))
            nmse = mse / np.var(ys)
            for k, v in (('MAE', mae), ('NMAE', mae / np.std(ys)), ('MSE', mse), ('NMSE', nmse),
                         ('MNLP', mnlp), ('SCORE', nmse / (1 + np.exp(-mnlp)))):
                self.evals[k][1].append(v)
        return mu_y, std_y

    # -- persistence -----------------------------------------------------------------------------------
    def save(self, path):
        """Portable checkpoint (arrays only; never pickles code).  The reference pickles the compiled
        train_iter_func itself (SCFGP/SCFGP.py:296-302), i.e. its Adam moments, step counter and Nesterov velocity
        (SCFGP/Optimizer.py:314-323,92-93); here they are saved as arrays next to the rule's name and keyword
        arguments, so load() + optimize(funcs=model.get_compiled_funcs()) continues the same trajectory."""
        import json
        sc = {}
        for tag, s in (('X', self.X_scaler), ('y', self.y_scaler)):
            sc[tag + '_algo'] = s.algo
            for k, v in s.data.items():
                sc['%s_scaler_%s' % (tag, k)] = np.asarray(v)
        ev = {'evals_' + k: np.asarray(v[1], dtype=np.float64) for k, v in self.evals.items()}
        opt = {}
        cf = getattr(self, '_compiled', None)
        owner = getattr(getattr(self, 'train_iter_func', None), '__self__', None)
        if isinstance(owner, CompiledFuncs):
            cf = owner                                     # a reused triple owns the state that is being trained
        if cf is not None:
            plain = {k: (v.item() if isinstance(v, np.generic) else v) for k, v in cf.algo_params.items()}   # np.float32(0.01) etc.
            opt = {'opt_algo': cf.algo, 'opt_kwargs': json.dumps(plain, sort_keys=True),
                   'opt_momentum': float(cf.momentum), 'opt_device': bool(cf.device_optimizer), 'dtype': str(cf.dtype)}
            for i, a in enumerate(cf.get_opt_state()):
                opt['opt_state_%d' % i] = a
        with open(path, 'wb') as f:
            np.savez(f, ID=self.ID, S=self.S, M=self.M, D=self.D, params=self.params.get_value(),
                     alpha=self.alpha, Li=self.Li, **sc, **ev, **opt)

    def load(self, path):
        import json
        algo, kwargs, opt_state, momentum = 'adam', dict(_ADAM_DEFAULTS), None, 0.9
        with np.load(path, allow_pickle=False) as z:
            self.ID = str(z['ID']); self.S = int(z['S']); self.M = int(z['M']); self.D = int(z['D'])
            self.params = Shared(z['params'])
            self.alpha, self.Li = z['alpha'], z['Li']
            for tag in ('X', 'y'):
                s = Scaler(str(z[tag + '_algo']))
                pre = '%s_scaler_' % tag
                for k in z.files:
                    if k.startswith(pre):
                        v = z[k]
                        s.data[k[len(pre):]] = [int(c) for c in v] if k.endswith('cols') else v
                setattr(self, tag + '_scaler', s)
            for k in self.evals:
                if 'evals_' + k in z.files:
                    self.evals[k][1] = list(z['evals_' + k])
            if 'opt_algo' in z.files:                       # checkpoints written before the optimiser was saved lack these
                algo, kwargs = str(z['opt_algo']), json.loads(str(z['opt_kwargs']))
                self.device_optimizer = bool(z['opt_device']); self.dtype = str(z['dtype'])
                momentum = float(z['opt_momentum']) if 'opt_momentum' in z.files else 0.9
                n = len([k for k in z.files if k.startswith('opt_state_')])
                opt_state = [z['opt_state_%d' % i] for i in range(n)]
        self.NAME = "SCFGP (Sparsity=%d, Fourier Features=%d)" % (self.S, self.M)
        self.build_hip_models(algo, kwargs, momentum)
        if opt_state is not None:
            self._compiled.set_opt_state(opt_state)

    def _print_current_evals(self):
        for metric in sorted(self.evals.keys()):
            if len(self.evals[metric][1]) < len(self.evals['COST'][1]):
                continue
            best_perform_eval = self.evals[metric][1][self.min_obj_ind]
            self.message(self.NAME, "%7s = %.4e" % (metric, best_perform_eval))
