"""
Update rules behind train_iter_func -- host-side counterpart of SCFGP/Optimizer.py.

The reference builds Lasagne-style *symbolic* update dictionaries on Theano shared
variables and bakes them into the compiled train_iter_func
(SCFGP/SCFGP.py:130-131,136-137).  The arithmetic is O(P) per step, so it stays on
the host above the C ABI; what is kept is the callback surface:

    updates = getattr(Optimizer, algo)(params, grads, **algo_params)   # SCFGP.py:130
    updates = Optimizer.apply_nesterov_momentum(updates, momentum=0.9) # SCFGP.py:131

with the same rule names, keyword arguments, defaults and state variables.  `params`
is a `Shared` (stand-in for theano.shared: get_value/set_value), `grads` any object
with get_value() returning the current gradient.  An update dictionary maps each
state variable to a thunk giving its next value; `apply_updates` evaluates every
thunk against the OLD state before assigning, which is Theano's simultaneous-update
semantics.
"""
from collections import OrderedDict

import numpy as np


class Shared(object):
    """Minimal stand-in for a Theano shared variable."""

    def __init__(self, value):
        self._v = np.array(value, dtype=np.float64)
        self.version = 0

    def get_value(self, borrow=False):
        return self._v if borrow else self._v.copy()

    def set_value(self, value):
        self._v = np.array(value, dtype=np.float64)
        self.version += 1

    def copy(self):
        return Shared(self._v)


def apply_updates(updates):
    new = [(var, np.asarray(thunk(), dtype=np.float64)) for var, thunk in updates.items()]
    for var, val in new:
        var.set_value(val)


def _state_like(params):
    return Shared(np.zeros_like(params.get_value(borrow=True)))


class Optimizer(object):

    # SCFGP/Optimizer.py:14-25 also lists norm_constraint / total_norm_constraint, which it
    # never defines; selecting them raises here instead of an AttributeError there.
    algos = ["apply_momentum", "apply_nesterov_momentum", "sgd", "adagrad", "rmsprop",
             "adadelta", "adam", "adamax"]

    # ---- momentum wrappers (SCFGP/Optimizer.py:27-60, :62-97) ------------------------------
    @staticmethod
    def apply_momentum(updates, momentum=0.9):
        """velocity := momentum*velocity + updates[param] - param ; param := that sum."""
        params = list(updates.keys())[0]
        updates = OrderedDict(updates)
        velocity = _state_like(params)
        step = updates[params]
        x = lambda: momentum * velocity.get_value() + step()
        updates[velocity] = lambda: x() - params.get_value()
        updates[params] = x
        return updates

    @staticmethod
    def apply_nesterov_momentum(updates, momentum=0.9):
        """velocity := momentum*velocity + updates[param] - param ;
        param := momentum*velocity_new + updates[param]          (SCFGP/Optimizer.py:92-96)"""
        params = list(updates.keys())[0]
        updates = OrderedDict(updates)
        velocity = _state_like(params)
        step = updates[params]
        x = lambda: momentum * velocity.get_value() + step() - params.get_value()
        updates[velocity] = x
        updates[params] = lambda: momentum * x() + step()
        return updates

    # ---- rules ------------------------------------------------------------------------------
    @staticmethod
    def sgd(params, grads, learning_rate=0.01, **args):
        """param := param - lr*g                                   (SCFGP/Optimizer.py:99-119)"""
        updates = OrderedDict()
        updates[params] = lambda: params.get_value() - learning_rate * grads.get_value()
        return updates

    @staticmethod
    def adagrad(params, grads, learning_rate=0.01, epsilon=1e-6, **args):
        """accu += g^2 ; param -= lr*g/sqrt(accu+eps)              (SCFGP/Optimizer.py:121-164)"""
        updates = OrderedDict()
        accu = _state_like(params)
        accu_new = lambda: accu.get_value() + grads.get_value() ** 2
        updates[accu] = accu_new
        updates[params] = lambda: params.get_value() - learning_rate * grads.get_value() / np.sqrt(accu_new() + epsilon)
        return updates

    @staticmethod
    def rmsprop(params, grads, learning_rate=0.01, rho=0.9, epsilon=1e-6, **args):
        """accu = rho*accu + (1-rho)*g^2 ; param -= lr*g/sqrt(accu+eps).
        The reference's version references an undefined name (SCFGP/Optimizer.py:210) and
        cannot run; this is the formula its docstring states."""
        updates = OrderedDict()
        accu = _state_like(params)
        accu_new = lambda: rho * accu.get_value() + (1 - rho) * grads.get_value() ** 2
        updates[accu] = accu_new
        updates[params] = lambda: params.get_value() - learning_rate * grads.get_value() / np.sqrt(accu_new() + epsilon)
        return updates

    @staticmethod
    def adadelta(params, grads, learning_rate=0.01, rho=0.95, epsilon=1e-6, **args):
        """SCFGP/Optimizer.py:215-276."""
        updates = OrderedDict()
        accu = _state_like(params)
        delta_accu = _state_like(params)
        accu_new = lambda: rho * accu.get_value() + (1 - rho) * grads.get_value() ** 2
        update = lambda: grads.get_value() * np.sqrt(delta_accu.get_value() + epsilon) / np.sqrt(accu_new() + epsilon)
        updates[accu] = accu_new
        updates[params] = lambda: params.get_value() - learning_rate * update()
        updates[delta_accu] = lambda: rho * delta_accu.get_value() + (1 - rho) * update() ** 2
        return updates

    @staticmethod
    def adam(params, grads, learning_rate=0.01, beta1=0.9, beta2=0.99, epsilon=1e-8, **args):
        """SCFGP/Optimizer.py:278-331 (note its default beta2=0.99; optimize() passes 0.999)."""
        t_prev = Shared(0.)
        updates = OrderedDict()
        m_prev = _state_like(params)
        v_prev = _state_like(params)
        t = lambda: t_prev.get_value() + 1
        a_t = lambda: learning_rate * np.sqrt(1 - beta2 ** t()) / (1 - beta1 ** t())
        m_t = lambda: beta1 * m_prev.get_value() + (1 - beta1) * grads.get_value()
        v_t = lambda: beta2 * v_prev.get_value() + (1 - beta2) * grads.get_value() ** 2
        updates[m_prev] = m_t
        updates[v_prev] = v_t
        updates[params] = lambda: params.get_value() - a_t() * m_t() / (np.sqrt(v_t()) + epsilon)
        updates[t_prev] = t
        return updates

    @staticmethod
    def adamax(params, grads, learning_rate=0.01, beta1=0.9, beta2=0.999, epsilon=1e-8, **args):
        """SCFGP/Optimizer.py:333-382."""
        t_prev = Shared(0.)
        updates = OrderedDict()
        m_prev = _state_like(params)
        u_prev = _state_like(params)
        t = lambda: t_prev.get_value() + 1
        a_t = lambda: learning_rate / (1 - beta1 ** t())
        m_t = lambda: beta1 * m_prev.get_value() + (1 - beta1) * grads.get_value()
        u_t = lambda: np.maximum(beta2 * u_prev.get_value(), np.abs(grads.get_value()))
        updates[m_prev] = m_t
        updates[u_prev] = u_t
        updates[params] = lambda: params.get_value() - a_t() * m_t() / (u_t() + epsilon)
        updates[t_prev] = t
        return updates
