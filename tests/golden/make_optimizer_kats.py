"""
Known-answer vectors for the update rules of SCFGP/Optimizer.py: three steps of every rule, written out as the literal
recurrences of the reference's update dictionaries on explicit state arrays -- no import of scfgp_amd/optimizer.py, no
thunks, no Shared: plain numpy on (theta, state, velocity) with Theano's simultaneous-update semantics (every new value
is computed from the OLD state, then all are assigned).

    python tests/golden/make_optimizer_kats.py        ->  tests/golden/optimizer_kats.npz

What is pinned, with the reference lines each formula is read from:
  sgd       SCFGP/Optimizer.py:117-118     adagrad   :157-163     rmsprop  :205-212 (its `grad` is an undefined name: the
  adadelta  :261-275                       adam      :315-330     formula of its docstring :196-198 is used, as in the product)
  adamax    :365-381
  apply_momentum :52-59, apply_nesterov_momentum :88-96 -- both wrap `list(updates.keys())[0]`, the FIRST key of the rule's
  dictionary: the parameter vector for sgd, the accumulator `accu` for adagrad / rmsprop / adadelta, the first moment
  `m_prev` for adam / adamax (their dictionaries insert those before `params`).

Keys of the .npz: '<rule>/<wrapper>/theta' (4, P) = theta_0..theta_3, '/s1', '/s2', '/vel' (4, P) = the rule's state
variables and the velocity after 0..3 steps ('s1' = accu or m, 's2' = delta_accu, v or u; zeros where a rule has none),
'grads' (3, P), 'theta0' (P), and '<rule>/kwargs' as a small float array [learning_rate, rho-or-beta1, beta2, epsilon].
wrapper is 'nesterov' (momentum 0.9: what SCFGP.py:131 hard-codes), 'plain' (no wrapper) or 'momentum' (apply_momentum 0.9).
"""
import os

import numpy as np

P = 7
MOM = 0.9
RULES = {   # learning_rate, rho / beta1, beta2, epsilon  (the defaults of SCFGP/Optimizer.py except the learning rates)
    'sgd': (0.05, 0.0, 0.0, 0.0),
    'adagrad': (0.05, 0.0, 0.0, 1e-6),
    'rmsprop': (0.01, 0.9, 0.0, 1e-6),
    'adadelta': (1.0, 0.95, 0.0, 1e-6),
    'adam': (0.01, 0.9, 0.99, 1e-8),
    'adamax': (0.02, 0.9, 0.999, 1e-8),
}


def inputs():
    theta0 = np.array([0.3, -1.2, 0.05, 2.0, -0.7, 1e-3, -4.0])
    grads = np.array([[0.5, -0.25, 0.0, 1.5, -2.0, 1e-4, 0.125],
                      [-0.3, -0.35, 0.2, 1.0, 0.4, -3e-4, 0.5],
                      [0.1, 0.45, -0.6, -0.8, 0.05, 2e-4, -0.25]])
    return theta0, grads


def rule_step(rule, kw, theta, s1, s2, t, g):
    """Plain new values (theta', s1', s2') of one rule from the OLD state; t = steps already taken."""
    lr, r1, b2, eps = kw
    if rule == 'sgd':                                          # :117-118
        return theta - lr * g, s1, s2
    if rule == 'adagrad':                                      # :160-162
        accu_new = s1 + g ** 2
        return theta - (lr * g / np.sqrt(accu_new + eps)), accu_new, s2
    if rule == 'rmsprop':                                      # :196-198, :209-211
        accu_new = r1 * s1 + (1 - r1) * g ** 2
        return theta - (lr * g / np.sqrt(accu_new + eps)), accu_new, s2
    if rule == 'adadelta':                                     # :268-274
        accu_new = r1 * s1 + (1 - r1) * g ** 2
        update = g * np.sqrt(s2 + eps) / np.sqrt(accu_new + eps)
        return theta - lr * update, accu_new, r1 * s2 + (1 - r1) * update ** 2
    if rule == 'adam':                                         # :318-329
        tt = t + 1
        a_t = lr * np.sqrt(1 - b2 ** tt) / (1 - r1 ** tt)
        m_t = r1 * s1 + (1 - r1) * g
        v_t = b2 * s2 + (1 - b2) * g ** 2
        return theta - a_t * m_t / (np.sqrt(v_t) + eps), m_t, v_t
    if rule == 'adamax':                                       # :368-380
        tt = t + 1
        a_t = lr / (1 - r1 ** tt)
        m_t = r1 * s1 + (1 - r1) * g
        u_t = np.maximum(b2 * s2, np.abs(g))
        return theta - a_t * m_t / (u_t + eps), m_t, u_t
    raise KeyError(rule)


def trajectory(rule, wrapper):
    kw = RULES[rule]
    theta0, grads = inputs()
    theta, s1, s2, vel = theta0.copy(), np.zeros(P), np.zeros(P), np.zeros(P)
    first_is_theta = rule == 'sgd'                             # first key of the rule's update dictionary
    out = {'theta': [theta.copy()], 's1': [s1.copy()], 's2': [s2.copy()], 'vel': [vel.copy()]}
    for t, g in enumerate(grads):
        th_n, s1_n, s2_n = rule_step(rule, kw, theta, s1, s2, t, g)
        old_first, new_first = (theta, th_n) if first_is_theta else (s1, s1_n)
        if wrapper == 'nesterov':                              # :92-95
            x = MOM * vel + new_first - old_first
            vel_n, first = x, MOM * x + new_first
        elif wrapper == 'momentum':                            # :56-58
            x = MOM * vel + new_first
            vel_n, first = x - old_first, x
        else:
            vel_n, first = vel, new_first
        if first_is_theta:
            th_n = first
        else:
            s1_n = first
        theta, s1, s2, vel = th_n, s1_n, s2_n, vel_n
        for k, v in (('theta', theta), ('s1', s1), ('s2', s2), ('vel', vel)):
            out[k].append(np.array(v, dtype=np.float64).copy())
    return {k: np.stack(v) for k, v in out.items()}


def build():
    theta0, grads = inputs()
    z = {'theta0': theta0, 'grads': grads, 'momentum': np.array(MOM)}
    for rule, kw in RULES.items():
        z[rule + '/kwargs'] = np.array(kw)
        for wrapper in ('nesterov', 'plain', 'momentum'):
            for k, v in trajectory(rule, wrapper).items():
                z['%s/%s/%s' % (rule, wrapper, k)] = v
    return z


if __name__ == '__main__':
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'optimizer_kats.npz')
    np.savez(path, **build())
    print('wrote', path)
