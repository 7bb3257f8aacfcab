"""
scfgp_amd: MI355X-native implementation of SCFGP's Fourier-feature marginal-likelihood
hot path (NLML + gradient + predictive moments) behind the reference's own
compiled-function boundary.  Export list follows SCFGP/__init__.py:7-9.
"""
from .model import SCFGP
from .scaler import Scaler
from .optimizer import Optimizer
from .funcs import CompiledFuncs
from .engine import HipEngine

__all__ = ['SCFGP', 'Scaler', 'Optimizer', 'CompiledFuncs', 'HipEngine']
