import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # a checkout without the built library (it is git-ignored): compile it once, in-tree, as build() does
    lib = os.path.join(ROOT, 'scfgp_amd', 'lib', 'libscfgp_hip.so')
    if not os.path.exists(lib) and os.path.exists('/opt/rocm/bin/hipcc'):
        import subprocess
        subprocess.call(['make', '-C', os.path.join(ROOT, 'scfgp_amd', 'csrc'), '-j4', 'ARCH=gfx950'],
                        stdout=subprocess.DEVNULL)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
