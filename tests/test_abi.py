"""The C-ABI library loads on a GPU-less box and exports every symbol include/scfgp_hip.h
declares (no compute calls here)."""
import ctypes
import os
import re

from scfgp_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'scfgp_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return set(re.findall(r'\b(scfgp_[a-z0-9_]+)\s*\(', text))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), 'missing export ' + n


def test_binding_table_matches_header():
    assert set(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    for n in _lib.SIGNATURES:
        assert getattr(lib, n).argtypes is not None


def test_bad_arguments_rejected_without_gpu():
    lib = _lib.load()
    ctx = ctypes.c_void_p()
    assert lib.scfgp_create(ctypes.byref(ctx), 0, 1, 1, 0, 0, None) == -1      # D < 1
    assert lib.scfgp_create(None, 4, 2, 3, 0, 0, None) == -1
    assert lib.scfgp_last_error(None) == b'null context'


def test_host_side_is_clean_under_asan_ubsan():
    """tools/asan_host.sh: the library's host code built with -fsanitize=address,undefined (device code untouched) and
    every GPU-less entry point driven through it; any report makes the run exit non-zero."""
    import shutil
    import subprocess
    if not os.path.exists('/opt/rocm/bin/hipcc') or shutil.which('make') is None:
        import pytest
        pytest.skip('no hipcc in this environment')
    r = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'asan_host.sh')], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       universal_newlines=True, timeout=900)
    assert r.returncode == 0 and 'ok' in r.stdout.splitlines()[-1], r.stdout[-2000:]
    assert 'ERROR: AddressSanitizer' not in r.stdout and 'runtime error' not in r.stdout
