import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['SCFGP_LIB_VARIANT'] = '_asan'
from scfgp_amd import _lib
lib = _lib.load()
ctx = C.c_void_p()
assert lib.scfgp_create(C.byref(ctx), 0, 1, 1, 0, 0, None) == -1
assert lib.scfgp_create(None, 4, 2, 3, 0, 0, None) == -1
assert lib.scfgp_last_error(None) == b'null context'
n = 0
for (D, S, M) in [(13, 8, 64), (64, 32, 1024), (512, 64, 2048), (3, 2, 3), (8, 2, 190)]:
    for N in (1, 257, 506, 100000, 1000000, 4000000):
        for dt in (0, 1):
            for ns in (0, 1, 7, 16, 48, 1000):
                for tp in (0, 1):
                    assert lib.scfgp_selftest_row_splits(D, S, M, N, dt, ns, tp) == 0
                    n += 1
# a context on a GPU-less box: create fails cleanly inside hipSetDevice / hipMalloc, the error text is readable, destroy is safe
rc = lib.scfgp_create(C.byref(ctx), 4, 2, 3, 0, 0, None)
print('create on a GPU-less box ->', rc, lib.scfgp_last_error(ctx) if ctx else None)
if ctx: lib.scfgp_destroy(ctx)
print('host-side calls under ASan/UBSan:', n + 4, 'ok')
