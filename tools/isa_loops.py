"""Per-loop instruction census of gfx950 ISA (CPU only): for every kernel whose mangled name contains `pattern`, every backward
branch's loop body with its MFMA / scratch / barrier / LDS / global counts -- where the spills are and what an inner loop
issues per MFMA.
   python tools/isa_loops.py gram gram_kernel [-DSOMETHING]      compile the source (Makefile flags + the -D)
   python tools/isa_loops.py --shipped gram_kernel               the code objects inside scfgp_amd/lib/libscfgp_hip.so"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_source                                                     # noqa: E402


def census(unit, pattern, extra=(), shipped=None):
    """[(kernel name, {'first', 'last', 'instr', 'mfma', 'scratch', 'barrier', 'ds_read', 'ds_write', 'global_load', 'waitcnt'})] for
    every MFMA-carrying loop (backward branch) of the kernels whose mangled name contains `pattern`; each kernel is preceded
    by (name, {'lines': n})"""
    out = []
    for name, body in isa_source.functions(unit, pattern, extra, shipped):
        name = re.sub(r'^void ', '', name)[:200]
        labels = {l[:-1]: k for k, l in enumerate(body) if l.endswith(':')}
        out.append((name, {'lines': len(body)}))
        for k, l in enumerate(body):
            m = re.match(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                seg = body[labels[m.group(1)]:k + 1]
                c = lambda pat: sum(pat in x for x in seg)
                if c('v_mfma'):
                    out.append((name, {
                        'first': labels[m.group(1)], 'last': k, 'instr': len(seg), 'mfma': c('v_mfma'), 'scratch': c('scratch_'),
                        'barrier': c('s_barrier'), 'ds_read': c('ds_read'), 'ds_write': c('ds_write'), 'global_load': c('global_load'),
                        'waitcnt': c('s_waitcnt')}))
    return out


def main(unit, pattern, extra, shipped=None):
    for name, b in census(unit, pattern, extra, shipped):
        if 'lines' in b:
            print(name, '--', b['lines'], 'lines')
        else:
            print('   loop %5d-%5d: %4d instr, mfma %3d, scratch %3d, barrier %d, ds_read %2d, ds_write %2d, global_load %2d, waitcnt %2d'
                  % (b['first'], b['last'], b['instr'], b['mfma'], b['scratch'], b['barrier'], b['ds_read'], b['ds_write'], b['global_load'], b['waitcnt']))


if __name__ == '__main__':
    if sys.argv[1] == '--shipped':
        main(None, sys.argv[2], [], isa_source.SHIPPED)
    else:
        main(sys.argv[1], sys.argv[2], sys.argv[3:])
