"""Per-loop instruction census of a HIP source's gfx950 ISA (CPU only): for every kernel whose mangled name contains `pattern`,
every backward branch's loop body with its MFMA / scratch / barrier / LDS / global counts -- where the spills are and what an
inner loop issues per MFMA.   python tools/isa_loops.py gram gram_kernel [-DSOMETHING]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'scfgp_amd', 'csrc')


def census(unit, pattern, extra):
    """[(kernel name, {'first', 'last', 'instr', 'mfma', 'scratch', 'barrier', 'ds_read', 'ds_write', 'global_load', 'waitcnt'})] for
    every MFMA-carrying loop (backward branch) of the kernels whose mangled name contains `pattern`"""
    asm = '/tmp/%s.isa.s' % unit
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '--cuda-device-only', '-S', '-I' + CSRC,
                    os.path.join(CSRC, unit + '.hip'), '-o', asm] + extra, check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split('\n')
    starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and pattern in l]
    names = subprocess.run(['c++filt'], input='\n'.join(n for _, n in starts), capture_output=True, text=True).stdout.split('\n')
    out = []
    for (i, _), name in zip(starts, names):
        j = i
        while not lines[j].startswith('.Lfunc_end'):
            j += 1
        body = lines[i:j]
        labels = {}
        for k, l in enumerate(body):
            m = re.match(r'^(\.LBB\d+_\d+):', l)
            if m:
                labels[m.group(1)] = k
        out.append((re.sub(r'^void ', '', name)[:200], {'lines': len(body)}))
        for k, l in enumerate(body):
            m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                seg = body[labels[m.group(1)]:k + 1]
                c = lambda pat: sum(pat in x.split(';')[0] for x in seg)
                if c('v_mfma'):
                    out.append((out[-1][0] if 'lines' not in out[-1][1] else name, {
                        'first': labels[m.group(1)], 'last': k, 'instr': len(seg), 'mfma': c('v_mfma'), 'scratch': c('scratch_'),
                        'barrier': c('s_barrier'), 'ds_read': c('ds_read'), 'ds_write': c('ds_write'), 'global_load': c('global_load'),
                        'waitcnt': c('s_waitcnt')}))
    return out


def main(unit, pattern, extra):
    for name, b in census(unit, pattern, extra):
        if 'lines' in b:
            print(name, '--', b['lines'], 'lines')
        else:
            print('   loop %5d-%5d: %4d instr, mfma %3d, scratch %3d, barrier %d, ds_read %2d, ds_write %2d, global_load %2d, waitcnt %2d'
                  % (b['first'], b['last'], b['instr'], b['mfma'], b['scratch'], b['barrier'], b['ds_read'], b['ds_write'], b['global_load'], b['waitcnt']))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
