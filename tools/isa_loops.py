"""Per-loop instruction census of a HIP source's gfx950 ISA (CPU only): for every kernel whose mangled name contains `pattern`,
every backward branch's loop body with its MFMA / scratch / barrier / LDS / global counts -- where the spills are and what an
inner loop issues per MFMA.   python tools/isa_loops.py gram gram_kernel [-DSOMETHING]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'scfgp_amd', 'csrc')


def main(unit, pattern, extra):
    asm = '/tmp/%s.isa.s' % unit
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '--cuda-device-only', '-S', '-I' + CSRC,
                    os.path.join(CSRC, unit + '.hip'), '-o', asm] + extra, check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split('\n')
    starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and pattern in l]
    names = subprocess.run(['c++filt'], input='\n'.join(n for _, n in starts), capture_output=True, text=True).stdout.split('\n')
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
        print(re.sub(r'^void ', '', name)[:200], '--', len(body), 'lines')
        for k, l in enumerate(body):
            m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                seg = body[labels[m.group(1)]:k + 1]
                c = lambda pat: sum(pat in x for x in seg)
                if c('v_mfma'):
                    print('   loop %5d-%5d: %4d instr, mfma %3d, scratch %3d, barrier %d, ds_read %2d, ds_write %2d, global_load %2d, waitcnt %2d'
                          % (labels[m.group(1)], k, len(seg), c('v_mfma'), c('scratch_'), c('s_barrier'), c('ds_read'), c('ds_write'),
                             c('global_load'), c('s_waitcnt')))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
