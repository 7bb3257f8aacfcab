"""VGPR / AGPR / scratch / occupancy / LDS of every kernel of a HIP source, from hipcc's -Rpass-analysis=kernel-resource-usage
remarks (CPU only: hipcc cross-compiles gfx950):  python tools/kernel_resources.py apply gram fmap kernels_kstage > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'scfgp_amd', 'csrc')


def demangle(names):
    try:
        out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
        return out[:len(names)]
    except OSError:
        return names


def main(units):
    for u in units:
        src = os.path.join(CSRC, u + '.hip')
        r = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950',
                            '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null'] + sys_extra, capture_output=True, text=True)
        blocks = re.split(r'remark: [^\n]*Function Name: ', r.stderr)[1:]
        names = demangle([b.split('\n')[0].strip() for b in blocks])
        print('== %s.hip' % u)
        for name, b in zip(names, blocks):
            g = lambda k: re.search(k + r': (\d+)', b).group(1)
            print('VGPR %3s AGPR %3s scratch %4s B/lane  occupancy %s  LDS %6s  %s'
                  % (g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]'),
                     re.sub(r'^void ', '', name)[:230]))


if __name__ == '__main__':
    args = sys.argv[1:]
    sys_extra = [a for a in args if a.startswith('-')]
    main([a for a in args if not a.startswith('-')] or ['apply', 'gram', 'fmap', 'kernels_kstage'])
