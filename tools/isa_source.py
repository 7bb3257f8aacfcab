"""Where the static ISA checks (isa_inflight.py, isa_loops.py) get their instruction streams from (CPU only).

functions(unit, pattern, extra)            compiles scfgp_amd/csrc/<unit>.hip to gfx950 assembly (hipcc -S, the Makefile's flags
                                           plus `extra`) -- for experiments with -D switches;
functions(None, pattern, shipped=<.so>)    disassembles the gfx950 code objects INSIDE a built library (default: the one that
                                           ships, scfgp_amd/lib/libscfgp_hip.so): what runs on the GPU box is what is checked,
                                           whatever flags built it.

Both return [(demangled kernel name, [instruction or label text, ...])] for the kernels whose mangled name contains `pattern`;
labels are `.LBB<f>_<n>:` lines, branch operands name them, comments are stripped."""
import os
import re
import shutil
import subprocess
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
CSRC = os.path.join(ROOT, 'scfgp_amd', 'csrc')
SHIPPED = os.path.join(ROOT, 'scfgp_amd', 'lib', 'libscfgp_hip.so')
LLVM = '/opt/rocm/lib/llvm/bin'
MAKEFILE_FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950']          # scfgp_amd/csrc/Makefile: CXXFLAGS


def _demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
    return out[:len(names)]


def _from_source(unit, pattern, extra):
    asm = os.path.join(tempfile.gettempdir(), '%s.isa_source.s' % unit)
    subprocess.run(['/opt/rocm/bin/hipcc'] + MAKEFILE_FLAGS + ['--cuda-device-only', '-S', '-I' + CSRC,
                    os.path.join(CSRC, unit + '.hip'), '-o', asm] + list(extra), check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split('\n')
    starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and pattern in l]
    out = []
    for (i, mangled), name in zip(starts, _demangle([n for _, n in starts])):
        j = i
        while not lines[j].startswith('.Lfunc_end'):
            j += 1
        body = []
        for l in lines[i + 1:j]:
            l = l.split(';')[0].strip()
            if re.match(r'^\.LBB\d+_\d+:', l) or (l and not l.startswith('.') and not l.endswith(':')):
                body.append(l)
        out.append((name, body))
    return out


def _from_library(path, pattern):
    tmp = tempfile.mkdtemp(prefix='isa_source_')
    try:
        so = os.path.join(tmp, 'lib.so')
        shutil.copy(path, so)                                  # llvm-objdump --offloading writes the bundles next to its input
        subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', so], check=True, capture_output=True)
        out = []
        for f in sorted(os.listdir(tmp)):
            if 'gfx950' not in f:
                continue
            dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--symbolize-operands', '--no-show-raw-insn',
                                  os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout.split('\n')
            cur, fno = None, len(out)
            for l in dis:
                m = re.match(r'^[0-9a-f]+ <(\w+)>:', l)
                if m:
                    if re.match(r'^L\d+$', m.group(1)):
                        if cur is not None:
                            cur[1].append('.LBB%d_%s:' % (fno, m.group(1)[1:]))
                        continue
                    cur = None
                    if m.group(1).startswith('_Z') and pattern in m.group(1):
                        cur = (m.group(1), [])
                        out.append(cur); fno = len(out)
                    continue
                if cur is None or not l.startswith('\t'):
                    continue
                l = l.split('//')[0].strip()
                if l:
                    cur[1].append(re.sub(r'\bL(\d+)\b', lambda mm: '.LBB%d_%s' % (fno, mm.group(1)), l))
        return [(n, b) for n, (_, b) in zip(_demangle([m for m, _ in out]), out)]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def functions(unit, pattern, extra=(), shipped=None):
    if unit is None or shipped is not None:
        return _from_library(shipped or SHIPPED, pattern)
    return _from_source(unit, pattern, extra)
