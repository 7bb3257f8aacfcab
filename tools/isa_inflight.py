"""Static check of the hand-counted LDS waits (CPU only): in the gfx950 ISA of every kernel whose mangled name contains
`pattern`, no instruction may read or overwrite the destination registers of a ds_read_* that the lgkmcnt waits seen since
have not yet covered.  LDS reads return in order, so the check replays the instruction stream in layout order with a queue of
outstanding reads; `s_waitcnt lgkmcnt(n)` retires all but the newest n.  Scalar loads share the counter and only make the
real wait longer, so they are ignored.  Layout order is not every path: the check also restarts with an empty queue at every
label only if `--reset-at-labels` is given; by default the queue is carried through labels (fall-through and loop bodies).
   python tools/isa_inflight.py apply apply_dma_kernel [-DSCFGP_APPLY_PIPE=1]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'scfgp_amd', 'csrc')


def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


def check(lines, name):
    queue, bad = [], []                                         # queue of (dst register set, line)
    for ln, l in lines:
        l = l.split(';')[0].strip()
        if not l or l.endswith(':') or l.startswith('.'):
            continue
        op = l.split()[0]
        if op == 's_waitcnt':
            m = re.search(r'lgkmcnt\((\d+)\)', l)
            if m:
                keep = int(m.group(1))
                queue = queue[len(queue) - keep:] if keep else []
            continue
        if op == 's_barrier' or op.startswith('s_'):
            continue
        inflight = set().union(*[q[0] for q in queue]) if queue else set()
        args = l[len(op):]
        first = args.split(',')[0]
        touched = regs(args)
        if op.startswith('ds_read'):
            touched = regs(first) | regs(','.join(args.split(',')[1:]))   # destination (overwrite) and address
        hit = touched & inflight
        if hit:
            bad.append((ln, l, sorted(hit)))
        if op.startswith('ds_read'):
            queue.append((regs(first), ln))
    return bad


def main(unit, pattern, extra):
    asm = '/tmp/%s.inflight.s' % unit
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '--cuda-device-only', '-S', '-I' + CSRC,
                    os.path.join(CSRC, unit + '.hip'), '-o', asm] + extra, check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split('\n')
    starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and pattern in l]
    names = subprocess.run(['c++filt'], input='\n'.join(n for _, n in starts), capture_output=True, text=True).stdout.split('\n')
    total = 0
    for (i, _), name in zip(starts, names):
        j = i
        while not lines[j].startswith('.Lfunc_end'):
            j += 1
        bad = check([(k - i, lines[k]) for k in range(i, j)], name)
        print('%s: %d reads of registers in flight' % (name.split('(')[0], len(bad)))
        for ln, l, hit in bad[:8]:
            print('    +%d  %s   <- v%s' % (ln, l, hit))
        total += len(bad)
    return total


if __name__ == '__main__':
    args = sys.argv[1:]
    sys.exit(1 if main(args[0], args[1], [a for a in args[2:] if a.startswith('-')]) else 0)
