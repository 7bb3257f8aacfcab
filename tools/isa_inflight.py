"""Static check of the hand-counted LDS waits (CPU only): in the gfx950 ISA of every kernel whose mangled name contains
`pattern`, no instruction may read or overwrite the destination registers of a ds_read_* that the lgkmcnt waits seen since
have not yet covered.  LDS reads return in order, so the check replays the instruction stream in layout order with a queue of
outstanding reads; `s_waitcnt lgkmcnt(n)` retires all but the newest n.  Scalar loads share the counter and only make the
real wait longer, so they are ignored.  Every path of the control-flow graph is followed.
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


def step(queue, l, ln, bad):
    """one instruction on the queue of outstanding reads (tuple of (frozenset of destination registers, line)); returns the new queue"""
    op = l.split()[0]
    if op == 's_waitcnt':
        m = re.search(r'lgkmcnt\((\d+)\)', l)
        if m:
            keep = int(m.group(1))
            queue = queue[len(queue) - keep:] if keep else ()
        return queue
    if op.startswith('s_'):
        return queue
    inflight = frozenset().union(*[q[0] for q in queue]) if queue else frozenset()
    args = l[len(op):]
    first = args.split(',')[0]
    hit = regs(args) & inflight
    if hit:
        bad[ln] = (ln, l, sorted(hit))
    if op.startswith('ds_read'):
        queue = queue + ((frozenset(regs(first)), ln),)
    return queue


def check(lines, name):
    """lines: [(line number, text)] of one function.  Every path through its control-flow graph is replayed (a block is visited
    once per distinct queue of outstanding reads that reaches it)."""
    ins = []
    for ln, l in lines:
        l = l.split(';')[0].strip()
        if l and not l.startswith('.') or re.match(r'^\.LBB\d+_\d+:', l):
            ins.append((ln, l))
    label_at = {l[:-1]: k for k, (_, l) in enumerate(ins) if l.endswith(':')}
    bad, seen = {}, set()
    work = [(0, ())]
    while work:
        k, queue = work.pop()
        while k < len(ins):
            ln, l = ins[k]
            if l.endswith(':'):
                key = (k, tuple(q[0] for q in queue))
                if key in seen:
                    break
                seen.add(key)
                k += 1
                continue
            op = l.split()[0]
            if op == 's_endpgm':
                break
            m = re.match(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
            if m:
                if m.group(1) in label_at:
                    work.append((label_at[m.group(1)], queue))
                if op == 's_branch':
                    break
                k += 1
                continue
            queue = step(queue, l, ln, bad)
            k += 1
    return [bad[k] for k in sorted(bad)]


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
