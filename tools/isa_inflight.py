"""Static check of the hand-counted LDS waits (CPU only): in the gfx950 ISA of every kernel whose mangled name contains
`pattern`, no instruction may read or overwrite the destination registers of a ds_read_* that the lgkmcnt waits seen since
have not yet covered.  LDS reads return in order, so the check replays the instruction stream in layout order with a queue of
outstanding reads; `s_waitcnt lgkmcnt(n)` retires all but the newest n.  Scalar loads share the counter and only make the
real wait longer, so they are ignored.  Every path of the control-flow graph is followed.

The same replay carries the counted `s_waitcnt vmcnt(n)` of the LDS-DMA rings (ring_check): inside every loop that issues
`global_load_lds` fetches, a counted wait lets exactly ONE later stage stay in flight (three-slot rings: apply.hip,
gram.hip), so every non-zero n waited on inside the loop must be the number of fetch instructions some path through the loop
body issues -- a vmcnt(3) in a loop whose waves fetch two instructions per stage (round 4's non-determinism at 6e-10) fails.

   python tools/isa_inflight.py apply apply_dma_kernel [-DSCFGP_APPLY_PIPE=1]      compile the source (Makefile flags + the -D)
   python tools/isa_inflight.py --shipped apply_dma_kernel                         the code objects inside scfgp_amd/lib/libscfgp_hip.so"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_source                                                     # noqa: E402


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
            if len(queue) > 48:                                         # more reads in flight than any counted wait can tell apart: a path
                bad[ln] = (ln, l, ['more than 48 LDS reads outstanding'])   # that issues reads and skips their waits (or a real leak)
                break
            k += 1
    return [bad[k] for k in sorted(bad)]


def loops_with_fetches(ins):
    """{head index: last index} of the loops (a head label with every backward branch to it: a rotated loop has several) whose
    body issues LDS-DMA fetches and MFMAs"""
    label_at = {l[:-1]: k for k, l in enumerate(ins) if l.endswith(':')}
    out = {}
    for k, l in enumerate(ins):
        m = re.match(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in label_at and label_at[m.group(1)] < k:
            j = label_at[m.group(1)]
            out[j] = max(out.get(j, 0), k)
    return {j: k for j, k in out.items()
            if any('global_load_lds' in x for x in ins[j:k + 1]) and any(x.startswith('v_mfma') for x in ins[j:k + 1])}


def ring_check(ins):
    """Counted vmcnt waits of the LDS-DMA ring.  For every INNERMOST fetch-carrying loop: F = {number of global_load_lds issued on
    a path from one `s_barrier` of the loop to the next (around the back edge if need be) -- a stage's fetches, however many stages
    the compiler or the source put into one trip}, W = {n of every `s_waitcnt vmcnt(n)` in the loop, n > 0}; returns
    [(first, last, sorted F, sorted W)] of the loops where W is not a subset of F (a wave would then let part of the stage it is about
    to read stay in flight, or wait for its newest stage too), and the number of loops examined.  States (position, fetches so far)
    are visited once: the bodies have a forward branch per predicated fetch."""
    loops = loops_with_fetches(ins)
    inner = [(j, k) for j, k in loops.items() if not any(j2 != j and j <= j2 and k2 <= k for j2, k2 in loops.items())]
    label_at = {l[:-1]: k for k, l in enumerate(ins) if l.endswith(':')}
    bad = []
    for j, k in inner:
        W = set()
        for l in ins[j:k + 1]:
            mm = re.search(r'vmcnt\((\d+)\)', l) if l.startswith('s_waitcnt') else None
            if mm and int(mm.group(1)) > 0:
                W.add(int(mm.group(1)))
        barriers = [p for p in range(j, k + 1) if ins[p].split()[0] == 's_barrier']
        F, seen = set(), set()
        work = [(b + 1, 0, False, 0) for b in barriers] if barriers else [(j + 1, 0, True, 0)]
        while work:
            pos, f, wrapped, out = work.pop()
            while True:
                # predicated fetches may sit in blocks laid out behind the loop that branch back into it: a path may stay outside
                # [j, k] for a few instructions
                out = 0 if j <= pos <= k else out + 1
                if pos >= len(ins) or out > 24 or (pos, f, wrapped) in seen:
                    break
                seen.add((pos, f, wrapped))
                l = ins[pos]
                op = l.split()[0]
                if op == 's_barrier':
                    F.add(f)
                    break
                if op == 's_endpgm':
                    break
                m = re.match(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
                if m:
                    t = label_at.get(m.group(1), -1)
                    if t == j:                                          # a back edge: on into the next trip (once)
                        if not barriers:
                            F.add(f)
                        elif not wrapped:
                            work.append((j + 1, f, True, 0))
                    elif t >= 0 and (t > pos or not (j <= t <= k)):
                        work.append((t, f, wrapped, out))               # forward inside the body, or out to / back from an out-of-line block
                    if op == 's_branch':
                        break
                elif 'global_load_lds' in l:
                    f += 1
                pos += 1
        if not W <= F:
            bad.append((j, k, sorted(F), sorted(W)))
    return bad, len(inner)


def run(unit, pattern, extra=(), shipped=None):
    """{'functions': n checked, 'inflight': total hazards, 'ring': total ring violations, 'ring_loops': loops examined, 'mfma_loops': ...}"""
    res = {'functions': 0, 'inflight': 0, 'ring': 0, 'ring_loops': 0, 'ds_reads': 0}
    for name, body in isa_source.functions(unit, pattern, extra, shipped):
        short = re.sub(r'^void ', '', name).split('(')[0]
        bad = check(list(enumerate(body)), name)
        rbad, nloops = ring_check(body)
        print('%s: %d reads of registers in flight; %d of %d fetch loops with a vmcnt count that is no path\'s fetch count'
              % (short, len(bad), len(rbad), nloops))
        for ln, l, hit in bad[:8]:
            print('    +%d  %s   <- v%s' % (ln, l, hit))
        for j, k, F, W in rbad:
            print('    loop +%d..+%d: fetch instructions per trip %s, counted waits %s' % (j, k, F, W))
        res['functions'] += 1; res['inflight'] += len(bad); res['ring'] += len(rbad); res['ring_loops'] += nloops
        res['ds_reads'] += sum(l.startswith('ds_read') for l in body)
    return res


def main(unit, pattern, extra, shipped=None):
    r = run(unit, pattern, extra, shipped)
    return r['inflight'] + r['ring']


if __name__ == '__main__':
    args = sys.argv[1:]
    if args[0] == '--shipped':
        sys.exit(1 if main(None, args[1], [], isa_source.SHIPPED) else 0)
    sys.exit(1 if main(args[0], args[1], [a for a in args[2:] if a.startswith('-')]) else 0)
