"""Static check of the split issue / wait LDS pipelines (GSE kernel, packed GEMM).

Those kernels issue `ds_read_b128` from inline asm and wait for the data later with `s_waitcnt lgkmcnt(0)`; between the two the
compiler does not know that the destination registers are still being filled.  This script proves, on the gfx950 ISA, that no
instruction reads or writes such a register while it may be in flight:

  * only reads inside `;;#ASMSTART` .. `;;#ASMEND` are unmanaged (the compiler tracks its own LDS loads with counted waits);
  * the "may be in flight" register set is propagated over the kernel's control-flow graph to a fixpoint (labels, s_cbranch_*,
    s_branch, fall-through), so loop back-edges and conditional issue blocks are handled exactly;
  * any `s_waitcnt` carrying lgkmcnt(0) -- or an s_barrier preceded by one, which the kernels use -- clears the set.

Round 4: the same for `global_load_dwordx4` issued from inline asm (the packed GEMM's weight fragments straight from L2, waited for
with a COUNTED `s_waitcnt vmcnt(N)`).  vmcnt retires in order, so the state carries the QUEUE of outstanding VMEM operations (every
global_ / buffer_ / flat_ / scratch_ instruction takes a slot -- the LDS DMA and the compiler's own loads and stores with an empty
register set); `s_waitcnt vmcnt(N)` retires all but the N youngest; at most 63 can be outstanding (the counter saturates).  Two paths
meeting at a label are merged aligned at the YOUNGEST operation (element-wise union): what vmcnt(N) retires on the merged queue it
retires on both.

usage: python scripts/check_inflight_regs.py file.s kernel_name_prefix [...]     (file.s: hipcc -S --cuda-device-only ...)
exit status 1 if any kernel has a violation.
"""
import re
import sys


def regs(tok):
    tok = tok.strip()
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return frozenset(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)\b', tok)
    return frozenset({int(m.group(1))}) if m else frozenset()


def parse(lines):
    """-> (blocks, order): blocks[label] = list of (kind, payload); kind in {'issue', 'wait', 'use', 'branch', 'jump', 'end'}"""
    blocks, order = {}, []
    cur = '<entry>'
    blocks[cur] = []
    order.append(cur)
    in_asm = False
    for ln in lines:
        t = ln.strip()
        if not t:
            continue
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        code = t.split(';')[0].strip()  # (a label may carry a comment: ".LBB8_26:    ; %.lr.ph" -- it is still a label)
        if not code or (code[0] == '.' and not code.endswith(':')):
            continue
        m = re.match(r'^(\.?[A-Za-z_][\w.$]*):', t)
        if m:
            cur = m.group(1)
            if cur not in blocks:
                blocks[cur] = []
                order.append(cur)
            continue
        t = t.split(';')[0].strip()
        if not t:
            continue
        op = t.split()[0]
        ops = [x for x in t[len(op):].split(',')]
        if op == 'ds_read_b128' and in_asm:
            blocks[cur].append(('issue', regs(ops[0]), t))
        elif op == 'global_load_dwordx4' and in_asm:
            addr = frozenset()
            for o in ops[1:]:
                addr |= regs(o.split()[0] if o.strip() else '')
            blocks[cur].append(('use', addr, t))  # the address is read at issue
            blocks[cur].append(('vissue', regs(ops[0]), t))
        elif op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', t)
            if m:
                blocks[cur].append(('vwait', int(m.group(1)), t))
            if 'lgkmcnt(0)' in t:
                blocks[cur].append(('wait', None, t))
        elif op.startswith('s_cbranch'):
            blocks[cur].append(('branch', ops[0].strip(), t))
        elif op == 's_branch':
            blocks[cur].append(('jump', ops[0].strip(), t))
        elif op == 's_endpgm':
            blocks[cur].append(('end', None, t))
        else:
            used = frozenset()
            for o in ops:
                used |= regs(o.split()[0] if o.strip() else '')
            blocks[cur].append(('use', used, t))
            if op.split('_')[0] in ('global', 'buffer', 'flat', 'scratch'):
                blocks[cur].append(('vissue', frozenset(), t))  # a slot of the vmcnt queue (the compiler waits for its own registers)
    return blocks, order


def merge_queues(a, b):
    """Right-aligned (youngest-aligned) element-wise union of two vmcnt queues."""
    if len(a) < len(b):
        a, b = b, a
    pad = len(a) - len(b)
    return tuple(x | (b[i - pad] if i >= pad else frozenset()) for i, x in enumerate(a))


def analyse(blocks, order):
    entry = {b: (frozenset(), ()) for b in order}  # (LDS registers in flight, vmcnt queue oldest -> youngest)
    violations = {}
    succ_fall = {b: (order[i + 1] if i + 1 < len(order) else None) for i, b in enumerate(order)}
    work = [order[0]]
    seen_once = set()
    while work:
        b = work.pop()
        live = set(entry[b][0])
        queue = list(entry[b][1])
        outs = []  # (target, state)
        fell = True

        def state():
            return frozenset(live), tuple(queue)

        for kind, payload, text in blocks[b]:
            if kind == 'issue':
                live |= payload
            elif kind == 'vissue':
                queue.append(payload)
                del queue[:-63]  # the counter saturates at 63: anything older has retired
            elif kind == 'wait':
                live.clear()
            elif kind == 'vwait':
                del queue[:max(0, len(queue) - payload)]
            elif kind == 'use':
                busy = live.union(*queue) if queue else live
                if payload & busy:
                    violations[text] = sorted(payload & busy)
            elif kind == 'branch':
                outs.append((payload, state()))
            elif kind == 'jump':
                outs.append((payload, state()))
                fell = False
                break
            elif kind == 'end':
                fell = False
                break
        if fell and succ_fall[b] is not None:
            outs.append((succ_fall[b], state()))
        for tgt, (st_live, st_queue) in outs:
            if tgt not in entry:
                continue  # branch out of the kernel text (should not happen)
            merged = (entry[tgt][0] | st_live, merge_queues(entry[tgt][1], st_queue))
            if merged != entry[tgt] or tgt not in seen_once:
                entry[tgt] = merged
                seen_once.add(tgt)
                work.append(tgt)
        seen_once.add(b)
    return violations


def kernels(text, prefix):
    for i, ln in enumerate(text):
        if ln.startswith(prefix) and re.match(r'^[\w.$]+:', ln):
            end = next(j for j in range(i, len(text)) if 's_endpgm' in text[j])
            yield ln.split(':')[0], text[i + 1:end + 1]


def main():
    text = open(sys.argv[1]).read().split('\n')
    rc = 0
    for prefix in sys.argv[2:]:
        found = False
        for name, body in kernels(text, prefix):
            found = True
            blocks, order = parse(body)
            issues = sum(1 for b in blocks.values() for k, p_, _ in b if k == 'issue' or (k == 'vissue' and p_))
            bad = analyse(blocks, order)
            print(f'{name[:78]}: {issues} asm ds_read_b128 / global_load_dwordx4, {len(bad)} instruction(s) touching a register that may be in flight')
            for t, r in list(bad.items())[:6]:
                print(f'    {t[:90]}   <- v{r}')
            rc |= bool(bad)
        if not found:
            print(f'no kernel matches {prefix!r}')
            rc = 1
    sys.exit(rc)


if __name__ == '__main__':
    main()
