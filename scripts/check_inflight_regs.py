"""Static check of the split issue / wait LDS pipelines (GSE kernel, packed GEMM).

Those kernels issue `ds_read_b128` from inline asm and wait for the data later with `s_waitcnt lgkmcnt(0)`; between the two the
compiler does not know that the destination registers are still being filled.  This script proves, on the gfx950 ISA, that no
instruction reads or writes such a register while it may be in flight:

  * only reads inside `;;#ASMSTART` .. `;;#ASMEND` are unmanaged (the compiler tracks its own LDS loads with counted waits);
  * the "may be in flight" register set is propagated over the kernel's control-flow graph to a fixpoint (labels, s_cbranch_*,
    s_branch, fall-through), so loop back-edges and conditional issue blocks are handled exactly;
  * any `s_waitcnt` carrying lgkmcnt(0) -- or an s_barrier preceded by one, which the kernels use -- clears the set.

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
        if t[0] == ';' or (t[0] == '.' and not t.endswith(':')):
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
        elif op == 's_waitcnt' and 'lgkmcnt(0)' in t:
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
    return blocks, order


def analyse(blocks, order):
    entry = {b: frozenset() for b in order}
    violations = {}
    succ_fall = {b: (order[i + 1] if i + 1 < len(order) else None) for i, b in enumerate(order)}
    work = [order[0]]
    seen_once = set()
    while work:
        b = work.pop()
        live = set(entry[b])
        outs = []  # (target, set)
        fell = True
        for kind, payload, text in blocks[b]:
            if kind == 'issue':
                live |= payload
            elif kind == 'wait':
                live.clear()
            elif kind == 'use':
                if payload & live:
                    violations[text] = sorted(payload & live)
            elif kind == 'branch':
                outs.append((payload, frozenset(live)))
            elif kind == 'jump':
                outs.append((payload, frozenset(live)))
                fell = False
                break
            elif kind == 'end':
                fell = False
                break
        if fell and succ_fall[b] is not None:
            outs.append((succ_fall[b], frozenset(live)))
        for tgt, st in outs:
            if tgt not in entry:
                continue  # branch out of the kernel text (should not happen)
            merged = entry[tgt] | st
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
            issues = sum(1 for b in blocks.values() for k, _, _ in b if k == 'issue')
            bad = analyse(blocks, order)
            print(f'{name[:78]}: {issues} asm ds_read_b128, {len(bad)} instruction(s) touching a register that may be in flight')
            for t, r in list(bad.items())[:6]:
                print(f'    {t[:90]}   <- v{r}')
            rc |= bool(bad)
        if not found:
            print(f'no kernel matches {prefix!r}')
            rc = 1
    sys.exit(rc)


if __name__ == '__main__':
    main()
