"""Audit of the K1 ISA (tools/k1_audit.py <k_buzhash .s file>): between a hand-written exec-masked ds_read_b128 and the s_waitcnt that
retires it, no instruction may read or write the destination registers (the compiler does not know the load is in flight; a copy
made in between reads the old value whenever the data has not landed)."""
import re, sys
s = open(sys.argv[1]).read()
bad = total = 0
for m in re.finditer(r'^(_ZN\S*k_buzhash_prefix\S*):', s, re.M):
    body = s[m.start():s.index('s_endpgm', m.start())].split('\n')
    pending = []  # (first reg, issue line)
    for n, l in enumerate(body):
        t = l.strip()
        mm = re.match(r'ds_read_b128 v\[(\d+):(\d+)\], v\d+ offset', t)
        if mm and 's_mov_b64 exec' in body[n - 1]:
            pending.append((int(mm.group(1)), n)); total += 1
            continue
        mw = re.match(r's_waitcnt lgkmcnt\((\d+)\)', t)
        if mw and ';;#ASMSTART' in body[n - 1]:
            keep = int(mw.group(1))
            pending = pending[len(pending) - keep:] if keep else []
            continue
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
            continue
        for r0, ln in pending:
            regs = set(range(r0, r0 + 4))
            used = set()
            for a, b in re.findall(r'v\[(\d+):(\d+)\]', t):
                used |= set(range(int(a), int(b) + 1))
            used |= {int(x) for x in re.findall(r'\bv(\d+)\b', t)}
            if used & regs:
                print(f"{m.group(1)[:70]}: line {n}: `{t}` touches v[{r0}:{r0+3}] loaded at line {ln} and not yet waited for")
                bad += 1
print(f"{total} hand-written loads checked, {bad} violations")
sys.exit(1 if bad else 0)
