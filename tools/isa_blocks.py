#!/usr/bin/env python3
"""Basic blocks of ONE kernel of a `hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S` listing with their instruction counts
(all / VALU / SALU / LDS) and branch targets: where a kernel that is bound by the number of vector instructions it issues spends them.
usage: isa_blocks.py <file.s> <substring of the mangled kernel name> [first_block last_block [x]]
  no range: the blocks that touch the private table (ds_read_u16 / ds_write_b16) -- the lane parser's probe round;
  with a range: a line per block, with a trailing argument also the instructions."""
import re,sys
lines=open(sys.argv[1]).read().split('\n')
key=sys.argv[2]
start=[i for i,l in enumerate(lines) if l.startswith('_ZN') and key in l.split(':')[0] and ': ' in l or (l.startswith('_ZN') and key in l and l.split(';')[0].rstrip().endswith(':'))][0]
end=next(i for i in range(start,len(lines)) if lines[i].startswith('.Lfunc_end'))
body=lines[start+1:end]
blocks=[['entry',[]]]
for l in body:
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m:
        blocks.append([m.group(1),[]]);continue
    t=l.split(';')[0].strip()
    if t and not t.startswith('.'):
        blocks[-1][1].append(t)
idx={b[0]:i for i,b in enumerate(blocks)}
tot=sum(len(b[1]) for b in blocks)
print('blocks',len(blocks),'instructions',tot)
def summary(i):
    name,ins=blocks[i]
    v=sum(1 for x in ins if x.startswith('v_')); s=sum(1 for x in ins if x.startswith('s_')); d=sum(1 for x in ins if x.startswith('ds_'))
    br=[x for x in ins if x.startswith('s_cbranch') or x.startswith('s_branch')]
    tg=[x.split()[-1] for x in br]
    return f"{i:4d} {name:12s} n={len(ins):4d} v={v:4d} s={s:4d} ds={d:3d} -> {','.join(tg)}"
if len(sys.argv)>3:
    a,b=int(sys.argv[3]),int(sys.argv[4])
    for i in range(a,b):
        print(summary(i))
        if len(sys.argv)>5:
            for x in blocks[i][1]: print('        ',x)
else:
    for i in range(len(blocks)):
        ins=blocks[i][1]
        if any('0x9e3779b1' in x for x in ins) or any('ds_read_u16' in x for x in ins) or any('ds_write_b16' in x for x in ins):
            print(summary(i))
