"""Static view of one kernel in a hipcc -S listing: per basic block the count of vector / memory instructions by opcode
(largest blocks first).   static_blocks.py <file.s> <mangled-name prefix> [blocks to show]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2])][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')][0]
show = int(sys.argv[3]) if len(sys.argv) > 3 else 12
blocks, cur, name = [], [], 'entry'
for l in lines[start:end]:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append((name, cur)); name = l.split(':')[0]; cur = []
    else:
        cur.append(l)
blocks.append((name, cur))
tot = collections.Counter()
rows = []
for name, b in blocks:
    c = collections.Counter()
    for l in b:
        t = l.strip().split()
        if t and t[0].startswith(('v_', 'ds_', 'global_', 'scratch_', 'buffer_', 's_waitcnt')):
            c[t[0]] += 1
    tot.update(c)
    rows.append((sum(v for k, v in c.items() if k.startswith('v_')), name, c))
print('total', sum(v for k, v in tot.items() if k.startswith('v_')), 'VALU;', {k: v for k, v in tot.most_common(14)})
for n, name, c in sorted(rows, reverse=True)[:show]:
    print(f'{name:12s} VALU {n:4d}', dict(c.most_common(9)))
