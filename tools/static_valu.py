"""Static count of the VALU instructions of one kernel in a hipcc -S listing, inside and outside its hottest loop
(the basic block with the most v_med3_f32):  static_valu.py <file.s> <mangled-name prefix>"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2])][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')][0]
blocks, cur, name = [], [], 'entry'
for l in lines[start:end]:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append((name, cur)); name = l.split(':')[0]; cur = []
    else:
        cur.append(l)
blocks.append((name, cur))
hot = max(blocks, key=lambda b: sum('v_med3_f32' in l for l in b[1]))
tot = collections.Counter()
for name, b in blocks:
    if name == hot[0]:
        continue
    for l in b:
        t = l.strip().split()
        if t and t[0].startswith('v_'):
            tot[t[0]] += 1
hc = collections.Counter(l.strip().split()[0] for l in hot[1] if l.strip() and l.strip().split()[0].startswith(('v_', 'ds_', 'global_')))
print('hot loop', hot[0], dict(hc))
print('static VALU outside the hot loop:', sum(tot.values()))
print(tot.most_common(16))
