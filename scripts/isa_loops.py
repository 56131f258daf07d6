"""usage: isa_loops.py file.s <mangled-kernel-prefix>  -- per loop (backward branch) of a kernel in hipcc -S output: the number
of instructions and their mix (VALU / SALU / LDS / VMEM), the most frequent opcodes of the large ones."""
import re, sys, collections
s = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(s) if l.startswith(sys.argv[2]) and ':' in l][0]
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
lines = [l.strip() for l in s[start:end]]
labels = {l.split(':')[0]: i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
for i, l in enumerate(lines):
    mm = re.match(r'^s_c?branch\w* (\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        seg = lines[labels[mm.group(1)]:i + 1]
        ins = [x.split()[0] for x in seg if x and not x.startswith('.') and not x.startswith(';')]
        cats = collections.Counter('valu' if k.startswith('v_') else 'salu' if k.startswith('s_') else 'lds' if k.startswith('ds_') else
                                   'vmem' if k.startswith(('global_', 'buffer_', 'scratch_', 'flat_')) else 'other' for k in ins)
        print('loop', mm.group(1), 'instructions', len(ins), dict(cats))
        if len(ins) > int(sys.argv[3]) if len(sys.argv) > 3 else 300:
            print('   ', collections.Counter(ins).most_common(45))
