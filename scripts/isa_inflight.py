"""usage: isa_inflight.py file.s <mangled-kernel-prefix>  -- flags instructions that read the destination register of a hand-issued
global load (inline asm, dp_quad.hip.h) before a s_waitcnt vmcnt has covered it: register copies the compiler inserts for such
registers read stale data.  Straight-line approximation (ignores control flow): check what it prints against the source."""
import re, sys
s = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(s) if l.startswith(sys.argv[2]) and ':' in l][0]
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
lines = [l.strip() for l in s[start:end]]
def regs(tok):
    out = []
    for m in re.finditer(r'v\[(\d+):(\d+)\]|v(\d+)', tok):
        out += [int(m.group(3))] if m.group(3) else list(range(int(m.group(1)), int(m.group(2)) + 1))
    return out
pending = {}
for i, l in enumerate(lines):
    if l.startswith(('global_load_dwordx4', 'global_load_ushort', 'global_load_dword ')):
        for r in regs(l.split(None, 1)[1].split(',')[0]): pending[r] = i
    elif l.startswith('s_waitcnt') and 'vmcnt' in l:
        n = int(re.search(r'vmcnt\((\d+)\)', l).group(1))
        for r, idx in list(pending.items()):
            if sum(1 for ll in lines[idx + 1:i] if ll.startswith(('global_', 'scratch_', 'buffer_'))) >= n: del pending[r]
    elif l and not l.startswith(('.', ';', 's_')):
        parts = l.split(None, 1)
        if len(parts) < 2: continue
        ops = parts[1].split(',')
        srcs = [r for o in (ops if l.startswith(('global_store', 'ds_write')) else ops[1:]) for r in regs(o)]
        bad = [r for r in srcs if r in pending]
        if bad: print(i, l, '  <- reads in-flight', bad, 'loaded at', sorted(set(pending[r] for r in bad)))
