#!/usr/bin/env python3
"""Basic-block instruction histogram of one kernel in a hipcc -S listing.

usage: isa_blocks.py listing.s kernel_substring [first_label last_label]
Prints, per basic block, the number of VALU / SALU / LDS / VMEM instructions and the
branch that ends it, so that the always-executed path of a loop can be added up by hand.
"""
import re, sys

def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and ':' in l)
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    blocks, cur = [], {'label': 'entry', 'v': 0, 's': 0, 'lds': 0, 'vm': 0, 'br': [], 'line': start}
    for i in range(start + 1, end + 1):
        l = lines[i].strip()
        if not l or l.startswith(';') or l.startswith('.') and not l.startswith('.LBB'):
            continue
        m = re.match(r'^(\.LBB\w+):', l)
        if m:
            blocks.append(cur)
            cur = {'label': m.group(1), 'v': 0, 's': 0, 'lds': 0, 'vm': 0, 'br': [], 'line': i}
            continue
        op = l.split()[0]
        if op.startswith('v_'):
            cur['v'] += 1
        elif op.startswith('ds_'):
            cur['lds'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
            cur['vm'] += 1
        elif op.startswith('s_'):
            cur['s'] += 1
            if 'branch' in op:
                cur['br'].append(op.replace('s_', '') + '->' + l.split()[-1])
    blocks.append(cur)
    lo = sys.argv[3] if len(sys.argv) > 3 else None
    hi = sys.argv[4] if len(sys.argv) > 4 else None
    on = lo is None
    for b in blocks:
        if b['label'] == lo:
            on = True
        if on:
            print(f"{b['label']:>12} L{b['line'] - start:<5} V{b['v']:<4} S{b['s']:<4} LDS{b['lds']:<3} VM{b['vm']:<3} {' '.join(b['br'])}")
        if b['label'] == hi:
            on = False

if __name__ == '__main__':
    main()
