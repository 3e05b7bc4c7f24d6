#!/usr/bin/env python3
"""Filtered listing of one kernel's disassembly: python tools/isa_path.py <dis.s> <kernel-substring> <lo> <hi> [all]
Prints instruction index, address, opcode for the transcendental / compare / move / branch / scratch instructions between
instruction indices lo..hi (all: every instruction) -- for walking a loop's hot path by hand."""
import re, sys
def load(path, filt):
    lines = open(path).read().split('\n')
    start = [i for i, l in enumerate(lines) if filt in l and l.endswith('>:')][0]
    end = next(i for i in range(start + 1, len(lines)) if re.match(r'^[0-9a-f]+ <', lines[i]))
    out = []
    for l in lines[start + 1:end]:
        m = re.match(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):', l)
        if m:
            out.append((m.group(3), m.group(1), m.group(2)))
    return out
if __name__ == "__main__":
    P = load(sys.argv[1], sys.argv[2])
    lo, hi = int(sys.argv[3]), int(sys.argv[4])
    every = len(sys.argv) > 5
    for k, (a, op, args) in enumerate(P):
        if lo <= k <= hi and (every or re.match(r'v_(rsq|rcp|sqrt|mov|cmp|cndmask)|scratch|s_cbranch|s_branch|s_and_saveexec|s_or_b64 exec', op + ' ' + args)):
            print(k, a[-5:], op, args[:70])
