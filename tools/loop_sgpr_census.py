#!/usr/bin/env python3
"""Census of a kernel's loops from llvm-objdump -d output: per backward branch, the VALU instructions of the loop body
and how many of them carry an SGPR source / are compares / are cndmasks with a scalar mask (4-cycle issue forms on gfx950,
profiles/r03_valu_sgpr_operand_ubench.txt) against the 2-cycle all-VGPR forms.

usage: python tools/loop_sgpr_census.py <disassembly.s> [kernel-name-substring]"""
import re, sys
from collections import Counter

def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    kern = None
    ins = {}
    for line in open(path):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            kern = m.group(1)
            ins[kern] = []
            continue
        m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-F]+):", line)
        if m and kern:
            ins[kern].append((int(m.group(3), 16), m.group(1), m.group(2), line))
    for k, lst in ins.items():
        if filt not in k or not lst:
            continue
        addr = {t[0]: i for i, t in enumerate(lst)}
        loops = []
        for i, (a, op, args, line) in enumerate(lst):
            if op.startswith("s_cbranch") or op == "s_branch":
                m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", line)
                if not m:
                    continue
                # target offset is relative to the symbol start
                tgt = lst[0][0] + int(m.group(1), 16)
                if tgt in addr and addr[tgt] < i:
                    loops.append((addr[tgt], i))
        print(f"== {k}: {len(lst)} instructions, {len(loops)} backward branches")
        for lo, hi in sorted(loops, key=lambda t: t[0] - t[1])[:6]:
            body = lst[lo:hi + 1]
            body = [t[:3] for t in body]
            valu = [(op, args) for _, op, args in body if op.startswith("v_")]
            trans = [op for op, _ in valu if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)", op)]
            cmp_ = [op for op, _ in valu if op.startswith("v_cmp")]
            cnd = [1 for op, args in valu if op.startswith("v_cndmask")]
            sg = []
            for op, args in valu:
                if op.startswith("v_cmp") or op.startswith("v_cndmask") or op.startswith("v_readfirstlane"):
                    continue
                parts = [p.strip() for p in args.split(",")]
                srcs = parts[1:]
                if any(re.match(r"^-?\|?s(\d+|\[\d+:\d+\])\|?$", s) or s in ("vcc", "vcc_lo", "vcc_hi", "exec") for s in srcs):
                    sg.append(op)
            smem = sum(1 for _, op, _ in body if op.startswith("s_") and not op.startswith("s_waitcnt") and not op.startswith("s_nop"))
            vmem = sum(1 for _, op, _ in body if re.match(r"(global|buffer|flat|scratch|ds)_", op))
            print(f"  loop [{lo},{hi}] {hi-lo+1} ins: VALU {len(valu)} (trans {len(trans)}, cmp {len(cmp_)}, cndmask {len(cnd)}, other-with-SGPR-src {len(sg)}), SALU {smem}, mem {vmem}")
            if sg:
                print("     sgpr-src ops:", dict(Counter(sg).most_common(12)))

if __name__ == "__main__":
    main()
