#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc -S, or llvm-objdump -d of a code object) for the exec-mask construct behind the round-5 wide-window bug
(profiles/round5_exec_hazard.md): LLVM's SILowerControlFlow drops the `s_or_b64 exec` that ends an inner `if` when the enclosing
divergent region ends right behind it ("redundant end-cf") and opens the inner `if` with a plain `s_and_b64 exec, exec, sN` instead of
s_and_saveexec; copies the register allocator places behind the inner `if` afterwards then run under the INNER mask, and lanes of the
outer region that skipped the inner `if` keep clobbered registers.  -mllvm -amdgpu-remove-redundant-endcf=0 removes the construct.
usage: exec_hazard.py file.s [...]   -> per kernel: plain narrowings, and those followed by vector writes before exec is restored"""
import re, sys

def scan(path):
    kern, out = None, {}
    lines = open(path, errors="replace").read().split("\n")
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^([.\w$]+):", l)
        if m:
            labels[m.group(1)] = i
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l) or re.match(r"^[0-9a-f]+ <(_Z\w+)>:", l)
        if m:
            kern = m.group(1)
        if re.search(r"\bs_and_b64 exec, exec, s\[", l):
            st = out.setdefault(kern, [0, 0])
            st[0] += 1
            # instructions from the narrowing to the next write of exec that widens it again
            j, valu = i + 1, 0
            while j < len(lines) and not re.search(r"\bs_or_b64 exec, exec|\bs_mov_b64 exec|\bs_endpgm", lines[j]):
                ins = lines[j].strip()
                # vector writes that are not the inner block's own work: count the ones behind the inner block's label
                j += 1
            # the inner block ends at the label the execz branch names; what follows it up to the s_or is what the allocator added
            k = i + 1
            tgt = None
            while k < j:
                mm = re.search(r"s_cbranch_execz (\S+)", lines[k])
                if mm:
                    tgt = mm.group(1); break
                k += 1
            if tgt and tgt in labels and labels[tgt] < j:
                for q in range(labels[tgt], j):
                    if re.match(r"^\s+v_(?!cmp|readlane|readfirstlane)", lines[q]):
                        valu += 1
            if valu:
                st[1] += 1
    return out

if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        res = scan(p)
        tot = sum(v[0] for v in res.values()); hz = sum(v[1] for v in res.values())
        print(f"{p}: {tot} plain exec narrowings, {hz} with vector writes behind the inner block under the narrowed mask")
        for k, v in sorted(res.items(), key=lambda kv: -kv[1][1])[:12]:
            if v[1]:
                print(f"   {v[1]:4d} of {v[0]:4d}  {k}")
        bad += tot
    sys.exit(1 if bad else 0)
