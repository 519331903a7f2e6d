#!/usr/bin/env python3
"""Static instruction mix per kernel of a hipcc -S listing:  python tools/isa_mix.py file.s [name-substring]
Classes: VALU, SALU (without waitcnt / branches / nops), WAIT, BRANCH, LDS, VMEM, other; plus the register / occupancy lines."""
import re, sys, collections
txt = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
cur, funcs = None, collections.OrderedDict()
for ln in txt:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur = m.group(1); funcs[cur] = collections.Counter(); continue
    if cur is None: continue
    if ln.startswith(".Lfunc_end"): cur = None; continue
    m = re.match(r"^\s+([a-z_0-9]+)\s", ln + " ")
    if not m: continue
    op = m.group(1)
    if op.startswith("v_"): c = "VALU"
    elif op in ("s_waitcnt",): c = "WAIT"
    elif op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_endpgm"): c = "BRANCH"
    elif op in ("s_nop", "s_barrier", "s_sleep"): c = "NOP/BAR"
    elif op.startswith("s_load") or op.startswith("s_buffer_load"): c = "SMEM"
    elif op.startswith("s_"): c = "SALU"
    elif op.startswith("ds_"): c = "LDS"
    elif op.startswith(("buffer_", "global_", "flat_", "scratch_")): c = "VMEM"
    else: continue
    funcs[cur][c] += 1
    funcs[cur]["op:" + op] += 1
for f, c in funcs.items():
    if want not in f or not c: continue
    print(f)
    print("  ", {k: v for k, v in c.items() if not k.startswith("op:")})
    if len(sys.argv) > 3:
        for k, v in sorted(((k, v) for k, v in c.items() if k.startswith("op:")), key=lambda kv: -kv[1])[:int(sys.argv[3])]:
            print("     ", k[3:], v)
