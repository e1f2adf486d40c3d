#!/usr/bin/env python3
"""Dev tool: per kernel of a gfx950 .s file, the instruction mix of its hottest basic block (most MFMAs).
    hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only x.hip -o x.s && python tools/isa_loop_mix.py x.s [name filter]"""
import re
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
heads = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", text, re.M)]
for i, (pos, name) in enumerate(heads):
    if flt not in name:
        continue
    body = text[pos:heads[i + 1][0] if i + 1 < len(heads) else len(text)]
    body = body.split("s_endpgm")[0]
    blocks = re.split(r"\n\.LBB\d+_\d+:", body)
    best = max(blocks, key=lambda b: len(re.findall(r"\bv_mfma", b)))
    lines = [l.strip() for l in best.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
    cnt = lambda pred: sum(1 for l in lines if pred(l))
    vg = re.search(r"\.vgpr_count:\s+(\d+)", text[pos:])
    print(f"{name}: mfma {cnt(lambda l: l.startswith('v_mfma'))}  valu {cnt(lambda l: l.startswith('v_') and not l.startswith('v_mfma'))}"
          f"  ds {cnt(lambda l: l.startswith('ds_'))}  vmem {cnt(lambda l: l.startswith(('global_', 'buffer_', 'flat_')))}"
          f"  s_nop {cnt(lambda l: l.startswith('s_nop'))}  waitcnt {cnt(lambda l: l.startswith('s_waitcnt'))}  total {len(lines)}")
