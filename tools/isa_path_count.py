"""Instruction count, by class, along a hand-specified path of basic blocks of one kernel's gfx950 ISA
(profiles/r05e_string_level.md).  usage: isa_path_count.py <kernel .s (one function)> block [block ...]
Blocks are named as the compiler's labels: .LBB0_17 or %bb.25."""
import re, sys
from collections import Counter
body = open(sys.argv[1]).read().split("\n")
blocks, cur = {}, None
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; (%bb\.\d+):", l)
    if m:
        cur = m.group(1); blocks[cur] = []; continue
    t = l.strip()
    if cur is None or not t or t[0] in ";." :
        continue
    blocks[cur].append(t.split()[0])
def cls(op):
    if op.startswith("s_load"): return "SMEM"
    if op.startswith(("s_waitcnt", "s_nop")): return "wait/nop"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "SALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_")): return "VMEM"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")): return "lane<->scalar"
    return "VALU" if op.startswith("v_") else "other"
tot = Counter()
print("| block | instructions | by class |\n|---|---:|---|")
for b in sys.argv[2:]:
    c = Counter(cls(o) for o in blocks[b]); tot += c
    print(f"| `{b}` | {sum(c.values())} | {', '.join(f'{k} {v}' for k, v in sorted(c.items()))} |")
print(f"| **total** | **{sum(tot.values())}** | {', '.join(f'{k} {v}' for k, v in sorted(tot.items()))} |")
