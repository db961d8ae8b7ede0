#!/usr/bin/env python3
"""Live-VGPR profile of one kernel's compiled ISA (backward liveness over the straight-line text of each basic block; a
value that is live into the next block in layout order or defined outside the block counts as live throughout).  Prints,
per block, the peak number of live VGPRs and every N-th instruction's count with a marker column for LDS / global / scratch /
swap instructions -- enough to see WHICH phase of a kernel sets its register demand.
    python tools/vgpr_pressure.py <file.s> <mangled-name-substring> [every=40]"""
import re
import sys

text = open(sys.argv[1]).read()
pat = sys.argv[2]
every = int(sys.argv[3]) if len(sys.argv) > 3 else 40
m = re.search(r"^(_Z\w*%s\w*):" % re.escape(pat), text, re.M)
body = text[m.start():text.index(".Lfunc_end", m.start())].split("\n")[1:]

NODEST = ("global_store", "buffer_store", "scratch_store", "ds_write", "ds_store", "s_", "v_cmp", "v_readlane", "v_readfirstlane",
          "v_cmpx", "ds_bpermute_dummy")
INOUT = ("v_permlane32_swap", "v_swap", "v_fmac", "v_mac", "v_mfma")   # first operand also read (mfma: only if it is srcC)


def regs(tok):
    out = []
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out += range(int(a), int(b) + 1)
    for a in re.findall(r"\bv(\d+)\b", re.sub(r"v\[\d+:\d+\]", "", tok)):
        out.append(int(a))
    return out


blocks, cur = [["entry", []]], None
for ln in body:
    mm = re.match(r"^(\.LBB\d+_\d+):", ln)
    if mm:
        blocks.append([mm.group(1), []])
        continue
    s = ln.split(";")[0].strip()
    if not s or s.startswith("."):
        continue
    op, _, rest = s.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    if op.startswith(NODEST):
        defs, uses = [], [r for o in ops for r in regs(o)]
        if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_cmp"):
            uses = [r for o in ops[1:] for r in regs(o)]
    else:
        ndst = 2 if op.startswith("v_permlane32_swap") or op.startswith("v_swap") else 1
        defs = [r for o in ops[:ndst] for r in regs(o)]
        uses = [r for o in ops[ndst:] for r in regs(o)]
        if op.startswith(INOUT[:4]):
            uses += defs
    blocks[-1][1].append((op, defs, uses, s))

# live-out of a block = registers used before being defined in ANY later-or-same block (loops: iterate to a fixed point over
# layout order with a crude "every block may follow every block of the function's loops" = union of all blocks' upward-exposed uses)
def upward(ins):
    seen_def, up = set(), set()
    for op, defs, uses, _ in ins:
        up |= set(u for u in uses if u not in seen_def)
        seen_def |= set(defs)
    return up


glob_live = set()
for name, ins in blocks:
    glob_live |= upward(ins)
for name, ins in blocks:
    if len(ins) < 30:
        continue
    live = set(glob_live)
    prof = []
    for op, defs, uses, s in reversed(ins):
        live -= set(defs)
        live |= set(uses)
        prof.append(len(live))
    prof.reverse()
    peak = max(prof)
    print("== %s: %d instructions, peak %d live VGPRs (of which %d live across blocks)" % (name, len(ins), peak, len(glob_live)))
    for i, (op, defs, uses, s) in enumerate(ins):
        tag = "L" if op.startswith("ds_") else "G" if op.startswith(("global_", "buffer_")) else "S" if op.startswith("scratch_") else \
              "P" if "permlane" in op else "B" if op.startswith(("s_barrier", "s_cbranch")) else " "
        if i % every == 0 or prof[i] == peak and (i == 0 or prof[i - 1] != peak):
            print("   %5d %s %4d  %s" % (i, tag, prof[i], s[:90]))
