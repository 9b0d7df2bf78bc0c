"""Instruction mix of a hipcc -S listing, per labelled block range: python isa_mix.py file.s <kernel-substring> [--blocks]
Used to check the Winograd F(4x4) kernel (csrc/conv3x3_wino4.hip): fillers per MFMA in the chunk loop, scratch traffic, accvgpr moves."""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_ZN") and key in l and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_load"): return "gload"
    if op.startswith("global_store"): return "gstore"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    return "other"
blocks, cur, name = [], collections.Counter(), "entry"
for i in range(start + 1, end + 1):
    l = src[i].strip()
    if not l or l.startswith(";") or l.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((name, cur)); cur, name = collections.Counter(), m.group(1) + (" " + l.split(";")[1].strip() if ";" in l else "")
        continue
    cur[cls(l.split()[0])] += 1
blocks.append((name, cur))
tot = collections.Counter()
for n, c in blocks:
    tot.update(c)
    if "--blocks" in sys.argv and sum(c.values()) >= 40:
        print(f"{n[:70]:70s}", dict(c))
print("TOTAL", dict(tot))
