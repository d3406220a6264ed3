#!/usr/bin/env python3
"""tools/isa_gaps.py FILE.s KERNEL_SUBSTR [LABEL] — issue-slot budget of a loop: per MFMA gap (the instructions between two
v_mfma), how many VALU / transcendental / LDS / VMEM (LDS-DMA) / SALU / waitcnt / nop / barrier instructions the wave issues.
LABEL = the loop header label (".LBB1_18"); default: the longest backward-branch loop of the kernel.  The guide's budget
for one wave per SIMD is <= 5 single-issue fillers per v_mfma_f32_32x32x16 gap (MI355X_MICROARCH.md, cycle constants)."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")): return "trans"
    if op.startswith("v_accvgpr"): return "accmov"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_load_lds", "buffer_load")) or ("global_load" in op and "lds" in op): return "dma"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    path, kern = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else None
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kern in l and ":" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    # loops: label ... s_cbranch* label (backward)
    pos = {}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: pos[m.group(1)] = i
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in pos: loops.append((i - pos[m.group(1)], m.group(1), pos[m.group(1)], i))
    if label is None:
        loops.sort()
        _, label, lo, hi = loops[-1]
    else:
        lo, hi = next((a, b) for _, lb, a, b in loops if lb == label)
    gaps, cur = [], collections.Counter()
    total = collections.Counter()
    for l in body[lo:hi + 1]:
        l = l.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."): continue
        op = l.split()[0]
        c = classify(op)
        if c == "dma" or (c == "vmem" and " lds" in l): c = "dma"
        total[c] += 1
        if c == "mfma":
            gaps.append(cur); cur = collections.Counter()
        else:
            cur[c] += 1
    n = total["mfma"]
    print(f"kernel *{kern}*, loop {label}: {hi - lo + 1} lines, {n} MFMAs")
    cats = ["valu", "trans", "accmov", "lds", "dma", "vmem", "salu", "wait", "nop", "barrier", "other"]
    print("per MFMA: " + "  ".join(f"{c} {total[c] / n:.2f}" for c in cats if total[c]) +
          f"  | all non-MFMA {sum(total[c] for c in cats) / n:.2f}")
    hist = collections.Counter(sum(g.values()) for g in gaps)
    print("gap size histogram (non-MFMA instructions in front of an MFMA -> gaps): " + " ".join(f"{k}:{hist[k]}" for k in sorted(hist)))
    over = sum(max(0, sum(g.values()) - 5) for g in gaps)
    print(f"instructions beyond the 5-filler budget, summed over gaps: {over} ({over / n:.2f} per MFMA)")


if __name__ == "__main__":
    main()
