#!/usr/bin/env python3
"""Instruction mix of the hottest loop of a kernel in a gfx950 ISA listing (hipcc -S --offload-device-only).
usage: isa_loop_mix.py file.s <kernel-name substring> [min loop length]
Prints every backward-branch loop (label .. s_cbranch back to it) with its count of instructions by class and an
issue-cycle estimate from the guide's per-instruction costs (4 cycles; 8 transcendental; 16 32-bit integer multiply
and v_mad_u64_u32 (quarter rate); MFMA 16x16x32: 8 issue of 16)."""
import re
import sys
from collections import Counter

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
QUARTER = ("v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_hi_i32", "v_mad_i64_i32")


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith(QUARTER):
        return "imul"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, kname = sys.argv[1], sys.argv[2]
    minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kname in l and re.match(r"^_Z\S+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    labels = {}
    insts = []
    for l in body:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        insts.append(s.split(";")[0].strip())
    print(f"{lines[start][:100]}  {len(insts)} instructions")
    for i, ins in enumerate(insts):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)", ins) or re.match(r"^s_branch\s+(\.LBB\d+_\d+)", ins)
        if m and m.group(1) in labels and labels[m.group(1)] <= i and i - labels[m.group(1)] >= minlen:
            lo = labels[m.group(1)]
            c = Counter(classify(x.split()[0]) for x in insts[lo:i + 1])
            ops = Counter(x.split()[0] for x in insts[lo:i + 1])
            cyc = 4 * c["valu"] + 8 * c["trans"] + 16 * c["imul"] + 8 * c["mfma"]
            print(f"loop {m.group(1)} [{lo}..{i}] len {i - lo + 1}: {dict(c)}  est VALU-issue cycles {cyc}")
            print("   top ops:", ", ".join(f"{k}:{v}" for k, v in ops.most_common(28)))


if __name__ == "__main__":
    main()
