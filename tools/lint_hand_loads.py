#!/usr/bin/env python3
"""Lint of the hand-issued loads of the gfx950 kernels (no GPU needed: reads the compiler's assembly).

Some kernels issue loads in inline assembly and wait for them by hand (`global_load_dwordx4 ... sc1`, `ds_read_b128`,
`ds_read_b64_tr_b16` followed later by an `s_waitcnt` statement that names the destination registers): the compiler's
own wait-count bookkeeping would drain the prefetch.  The compiler does not know that such a register is
the target of a load in flight: it is free to spill it, or to hand it to another value on a path where the wait
statement does not consume it -- the first wrote an address register over by a late-landing load (a GPU memory fault in
round 3), the second overwrote the generator's 2^-33 constant in a peeled loop tail.

Invariant checked here, per kernel: between a hand-issued load with VGPR destinations and the `s_waitcnt` that retires
it, NO instruction reads or writes those destination registers (scratch stores / loads of them included).

Model: instructions in textual order; vector-memory operations (loads, stores, atomics, LDS-DMA) retire in issue order
under `vmcnt`, LDS operations in issue order under `lgkmcnt` (scalar loads share that counter and may return out of
order: a counted wait then guarantees AT LEAST the LDS retirements the in-order model gives, see below); `s_waitcnt
vmcnt(N)` / `lgkmcnt(N)` retires all but the N youngest entries of its queue.  With k scalar loads among the
outstanding lgkm operations a wait for N leaves at most N operations of any kind, so at least (total - N) - k of the
LDS reads have returned, the oldest first: the model subtracts nothing for scalar loads issued BEFORE the reads (they
only make the wait stricter) and treats a scalar load issued after a hand-issued read as one more outstanding entry.
Hand-issued = inside `;;#ASMSTART` .. `;;#ASMEND`.  Queues are cleared at `s_endpgm`; a kernel's loops are scanned in
textual order, which matches execution order inside a loop body and across the fall-through edges.

usage: python tools/lint_hand_loads.py [file.s ...]     (no arguments: compiles csrc/bbb_linear.hip and lr_linear.hip)
exit status 1 when a violation is found; `check(path)` returns (violations, hand-issued loads seen per kernel)."""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "bayesian-neural-network_amd", "csrc")
SOURCES = ("bbb_linear.hip", "lr_linear.hip")

_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(text):
    """Set of vector / accumulator registers named in an operand string: {'v12', 'a3', ...}."""
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.add(m.group(1) + m.group(2))
        else:
            out.update(m.group(3) + str(i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def _split(ins):
    parts = ins.split(None, 1)
    return parts[0], (parts[1] if len(parts) > 1 else "")


def _dest_regs(op, operands):
    """Destination VGPRs of a load instruction (first operand), or empty for LDS-DMA / stores."""
    if op.startswith("global_load_lds") or (op.startswith("buffer_load") and re.search(r"\blds\b", operands)):
        return set()                     # LDS-DMA: the vector operand is the lane OFFSET, the data lands in LDS
    return regs_of(operands.split(",")[0])


def is_vmem(op):
    return op.startswith(("global_", "buffer_", "scratch_", "flat_"))


def is_lds(op):
    return op.startswith("ds_")


def is_smem(op):
    return op.startswith(("s_load", "s_buffer_load"))


def check(path):
    violations, seen = [], {}
    kernel, in_asm = None, False
    vm, lgkm = [], []            # outstanding operations, oldest first: (hand dest regs | None, text, line number)

    def retire(queue, n):
        del queue[:max(0, len(queue) - n)]

    for ln, raw in enumerate(open(path), 1):
        line = raw.strip()
        m = re.match(r"^(_Z\S+):", line)
        if m:
            kernel, vm, lgkm = m.group(1), [], []
            continue
        if line.startswith(";;#ASMSTART") or line.startswith(";#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND") or line.startswith(";#ASMEND"):
            in_asm = False
            continue
        if not line or line.startswith((";", ".")) or line.endswith(":") or kernel is None:
            continue
        ins = line.split(";")[0].strip()
        if not ins:
            continue
        op, operands = _split(ins)
        if op == "s_endpgm":
            vm, lgkm = [], []
            continue
        if op == "s_waitcnt":
            mv = re.search(r"vmcnt\((\d+)\)", operands)
            ml = re.search(r"lgkmcnt\((\d+)\)", operands)
            if mv:
                retire(vm, int(mv.group(1)))
            if ml:
                retire(lgkm, int(ml.group(1)))
            if not mv and not ml and re.match(r"^(0x[0-9a-fA-F]+|\d+)$", operands.strip()):
                vm, lgkm = [], []                                  # a raw immediate: treat as a full drain
            continue
        # every other instruction: does it touch a register with a hand-issued load in flight?
        touched = regs_of(operands)
        for queue in (vm, lgkm):
            for dest, text, l0 in queue:
                if dest and (dest & touched):
                    violations.append((kernel, ln, ins, l0, text, sorted(dest & touched)))
        if is_vmem(op):
            hand = in_asm and ("load" in op) and not op.startswith("scratch_")
            dest = _dest_regs(op, operands) if hand else None
            vm.append((dest or None, ins, ln))
            if dest:
                seen[kernel] = seen.get(kernel, 0) + 1
        elif is_lds(op):
            hand = in_asm and op.startswith("ds_read")
            dest = _dest_regs(op, operands) if hand else None
            lgkm.append((dest or None, ins, ln))
            if dest:
                seen[kernel] = seen.get(kernel, 0) + 1
        elif is_smem(op):
            lgkm.append((None, ins, ln))
    return violations, seen


def compile_to_asm(src, out):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-S",
                    "--offload-device-only", os.path.join(CSRC, src), "-o", out], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)


def main(argv):
    paths = list(argv)
    tmp = None
    if not paths:
        tmp = tempfile.TemporaryDirectory()
        for src in SOURCES:
            paths.append(os.path.join(tmp.name, src + ".s"))
            compile_to_asm(src, paths[-1])
    bad = 0
    for p in paths:
        violations, seen = check(p)
        print(f"{os.path.basename(p)}: {sum(seen.values())} hand-issued loads with register destinations in {len(seen)} kernels, "
              f"{len(violations)} violations")
        for k, ln, ins, l0, text, regs in violations[:40]:
            print(f"  {k[:60]} line {ln}: `{ins}` touches {regs} of `{text}` (line {l0}) before its wait")
        bad += len(violations)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
