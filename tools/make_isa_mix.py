#!/usr/bin/env python3
"""Instruction mix of the steady-state k-step loop of the throughput kernels -> profiles/isa_mix.json.

bench.py prices the kernel that is VALU-bound (K1b: the on-chip Philox / Box-Muller generator feeding the matrix core)
against the vector-issue roof with these counts: issue cycles per k-step and wave = 2 x plain VALU + 8 x transcendental
+ 7 x 32-bit integer multiply (v_mad_u64_u32 ...) wave-instructions, the per-SIMD throughput costs measured by
tools/ubench.hip (profiles/r02_ubench_generator.log: 16 fma + 16 add = 53 cycles per SIMD at 3 waves -> 1.7 per
instruction; box_muller x2 = 8 transcendentals + ~16 VALU = 99; Philox4x32-10 = 20 multiplies + 40 xor + ... = 164).
Compiles the kernel sources to gfx950 assembly here (hipcc -S, no GPU needed) and picks, per kernel, the loop with
the most MFMAs per iteration among the backward-branch loops -- the k-step loop.  Each entry carries the hash of the
source files it was made from; bench.py ignores stale entries.
usage: python tools/make_isa_mix.py"""
import json
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
from bench import source_hash, KERNEL_SOURCES          # noqa: E402
from isa_loop_mix import classify                       # noqa: E402

CSRC = os.path.join(REPO, "bayesian-neural-network_amd", "csrc")
KERNELS = {   # key -> (source file, mangled-name substring, family of bench.KERNEL_SOURCES)
    "bbb_fwd_gemm2_kernel<4,2,philox>": ("bbb_linear.hip", "bbb_fwd_gemm2_kernelILi4ELi2ELi0ELi2ELb0E", "bbb"),
    "bbb_fwd_gemm2_kernel<4,2,philox,x3>": ("bbb_linear.hip", "bbb_fwd_gemm2_kernelILi4ELi2ELi0ELi2ELb1E", "bbb"),
    "bbb_fwd_gemm_kernel<4,true,philox>": ("bbb_linear.hip", "bbb_fwd_gemm_kernelILi4ELb1ELi0", "bbb"),
    "bbb_fwd_gemm_kernel<4,false,philox>": ("bbb_linear.hip", "bbb_fwd_gemm_kernelILi4ELb0ELi0", "bbb"),
    "lr_fwd_gemm_kernel<16,true,2>": ("lr_linear.hip", "lr_fwd_gemm_kernelILi16ELb1ELi2", "lr"),
}
CYCLES = {"valu": 2, "trans": 8, "imul": 7}   # ubench: 16 v_mad_u64_u32 + 32 v_xor = 170 cycles per SIMD wave-slot -> 6.6 per multiply


def loops(path, kname):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l) and kname in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    labels, insts = {}, []
    for l in lines[start:end + 1]:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        insts.append(s.split(";")[0].strip())
    out = []
    for i, ins in enumerate(insts):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)", ins) or re.match(r"^s_branch\s+(\.LBB\d+_\d+)", ins)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            lo = labels[m.group(1)]
            out.append((Counter(classify(x.split()[0]) for x in insts[lo:i + 1]), Counter(x.split()[0] for x in insts[lo:i + 1]), i - lo + 1))
    return out


def main():
    data = {}
    asm = {}
    with tempfile.TemporaryDirectory() as td:
        for key, (src, kname, fam) in KERNELS.items():
            if src not in asm:
                asm[src] = os.path.join(td, src + ".s")
                subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-S",
                                "--offload-device-only", os.path.join(CSRC, src), "-o", asm[src]], check=True,
                               stderr=subprocess.DEVNULL)
            cand = [c for c in loops(asm[src], kname) if c[0]["mfma"] > 0]
            if key.startswith("bbb"):      # the Philox k-step loop: the shortest loop holding both the multiplies and the MFMAs
                cls, ops, length = min((c for c in cand if c[0]["imul"] > 0), key=lambda c: c[2])
            else:                          # K3b: the unrolled steady-state body (six k-steps)
                cls, ops, length = max(cand, key=lambda c: c[0]["mfma"])
            # the static body holds branches the benchmark configuration does not take: the scale-mixture prior (2 v_exp + 1
            # v_log + ~3 plain VALU per weight) and the log sigma sum of sample 0 (8 v_log + 8 fma): remove them for the
            # EXECUTED path of a Gaussian-prior step of a sample other than 0
            n_exp = sum(v for k, v in ops.items() if k.startswith("v_exp_"))
            n_log = sum(v for k, v in ops.items() if k.startswith("v_log_"))
            mix_logs = n_exp // 2
            ls_logs = 8 if key.startswith("bbb") and n_log - mix_logs >= 12 else 0
            execd = {"valu": cls["valu"] - 3 * n_exp - ls_logs, "trans": cls["trans"] - n_exp - mix_logs - ls_logs, "imul": cls["imul"]}
            cyc = sum(CYCLES[k] * execd[k] for k in CYCLES)
            data[key] = {"instructions": length, "classes": dict(cls), "executed_classes": execd, "mfma_per_iteration": cls["mfma"],
                         "valu_issue_cycles_per_iteration": cyc, "cycle_costs": CYCLES,
                         "top_ops": dict(ops.most_common(16)), "source_hash": source_hash(KERNEL_SOURCES[fam])}
            print(key, json.dumps({k: v for k, v in data[key].items() if k != "top_ops"}))
    json.dump(data, open(os.path.join(REPO, "profiles", "isa_mix.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
