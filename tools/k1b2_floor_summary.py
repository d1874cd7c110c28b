"""Per-wave-step figures of tools/k1b2_floor_probe.py's variants from a `rocprofv3 --pmc` pass over it.
usage: k1b2_floor_summary.py <rocprof output dir> <probe stdout log> [more rocprof dirs of other counter passes]"""
import collections
import csv
import glob
import re
import sys

dirs, log = [sys.argv[1]] + sys.argv[3:], sys.argv[2]
names = [re.match(r"VARIANT (.*?)\s+tune", l).group(1) for l in open(log) if l.startswith("VARIANT")]
WAVE_STEPS = 75 * 256 * 38            # 16-feature tiles x pairs x k-steps of the 1200 x 1200 layer
SIMDS = 1024
rows = collections.defaultdict(dict)  # dispatch id -> counter -> value
meta = {}
for d in dirs:
    per = collections.defaultdict(dict)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "bbb_fwd_gemm2_kernel" not in r["Kernel_Name"]:
                continue
            i = int(r["Dispatch_Id"])
            per[i][r["Counter_Name"]] = float(r["Counter_Value"])
            per[i]["_us"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    for rank, i in enumerate(sorted(per)):          # dispatch order = launch order
        for k, v in per[i].items():
            if k == "_us":
                rows[rank].setdefault("_us", []).append(v)
            else:
                rows[rank][k] = v
n = len(rows)
per_variant = n // max(1, len(names))
print(f"{n} dispatches, {len(names)} variants, {per_variant} each")
for vi, name in enumerate(names):
    sel = [rows[i] for i in range(vi * per_variant + 1, (vi + 1) * per_variant)]      # skip each variant's first launch
    m = collections.defaultdict(float)
    for r in sel:
        for k, v in r.items():
            m[k] += (sum(v) / len(v) if isinstance(v, list) else v) / len(sel)
    us = m["_us"]
    clk = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / us / 1e3 if us else 0.0          # GHz
    line = f"{name:30s} {us:7.1f} us  clock {clk:4.2f} GHz"
    if "SQ_INSTS_VALU" in m:
        line += f" | per wave-step: VALU insts {m['SQ_INSTS_VALU'] / WAVE_STEPS:6.1f}"
    if "SQ_INSTS_SALU" in m:
        line += f" SALU {m['SQ_INSTS_SALU'] / WAVE_STEPS:5.1f}"
    if "SQ_ACTIVE_INST_VALU" in m and m.get("SQ_INSTS_VALU"):
        line += f" | VALU busy cycles per inst {m['SQ_ACTIVE_INST_VALU'] * 4 / m['SQ_INSTS_VALU']:4.2f}"
        line += f", per wave-step {m['SQ_ACTIVE_INST_VALU'] * 4 / WAVE_STEPS:6.0f}"
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"] * 4
        line += f" | wave cycles per wave-step {wc / WAVE_STEPS:6.0f} (waves per SIMD {wc / (clk * 1e3 * us * SIMDS + 1e-9):4.2f})"
        for k, lab in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "issue-stalled"), ("SQ_WAIT_ANY", "parked")):
            if k in m:
                line += f" {lab} {m[k] / m['SQ_WAVE_CYCLES']:4.2f}"
    if "SQ_BUSY_CYCLES" in m:
        line += f" | SIMD-time per wave-step {clk * 1e3 * us * SIMDS / WAVE_STEPS:6.0f} cycles"
    print(line)
