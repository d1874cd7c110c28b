"""Summarise one `rocprofv3 --pmc ...` pass of `bench.py --roofline-only` into profiles/pmc.json:
per bnn:: kernel the mean counters per dispatch and the utilisations derived from them with the
gfx94x formulas (MI355X_MICROARCH.md: ROCm 7.2 has no gfx950 derived-counter section):
   MFMA util  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs)
   VALU busy  = SQ_ACTIVE_INST_VALU * 4 / SQ_BUSY_CU_CYCLES-equivalent  (quad-cycle units -> x4)
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs.  clock_ghz = (GRBM_GUI_ACTIVE / 8) / the dispatch's duration in the same pass.
usage: collect_pmc.py <pmc_dir> <key>"""
import collections, csv, glob, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import source_hash, KERNEL_SOURCES      # entries carry the hash of the kernel sources they were measured on
d, key = sys.argv[1:3]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)                           # kernel -> dispatch id -> us (the counter pass's own timestamps)
for r in csv.DictReader(open(f)):
    if "bnn::" in r["Kernel_Name"]:
        kern = r["Kernel_Name"].split("(")[0]
        agg[kern][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur[kern][r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
out_path = os.path.join(os.environ.get("BNN_PROFILES_DIR") or os.path.join(REPO, "profiles"), "pmc.json")
data = json.load(open(out_path)) if os.path.exists(out_path) else {}
for kern, cs in agg.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    fam = "lr" if "lr_" in kern else "block_gemm" if "block_gemm" in kern else "bbb"
    e = {"dispatches_averaged": len(next(iter(cs.values()))), "counters_mean_per_dispatch": m, "source_hash": source_hash(KERNEL_SOURCES[fam])}
    if "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0                      # shader cycles the dispatch was active
        simd_cycles = cyc * 256 * 4
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            e["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
        if "SQ_ACTIVE_INST_VALU" in m:
            e["valu_busy"] = m["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles
        if "SQ_WAVE_CYCLES" in m:
            e["waves_per_simd"] = m["SQ_WAVE_CYCLES"] * 4.0 / simd_cycles
        e["active_cycles"] = cyc
        if dur.get(kern) and sum(dur[kern].values()) / len(dur[kern]) >= 100.0:      # (short dispatches: the counter window is not the kernel)
            # DVFS: the shader clock this kernel ran at UNDER THE COUNTER PASS (report wall and cycles) --
            # active cycles over the same dispatches' durations
            e["dispatch_us"] = sum(dur[kern].values()) / len(dur[kern])
            e["clock_ghz"] = cyc / e["dispatch_us"] / 1e3
    data[f"{key}:{kern}"] = e
    print(key, kern, json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items() if k != "counters_mean_per_dispatch"}))
json.dump(data, open(out_path, "w"), indent=1)
