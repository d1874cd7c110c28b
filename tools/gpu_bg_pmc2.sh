#!/bin/bash
# clock + MFMA utilisation of harness binaries: usage tools/gpu_bg_pmc2.sh "<args>" bin...
A=$1; shift
ROOT=$(pwd); export TMPDIR=/tmp
for B in "$@"; do
  BIN=$(realpath $B); OUT=$ROOT/gpurun_out/pmc2_$(basename $B .out); rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && timeout -k 10 200 rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/p -o p -- $BIN $A > $OUT/log 2> $OUT/err)
  (cd /tmp && timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/t -o t -- $BIN $A > $OUT/tlog 2> $OUT/terr)
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "block_gemm" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in agg.items()}
dur = None
for f in glob.glob("$OUT/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "block_gemm" in r["Name"]: dur = float(r["AverageNs"]) / 1e3
cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
print("$B", "us(trace)=%.1f" % (dur or -1), "active_cycles=%.0f" % cyc, "clock_GHz=%.2f" % (cyc / (dur or 1) / 1e3), "mfma_util=%.3f" % (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024 + 1)),
      "wait_any=%.2f wait_inst=%.2f active=%.2f" % tuple(m.get(k, 0) / (m.get("SQ_WAVE_CYCLES", 1)) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")))
PY
done
