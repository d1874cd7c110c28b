#!/bin/bash
# clock + matrix-pipe utilisation per kernel of any harness binary (one --pmc pass; durations from the same dispatches):
# usage tools/gpu_pmc_any.sh <tag> <bin> [args...]
TAG=$1; shift; BIN=$(realpath $1); shift
ROOT=$(pwd); export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmcany_$TAG; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout -k 10 300 rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/p -o p -- $BIN "$@" > $OUT/log 2> $OUT/err)
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
order = []
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"], r["Grid_Size"], r["LDS_Block_Size"])
        if key not in order: order.append(key)
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key][r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
with open("$OUT/summary.txt", "w") as out:
    for key in order:
        m = {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in agg[key].items()}
        d = list(dur[key].values())
        us = sum(d[1:]) / max(1, len(d) - 1)
        cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
        line = "%-62s grid %s lds %s: %.1f us (%d launches), clock %.2f GHz, matrix pipe busy %.3f, wave wait_inst %.2f active %.2f" % (
            key[0][:62], key[1], key[2], us, len(d), cyc / max(us, 1e-9) / 1e3, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024 + 1),
            m.get("SQ_WAIT_INST_ANY", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1)), m.get("SQ_ACTIVE_INST_ANY", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1)))
        print(line); out.write(line + "\n")
PY
