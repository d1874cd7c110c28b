#!/bin/bash
# PMC passes over the K1g harness: usage  tools/gpu_block_gemm_pmc.sh <binary> "<harness args>" <tag>
BIN=$(realpath $1); ARGS=$2; TAG=$3
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --output-format csv --pmc $ctr -d $OUT/p$i -o p -- $BIN $ARGS > $OUT/p$i.log 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "block_gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in agg.items()}   # skip the first (cold) dispatch
for k in sorted(m): print(f"{k:36s} {m[k]:.4g}")
if "GRBM_GUI_ACTIVE" in m:
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print("active cycles", cyc, "mfma_util", m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024))
PY
