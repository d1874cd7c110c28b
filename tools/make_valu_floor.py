#!/usr/bin/env python3
"""profiles/r03_k1b_ablate.log (tools/k1b_ablate.py on the tuning build, run by tools/profile_final.sh) -> profiles/valu_floor.json:
the headline kernel's IN-SITU vector floor -- K1b2 with its DMA, x reads, MFMAs, barrier and waits compiled out (tune 59), i.e.
nothing left but the generator, w, the statistics and the epilogue -- beside the full launch of the same build (tune 0).
bench.py attaches it to roofline.valu when the kernel sources have not changed since.  usage: python tools/make_valu_floor.py [log]"""
import json, os, re, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import source_hash, KERNEL_SOURCES          # noqa: E402

log = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "profiles", "r03_k1b_ablate.log")
section, rows = None, {}
for line in open(log):
    m = re.match(r"== BNN_TUNE_PAIRS=(\d)", line)
    if m:
        section = m.group(1)
        continue
    m = re.match(r"(philox, stats, hoisted sigma)\s+tune\s+(\d+) form \d+ blocks\s+(\d+):\s+([\d.]+) us per launch", line)
    if m and section == "1":
        rows[int(m.group(2))] = (float(m.group(4)), int(m.group(3)))
full, floor = rows[0][0], rows[59][0]
out = {"kernel": "K1b2 bbb_fwd_gemm2_kernel<4,2,philox>, 1200 x 1200, 256 pairs, on-chip eps + statistics, hoisted sigma",
       "full_launch_us": full, "vector_work_only_us": floor, "without_dma_us": rows[8][0], "frac": floor / full, "blocks": rows[0][1],
       "source": "profiles/r03_k1b_ablate.log: tuning build (make tune), BNN_TUNE_K1B 0 against 59 = no LDS-DMA staging, no x-fragment "
                 "reads, no MFMAs, no barrier, no waits; the tuning build's full launch is a few % slower than the product's",
       "source_hash": source_hash(KERNEL_SOURCES["bbb"])}
json.dump(out, open(os.path.join(REPO, "profiles", "valu_floor.json"), "w"), indent=1)
print(json.dumps(out))
