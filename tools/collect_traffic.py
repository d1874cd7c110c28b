"""Turn two rocprofv3 --pmc passes (FETCH_SIZE ; WRITE_SIZE) of bench.py into the per-launch HBM
traffic of the dominant kernel, applying the gfx950 correction of MI355X_MICROARCH.md §HBM
(FETCH_SIZE reports 1/2 of wide coalesced reads; counter unit = KiB).  Writes/updates
profiles/traffic.json, which bench.py reads to fill roofline.traffic (only while the kernel sources still hash
to the value recorded here).
usage: collect_traffic.py <fetch_dir> <write_dir> <key> <kernel-substring>"""
import csv, glob, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import source_hash, KERNEL_SOURCES   # entries are stamped with the kernel sources they were measured on

def mean_counter(d, counter, kern):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter and kern in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)

fetch_dir, write_dir, key, kern = sys.argv[1:5]
fetch_kib, nf = mean_counter(fetch_dir, "FETCH_SIZE", kern)
write_kib, nw = mean_counter(write_dir, "WRITE_SIZE", kern)
hbm = 2.0 * fetch_kib * 1024.0 + write_kib * 1024.0
out_path = os.path.join(os.environ.get("BNN_PROFILES_DIR") or os.path.join(REPO, "profiles"), "traffic.json")
data = json.load(open(out_path)) if os.path.exists(out_path) else {}
data[key] = {"kernel": kern, "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
             "hbm_bytes_per_launch": hbm, "dispatches_averaged": [nf, nw], "source_hash": source_hash(KERNEL_SOURCES["lr" if key.startswith("lr") else "block_gemm" if key.startswith("block") else "bbb"]),
             "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); reads x2 (gfx950 FETCH_SIZE "
                       "counts 64 B per 128 B request); fabric-side counter: Infinity-Cache hits are included"}
json.dump(data, open(out_path, "w"), indent=1)
print(key, json.dumps(data[key]))
