#!/bin/bash
# Profiles of one round on the GPU box: rocprofv3 kernel-trace summaries and PMC traffic passes of the dominant
# kernels, written under gpurun_out/prof_<tag>/ and summarised into profiles/ (copy the files you want judged).
# usage: tools/profile_round.sh <round tag, e.g. r02> [extra bench args for the headline run]
# rocprofv3 gets `python3 <script>` directly after `--` (no env / shell hop: the profiler initialises the GPU first).
set -o pipefail
TAG=${1:-rXX}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
# only gpurun_out/ travels back from the GPU box: summaries go to gpurun_out/profiles/ (copy them into profiles/ afterwards)
PROF=$ROOT/gpurun_out/profiles
mkdir -p "$OUT" "$PROF"
for f in traffic.json pmc.json; do [ -f "$PROF/$f" ] || cp "$ROOT/profiles/$f" "$PROF/$f" 2>/dev/null; done
export BNN_PROFILES_DIR=$PROF
export TMPDIR=/tmp
cd /tmp

stats() {   # stats <name> <bench args...>: kernel-trace summary of bench.py --roofline-only
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/$name" -o "$name" -- python3 "$ROOT/bench.py" --roofline-only "$@" \
      > "$OUT/$name.json" 2> "$OUT/$name.err" || return $?
  cp "$(find "$OUT/$name" -name '*kernel_stats.csv' | head -1)" "$PROF/${TAG}_${name}_kernel_stats.csv"
  tail -1 "$OUT/$name.json" > "$PROF/${TAG}_${name}_roofline.json"
  echo "== $name"; head -4 "$PROF/${TAG}_${name}_kernel_stats.csv"
}

statsfull() {   # statsfull <name> <bench args...>: kernel-trace summary of a whole bench.py run (the driver's command)
  local name=$1; shift
  timeout -k 10 500 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/$name" -o "$name" -- python3 "$ROOT/bench.py" "$@" \
      > "$OUT/$name.json" 2> "$OUT/$name.err" || return $?
  cp "$(find "$OUT/$name" -name '*kernel_stats.csv' | head -1)" "$PROF/${TAG}_${name}_kernel_stats.csv"
  tail -1 "$OUT/$name.json" > "$PROF/${TAG}_bench_${name}.json"
  echo "== $name"; head -5 "$PROF/${TAG}_${name}_kernel_stats.csv"
}

traffic() { # traffic <name> <key> <kernel substring> <bench args...>: FETCH_SIZE / WRITE_SIZE in separate passes
  local name=$1 key=$2 kern=$3; shift 3
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --output-format csv --pmc $c -d "$OUT/${name}_$c" -o "$name" -- python3 "$ROOT/bench.py" --roofline-only "$@" \
        > /dev/null 2> "$OUT/${name}_$c.err" || return $?
  done
  python3 "$ROOT/tools/collect_traffic.py" "$OUT/${name}_FETCH_SIZE" "$OUT/${name}_WRITE_SIZE" "$key" "$kern"
}

pmc() {     # pmc <name> <key> <counters> <bench args...>
  local name=$1 key=$2 ctr=$3; shift 3
  timeout -k 10 300 rocprofv3 --output-format csv --pmc $ctr -d "$OUT/${name}_pmc" -o "$name" -- python3 "$ROOT/bench.py" --roofline-only "$@" \
      > /dev/null 2> "$OUT/${name}_pmc.err" || return $?
  python3 "$ROOT/tools/collect_pmc.py" "$OUT/${name}_pmc" "$key"
}

statspy() { # statspy <name> <script> <args...>: kernel-trace summary of any script of this repo
  local name=$1 script=$2; shift 2
  timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/$name" -o "$name" -- python3 "$ROOT/$script" "$@" \
      > "$OUT/$name.log" 2> "$OUT/$name.err" || return $?
  cp "$(find "$OUT/$name" -name '*kernel_stats.csv' | head -1)" "$PROF/${TAG}_${name}_kernel_stats.csv"
  cp "$OUT/$name.log" "$PROF/${TAG}_${name}.log"
  echo "== $name"; head -4 "$PROF/${TAG}_${name}_kernel_stats.csv"; tail -4 "$OUT/$name.log"
}

"$@"
