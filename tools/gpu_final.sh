#!/bin/bash
# The round's record runs on ONE box (so that the numbers belong together):  tools/gpu_final.sh <tag> [part ...]
#   rocprof   tools/profile_round.sh: rocprofv3 kernel stats + three PMC passes of bench.py
#   bench     bench.py with the driver's flags (full CPU oracle leg, full-grid parity)
#   ranks     virtual ranks 2 4 8 of the headline config + the other sharded configs at 8
#   configs   tools/bench_configs.py (roll-out eager / graph / to_host, AuroraHighRes, AuroraAirPollution)
#   kinds     tools/rank_kinds.py 8 4 + tools/step_shapes.py
cd "$(dirname "$0")/.." || exit 1
TAG=${1:-r04}; shift
WHAT=${*:-rocprof bench ranks configs kinds}
OUT=gpurun_out; mkdir -p $OUT
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has kinds; then
  timeout 600 python tools/step_shapes.py > $OUT/${TAG}_step_shapes.txt 2>/dev/null; head -3 $OUT/${TAG}_step_shapes.txt
  timeout 600 python tools/rank_kinds.py 8 4 > $OUT/${TAG}_rank_kinds_8_4.json 2> $OUT/${TAG}_rank_kinds.err; grep -o '"rank_total_ms": [0-9.]*' $OUT/${TAG}_rank_kinds_8_4.json
  grep '^{"by_shape"' $OUT/${TAG}_rank_kinds.err > $OUT/${TAG}_rank_shapes_8_4.json
fi
if has ranks; then
  timeout 900 python tools/bench_virtual_ranks.py 2 4 8 > $OUT/${TAG}_virtual_ranks.json 2>/dev/null; cat $OUT/${TAG}_virtual_ranks.json
  : > $OUT/${TAG}_virtual_ranks_configs.jsonl
  timeout 600 python tools/bench_virtual_ranks.py --model AuroraHighRes --grid 1801x3600 8 >> $OUT/${TAG}_virtual_ranks_configs.jsonl 2>/dev/null
  timeout 600 python tools/bench_virtual_ranks.py --model AuroraAirPollution --grid 451x900 8 >> $OUT/${TAG}_virtual_ranks_configs.jsonl 2>/dev/null
  cat $OUT/${TAG}_virtual_ranks_configs.jsonl
fi
if has configs; then timeout 900 python tools/bench_configs.py > $OUT/${TAG}_bench_configs.json 2>/dev/null; cat $OUT/${TAG}_bench_configs.json; fi
if has bench; then
  timeout 1500 python bench.py --steps 20 --warmup 3 2> $OUT/${TAG}_bench.err | grep '^{"metric"' | tail -1 > $OUT/${TAG}_bench.json
  python - <<PY
import json
d = json.load(open("$OUT/${TAG}_bench.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["frac_all_matrix_launches"], d["roofline"]["attention"]["frac"], d.get("cpu_baseline", {}).get("value"), d.get("parity_full_grid"))
PY
fi
if has rocprof; then bash tools/profile_round.sh $TAG > $OUT/${TAG}_profile_round.log 2>&1; tail -14 $OUT/${TAG}_profile_round.log; fi
