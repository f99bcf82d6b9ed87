#!/bin/bash
# One GPU session while developing: GPU tests, a short bench (no CPU leg), virtual ranks, rank kinds.
#   tools/gpu_round.sh <tag> [tests|bench|ranks|all ...]
cd "$(dirname "$0")/.." || exit 1
TAG=${1:-r04}; shift
WHAT=${*:-all}
OUT=gpurun_out; mkdir -p $OUT
has() { [[ " $WHAT " == *" $1 "* || " $WHAT " == *" all "* ]]; }
if has tests; then timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/${TAG}_gpu_tests.log; tail -3 $OUT/${TAG}_gpu_tests.log; fi
if has optests; then timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sharded.py -m gpu -x -q > $OUT/${TAG}_gpu_optests.log 2>&1; echo "optests rc=$?" | tee -a $OUT/${TAG}_gpu_optests.log; tail -3 $OUT/${TAG}_gpu_optests.log; fi
if has bench; then
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2> $OUT/${TAG}_bench.err | grep '^{"metric"' | tail -1 > $OUT/${TAG}_bench_nocpu.json
  python - <<PY
import json
d = json.load(open("$OUT/${TAG}_bench_nocpu.json"))
print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["attention"]["frac"], {k: round(v, 2) for k, v in d.get("kernel_ms_per_step", {}).items()})
PY
fi
if has ranks; then timeout 900 python tools/bench_virtual_ranks.py ${RANKS:-8} > $OUT/${TAG}_virtual_ranks.json 2> $OUT/${TAG}_virtual_ranks.err; cat $OUT/${TAG}_virtual_ranks.json; fi
if has kinds; then timeout 600 python tools/rank_kinds.py 8 4 > $OUT/${TAG}_rank_kinds_8_4.json 2> $OUT/${TAG}_rank_kinds.err; python - <<PY
import json
d = json.load(open("$OUT/${TAG}_rank_kinds_8_4.json"))
print(d["rank_total_ms"], d["share_of_unsharded_total_ms"])
for r in d["per_kind"][:9]: print(r)
PY
fi
