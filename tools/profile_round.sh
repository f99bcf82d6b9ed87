#!/bin/bash
# Round-end profiling ON the GPU box: kernel-trace stats + three PMC passes of bench.py, summarised into gpurun_out/.
#   tools/profile_round.sh <tag>
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
# what exactly is being profiled: SHA-256 of every kernel source and of the library the runs below load (recorded ON this
# box, before the runs; tools/rocprof_summary.py and tools/pmc_rollup.py carry them into the files kept under profiles/)
(cd $ROOT && sha256sum aurora_amd/csrc/*.hip aurora_amd/csrc/*.h aurora_amd/_lib/libaurora_hip.so) > $OUT/${TAG}_sources.sha256
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write /tmp/prof_sq
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o st -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_stats_bench.log 2>&1
DB=$(find /tmp/prof_stats -name "*.db" | head -1)
python $ROOT/tools/rocprof_summary.py $DB $OUT/${TAG}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline" $OUT/${TAG}_sources.sha256 > /dev/null 2>$OUT/${TAG}_summary.err
grep '^{"metric"' $OUT/${TAG}_stats_bench.log | tail -1 > $OUT/${TAG}_bench_under_rocprof.json
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_fetch -o fetch -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -o write -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d /tmp/prof_sq -o sq -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/sq.log 2>&1
F=$(find /tmp/prof_fetch -name "*counter_collection.csv" | head -1)
W=$(find /tmp/prof_write -name "*counter_collection.csv" | head -1)
S=$(find /tmp/prof_sq -name "*counter_collection.csv" | head -1)
python $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_summary.json fetch=$F write=$W sq=$S > /dev/null 2>$OUT/${TAG}_pmc.err
ls -la $OUT | tail -12
head -12 $OUT/${TAG}_kernel_stats.txt
