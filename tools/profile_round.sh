#!/bin/bash
# The round's rocprofv3 evidence, run on the GPU box from the repository root:
#   bash tools/profile_round.sh r3   ->  gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc_traffic.json, <tag>_sq_counters.txt,
#                                        <tag>_stats_bench.json (the bench line of the traced run)
# Kernel trace + stats in one run, every --pmc counter set in a run of its own (gpurun refuses --pmc together with the
# trace domains); the chain is launched eagerly (MG_NO_GRAPH=1) so that every kernel is a dispatch of its own.  Only the
# summaries stay under gpurun_out/ (the raw traces are hundreds of megabytes).
set -e
TAG=${1:-r4}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
RAW=/tmp/mg_prof_$TAG
rm -rf $RAW; mkdir -p $RAW $OUT
# MG_PLACEMENT_TRIES=0: the first call's placement trial launches the flat-field and ROI passes a few times more --
# the per-step division of the counters below would count them as steps' traffic
export TMPDIR=/tmp MG_NO_GRAPH=1 MG_PLACEMENT_TRIES=0
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu > $OUT/${TAG}_stats_bench.json 2> $OUT/${TAG}_stats.log
cp $(find $RAW/stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
echo stats done
STEPS=3
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $RAW/fetch -- python3 $ROOT/bench.py --steps $STEPS --warmup 1 --no-cpu > /dev/null 2> $OUT/${TAG}_pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $RAW/write -- python3 $ROOT/bench.py --steps $STEPS --warmup 1 --no-cpu > /dev/null 2> $OUT/${TAG}_pmc_write.log
echo pmc done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $RAW/sq -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $OUT/${TAG}_sq.log || \
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $RAW/sq -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $OUT/${TAG}_sq.log
echo sq done
cd $ROOT
# bench runs warmup + steps + the per-stage profile pass (min(steps, 5), at least 2) + 1 result step per process
python3 tools/summarize_pmc.py $TAG $RAW/fetch $RAW/write $((1 + STEPS + STEPS + 1)) > $OUT/${TAG}_pmc_traffic.json
python3 tools/pmc_table.py $RAW/sq > $OUT/${TAG}_sq_counters.txt
rm -rf $RAW
ls -la $OUT | grep ${TAG}_
