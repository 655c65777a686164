#!/bin/bash
# The round's rocprofv3 evidence, run on the GPU box from the repository root:
#   bash tools/profile_round.sh r3        ->  gpurun_out/<tag>_{stats,pmc_fetch,pmc_write,sq1,sq2}/ + bench JSON lines
# Kernel trace + stats in one run, every --pmc counter set in a run of its own (gpurun refuses --pmc together with the
# trace domains); the chain is launched eagerly (MG_NO_GRAPH=1) so that every kernel is a dispatch of its own.
set -e
TAG=${1:-r3}
ROOT=$(pwd)
export TMPDIR=/tmp MG_NO_GRAPH=1
cd /tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/${TAG}_stats -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu > $ROOT/gpurun_out/${TAG}_stats_bench.json 2> $ROOT/gpurun_out/${TAG}_stats.log
echo stats done
rocprofv3 --pmc FETCH_SIZE -d $ROOT/gpurun_out/${TAG}_pmc_fetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2> $ROOT/gpurun_out/${TAG}_pmc_fetch.log
echo fetch done
rocprofv3 --pmc WRITE_SIZE -d $ROOT/gpurun_out/${TAG}_pmc_write -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2> $ROOT/gpurun_out/${TAG}_pmc_write.log
echo write done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $ROOT/gpurun_out/${TAG}_sq1 -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --timepoints 16 > /dev/null 2> $ROOT/gpurun_out/${TAG}_sq1.log
echo sq done
