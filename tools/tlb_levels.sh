#!/bin/bash
# the ROI and correction passes run at one of two levels per PROCESS (4.3 vs 4.6-4.9 ms, 3.30 vs 3.46).
# Suspect: address translation.  Several processes in a row, each under rocprofv3 --pmc with the UTCL1 (per-CU TLB)
# counters on those two kernels; per process: the kernels' durations and counter means.
#   bash tools/tlb_levels.sh <tag> [runs] [extra env assignment, e.g. PYTORCH_HIP_ALLOC_CONF=expandable_segments:True]
TAG=${1:-r4}
RUNS=${2:-5}
EXTRA=${3:-MG_DUMMY=1}
COUNTERS=${COUNTERS:-TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${TAG}_tlb_levels.txt
export TMPDIR=/tmp MG_NO_GRAPH=1
cd /tmp
echo "# $EXTRA" >> $OUT
for i in $(seq 1 $RUNS); do
  rm -rf /tmp/tlb_$i
  echo "process $i ..."
  env $EXTRA timeout -k 10 240 rocprofv3 --pmc $COUNTERS --kernel-trace --kernel-include-regex "k_roi_u16_even|k_apply_stitch_aligned|k_flatfield_max_lean" --output-format csv -d /tmp/tlb_$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-isolated > /tmp/tlb_bench_$i.json 2> /tmp/tlb_err_$i.log
  echo "== process $i" >> $OUT
  python3 $ROOT/tools/pmc_table.py /tmp/tlb_$i k_roi_u16_even k_apply_stitch_aligned k_flatfield_max_lean >> $OUT 2>&1 || tail -3 /tmp/tlb_err_$i.log >> $OUT
done
