set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "roi or labels" > gpurun_out/r3_t1.log 2>&1 || { tail -30 gpurun_out/r3_t1.log; exit 1; }
tail -3 gpurun_out/r3_t1.log
for v in 0 1 2 3 4; do
  MG_ROI_VARIANT=$v python bench.py --timepoints 16 --steps 10 --warmup 3 --no-cpu > gpurun_out/r3_roi_v$v.json 2> gpurun_out/r3_roi_v$v.err || { tail -20 gpurun_out/r3_roi_v$v.err; exit 1; }
  python - <<PY
import json
r=json.load(open("gpurun_out/r3_roi_v$v.json"))
print("variant $v", r["ms_per_step"], r["stages"]["mg_roi_segment_reduce"])
PY
done
