set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r3_t2.log 2>&1 || { tail -40 gpurun_out/r3_t2.log; exit 1; }
tail -3 gpurun_out/r3_t2.log
for tp in 64 8; do
for v in full sweeps; do
  if [ $v = sweeps ]; then export MG_HYST_SWEEPS=1; else unset MG_HYST_SWEEPS; fi
  timeout -k 10 300 python bench.py --timepoints $tp --steps 10 --warmup 3 --no-cpu > gpurun_out/r3_hy_${v}_$tp.json 2> gpurun_out/r3_hy_${v}_$tp.err || { tail -20 gpurun_out/r3_hy_${v}_$tp.err; exit 1; }
  python - <<PY
import json
r=json.load(open("gpurun_out/r3_hy_${v}_$tp.json"))
print("$v $tp", round(r["ms_per_step"],3), {k:(v["ms_per_step"],v["launches_per_step"]) for k,v in r["stages"].items()})
PY
done
done
