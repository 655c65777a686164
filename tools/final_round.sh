#!/bin/bash
# The round's bench lines, run on the GPU box from the repository root:  bash tools/final_round.sh r3
# -> gpurun_out/<tag>_final_bench.json (the default command, CPU baseline included), _bench_8tp / _bench_1tp /
#    _noiseless_bench.json, _configs.json (C1-C3 through the drop-in API), _c5_*.json (pinned memory / files, sinks, two ranks),
#    _mode_r.json
set -e
TAG=${1:-r4}
OUT=gpurun_out
# (the driver's command of rounds 1-3; `python bench.py` alone runs 10 steps after 6 warm-up calls)
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_final_bench.json 2> $OUT/${TAG}_final_bench.err || { tail -20 $OUT/${TAG}_final_bench.err; exit 1; }
echo default done
python bench.py --timepoints 8 --steps 40 --warmup 3 --no-cpu --no-isolated > $OUT/${TAG}_bench_8tp.json 2> $OUT/${TAG}_bench_8tp.err
python bench.py --timepoints 1 --steps 40 --warmup 3 --no-cpu --no-isolated > $OUT/${TAG}_bench_1tp.json 2> $OUT/${TAG}_bench_1tp.err
python bench.py --noiseless --steps 10 --warmup 2 --no-cpu --no-isolated > $OUT/${TAG}_noiseless_bench.json 2> $OUT/${TAG}_noiseless_bench.err
python bench.py --steps 20 --warmup 3 --no-cpu --no-isolated > $OUT/${TAG}_bench_20steps.json 2> $OUT/${TAG}_bench_20steps.err
echo benches done
timeout -k 10 600 python tests/config_table.py --out $OUT/${TAG}_configs.json > $OUT/${TAG}_configs.log 2>&1
echo configs done
timeout -k 10 500 python tools/c5_stream_bench.py --timepoints 64 > $OUT/${TAG}_c5_mem.json 2> $OUT/${TAG}_c5_mem.err
timeout -k 10 500 python tools/c5_stream_bench.py --timepoints 64 --sink save > $OUT/${TAG}_c5_mem_save.json 2> $OUT/${TAG}_c5_mem_save.err
timeout -k 10 800 python tools/c5_stream_bench.py --timepoints 32 --files /tmp/c5_series --reader-only > $OUT/${TAG}_c5_files.json 2> $OUT/${TAG}_c5_files.err
timeout -k 10 800 python tools/c5_stream_bench.py --timepoints 32 --files /tmp/c5_series --sink save > $OUT/${TAG}_c5_files_save.json 2> $OUT/${TAG}_c5_files_save.err
timeout -k 10 800 python tools/c5_stream_bench.py --timepoints 32 --files /tmp/c5_series --sink save --want-roi > $OUT/${TAG}_c5_files_save_roi.json 2> $OUT/${TAG}_c5_files_save_roi.err
timeout -k 10 800 python tools/c5_stream_bench.py --timepoints 32 --files /tmp/c5_series2 --sink save --gpus 2 > $OUT/${TAG}_c5_files_save_2ranks.json 2> $OUT/${TAG}_c5_files_save_2ranks.err
echo c5 done
timeout -k 10 300 python tools/mode_r_bench.py > $OUT/${TAG}_mode_r.json 2> $OUT/${TAG}_mode_r.err || tail -3 $OUT/${TAG}_mode_r.err
cat $OUT/${TAG}_mode_r.json | cut -c 1-600
python - <<PY
import json
for n in ("final_bench", "bench_20steps", "bench_8tp", "bench_1tp", "noiseless_bench"):
    r = json.load(open("$OUT/${TAG}_%s.json" % n))
    print(n, round(r["ms_per_step"], 3), round(r["value"]), r["roofline"]["frac"] and round(r["roofline"]["frac"], 3), r.get("cpu_baseline") and r["cpu_baseline"]["value"])
for n in ("c5_mem", "c5_mem_save", "c5_files", "c5_files_save", "c5_files_save_roi", "c5_files_save_2ranks"):
    r = json.load(open("$OUT/${TAG}_%s.json" % n))
    print(n, {k: r[k] for k in r if k in ("ms_per_timepoint", "stitched_MPs", "host_to_device_GBs", "sink_writer", "reader_alone")})
PY
