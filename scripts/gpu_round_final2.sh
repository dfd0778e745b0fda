#!/bin/bash
# Last check of the round: distributed GPU tests (they run bench.py under torchrun), pipeline tests, the driver's bench command.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/final2
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_dist_gpu.py tests/test_pipeline_gpu.py tests/test_abi.py -m gpu -q -p no:cacheprovider --timeout 600 > $OUT/tests.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench rc=$?"; grep "quarter\|NUMA\|timed region" $OUT/bench_driver_cmd.err | cut -c1-200
python -c "
import json; d=json.load(open('$OUT/bench_driver_cmd.json')); print('driver cmd', d['value'], d['ms_per_step'], 'parity', d['parity']['ok'], 'cpu', d['cpu_baseline']['value'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'], d['config']['host_binding'])"
