#!/bin/bash
# Round-2 call Y: is the first bench run on a box slower than the following ones?  (911 after the test suite vs 1050 minutes later)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/y
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-fp32-mode "$@" > $OUT/$name.json 2> $OUT/$name.err; python -c "
import json
try:
    d=json.load(open('$OUT/$name.json')); print('$name', d['value'], 'pairs/s', d['ms_per_step'],'ms/step')
except Exception as e: print('$name FAILED', e)" | tee -a $OUT/runs.txt; grep quarter $OUT/$name.err | tee -a $OUT/runs.txt; }
run first --gpus 1 --steps 20 --warmup 5
run second --gpus 1 --steps 20 --warmup 5
run third --gpus 1 --steps 20 --warmup 5
cd $ROOT
timeout 600 python -m pytest tests/test_bench_config_gpu.py tests/test_dist_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 600 > $OUT/tests.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/runs.txt; tail -3 $OUT/tests.log
cd /tmp
run after_tests --gpus 1 --steps 20 --warmup 5
run after_tests_2 --gpus 1 --steps 20 --warmup 5
run long --gpus 1 --steps 100 --warmup 10
