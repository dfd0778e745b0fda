#!/bin/bash
# Round-2 call V: first-layer fused KPConv validation, kitti bench line (row capacity 512), default launch shape (4 x 16) A/B.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/v
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_bench_config_gpu.py tests/test_pipeline_gpu.py tests/test_neighbors_gpu.py tests/test_model_oracle.py -m gpu -q -x -p no:cacheprovider --timeout 600 > $OUT/tests.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 700 python $ROOT/bench.py --config kitti --steps 5 --warmup 1 --pairs 4 --no-fp32-mode > $OUT/bench_kitti.json 2> $OUT/bench_kitti.err; echo "kitti rc=$?"; tail -3 $OUT/bench_kitti.err
ab() { name=$1; shift; timeout 300 env ${ENVV:-X=1} python $ROOT/bench.py --no-cpu-baseline --no-fp32-mode "$@" > $OUT/ab_$name.json 2> $OUT/ab_$name.err; python -c "
import json
try:
    d=json.load(open('$OUT/ab_$name.json')); print('$name', d['value'], 'pairs/s', d['ms_per_step'],'ms/step')
except Exception as e: print('$name FAILED', e)" | tee -a $OUT/ab_runs.txt; }
ab default
ENVV="GEOTR_KPCONV_C1_FUSED=0" ab c1_two_kernel
ab default_again
ab stack8 --stack 8
ab lanes1 --lanes 1
