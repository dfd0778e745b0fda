#!/bin/bash
# Round-2 call G: shortcut GroupNorm applied inside the main branch's apply pass (geotr_group_norm_shortcut) -- tests + A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/g
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_pipeline_gpu.py tests/test_configs_gpu.py tests/test_bench_config_gpu.py tests/test_heads_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 600 > $OUT/tests.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 env ${ENVV:-X=1} python $ROOT/bench.py --no-cpu-baseline --no-fp32-mode --gpus 1 --steps 20 --warmup 5 "$@" > $OUT/$name.json 2> $OUT/$name.err; python -c "
import json
try:
    d=json.load(open('$OUT/$name.json')); print('$name', d['value'], 'pairs/s', d['ms_per_step'],'ms/step')
except Exception as e: print('$name FAILED', e)" | tee -a $OUT/runs.txt; }
run fused
ENVV="GEOTR_GN_SHORTCUT_FUSED=0" run two_pass
run fused_again
ENVV="GEOTR_GN_SHORTCUT_FUSED=0" run two_pass_again
