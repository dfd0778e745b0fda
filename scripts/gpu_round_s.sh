#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/s
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_heads_gpu.py tests/test_pipeline_gpu.py tests/test_bench_config_gpu.py tests/test_configs_gpu.py -m gpu -q -p no:cacheprovider --timeout 600 > $OUT/gputests.log 2>&1
echo "pytest rc=$?" >> $OUT/gputests.log
tail -6 $OUT/gputests.log
cd /tmp && export TMPDIR=/tmp
ab() { name=$1; shift; timeout 300 env "$@" python $ROOT/bench.py --no-cpu-baseline --no-fp32-mode ${EXTRA:-} > $OUT/ab_$name.json 2> $OUT/ab_$name.err; python -c "
import json,sys
try:
    d=json.load(open('$OUT/ab_$name.json')); r=d['roofline']; k=[x for x in [r, r.get('other')] if x and 'kpconv' in x['kernel']]
    print('$name', d['value'], 'pairs/s', d['ms_per_step'],'ms/step', [(x['avg_launch_us'], x['frac']) for x in k])
except Exception as e: print('$name FAILED', e)"; }
EXTRA="" ab default X=1
EXTRA="" ab unfused_kpconv GEOTR_KPCONV_FUSED=0
EXTRA="--lanes 1" ab lanes1 X=1
EXTRA="--lanes 1" ab lanes1_unfused GEOTR_KPCONV_FUSED=0
EXTRA="--lanes 3" ab lanes3 X=1
EXTRA="--lanes 5" ab lanes5 X=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_l1 -o bench -- python $ROOT/bench.py --steps 6 --warmup 2 --lanes 1 --no-cpu-baseline --no-fp32-mode > $OUT/bench_l1_under_rocprof.json 2>/dev/null
