#!/bin/bash
# Round-2 call W: visiting order (grid order + XCD-contiguous tiles) for the gather kernels, adaptive GroupNorm statistics blocks.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/w
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_bench_config_gpu.py tests/test_pipeline_gpu.py tests/test_gemm_gpu.py tests/test_model_oracle.py -m gpu -q -x -p no:cacheprovider --timeout 600 > $OUT/tests.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
ab() { name=$1; shift; timeout 300 env ${ENVV:-X=1} python $ROOT/bench.py --no-cpu-baseline --no-fp32-mode "$@" > $OUT/ab_$name.json 2> $OUT/ab_$name.err; python -c "
import json
try:
    d=json.load(open('$OUT/ab_$name.json')); print('$name', d['value'], 'pairs/s', d['ms_per_step'],'ms/step')
except Exception as e: print('$name FAILED', e)" | tee -a $OUT/ab_runs.txt; }
ab default
ENVV="GEOTR_SPATIAL_ORDER=0" ab row_order
ab default_again
ENVV="GEOTR_SPATIAL_ORDER=0" ab row_order_again
ab lanes1 --lanes 1
ENVV="GEOTR_SPATIAL_ORDER=0" ab lanes1_row_order --lanes 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_l1 -o bench -- python $ROOT/bench.py --steps 6 --warmup 2 --lanes 1 --no-cpu-baseline --no-fp32-mode > $OUT/bench_l1_under_rocprof.json 2>/dev/null
python - <<'P'
import csv, glob, os
f = glob.glob(os.environ.get('GRAFT_REPO_ROOT','/root/repo') + '/gpurun_out/w/stats_l1/**/*kernel_stats.csv', recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:22]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} total ms {float(r['TotalDurationNs'])/1e6:9.2f} avg us {float(r['AverageNs'])/1e3:9.1f}")
P
