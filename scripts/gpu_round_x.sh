#!/bin/bash
# Round-2 call X: visiting order with the rows prefetched one tile ahead -- A/B against row order, per-kernel durations on one lane.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/x
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_backbone_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 600 > $OUT/tests.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/tests.log
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
for mode in 1 0; do
GEOTR_SPATIAL_ORDER=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_l1_$mode -o bench -- python $ROOT/bench.py --steps 6 --warmup 2 --lanes 1 --no-cpu-baseline --no-fp32-mode > $OUT/bench_l1_$mode.json 2>/dev/null
python - $mode <<'P'
import csv, glob, os, sys
f = glob.glob(os.environ.get('GRAFT_REPO_ROOT','/root/repo') + '/gpurun_out/x/stats_l1_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True)
print('order mode', sys.argv[1])
if f:
    rows = list(csv.DictReader(open(f[0])))
    for r in rows:
        if any(k in r['Name'] for k in ('kpconv', 'maxpool', 'rg_order')):
            print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5s} total ms {float(r['TotalDurationNs'])/1e6:8.2f} avg us {float(r['AverageNs'])/1e3:8.1f}")
P
done
