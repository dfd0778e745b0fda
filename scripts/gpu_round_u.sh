#!/bin/bash
# configs[3] / configs[4] bench lines (own parity blocks) + stack / lane shape experiments
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/u
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 500 python $ROOT/bench.py --config lomatch --precision bf16 --no-fp32-mode > $OUT/bench_lomatch_bf16.json 2> $OUT/bench_lomatch_bf16.err; echo "lomatch rc=$?"; tail -2 $OUT/bench_lomatch_bf16.err
timeout 500 python $ROOT/bench.py --config lomatch --no-fp32-mode --no-cpu-baseline > $OUT/bench_lomatch_bf16x3.json 2> $OUT/bench_lomatch_bf16x3.err; echo "lomatch bf16x3 rc=$?"
timeout 700 python $ROOT/bench.py --config kitti --lanes 2 --stack 4 --batch 8 --steps 5 --warmup 1 --pairs 4 --no-fp32-mode > $OUT/bench_kitti.json 2> $OUT/bench_kitti.err; echo "kitti rc=$?"; tail -3 $OUT/bench_kitti.err
timeout 300 python $ROOT/bench.py --config modelnet --no-fp32-mode --no-cpu-baseline > $OUT/bench_modelnet.json 2> $OUT/bench_modelnet.err; echo "modelnet rc=$?"
ab() { name=$1; shift; timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-fp32-mode "$@" > $OUT/ab_$name.json 2> $OUT/ab_$name.err; python -c "
import json
try:
    d=json.load(open('$OUT/ab_$name.json')); print('$name', d['value'], 'pairs/s', d['ms_per_step'],'ms/step')
except Exception as e: print('$name FAILED', e)" | tee -a $OUT/ab_runs.txt; }
ab default
ab stack16_lanes4 --stack 16 --batch 64
ab stack16_lanes3 --stack 16 --batch 48 --lanes 3
ab stack4_lanes4 --stack 4 --batch 32
ab stack8_lanes3 --lanes 3
ab stack8_lanes5 --lanes 5 --batch 40
for f in lomatch_bf16 lomatch_bf16x3 kitti modelnet; do python -c "
import json
try:
    d=json.load(open('$OUT/bench_$f.json')); print('$f', d['value'], d['unit'], d['ms_per_step'], 'ms/step', 'parity ok:', d.get('parity',{}).get('ok'), 'cpu:', d.get('cpu_baseline',{}).get('value'))
except Exception as e: print('$f FAILED', e)"; done
