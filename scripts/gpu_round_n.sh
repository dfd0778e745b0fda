#!/bin/bash
# Round-2 call N: process bound to the GPU's NUMA node (default) vs unbound, alternating, fresh box; host time per stack
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/n
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 env GEOTR_HOST_TIMING=1 python $ROOT/bench.py --no-cpu-baseline --no-fp32-mode --gpus 1 --steps 20 --warmup 5 "$@" > $OUT/$name.json 2> $OUT/$name.err; python -c "
import json
try:
    d=json.load(open('$OUT/$name.json')); print('$name', d['value'], 'pairs/s', d['ms_per_step'],'ms/step')
except Exception as e: print('$name FAILED', e)" | tee -a $OUT/runs.txt; grep "host ms\|NUMA\|quarter" $OUT/$name.err | cut -c1-260 | tee -a $OUT/runs.txt; }
run bound_1
run unbound_1 --no-numa-bind
run bound_2
run unbound_2 --no-numa-bind
run bound_3
run unbound_3 --no-numa-bind
run bound_4
