#!/bin/bash
# Round-2 call Q: pyramid on a high-priority queue per lane (GEOTR_PYRAMID_PRIORITY=1) vs the default
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 env ${ENVV:-X=1} python $ROOT/bench.py --no-cpu-baseline --no-fp32-mode --profile-events 0 "$@" > $OUT/$name.json 2> $OUT/$name.err; python -c "
import json
try:
    d=json.load(open('$OUT/$name.json')); print('$name', d['value'], 'pairs/s', d['ms_per_step'],'ms/step')
except Exception as e: print('$name FAILED', e)" | tee -a $OUT/runs.txt; grep "host ms" $OUT/$name.err | tee -a $OUT/runs.txt; }
ENVV="GEOTR_HOST_TIMING=1" run default
ENVV="GEOTR_HOST_TIMING=1 GEOTR_PYRAMID_PRIORITY=1" run priority
ENVV="GEOTR_HOST_TIMING=1" run default_again
ENVV="GEOTR_HOST_TIMING=1 GEOTR_PYRAMID_PRIORITY=1" run priority_again
ENVV="GEOTR_HOST_TIMING=1 GEOTR_PYRAMID_PRIORITY=1" run priority_lanes3 --lanes 3
ENVV="GEOTR_HOST_TIMING=1 GEOTR_PYRAMID_PRIORITY=1" run priority_lanes5 --lanes 5
