#!/bin/bash
# Round-2 call Z: bench with the event sample spread over the region (stride 8) vs every launch of its start (stride 1), first runs on a
# fresh box; host time of the lane threads per stack (GEOTR_HOST_TIMING=1) at 4 lanes and 1 lane.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/z
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 env ${ENVV:-X=1} python $ROOT/bench.py --no-cpu-baseline --no-fp32-mode "$@" > $OUT/$name.json 2> $OUT/$name.err; python -c "
import json
try:
    d=json.load(open('$OUT/$name.json')); print('$name', d['value'], 'pairs/s', d['ms_per_step'],'ms/step', '| roofline', d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['launches'])
except Exception as e: print('$name FAILED', e)" | tee -a $OUT/runs.txt; grep "quarter\|host ms" $OUT/$name.err | tee -a $OUT/runs.txt; }
run first --gpus 1 --steps 20 --warmup 5
run second --gpus 1 --steps 20 --warmup 5
run third --gpus 1 --steps 20 --warmup 5
run stride1 --gpus 1 --steps 20 --warmup 5 --profile-stride 1
run no_events --gpus 1 --steps 20 --warmup 5 --profile-events 0
ENVV="GEOTR_HOST_TIMING=1" run host_4lanes --gpus 1 --steps 20 --warmup 5 --profile-events 0
ENVV="GEOTR_HOST_TIMING=1" run host_1lane --gpus 1 --steps 20 --warmup 5 --profile-events 0 --lanes 1
ENVV="GEOTR_HOST_TIMING=1" run host_2lanes --gpus 1 --steps 20 --warmup 5 --profile-events 0 --lanes 2
