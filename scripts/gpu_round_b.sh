#!/bin/bash
# GPU call B of round 2: full GPU suite on the new kernels, default bench line, A/B runs of each change, kernel stats.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/b
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 > $OUT/gputests.log 2>&1
echo "pytest rc=$?" >> $OUT/gputests.log
tail -8 $OUT/gputests.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python $ROOT/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo "bench rc=$?"; tail -3 $OUT/bench_n1.err; head -c 400 $OUT/bench_n1.json; echo
ab() { name=$1; shift; timeout 300 env "$@" python $ROOT/bench.py --no-cpu-baseline --no-fp32-mode ${EXTRA:-} > $OUT/ab_$name.json 2> $OUT/ab_$name.err; python -c "
import json,sys
try:
    d=json.load(open('$OUT/ab_$name.json')); print('$name', d['value'], 'pairs/s', d['ms_per_step'],'ms/step')
except Exception as e: print('$name FAILED', e)"; }
EXTRA="" ab default X=1
EXTRA="--gse mfma" ab gse_mfma X=1
EXTRA="" ab no_splitk GEOTR_SPLITK=0
EXTRA="" ab no_chunk GEOTR_KPCONV_CHUNK_MB=0
EXTRA="" ab chunk16 GEOTR_KPCONV_CHUNK_MB=16
EXTRA="--lanes 3" ab lanes3 X=1
EXTRA="--lanes 6" ab lanes6 X=1
EXTRA="--lanes 2 --stack 16" ab lanes2_stack16 X=1
EXTRA="--lanes 1" ab lanes1 X=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-mode > $OUT/bench_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_l1 -o bench -- python $ROOT/bench.py --steps 6 --warmup 2 --lanes 1 --no-cpu-baseline --no-fp32-mode > $OUT/bench_l1_under_rocprof.json 2>/dev/null
ls $OUT | head -40
