#!/bin/bash
# GPU call P of round 2: full GPU suite with the fused KPConv kernel, default bench line, A/B runs, kernel stats.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/p
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 > $OUT/gputests.log 2>&1
echo "pytest rc=$?" >> $OUT/gputests.log
tail -12 $OUT/gputests.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python $ROOT/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo "bench rc=$?"; tail -3 $OUT/bench_n1.err; head -c 300 $OUT/bench_n1.json; echo
ab() { name=$1; shift; timeout 300 env "$@" python $ROOT/bench.py --no-cpu-baseline --no-fp32-mode ${EXTRA:-} > $OUT/ab_$name.json 2> $OUT/ab_$name.err; python -c "
import json,sys
try:
    d=json.load(open('$OUT/ab_$name.json')); print('$name', d['value'], 'pairs/s', d['ms_per_step'],'ms/step')
except Exception as e: print('$name FAILED', e)"; }
EXTRA="" ab default X=1
EXTRA="" ab unfused_kpconv GEOTR_KPCONV_FUSED=0
EXTRA="" ab unfused_nosplitk GEOTR_KPCONV_FUSED=0 GEOTR_SPLITK=0
EXTRA="--lanes 1" ab lanes1 X=1
EXTRA="--lanes 1" ab lanes1_unfused GEOTR_KPCONV_FUSED=0
EXTRA="--lanes 1" ab lanes1_unfused_nosplitk GEOTR_KPCONV_FUSED=0 GEOTR_SPLITK=0
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-mode > $OUT/bench_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_l1 -o bench -- python $ROOT/bench.py --steps 6 --warmup 2 --lanes 1 --no-cpu-baseline --no-fp32-mode > $OUT/bench_l1_under_rocprof.json 2>/dev/null
cd $ROOT
bad=0; for i in 1 2 3 4 5 6 7 8; do n=$(GEOTR_POISON_WS=1 LABEL=m GSE=table python scripts/determinism_bisect.py bisect 1 2>&1 | grep -c "DIFFERENCES"); bad=$((bad + n)); done; echo "concurrency determinism (4 lanes x 4 rotated stacks, poisoned workspaces): $bad of 8 runs nondeterministic" | tee $OUT/determinism.txt
ls $OUT | head -40
