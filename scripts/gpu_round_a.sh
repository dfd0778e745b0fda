#!/bin/bash
# GPU call A of round 2: full GPU suite, default bench line, kernel stats of the bench command (single lane + default).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/a
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $OUT/gputests.log 2>&1
echo "pytest rc=$?" >> $OUT/gputests.log
tail -5 $OUT/gputests.log
cd /tmp && export TMPDIR=/tmp
timeout 900 python $ROOT/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo "bench rc=$?"; tail -c 1500 $OUT/bench_n1.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-mode > $OUT/bench_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_l1 -o bench -- python $ROOT/bench.py --steps 6 --warmup 2 --lanes 1 --no-cpu-baseline --no-fp32-mode > $OUT/bench_l1_under_rocprof.json 2>/dev/null
ls $OUT $OUT/stats | head -30
