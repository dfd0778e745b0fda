#!/bin/bash
# Round-end check of the final tree: full GPU suite, smoke(), the driver's bench command, the default bench line (with CPU baseline + parity).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 > $OUT/gputests.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/gputests.log; tail -3 $OUT/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "driver-cmd bench rc=$?"; grep "quarter\|timed region" $OUT/bench_driver_cmd.err
python -c "
import json; d=json.load(open('$OUT/bench_driver_cmd.json')); print('driver cmd', d['value'], d['ms_per_step'], 'parity', d['parity']['ok'], 'cpu', d['cpu_baseline']['value'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'])"
timeout 600 python $ROOT/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "default bench rc=$?"; grep "quarter\|timed region" $OUT/bench_n1.err
python -c "
import json; d=json.load(open('$OUT/bench_n1.json')); print('default', d['value'], d['ms_per_step'], 'parity', d['parity']['ok'])"
