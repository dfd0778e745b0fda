#!/bin/bash
# Round artifacts on the GPU box: bench line (with CPU baseline), rocprofv3 kernel stats of the same command, PMC passes.
# usage (from the repo root, via gpurun): bash scripts/refresh_artifacts.sh r01
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/artifacts_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --lanes 1 --batch 8 --no-cpu-baseline > /dev/null 2>&1
done
python $ROOT/scripts/pmc_summary.py $OUT $OUT/pmc_hbm_traffic.md $OUT/pmc_hbm_traffic.json "python bench.py --steps 2 --warmup 1 --lanes 1 --batch 8"
ls -la $OUT $OUT/stats | head -30
tail -c 600 $OUT/bench_n1.json
