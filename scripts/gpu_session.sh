set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_neighbors_gpu.py tests/test_datasets_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 600 2>&1 | tail -25
timeout 900 python -m pytest tests/test_bench_config_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 900 -k "bench_workload or kitti" 2>&1 | tail -25
cd /tmp && export TMPDIR=/tmp
b() { name=$1; shift; env "$@" timeout 300 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-mode ${EXTRA:-} 2>$GRAFT_REPO_ROOT/$O/bench_$name.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'])" || tail -5 $GRAFT_REPO_ROOT/$O/bench_$name.err; }
( EXTRA="" b pipelined X=1; EXTRA="" b sync_lanes GEOTR_PIPELINED=0; EXTRA="--lanes 1" b pipelined_l1 X=1; EXTRA="--lanes 1" b sync_l1 GEOTR_PIPELINED=0
  EXTRA="--lanes 2" b pipelined_l2 X=1; EXTRA="--lanes 3" b pipelined_l3 X=1; EXTRA="--lanes 6" b pipelined_l6 X=1; EXTRA="" b pipelined_again X=1 ) | tee $GRAFT_REPO_ROOT/$O/ab_pipelined.txt
cd $GRAFT_REPO_ROOT
for t in r2orig r2A r2B; do ( cd .ab_old/$t && GEOTR_P2N_MODE=7 TRUTH=1 DISSECT=1 LABEL=$t timeout 400 python scripts/debug_c.py bisect 12 > $GRAFT_REPO_ROOT/$O/hz_$t.txt 2>&1; echo "$t: $(grep -c DIFFERENCES $GRAFT_REPO_ROOT/$O/hz_$t.txt) of 12 runs differ" ); done
grep -A12 "dissect" $O/hz_r2orig.txt | head -60
