set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3j; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $O/gputests.log 2>&1; echo "pytest rc=$?"; tail -30 $O/gputests.log
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
b() { name=$1; shift; env "$@" timeout 300 python $B --no-cpu-baseline --no-fp32-mode ${EXTRA:-} 2>$O/bench_$name.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'])" || tail -5 $O/bench_$name.err; }
( EXTRA="" b default X=1; EXTRA="" b pyramid_graph GEOTR_PYRAMID_GRAPH=1; EXTRA="" b default_again X=1; EXTRA="" b pyramid_graph_again GEOTR_PYRAMID_GRAPH=1; EXTRA="--lanes 1" b l1_graph GEOTR_PYRAMID_GRAPH=1; EXTRA="--lanes 1" b l1 X=1 ) | tee $O/ab_graph.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_l1 -o bench -- python $B --steps 6 --warmup 2 --lanes 1 --no-cpu-baseline --no-fp32-mode > $O/bench_l1_under_rocprof.json 2>/dev/null
grep -E "rg_query|gemm_packed_kernel<2, 2, 3>|gs_|scan_" $(find $O/stats_l1 -name "*kernel_stats.csv" | head -1) | cut -c1-160 | head -12
timeout 500 python $B --config kitti --steps 5 --warmup 1 --pairs 4 --no-fp32-mode > $O/bench_kitti.json 2> $O/bench_kitti.err; echo "kitti rc=$?"; head -c 300 $O/bench_kitti.json; echo
timeout 400 python $B --config lomatch --precision bf16 --no-fp32-mode > $O/bench_lomatch_bf16.json 2> $O/bench_lomatch_bf16.err; echo "lomatch bf16 rc=$?"; head -c 300 $O/bench_lomatch_bf16.json; echo
