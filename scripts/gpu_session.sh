set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3k; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $O/gputests.log 2>&1; echo "pytest rc=$?"; tail -30 $O/gputests.log
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
b() { name=$1; shift; env "$@" timeout 300 python $B --no-cpu-baseline --no-fp32-mode ${EXTRA:-} 2>$O/bench_$name.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'])" || tail -5 $O/bench_$name.err; }
( EXTRA="" b default X=1; EXTRA="" b concat_decoder GEOTR_DECODER_SPLIT=0; EXTRA="" b default_again X=1; EXTRA="" b concat_decoder_again GEOTR_DECODER_SPLIT=0; EXTRA="--lanes 1" b l1 X=1; EXTRA="--lanes 3" b l3 X=1 ) | tee $O/ab_decoder.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_l1 -o bench -- python $B --steps 6 --warmup 2 --lanes 1 --no-cpu-baseline --no-fp32-mode > $O/bench_l1_under_rocprof.json 2>/dev/null
python - <<PY
import csv,glob,collections
f=glob.glob('$O/stats_l1/**/*kernel_trace.csv',recursive=True)[0]
by=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'rg_query' in r['Kernel_Name']: by[int(r['Grid_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for g in sorted(by): print('rg_query grid', g, 'n', len(by[g]), 'avg us %.1f'%(sum(by[g])/len(by[g])))
PY
grep -E "rg_query|gemm_packed_kernel<2, 2, 3>|upsample" $(find $O/stats_l1 -name "*kernel_stats.csv" | head -1) | cut -c1-60,150-230 | head
timeout 500 python $B --config kitti --steps 5 --warmup 1 --pairs 4 --no-fp32-mode --no-cpu-baseline > $O/bench_kitti.json 2> $O/bench_kitti.err; echo "kitti rc=$?"; head -c 200 $O/bench_kitti.json; echo
