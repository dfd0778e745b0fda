O=$GRAFT_REPO_ROOT/gpurun_out/s12; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_transformer_gpu.py tests/test_neighbors_gpu.py tests/test_heads_gpu.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-sibling-mode"
run() { name=$1; shift; timeout 300 env "$@" python $R/bench.py $B ${EXTRA:-} > $O/$name.json 2> $O/$name.err; python -c "
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['value'], 'pairs/s', d['ms_per_step'],'ms/step', d['config'].get('untimed_prewarm'))
except Exception as e: print('$name FAILED', e)"; }
EXTRA="" run first_run_with_prewarm X=1
EXTRA="" run nosplit_on X=1
EXTRA="" run nosplit_off GEOTR_SKINNY_NOSPLIT=0
EXTRA="" run nosplit_on2 X=1
EXTRA="" run stagger16 GEOTR_LANE_STAGGER_MS=16
EXTRA="" run stagger8 GEOTR_LANE_STAGGER_MS=8
EXTRA="--config kitti --steps 5 --warmup 1 --pairs 8" run kitti_base X=1
EXTRA="--config kitti --steps 5 --warmup 1 --pairs 8" run kitti_stagger13 GEOTR_LANE_STAGGER_MS=13
EXTRA="--config kitti --steps 5 --warmup 1 --pairs 8 --lanes 3" run kitti_3lanes X=1
EXTRA="--config kitti --steps 5 --warmup 1 --pairs 8 --lanes 1" run kitti_1lane X=1
timeout 400 python $R/bench.py --config lomatch --precision bf16 --no-sibling-mode > $O/bench_lomatch_bf16.json 2> $O/bench_lomatch_bf16.err; python -c "
import json; d=json.load(open('$O/bench_lomatch_bf16.json')); print('lomatch bf16', d['value'], d['parity']['ok'])"
