set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3m; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_heads_gpu.py tests/test_backbone_gpu.py tests/test_configs_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 900 2>&1 | tail -15
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
b() { name=$1; shift; env "$@" timeout 300 python $B --no-cpu-baseline --no-fp32-mode ${EXTRA:-} 2>$O/bench_$name.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'])" || tail -5 $O/bench_$name.err; }
( EXTRA="" b default X=1; EXTRA="" b tail_apply_pass GEOTR_TAIL_FUSED=0; EXTRA="" b default_again X=1; EXTRA="" b tail_apply_pass_again GEOTR_TAIL_FUSED=0; EXTRA="--lanes 1" b l1 X=1; EXTRA="--lanes 1" b l1_tail_apply_pass GEOTR_TAIL_FUSED=0 ) | tee $O/ab_tail.txt
cd $GRAFT_REPO_ROOT && timeout 900 python -m pytest tests/test_bench_config_gpu.py tests/test_reference_forward_gpu.py tests/test_pipeline_gpu.py tests/test_bf16_gpu.py -m gpu -q -p no:cacheprovider --timeout 900 2>&1 | tail -8
