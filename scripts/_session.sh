O=$GRAFT_REPO_ROOT/gpurun_out/s2; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_backbone_gpu.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1; tail -5 $O/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --precision fp32 --no-cpu-baseline --steps 10 --warmup 2 > $O/fp32_4l.json 2> $O/fp32_4l.err
timeout 300 python $R/bench.py --precision fp32 --no-cpu-baseline --steps 6 --warmup 2 --lanes 1 > $O/fp32_1l.json 2> $O/fp32_1l.err
timeout 300 python $R/bench.py --precision bf16x3 --no-cpu-baseline --no-fp32-mode --steps 10 --warmup 2 > $O/bf16x3_4l.json 2> $O/bf16x3_4l.err
timeout 300 python $R/bench.py --precision fp32 --steps 10 --warmup 2 > $O/fp32_4l_parity.json 2> $O/fp32_4l_parity.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fp32_l1 -o bench -- python $R/bench.py --precision fp32 --steps 4 --warmup 1 --lanes 1 --no-cpu-baseline > $O/rocprof_fp32_l1.json 2>/dev/null
find $O -name "*kernel_trace.csv" -size +20M -delete
for f in $O/*.json; do echo $f; head -c 160 $f; echo; done
