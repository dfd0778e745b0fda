O=$GRAFT_REPO_ROOT/gpurun_out/s5; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q -p no:cacheprovider > $O/tests_gemm.log 2>&1; tail -2 $O/tests_gemm.log
{
for shp in "43826 512 512" "179984 256 256" "640000 128 64"; do
  scripts/abi_bench.bin gemm $shp fp32 20
  scripts/abi_bench.bin gemm $shp fp32 20 nostore
  GEOTR_GEMM_LDS_PAD_KB=24 scripts/abi_bench.bin gemm $shp fp32 20
  scripts/abi_bench.bin gemm $shp bf16x3 20
  scripts/abi_bench.bin gemm $shp bf16x3 20 nostore
done
} 2>&1 | grep -v amdgpu.ids > $O/gemm_experiments.txt
cat $O/gemm_experiments.txt | cut -c1-210
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "mfma|SQ_BUSY|SQ_WAVE_CYCLES|SQ_WAIT|GRBM_GUI|LDS_BANK|SQ_INSTS_VALU |SQ_ACTIVE" | head -60 > $O/counters_avail.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_gemm -o pmc -- $R/scripts/abi_bench.bin gemm 43826 512 512 fp32 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_gemm2 -o pmc -- $R/scripts/abi_bench.bin gemm 43826 512 512 fp32 5 > /dev/null 2>&1
ls $O/pmc_gemm $O/pmc_gemm2 2>/dev/null | head
# KITTI: rounds 2 / 3 / this tree on the same box
for t in r2 r3; do
  (cd $R/.ab_old/$t && timeout 300 python bench.py --config kitti --steps 5 --warmup 1 --pairs 4 --no-cpu-baseline --no-fp32-mode > $O/kitti_$t.json 2> $O/kitti_$t.err)
done
timeout 300 python $R/bench.py --config kitti --precision bf16x3 --steps 5 --warmup 1 --pairs 4 --no-cpu-baseline --no-sibling-mode > $O/kitti_r4_bf16x3.json 2> $O/kitti_r4_bf16x3.err
timeout 300 python $R/bench.py --config kitti --steps 5 --warmup 1 --pairs 4 --no-cpu-baseline --no-sibling-mode > $O/kitti_r4_fp32.json 2> $O/kitti_r4_fp32.err
(cd $R/.ab_old/r2 && timeout 300 python bench.py --config kitti --steps 5 --warmup 1 --pairs 4 --no-cpu-baseline --no-fp32-mode > $O/kitti_r2_again.json 2> $O/kitti_r2_again.err)
timeout 400 python $R/bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
for f in $O/kitti_*.json $O/bench_default.json; do python -c "
import json,sys
try:
    d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'])
except Exception as e: print('$f FAILED', e)"; done
