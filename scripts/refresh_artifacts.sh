#!/bin/bash
# Round artifacts on the GPU box: full GPU suite, bench line (with CPU baseline + parity), rocprofv3 kernel stats of the same
# command (4 lanes and 1 lane), PMC passes (HBM traffic), A/B runs of the round's switches.
# usage (from the repo root, via gpurun): bash scripts/refresh_artifacts.sh r03   (SUITE=0 skips the GPU test suite)
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/artifacts_$TAG
mkdir -p $OUT
cd $ROOT
if [ "${SUITE:-1}" = "1" ]; then
  timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $OUT/gputests.log 2>&1
  echo "pytest rc=$?" >> $OUT/gputests.log
  tail -4 $OUT/gputests.log
fi
cd /tmp && export TMPDIR=/tmp
timeout 600 python $ROOT/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo "bench rc=$?"; head -c 200 $OUT/bench_n1.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-mode > $OUT/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_l1 -o bench -- python $ROOT/bench.py --steps 6 --warmup 2 --lanes 1 --no-cpu-baseline --no-fp32-mode > $OUT/bench_l1_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --lanes 1 --stack 8 --batch 8 --no-cpu-baseline --no-fp32-mode > /dev/null 2>&1
done
python $ROOT/scripts/pmc_summary.py $OUT $OUT/pmc_hbm_traffic.md $OUT/pmc_hbm_traffic.json "python bench.py --steps 2 --warmup 1 --lanes 1 --stack 8 --batch 8"
python $ROOT/scripts/kernel_trace_summary.py $OUT/kernel_trace.md $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_under_rocprof.json $(find $OUT/stats_l1 -name "*kernel_stats.csv" | head -1) $OUT/bench_l1_under_rocprof.json $OUT/bench_n1.json
# SQ counters of the radius query: two passes of 8 SQ counters each, counters only (no trace domains besides --kernel-trace)
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq1 -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --lanes 1 --stack 8 --batch 8 --no-cpu-baseline --no-fp32-mode > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --lanes 1 --stack 8 --batch 8 --no-cpu-baseline --no-fp32-mode > /dev/null 2>&1
python $ROOT/scripts/sq_counters_summary.py $(find $OUT/pmc_sq1 $OUT/pmc_sq2 -name "*counter_collection.csv") rg_query $OUT/rg_query_counters.md
ab() { name=$1; shift; timeout 300 env "$@" python $ROOT/bench.py --no-cpu-baseline --no-fp32-mode ${EXTRA:-} > $OUT/ab_$name.json 2> $OUT/ab_$name.err; python -c "
import json
try:
    d=json.load(open('$OUT/ab_$name.json')); print('$name', d['value'], 'pairs/s', d['ms_per_step'],'ms/step')
except Exception as e: print('$name FAILED', e)" | tee -a $OUT/ab_runs.txt; }
EXTRA="" ab default X=1
EXTRA="" ab gn_statistics_pass GEOTR_GN_EPILOGUE_STATS=0
EXTRA="" ab concatenated_decoder GEOTR_DECODER_SPLIT=0
EXTRA="" ab synchronous_lanes GEOTR_PIPELINED=0
EXTRA="" ab pyramid_graph GEOTR_PYRAMID_GRAPH=1
EXTRA="--lanes 1" ab one_lane X=1
EXTRA="--lanes 1" ab one_lane_synchronous GEOTR_PIPELINED=0
EXTRA="--lanes 2" ab two_lanes X=1
EXTRA="--lanes 3" ab three_lanes X=1
EXTRA="--lanes 6" ab six_lanes X=1
EXTRA="" ab default_again X=1
# the driver's smoke entry point
cd $ROOT && python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
# the other BASELINE configurations, each with its own parity block and CPU baseline
timeout 400 python $ROOT/bench.py --config lomatch --precision bf16 --no-fp32-mode > $OUT/bench_lomatch_bf16.json 2> $OUT/bench_lomatch_bf16.err; echo "lomatch bf16 rc=$?"
timeout 300 python $ROOT/bench.py --config modelnet --no-fp32-mode > $OUT/bench_modelnet.json 2> $OUT/bench_modelnet.err; echo "modelnet rc=$?"
# KITTI last (1.5 min): skipped when the session is already long (KITTI=0, or more than ${KITTI_AFTER:-420} s in)
if [ "${KITTI:-1}" = "1" ] && [ $SECONDS -lt ${KITTI_AFTER:-420} ]; then timeout 300 python $ROOT/bench.py --config kitti --steps 5 --warmup 1 --pairs 4 --no-fp32-mode > $OUT/bench_kitti.json 2> $OUT/bench_kitti.err; echo "kitti rc=$?"; fi
KARG=""; [ -s $OUT/bench_kitti.json ] && KARG="kitti=$OUT/bench_kitti.json"
python $ROOT/scripts/other_configs_summary.py $OUT/other_configs.md modelnet=$OUT/bench_modelnet.json $KARG lomatch_bf16=$OUT/bench_lomatch_bf16.json
head -12 $OUT/other_configs.md
ls $OUT | head -60
