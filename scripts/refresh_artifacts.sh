#!/bin/bash
# Round artifacts on the GPU box: full GPU suite, bench line (fp32 headline + split-bf16 sibling, CPU baseline, 4-pair parity), rocprofv3
# kernel stats of the same command (4 lanes and 1 lane), PMC passes (HBM traffic, fp32), per-instantiation GEMM traffic table,
# the other BASELINE configurations.
# usage (from the repo root, via gpurun): bash scripts/refresh_artifacts.sh r06   (SUITE=0 skips the GPU test suite, OTHERS=0 the other configs,
# PROF=0 the rocprofv3 passes)
set -u
TAG=${1:-r06}
export ROUND=${TAG#r}
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
timeout 600 python $ROOT/bench.py --detail $OUT/bench_n1_detail.json > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo "bench rc=$?"; head -c 200 $OUT/bench_n1.json; echo
B="--no-cpu-baseline --no-sibling-mode"
if [ "${PROF:-1}" = "1" ]; then  # (PROF=0: bench lines only -- host-side changes leave the kernel traces and the PMC passes of the previous call valid)
for try in 1 2 3; do  # (rocprofv3 itself crashed once in round 6: "Segmentation fault" after the run, no stats written)
  rm -rf $OUT/stats
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 $B > $OUT/bench_under_rocprof.json 2>/dev/null
  [ -n "$(find $OUT/stats -name '*kernel_stats.csv' 2>/dev/null | head -1)" ] && break
  echo "rocprofv3 four-lane trace failed (try $try)"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_l1 -o bench -- python $ROOT/bench.py --steps 6 --warmup 2 --lanes 1 $B > $OUT/bench_l1_under_rocprof.json 2>/dev/null
# HBM traffic: counters only (no trace domains besides --kernel-trace), one counter per pass; a launch covers 16 stacked pairs, as in the bench
P="--steps 2 --warmup 1 --lanes 1 --stack 16 --batch 16 $B"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $ROOT/bench.py $P > /dev/null 2>&1
done
python $ROOT/scripts/pmc_summary.py $OUT $OUT/pmc_hbm_traffic_fp32.md $OUT/pmc_hbm_traffic_fp32.json "python bench.py $P (fp32)"
rm -f $OUT/shapes_pmc.jsonl
python $ROOT/bench.py --steps 2 --warmup 1 --lanes 1 --stack 16 --batch 16 --profile-stride 1 --profile-events 8192 --dump-shapes $OUT/shapes_pmc.jsonl $B > /dev/null 2>&1
python $ROOT/scripts/gemm_traffic_table.py $OUT/pmc_hbm_traffic_fp32.json $OUT/shapes_pmc.jsonl fp32 $OUT/gemm_traffic_fp32.md
python $ROOT/scripts/kernel_trace_summary.py $OUT/kernel_trace.md $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_under_rocprof.json \
  $(find $OUT/stats_l1 -name "*kernel_stats.csv" | head -1) $OUT/bench_l1_under_rocprof.json 2>&1 | tail -3
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
fi
ab() { name=$1; shift; timeout 300 env "$@" python $ROOT/bench.py $B ${EXTRA:-} > $OUT/ab_$name.json 2> $OUT/ab_$name.err; python -c "
import json
try:
    d=json.load(open('$OUT/ab_$name.json')); print('$name', d['value'], 'pairs/s', d['ms_per_step'],'ms/step')
except Exception as e: print('$name FAILED', e)" | tee -a $OUT/ab_runs.txt; }
if [ "${AB:-1}" = "1" ]; then
  EXTRA="" ab default X=1
  EXTRA="" ab sinkhorn_block GEOTR_SINKHORN_FORM=block
  EXTRA="" ab default_again X=1
  EXTRA="--precision bf16x3" ab bf16x3 X=1
  EXTRA="--precision bf16" ab bf16 X=1
  EXTRA="--lanes 5" ab lanes5 X=1
  EXTRA="" ab default_third X=1
fi
# the driver's smoke entry point
cd $ROOT && python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
cd /tmp
if [ "${OTHERS:-1}" = "1" ]; then
  timeout 400 python $ROOT/bench.py --config lomatch --precision bf16 --detail $OUT/bench_lomatch_bf16_detail.json > $OUT/bench_lomatch_bf16.json 2> $OUT/bench_lomatch_bf16.err; echo "lomatch bf16 rc=$?"
  timeout 300 python $ROOT/bench.py --config modelnet --detail $OUT/bench_modelnet_detail.json > $OUT/bench_modelnet.json 2> $OUT/bench_modelnet.err; echo "modelnet rc=$?"
  timeout 400 python $ROOT/bench.py --config kitti --steps 20 --warmup 2 --pairs 8 --detail $OUT/bench_kitti_detail.json > $OUT/bench_kitti.json 2> $OUT/bench_kitti.err; echo "kitti rc=$?"
  timeout 200 env GEOTR_SINKHORN_FORM=block GEOTR_RG_DENSE_ORDER=0 python $ROOT/bench.py --config kitti --steps 5 --warmup 1 --pairs 8 $B > $OUT/ab_kitti_r5_forms.json 2> $OUT/ab_kitti_r5_forms.err
  timeout 200 python $ROOT/bench.py --config kitti --steps 5 --warmup 1 --pairs 8 $B > $OUT/ab_kitti_default.json 2> $OUT/ab_kitti_default.err
  python -c "
import json
for n in ('default', 'r5_forms'):
    try: d = json.load(open('$OUT/ab_kitti_%s.json' % n)); print('kitti', n, d['value'], 'pairs/s')
    except Exception as e: print('kitti', n, 'FAILED', e)" | tee -a $OUT/ab_runs.txt
  KARG=""; [ -s $OUT/bench_kitti.json ] && KARG="kitti=$OUT/bench_kitti.json"
  python $ROOT/scripts/other_configs_summary.py $OUT/other_configs.md modelnet=$OUT/bench_modelnet.json $KARG lomatch_bf16=$OUT/bench_lomatch_bf16.json
  head -12 $OUT/other_configs.md
fi
ls $OUT | head -60
