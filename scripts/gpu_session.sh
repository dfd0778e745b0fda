set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3i; mkdir -p $O
( cd .ab_old/r2C && GEOTR_P2N_MODE=7 LABEL=r2C_noslp timeout 300 python scripts/debug_c.py bisect 12 2>&1 | grep -v "^ " > $O/hz_r2C.txt; echo "r2C (no SLP vectorisation): $(grep -c DIFFERENCES $O/hz_r2C.txt) of 12 runs differ" )
( cd .ab_old/r2orig && GEOTR_P2N_MODE=7 LABEL=r2orig timeout 300 python scripts/debug_c.py bisect 12 2>&1 | grep -v "^ " > $O/hz_r2orig.txt; echo "r2orig: $(grep -c DIFFERENCES $O/hz_r2orig.txt) of 12 runs differ" )
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $B --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-mode > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_l1 -o bench -- python $B --steps 6 --warmup 2 --lanes 1 --no-cpu-baseline --no-fp32-mode > $O/bench_l1_under_rocprof.json 2>/dev/null
find $O/stats $O/stats_l1 -name "*kernel_stats.csv" | head; 
python $GRAFT_REPO_ROOT/scripts/kernel_trace_summary.py $O/kernel_trace.md $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_under_rocprof.json $(find $O/stats_l1 -name "*kernel_stats.csv" | head -1) $O/bench_l1_under_rocprof.json && head -45 $O/kernel_trace.md
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python $B --steps 2 --warmup 1 --lanes 1 --stack 8 --batch 8 --no-cpu-baseline --no-fp32-mode > /dev/null 2>&1
done
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $O $O/pmc_hbm_traffic.md $O/pmc_hbm_traffic.json "python bench.py --steps 2 --warmup 1 --lanes 1 --stack 8 --batch 8" 2>&1 | tail -3
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq1 -o pmc -- python $B --steps 2 --warmup 1 --lanes 1 --stack 8 --batch 8 --no-cpu-baseline --no-fp32-mode > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc_sq2 -o pmc -- python $B --steps 2 --warmup 1 --lanes 1 --stack 8 --batch 8 --no-cpu-baseline --no-fp32-mode > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/sq_counters_summary.py $(find $O/pmc_sq1 $O/pmc_sq2 -name "*counter_collection.csv") rg_query $O/sq_rg_query.md; cat $O/sq_rg_query.md
ls $O; du -sh $O
