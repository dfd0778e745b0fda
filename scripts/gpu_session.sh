set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
rocm-smi --showuniqueid --showfwinfo 2>/dev/null | grep -i "unique\|MEC\|SMC\|CP " | head -8 > $O/box.txt; cat $O/box.txt
# same box: the round-2 tree with its ORIGINAL matching.hip (plain loads) vs the same tree with this round's matching.hip (plain loads) vs HEAD
( cd .ab_old/r2orig && LABEL=orig timeout 300 python scripts/debug_c.py bisect 12 2>&1 | grep -v "^ " | tail -13 ) | tee $O/old_orig_debug_c.txt
( cd .ab_old/r2fix && GEOTR_P2N_MODE=7 LABEL=newmatching timeout 300 python scripts/debug_c.py bisect 12 2>&1 | grep -v "^ " | tail -13 ) | tee $O/old_newmatching_debug_c.txt
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider --timeout 600 2>&1 | tail -15
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 --deselect tests/test_gemm_gpu.py > $O/gputests.log 2>&1; echo "pytest rc=$?"; tail -25 $O/gputests.log
cd /tmp && export TMPDIR=/tmp
for v in 1 0 1 0; do GEOTR_GN_EPILOGUE_STATS=$v timeout 300 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gn_epilogue_stats=$v', d['value'], d['ms_per_step'])"; done | tee $GRAFT_REPO_ROOT/$O/ab_gn_stats.txt
