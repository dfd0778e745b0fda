set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
# (1) the round-2 tree at the commit that shipped the agent-scope loads, with those loads made plain again: does TODAY's box reproduce the failure?
( cd .ab_old/r2fix && LABEL=old_plain timeout 300 python scripts/debug_c.py bisect 8 > $GRAFT_REPO_ROOT/$O/old_tree_plain.txt 2>&1; echo "old rc=$?"; grep -c DIFFERENCES $GRAFT_REPO_ROOT/$O/old_tree_plain.txt; tail -12 $GRAFT_REPO_ROOT/$O/old_tree_plain.txt )
hz() { name=$1; shift; ( env LABEL=$name "$@" timeout 300 python scripts/hazard_probe.py > $O/hz_$name.json 2> $O/hz_$name.err; echo "$name rc=$?"; head -c 1800 $O/hz_$name.json; echo ) ; }
hz plain GEOTR_P2N_MODE=1 REPS=40
hz probe GEOTR_P2N_PROBE=1 GEOTR_ALLOC_LOG=1 REPS=40
tail -30 $O/hz_probe.err
timeout 600 python -m pytest tests/test_reference_forward_gpu.py tests/test_bench_config_gpu.py -m gpu -q -s -p no:cacheprovider --timeout 900 > $O/tests_verbose.log 2>&1; echo "pytest rc=$?"; tail -5 $O/tests_verbose.log
