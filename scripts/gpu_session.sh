set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
OLD=$GRAFT_REPO_ROOT/.ab_old/r2fix
hz() { name=$1; shift; ( env LABEL=$name "$@" timeout 400 python scripts/hazard_probe.py > $O/hz_$name.json 2> $O/hz_$name.err; echo "$name rc=$?"; head -c 4000 $O/hz_$name.json; echo; tail -3 $O/hz_$name.err ) ; }
hz old_fresh_plain GEOTR_TREE=$OLD GEOTR_P2N_MODE=7 FRESH_PIPELINE=1 REPS=64
hz old_fresh_plain_probe GEOTR_TREE=$OLD GEOTR_P2N_MODE=7 FRESH_PIPELINE=1 GEOTR_P2N_PROBE=1 REPS=64
tail -40 $O/hz_old_fresh_plain_probe.err
hz head_fresh_plain GEOTR_P2N_MODE=7 FRESH_PIPELINE=1 REPS=64
hz head_fresh_plain_probe GEOTR_P2N_MODE=7 FRESH_PIPELINE=1 GEOTR_P2N_PROBE=1 GEOTR_ALLOC_LOG=1 REPS=64
( cd $OLD && GEOTR_P2N_MODE=7 LABEL=old_debug_c timeout 300 python scripts/debug_c.py bisect 8 2>&1 | grep -v "^ " | tail -10 )
