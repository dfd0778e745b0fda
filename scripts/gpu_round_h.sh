#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
run() { name=$1; shift; bad=0; for i in 1 2 3 4 5 6 7 8 9 10 11 12; do n=$(env "$@" GEOTR_KPCONV_FUSED=0 GEOTR_POISON_WS=1 LABEL=m GSE=table python scripts/determinism_bisect.py bisect 1 2>&1 | grep -c "DIFFERENCES"); bad=$((bad + n)); done; echo "$name: $bad of 12 runs nondeterministic"; }
run "no runtime blits, plain loads (mode 0)" GEOTR_P2N_MODE=0
