#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/e
mkdir -p $OUT
cd $ROOT
LABEL=table_splitk GSE=table python scripts/determinism_bisect.py bisect 3 > $OUT/bisect.log 2>&1
LABEL=table_nosplitk GSE=table GEOTR_SPLITK=0 python scripts/determinism_bisect.py bisect 3 >> $OUT/bisect.log 2>&1
LABEL=mfma_splitk GSE=mfma python scripts/determinism_bisect.py bisect 3 >> $OUT/bisect.log 2>&1
LABEL=mfma_nosplitk GSE=mfma GEOTR_SPLITK=0 python scripts/determinism_bisect.py bisect 3 >> $OUT/bisect.log 2>&1
grep -v amdgpu.ids $OUT/bisect.log
GEOTR_DIST_BACKEND=gloo GEOTR_ALLOW_SHARED_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --batch 8 --pairs 2 --points 6000 --lanes 2 --stack 4 > $OUT/gloo_bench.out 2> $OUT/gloo_bench.err
echo "gloo bench rc=$?"; grep -v "amdgpu.ids\|socket.cpp" $OUT/gloo_bench.err | tail -30; cat $OUT/gloo_bench.out | head -c 300
