O=$GRAFT_REPO_ROOT/gpurun_out/s10; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
(cd $R/.ab_old/r2 && timeout 300 python bench.py --config kitti --steps 3 --warmup 1 --pairs 4 --no-cpu-baseline --no-fp32-mode > /dev/null 2>&1)
for t in r2 c06 c07 c14 c18 c19 r3; do
  (cd $R/.ab_old/$t && timeout 300 python bench.py --config kitti --steps 5 --warmup 1 --pairs 4 --no-cpu-baseline --no-fp32-mode > $O/kitti_$t.json 2> $O/kitti_$t.err)
  python -c "
import json
try:
    d=json.load(open('$O/kitti_$t.json')); print('$t', d['value'], d['ms_per_step'])
except Exception as e: print('$t FAILED', e)"
done
