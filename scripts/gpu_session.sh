set -u
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_concurrency_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | cut -c1-400
bash scripts/refresh_artifacts.sh r03
