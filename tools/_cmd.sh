cd $GRAFT_REPO_ROOT
python tools/stage_bench.py 2>&1 | tail -1
python -m pytest tests/test_fullsize_gpu.py tests/test_e2e_gpu.py tests/test_operating_point_gpu.py -q -m gpu -x 2>&1 | tail -3
