set -e
timeout -k 10 600 python -u -m pytest tests/test_gpu_ba.py tests/test_gpu_ba_fullsize.py tests/test_gpu_shard_solve.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
timeout -k 10 300 python3 tools/time_lm.py c4 c5shard c5
