set -e
B="python -u bench.py --no-extra --no-cpu --steps 200 --warmup 20"
pick() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['kernels_ms'])"; }
timeout -k 10 600 python -u -m pytest tests/test_gpu_ba_fullsize.py tests/test_gpu_ba.py tests/test_gpu_ba_c5.py tests/test_gpu_shard_solve.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
timeout -k 10 120 $B --workload c4 | pick c4_lds
VS_BA_NO_LDS_CAMS=1 timeout -k 10 120 $B --workload c4 | pick c4_gather
timeout -k 10 120 $B --workload c2 | pick c2
timeout -k 10 120 $B --workload c5shard | pick c5shard
