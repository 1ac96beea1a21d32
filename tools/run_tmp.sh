set -e
B="python -u bench.py --workload c4 --no-extra --no-cpu --steps 300 --warmup 30"
pick() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['kernels_ms'])"; }
timeout -k 10 120 $B | pick base
for v in e3 e4 e4w2 e3w2; do
VSLAM_HIP_LIB=visual_slam_amd/lib/variants/libvslam_hip_$v.so timeout -k 10 120 $B | pick $v
done
