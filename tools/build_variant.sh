#!/bin/bash
# Builds a tuning variant of the library: tools/build_variant.sh <name> <extra hipcc flags...>
# -> visual_slam_amd/lib/variants/libvslam_hip_<name>.so  (select it with VSLAM_HIP_LIB=<path>)
set -e
name=$1; shift
d=visual_slam_amd/lib/variants
mkdir -p $d /tmp/vsvar_$name
for f in ba match track comm capi; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Iinclude -Ivisual_slam_amd/csrc -Wno-unused-function "$@" -c visual_slam_amd/csrc/$f.hip -o /tmp/vsvar_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libvslam_hip_$name.so /tmp/vsvar_$name/*.o -ldl
echo built $d/libvslam_hip_$name.so
