# A/B builds of libvd3d_hip.so with one compile-time difference (same ABI; select with VD3D_LIB_PATH).  usage: bash tools/build_ab.sh NAME -DFLAG...
# Development knobs (vd3d_debug_tune 0 - 2 / 5 - 9, the VD3D_TUNE variable, the parked persistent kernels) exist only in builds with -DVD3D_DEV_KNOBS
# (round 6): `bash tools/build_ab.sh dev -DVD3D_DEV_KNOBS` -> VD3D_LIB_PATH=visiondepth3d_amd/ab/libvd3d_hip_dev.so; tools/r05/call*.sh, tools/probe_step.py and
# tools/gpu_ab.cpp need that library.
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/gpurun_out/ab_build_$NAME; mkdir -p $B $R/visiondepth3d_amd/ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -Wno-unused-function"
cd $R/visiondepth3d_amd/csrc
for f in *.hip; do /opt/rocm/bin/hipcc $FLAGS "$@" -c $f -o $B/${f%.hip}.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/visiondepth3d_amd/ab/libvd3d_hip_$NAME.so $B/*.o
rm -rf $B
ls -la $R/visiondepth3d_amd/ab/libvd3d_hip_$NAME.so
