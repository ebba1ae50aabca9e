export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c11; mkdir -p $O
cd $R
VD3D_LIB_PATH=$R/visiondepth3d_amd/ab/libvd3d_hip_stamps.so timeout 300 python tools/probe_phases.py > $O/phases.log 2>&1
cat $O/phases.log
