# round-4 GPU call 11: full GPU suite on the W1 chunk skipping, phase stamps of W1 / E1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4c11; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
VD3D_LIB_PATH=$R/visiondepth3d_amd/ab/libvd3d_hip_stamps.so timeout 200 python tools/probe_phases.py > $O/phases.log 2>&1; tail -22 $O/phases.log
