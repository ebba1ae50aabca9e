# development: timing ablations of k_gemm_bf16x3 (A/B build with -DVD_GEMM_ABLATE; results of the ablated kernels are wrong by construction)
R=$GRAFT_REPO_ROOT; cd $R
bash tools/build_ab.sh gemmab -DVD_GEMM_ABLATE > /dev/null 2>&1
for d in ${ABL:-0 1 2 4 8 5 3 9 11 16 27}; do
  echo "== VD3D_GEMM_DBG=$d (1 no DMA, 2 no barrier, 4 no MFMA, 8 no split, 16 no stores)"
  VD3D_GEMM_DBG=$d VD3D_LIB_PATH=$R/visiondepth3d_amd/ab/libvd3d_hip_gemmab.so python tools/probe_gemm_x3.py 39088 2>&1 | grep "^M 39088 K 768 N 2304\|^M 39088 K 3072" | cut -c1-110
done
