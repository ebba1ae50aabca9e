# round-4 GPU call 13: the rocprofv3 evidence of the final kernels (kernel traces of the sequential 4K DIBR run, the headline and the configs[4] chain; three PMC passes)
export VD3D_COMMIT=fad9ee2
bash tools/make_profiles.sh r04 > gpurun_out/make_profiles_r04.log 2>&1
tail -12 gpurun_out/make_profiles_r04.log
