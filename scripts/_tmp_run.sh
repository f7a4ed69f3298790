FG_LIB_VARIANT=freegaussian_amd/libfgraster_noskew.so bash scripts/gpu_r04_binprof.sh r04_binprof6 "0.8 0.2" 2>&1 | grep -v "raster_\|preprocess\|amdgpu.ids\|^{"
