mkdir -p gpurun_out/r3h
timeout -k 10 200 python -u tools/launch_overhead.py > gpurun_out/r3h/launch_overhead.log 2>&1 || exit 1
RT_AMD_LIB=$PWD/raytracing_amd/librt_amd_tail.so timeout -k 10 200 python -u tools/tail.py 120 961 > gpurun_out/r3h/tail.log 2>&1 || exit 1
cat gpurun_out/r3h/launch_overhead.log gpurun_out/r3h/tail.log
