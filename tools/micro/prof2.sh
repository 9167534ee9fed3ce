cd $GRAFT_REPO_ROOT
TESTS="tests/test_gpu_scale.py tests/test_gpu_parity.py" TMO=600 tools/micro/t.sh
tools/gpu_profile.sh c_step > /dev/null 2>&1
tools/gpu_profile.sh c_sync --sync > /dev/null 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/c_driver_line.json 2> gpurun_out/c_driver_line.err
timeout 400 python bench.py > gpurun_out/c_default_line.json 2> gpurun_out/c_default_line.err
