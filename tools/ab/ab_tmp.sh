cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_golden.py tests/test_gpu_stream.py tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_abr_device.py -x -q 2>&1 | tail -2
AB_STEPS=4 tools/ab/run_variants.sh "k_hz_" hzhead base hzhead base
