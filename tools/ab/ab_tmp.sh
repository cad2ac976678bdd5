cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_recon.py tests/test_gpu_golden.py tests/test_gpu_stream.py tests/test_gpu_chain.py tests/test_gpu_ops.py tests/test_gpu_decode_escape.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
AB_STEPS=4 tools/ab/run_variants.sh "k_inv_haar_tile k_fwd_b4t k_inv_tile54" sbthead base sbthead base
