cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_recon.py tests/test_gpu_golden.py tests/test_gpu_stream.py tests/test_gpu_chain.py tests/test_gpu_halfpel.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
AB_STEPS=4 tools/ab/run_variants.sh "k_fwd_mc_fast k_inv_p_tile k_inv_patch_c k_inv_b4t" sbthead base sbthead base
