cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_golden.py tests/test_gpu_stream.py tests/test_gpu_blocksize.py -x -q 2>&1 | tail -2
AB_STEPS=4 tools/ab/run_variants.sh "k_hme_level" prevhme base prevhme base
