cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_golden.py tests/test_gpu_stream.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -5
for v in prevhz base prevhz base; do
  if [ "$v" = base ]; then unset DSV1_SO; else export DSV1_SO=$GRAFT_REPO_ROOT/digital-subband-video-1_amd/variants/$v/libdsv1_mi355x.so; fi
  echo "== $v"; SHAPE_PROF=1 python3 tools/bench_shape.py 1920 1080 2 256 0 85 1 0 4 2>&1 | head -9
done
AB_STEPS=4 tools/ab/run_variants.sh "k_hz_collect k_hz_emit" prevhz base prevhz base
