cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
( echo "# same box: sources before csrc/decode.hip (commit dfa6db6, ABI 6) vs final sources (ABI 7)"
  (cd tools/gpu/old_tree && timeout -k 5 80 python tools/step_probe.py audioldm2-full 2 < /dev/null 2>&1 | grep "unet step" | sed 's/^/before decode.hip: /')
  timeout -k 5 80 python tools/step_probe.py audioldm2-full 2 < /dev/null 2>&1 | grep "unet step" | sed 's/^/final sources:     /'
  (cd tools/gpu/old_tree && timeout -k 5 80 python tools/step_probe.py audioldm2-full 2 < /dev/null 2>&1 | grep "unet step" | sed 's/^/before decode.hip: /')
  timeout -k 5 80 python tools/step_probe.py audioldm2-full 2 < /dev/null 2>&1 | grep "unet step" | sed 's/^/final sources:     /'
) | tee gpurun_out/r4/step_ab_decode_hip.txt
