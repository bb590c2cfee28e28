#!/bin/bash
# round 3, call 29: GPU tests that touch the host-side additions of the end of the round (batch construction, save paths,
# sampler decode / stochastic_encode, schedule helpers)
mkdir -p gpurun_out/r3
timeout 500 python -m pytest tests/test_model_gpu.py tests/test_htsat.py tests/test_reference_binding.py -q -m gpu -x -k "timesteps_subset or e2e_5step or sharded or masked or rerank or candidates or hip_conditioner or two_pass" 2>&1 | tail -4
