#!/bin/bash
# Round 5, session N: output_attentions (tiny + real head geometry) and the training entry end to end
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_train_entry_gpu.py tests/test_model_gpu.py tests/test_real_geometry_gpu.py tests/test_train_step_gpu.py tests/test_api_surface_gpu.py -m gpu -q -p no:cacheprovider -k "attentions or train_entry or hidden_states or train_step or api_surface" 2>&1 | tail -25 ) > gpurun_out/r05_n_tests.txt 2>&1
tail -25 gpurun_out/r05_n_tests.txt
