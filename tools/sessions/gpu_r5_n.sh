#!/bin/bash
# Round 5, session N: the training entry end to end (output_attentions tests ran green in the first pass of this session: 34 passed)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_train_entry_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -60 ) > gpurun_out/r05_n_tests.txt 2>&1
grep -E "^E |passed|failed|losses" gpurun_out/r05_n_tests.txt | head -30
