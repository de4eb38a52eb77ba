#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_decode_persist_gpu.py -x -q -m gpu 2>&1 | tail -4
run() { env "$@" timeout 200 python tools/mb_persist.py 2>&1 | grep -E "persist_step|fault|Error" | cut -c1-220; }
for k in 1 2 3 6 161; do run LMX_DECODE_PERSIST=1 LMX_DECODE_PERSIST_STEPS=$k; done
run LMX_DECODE_PERSIST=1 LMX_DECODE_PERSIST_GRID=256
run LMX_DECODE_PERSIST=0
