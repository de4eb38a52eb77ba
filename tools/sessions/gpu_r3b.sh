#!/bin/bash
# Round-3 GPU session B: flow parity tests + decode ms/token over the configurations given as arguments (tools/mb_decode.py).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${OUT:-r3b}; mkdir -p $O
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 900 python -m pytest tests/test_decode_flow_gpu.py -x -q > $O/pytest_flow.txt 2>&1; echo "pytest flow rc=$?"; tail -5 $O/pytest_flow.txt
fi
timeout 900 python tools/mb_decode.py "$@" > $O/mb_decode.jsonl 2> $O/mb_decode.err; echo "mb rc=$?"; cat $O/mb_decode.jsonl; tail -3 $O/mb_decode.err
