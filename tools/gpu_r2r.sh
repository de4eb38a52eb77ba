#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" timeout 120 python tools/dbg_persist.py 2>&1 | grep -E "^ok|fault|LmxError" | cut -c1-200 | head -2; }
run LMX_DECODE_PERSIST_STEPS=5
run LMX_DECODE_PERSIST_STEPS=0
run LMX_DECODE_PERSIST=0
