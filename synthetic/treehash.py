"""Identity of the kernel sources a measurement was taken on: sha256 over csrc/* and the C-ABI header (16 hex digits).  tests/test_full_depth_gpu.py stamps its
reports with it and bench.py's `parity` object compares the stamp of the committed report with the tree it runs on (committed parity figures must not
pass for figures of a later build)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16() -> str:
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "llava-plus-codebase_amd", "csrc")
    files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile"]
    files.append(os.path.join(ROOT, "include", "llava_mi355x.h"))
    for p in files:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]
