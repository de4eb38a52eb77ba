"""The prefill GEMM's hot instantiations must stay free of scratch (register spills): gemm8p_kernel sits at 251 of 256 VGPRs, and a few more live values in a
rarely taken branch once doubled the whole prefill (round 4: a runtime `skip X half 1` flag in the ragged loop -> 88 bytes of spills per lane -> ragged tiles
took two tile-times).  hipcc reports the numbers at compile time; this test recompiles the one file (~1 min, no GPU needed)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "llava-plus-codebase_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_hot_gemm8p_instantiations_have_no_scratch(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I.", "-I../../include", "-c", "gemm8p.hip", "-o", str(tmp_path / "g.o"),
                        "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    usage = {}
    name = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            usage[name] = int(m.group(1))
    # every instantiation of the ping-pong kernel: <T, SPLIT> = {bf16, f16} x {1 (q|k|v, gate|up), 2, 3 (o_proj, down_proj)}
    hot = [k for k in usage if "gemm8p_kernel" in k]
    assert len(hot) == 6, sorted(usage)
    bad = {k: usage[k] for k in hot if usage[k] != 0}
    assert not bad, bad
