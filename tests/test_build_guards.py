"""Build-time guards on the recurrence kernels: no private (scratch) memory.

The row-parallel LSTM kernels keep a whole timestep's operands in registers and request the next step's operands a step ahead; an array that the
compiler leaves in scratch memory turns those prefetches into store-and-reload round trips (measured: backward 3.8 -> 6.6 ms per train step at the
bench shape when a `float4[4]` staging array went to scratch).  hipcc cross-compiles without a GPU, so this runs in the CPU suite."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_rowpar_kernels_use_no_scratch_memory(tmp_path):
    from multinn_amd import build
    src = os.path.join(build.CSRC, "lstm_rowpar.hip")
    out = str(tmp_path / "rp.s")
    subprocess.check_call([HIPCC] + build.FLAGS + ["-S", "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read()
    sizes = {m.group(1): int(m.group(2))
             for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", text)}
    kernels = {k: v for k, v in sizes.items() if "lstm_rowpar_" in k and "kernel" in k}
    assert len(kernels) >= 12, sizes                      # three widths x forward / backward x (one wave | a wave pair) per item
    assert all(v == 0 for v in kernels.values()), kernels
