"""Build a second copy of the library with a development macro (e.g. -DRES_TRACE, -DRP_TRACE) into scratch/<name>.so:
    python profiles/tools/build_trace_lib.py RES_TRACE scratch/lib_res_trace.so
Use it with MULTINN_HIP_LIB=<path> (the loader checks the ABI version as for the product library)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multinn_amd import build  # noqa: E402

macro, out = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
procs = []
for src in build.SOURCES:
    obj = "/tmp/tr_" + macro + "_" + src.replace(".hip", ".o")
    procs.append((obj, subprocess.Popen([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + build.flags_for(src) + ["-D" + macro, "-c", os.path.join(build.CSRC, src), "-o", obj],
                                        stderr=subprocess.DEVNULL)))
for o, p in procs:
    assert p.wait() == 0, o
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [o for o, _ in procs] + ["-ldl"],
                      stderr=subprocess.DEVNULL)
print(out)
