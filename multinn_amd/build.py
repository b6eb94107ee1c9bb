"""In-tree build of libmultinn_hip.so (hipcc, gfx950 only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmultinn_hip.so")
SOURCES = ["gemm.hip", "gemm_bres.hip", "lstm_persist.hip", "lstm_rowpar.hip", "lstm_resident.hip", "lstm_cluster.hip", "elementwise.hip", "nade.hip", "nade_mfma.hip", "rbm.hip", "musical.hip", "det_step.hip", "comm.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wno-unused-result"]
# per-source additions.  lstm_resident.hip: MFMA results in VGPRs (the pointwise reads them there: no v_accvgpr_read per accumulator register),
# which leaves the AGPRs to the recurrent weights the matrix cores read in place
EXTRA_FLAGS = {"lstm_resident.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "lstm_cluster.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "gemm_bres.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def flags_for(src):
    return FLAGS + EXTRA_FLAGS.get(os.path.basename(src), [])


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "multinn_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the C-ABI shared library."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc] + flags_for(src) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
