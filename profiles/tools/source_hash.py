"""sha256 (first 16 hex digits) over the kernel sources and the C-ABI header: the identity of the BUILD a counter recording belongs to.
bench.py compares it with the tree it runs from before it lets recorded rocprofv3 counters stand beside its live timings (the GPU box has no
.git, so a commit id cannot be used).  Usage: python profiles/tools/source_hash.py"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def source_hash():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "multinn_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "multinn_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
