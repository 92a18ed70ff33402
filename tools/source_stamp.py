"""Hash of the kernel sources: identifies what a counter collection (profiles/pmc_latest.json) was measured on.
The GPU box receives the repository without .git, so a commit id is not available there; the sources are."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_stamp():
    csrc = os.path.join(ROOT, "gaussian-splatterer_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(csrc, "Makefile"),
                   os.path.join(ROOT, "include", "gsplat.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(source_stamp())
