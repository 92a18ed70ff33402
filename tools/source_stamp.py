"""Hash of the kernel sources: identifies what a counter collection (profiles/pmc_latest.json) was measured on.
The GPU box receives the repository without .git, so a commit id is not available there; the sources are.
Comments and white space do not count: the hash is taken over the token stream (comments stripped, runs of white space
collapsed), so that editing a comment does not invalidate a collection while any change to the code does."""
import glob
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_COMMENT = re.compile(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\])*"', re.S)


def _code_only(text):
    text = _COMMENT.sub(lambda m: m.group(0) if m.group(0).startswith('"') else " ", text)   # keep string literals, drop comments
    return re.sub(r"\s+", " ", text).strip()


def source_stamp(root=ROOT):
    csrc = os.path.join(root, "gaussian-splatterer_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(csrc, "Makefile"),
                   os.path.join(root, "include", "gsplat.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        text = open(f, encoding="utf-8", errors="replace").read()
        h.update((_code_only(text) if not f.endswith("Makefile") else text).encode())
    return h.hexdigest()


if __name__ == "__main__":
    import sys
    print(source_stamp(sys.argv[1] if len(sys.argv) > 1 else ROOT))
