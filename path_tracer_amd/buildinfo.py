"""Identity of the kernel sources a measurement belongs to."""
from __future__ import annotations

import glob
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def kernel_source_hash() -> str:
    """sha256 (first 16 hex digits) over the library's sources (csrc/*.hip, *.h, *.cpp, Makefile) in
    name order.  Counter summaries under profiles/ carry the hash of the sources they were measured
    on; bench.py reports them only while it still matches."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_CSRC, "*.hip")) + glob.glob(os.path.join(_CSRC, "*.h")) + glob.glob(os.path.join(_CSRC, "*.cpp")) +
                   [os.path.join(_CSRC, "Makefile")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
