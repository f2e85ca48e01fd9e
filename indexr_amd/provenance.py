"""Which kernel sources a measurement belongs to: a hash over the files libixrgpu.so is built from.

Counter files under profiles/ are measured in one run and quoted by later ones; the hash lets a later run tell whether
the kernels it just timed are the kernels the counters were collected on."""
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def csrc_hash():
    """sha256 over name and content of every source file in indexr_amd/csrc (sorted), first 16 hex digits"""
    h = hashlib.sha256()
    for name in sorted(os.listdir(_CSRC)):
        if not name.endswith((".hip", ".h", ".cpp")) and name != "Makefile":
            continue
        with open(os.path.join(_CSRC, name), "rb") as fh:
            data = fh.read()
        h.update(name.encode() + b"\0" + str(len(data)).encode() + b"\0" + data)
    return h.hexdigest()[:16]
