#!/usr/bin/env python3
"""A fingerprint of the kernel sources of the inference path: sha256 over yolo_fastestv2_amd/csrc/*.hip (without the training
path's yfv2_train.hip / yfv2_loss.hip), *.h, the Makefile and include/yfv2.h
(sorted by name, CRLF-free bytes as they are on disk), first 16 hex digits.  Profiles written by tools/*_summary.py carry
it, and bench.py quotes a profile's counters (HBM traffic, MFMA busy) in its JSON line ONLY when the profile's fingerprint
equals the one of the tree it runs from - a profile of an older build is evidence for that build, not for this one.
(`.git` does not travel to the GPU box, so a commit hash is not available there; this is computable on both sides.)"""
import glob
import hashlib
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash(repo=REPO):
    csrc = os.path.join(repo, "yolo_fastestv2_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(csrc, "Makefile")])
    # the training path (yfv2_train.hip, yfv2_loss.hip) shares no kernel with what bench.py times and the profiles measure
    files = [f for f in files if os.path.basename(f) not in ("yfv2_train.hip", "yfv2_loss.hip")]
    files.append(os.path.join(repo, "include", "yfv2.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
