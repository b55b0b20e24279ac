"""Content hash of the kernel sources (quandary_amd/csrc/*.hip, *.h, *.cpp, Makefile): what a committed profile was taken on.  The GPU box
has no .git, so provenance is carried by this hash: profiles/summarize.py stamps it into profiles/pmc_latest.json, bench.py compares it
with the tree it runs from and marks a profile of other sources as stale."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash(root=ROOT):
    h = hashlib.sha1()
    d = os.path.join(root, "quandary_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.cpp")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_hash())
