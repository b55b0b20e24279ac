"""Copies golden DATA (configs, input vectors, expected output numbers) of the reference's own
regression tests into tests/golden/.  Run once in the build container, where the reference is
mounted at /root/reference; the copies are committed so the tests never read the reference tree.

Only data files are copied: <case>.cfg (the test's input), auxiliary input vectors, and
base/*.dat (expected outputs).  Large trajectory files are row-subsampled; the stride is
recorded in manifest.json.  No source code is copied.
"""
import json
import os
import shutil
import sys

REF = "/root/reference/tests/regression"
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ["AxC", "AxC_grad_initBasis0", "AxC_grad_schroedinger", "AxC_initDiag0", "AxC_initEnsemble", "AxC_initFile",
         "cnot", "pipulse", "xgate", "xgate_sparsemat", "state-to-state_spline0", "nlevels_4_4_4_4", "spinchain_N8",
         "hamiltonian-reader", "hamiltonian-reader-lindblad"]
MAX_BYTES = 120_000


def main():
    manifest = {}
    for case in CASES:
        src = os.path.join(REF, case)
        dst = os.path.join(HERE, case)
        os.makedirs(os.path.join(dst, "base"), exist_ok=True)
        for f in sorted(os.listdir(src)):
            p = os.path.join(src, f)
            if os.path.isfile(p):
                shutil.copy(p, os.path.join(dst, f))
        for f in sorted(os.listdir(os.path.join(src, "base"))):
            p = os.path.join(src, "base", f)
            size = os.path.getsize(p)
            with open(p) as fh:
                lines = fh.readlines()
            header = [l for l in lines if l.startswith("#")]
            rows = [l for l in lines if not l.startswith("#")]
            stride = 1
            while size / stride > MAX_BYTES and len(rows) // (stride * 2) >= 3:
                stride *= 2
            keep = rows[::stride]
            # always keep the last row as well (final time)
            last_extra = (len(rows) - 1) % stride != 0
            with open(os.path.join(dst, "base", f), "w") as fh:
                fh.writelines(header + keep + ([rows[-1]] if last_extra else []))
            manifest[f"{case}/base/{f}"] = {"row_stride": stride, "nrows_full": len(rows), "last_row_appended": last_extra}
    with open(os.path.join(HERE, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())
