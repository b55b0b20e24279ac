"""Timing of the global-memory sweeps (qd_big.h) on single large states: one workgroup per initial condition against teams of
workgroups (QD_BIG_TEAM / QD_BIG_SPREAD).  Usage: python profiles/big_probe.py [out.jsonl]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from helpers import synthetic_spec  # noqa: E402
from quandary_amd import capi  # noqa: E402

CASES = [
    ("10x10-lindblad-dim1e4", dict(nlevels=[10, 10], lindblad=True, target="pure", objective="Jfrobenius", init="pure, 0, 1"), 50),
    ("20x20-lindblad-dim1.6e5", dict(nlevels=[20, 20], lindblad=True, target="pure", objective="Jfrobenius", init="pure, 0, 1"), 20),
    ("4^4-lindblad-dim65536", dict(nlevels=[4, 4, 4, 4], lindblad=True, nessential=[2, 2, 2, 2], init="pure, 1, 0, 1, 0"), 20),
    ("3^5-lindblad-dim59049", dict(nlevels=[3, 3, 3, 3, 3], lindblad=True, nessential=[2, 2, 2, 2, 2], init="pure, 1, 0, 1, 0, 1"), 20),
]
TEAMS = [(1, 0), (2, 0), (4, 0), (8, 0), (16, 0), (32, 0), (8, 1), (32, 1), (64, 1), (128, 1), (256, 1)]

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
for name, kw, ntime in CASES:
    for linsolve in ("neumann", "gmres"):
        sp = synthetic_spec(**kw, ntime=ntime, nspline=5, linsolve=linsolve)
        ref = None
        for team, spread in TEAMS:
            os.environ["QD_BIG_TEAM"] = str(team)
            os.environ["QD_BIG_SPREAD"] = str(spread)
            h = capi.Handle(sp)
            opt = capi.Optim(h, sp)
            try:
                opt.evalF(sp.params0)
                t0 = time.perf_counter()
                val = opt.evalF(sp.params0)
                wall = 1e3 * (time.perf_counter() - t0)
                row = dict(case=name, dim=h.dim, linsolve=linsolve, team_asked=team, spread=spread, team=h.last_team, fwd_ms=h.forward_ms,
                           wall_ms=wall, applies=h.mean_applies, us_per_apply=1e3 * h.forward_ms / (ntime * h.mean_applies),
                           objective=val["objective"])
                if h.last_team == team:
                    if ref is None:
                        ref = val["objective"]
                    row["rel_diff_to_team1"] = abs(val["objective"] - ref) / abs(ref)
                    print(json.dumps(row), flush=True)
                    if out:
                        out.write(json.dumps(row) + "\n")
                        out.flush()
            except Exception as e:  # noqa: BLE001
                print(name, linsolve, team, spread, "failed:", e, flush=True)
            opt.close(); h.close()
