"""The tolerance model may not grow silently.  `helpers.parity_tolerance` declares a sample ill-conditioned
(tolerance +inf: only finiteness and flags are compared) when a one-ulp perturbation of an intermediate moves the result by more than
0.1 % or a selecting operator sits on its edge.  Round 4 moved the model five times, always towards "more samples incomparable".  This
test pins, per MODEL_VERSION, the share of samples the model declares ill-conditioned on three fixed populations
(tests/golden/tolerance_model_shares.json): a change of the model that raises a share fails here until the record is re-made
(`python tests/test_tolerance_model.py --record`) together with a MODEL_CHANGELOG entry citing the traced finding.
No GPU involved: the model is the float64 interpreter of the lowered program (tests/prog_interp.py)."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynamicexpressions_jl_amd as de
import fuzzlib as FZ
import helpers

RECORD = os.path.join(helpers.ROOT, "tests", "golden", "tolerance_model_shares.json")
POPULATIONS = {
    "hot_f32": dict(ops="hot", dtype="float32", seed=901, trees=48, N=384, scale=1.0),
    "wide_f32": dict(ops="wide", dtype="float32", seed=902, trees=48, N=384, scale=1.0),
    "hot_f64": dict(ops="hot", dtype="float64", seed=903, trees=48, N=384, scale=3.0),
}


def shares(key):
    cfg = POPULATIONS[key]
    ops = FZ.OPS_HOT if cfg["ops"] == "hot" else FZ.OPS_WIDE
    dtype = np.float32 if cfg["dtype"] == "float32" else np.float64
    rng = de.synth.Xoshiro256ss(cfg["seed"])
    F = 3
    trees = FZ.random_trees(rng, ops, F, dtype, cfg["trees"], 24)
    g = np.random.Generator(np.random.PCG64(cfg["seed"]))
    X = np.asfortranarray((g.standard_normal((F, cfg["N"])) * cfg["scale"]).astype(dtype))
    ill = total = 0
    for t in trees:
        tol = helpers.parity_tolerance(t, ops, X, dtype)
        ill += int(np.isinf(tol).sum())
        total += tol.size
    return ill / total


def test_model_has_a_version_and_a_changelog_entry_for_it():
    versions = [v for v, _ in helpers.MODEL_CHANGELOG]
    assert versions == sorted(set(versions)) and versions[-1] == helpers.MODEL_VERSION
    assert all(("profiles/" in text or "tools/" in text or v == 1) for v, text in helpers.MODEL_CHANGELOG), "every change cites its traced finding"


@pytest.mark.parametrize("key", sorted(POPULATIONS))
def test_ill_conditioned_share_does_not_grow(key):
    with open(RECORD) as fh:
        rec = json.load(fh)
    assert rec["model_version"] == helpers.MODEL_VERSION, "the model changed: add a MODEL_CHANGELOG entry and re-record (python tests/test_tolerance_model.py --record)"
    got = shares(key)
    print(f"[tolerance model v{helpers.MODEL_VERSION}] {key}: {100 * got:.2f} % of the samples ill-conditioned (recorded {100 * rec['shares'][key]:.2f} %)")
    assert got <= rec["shares"][key] + 1e-9, f"{key}: the model declares {got:.4f} of the samples ill-conditioned, {rec['shares'][key]:.4f} were recorded for version {rec['model_version']}"


if __name__ == "__main__" and "--record" in sys.argv:
    out = {"model_version": helpers.MODEL_VERSION, "made_by": "python tests/test_tolerance_model.py --record",
           "populations": POPULATIONS, "shares": {k: shares(k) for k in sorted(POPULATIONS)}}
    with open(RECORD, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(out["shares"], indent=1))
