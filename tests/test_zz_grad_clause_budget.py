"""Runs LAST (file name): the gradient tests pass a tensor on rel-L2 <= 2e-4 against the fp64 oracle, OR -- BatchNorm models --
EXACT (rel-L2 <= 5e-5) against the fp64 backward evaluated on the relu branches the device took ("branch_exact": a strict
comparison without kink ambiguity, not an escape), OR under an absolute floor, OR -- LayerNorm models only, where there is no
branch-exact oracle -- within 3x the reference's own fp32 noise.  Rounds 1-4 allowed the noise clause for every model and raised
its budget when kernels changed ("a budget that is raised whenever the kernels change bounds nothing"); round 5
removed it for BatchNorm models, which were the only ones that ever used it.
This test makes what is left visible and bounded: the per-test tally of which clause decided each tensor is written to
gpurun_out/grad_clauses.{json,txt} (committed copies under profiles/), and the suite FAILS when a test took more "noise" or
"floor" escapes than tests/golden/grad_clause_baseline.json allows.  The baseline file is never edited in a commit that also
touches a kernel."""
import json
import os

import pytest

from helpers import GRAD_CLAUSES

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASELINE = os.path.join(REPO, "tests", "golden", "grad_clause_baseline.json")
ESCAPES = ("noise", "floor")          # budgeted; "branch_exact" is tallied and reported, it is an exact comparison


def test_gradient_escape_clauses_stay_within_the_committed_baseline():
    if not GRAD_CLAUSES:
        pytest.skip("no gradient test ran in this session")
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    tally = {k: {c: int(n) for c, n in sorted(v.items()) if n} for k, v in sorted(GRAD_CLAUSES.items())}
    json.dump(tally, open(os.path.join(out, "grad_clauses.json"), "w"), indent=1)
    cols = ("l2", "branch_exact") + ESCAPES + ("miss",)
    lines = [f"{'test':110s} " + " ".join(f"{c:>12s}" for c in cols)]
    for k, v in tally.items():
        lines.append(f"{k[-110:]:110s} " + " ".join(f"{v.get(c, 0):12d}" for c in cols))
    tot = {c: sum(v.get(c, 0) for v in tally.values()) for c in cols}
    lines.append(f"{'TOTAL':110s} " + " ".join(f"{tot[c]:12d}" for c in cols))
    open(os.path.join(out, "grad_clauses.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-1:]))
    assert all(v.get("miss", 0) == 0 for v in tally.values())          # a miss fails its own test; belt and braces
    if not os.path.exists(BASELINE):
        pytest.skip("no committed baseline yet: copy gpurun_out/grad_clauses.json to tests/golden/grad_clause_baseline.json")
    base = json.load(open(BASELINE))
    worse = []
    for k, v in tally.items():
        if k not in base:
            continue                       # a new test: add it to the baseline when it is committed
        for c in ESCAPES:
            if v.get(c, 0) > base[k].get(c, 0):
                worse.append((k, c, v.get(c, 0), base[k].get(c, 0)))
    assert not worse, f"gradient tensors passing by an escape clause grew beyond the committed baseline: {worse}"
