"""Oracle end-to-end (per-read scoring + reads2 gather + global rank/cut) against the outputs of the
real reference binary recorded in tests/golden/e2e.json: exact ordered pass sets, trim/split child
coordinates (they are part of the child names), target / keeping numbers."""
import numpy as np

import _e2e_checks
import _oracle
import _pipeline


def test_oracle_matches_reference_binary_outputs():
    n = _e2e_checks.check_all(_pipeline.OracleBackend())
    assert n >= 18 + 2 + 16 + 10 + 6


def test_oracle_rank_tie_and_nan_do_not_crash():
    # all reads identical -> stdev == 0 -> NaN scores (main.cpp:192-195,206); reference prints -nan, must not crash
    r = _oracle.rank_and_cut(np.full(5, 80.0), np.full(5, 70.0), np.full(5, 1000, dtype=np.int32), np.ones(5, np.uint8),
                             target_bases=2500)
    assert r["outcome"] == 3 and int(r["passed"].sum()) == 3 and np.isnan(r["final_score"]).all()
