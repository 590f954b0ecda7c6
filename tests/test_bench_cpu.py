"""bench.py's bookkeeping of the parity claims (no GPU): what `value_sets_exactly_equal`, `value_sets_detail` and
`images_per_s_with_exact_sets` say for a given `parity_vs_oracle_full_depth` object, and `compare_contacts` itself on hand-made vectors
(north star: probabilities within 1e-3, vertex-id sets bit-exact; sets: SURVEY.md App. A - eval_utils.py:75, run_demo.py:459)."""
import importlib.util
import os

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(REPO, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_compare_contacts_counts_flips_inside_the_error_band_only():
    b = _bench()
    ref = torch.tensor([[0.10, 0.2995, 0.3005, 0.4998, 0.5002, 0.90, 0.50]])
    got = ref.clone()
    got[0, 3] = 0.5001  # crosses 0.5 (3e-4 away from the oracle's value: inside the band)
    got[0, 1] = 0.2999  # stays below 0.3
    nv = np.array([[1, 0, 2, 1, 1, 4, 1]])
    c = b.compare_contacts(got, ref, torch.from_numpy(nv), nv)
    assert abs(c["max_abs_dp"] - 4e-4) < 1e-6 and c["within_1e-3"] and c["visibility_set_equal"]  # (the 0.2995 -> 0.2999 move)
    s5, s3 = c["threshold_sets"]["ge_0.5"], c["threshold_sets"]["gt_0.3"]
    assert s5["mismatches_in_band"] == 1 and not s5["set_exactly_equal"] and s5["equal_outside_error_band"]
    assert s3["mismatches_in_band"] == 0 and s3["set_exactly_equal"]
    # a different visibility set is reported as such
    nv2 = nv.copy()
    nv2[0, 1] = 1
    assert not b.compare_contacts(got, ref, torch.from_numpy(nv2), nv)["visibility_set_equal"]


def test_set_equality_fields_of_the_bench_line():
    b = _bench()

    def mode(vis, e5, e3, ips, ok=True, f5=0, f3=0):
        return {"visibility_set_equal": vis, "within_1e-3": ok, "images_per_s": ips,
                "threshold_sets": {"ge_0.5": {"set_exactly_equal": e5, "mismatches_in_band": f5},
                                   "gt_0.3": {"set_exactly_equal": e3, "mismatches_in_band": f3}}}
    pf = {"config": "x", "default": mode(True, False, True, 10.4, f5=1), "bf16": mode(True, False, False, 10.8, ok=False, f5=22, f3=3),
          "parity-fast": mode(True, True, True, 8.3), "parity": mode(True, True, True, 7.5)}
    assert b._sets_exact(pf, "default") is False and b._sets_exact(pf, "parity") is True
    assert b._sets_detail(pf, "default") == {"visibility": True, "ge_0.5": False, "gt_0.3": True, "flips_ge_0.5": 1, "flips_gt_0.3": 0}
    assert b._exact_sets_rate(pf) == {"mode": "parity-fast", "images_per_s": 8.3}  # the FASTEST mode with exact sets inside 1e-3
    pf["parity-fast"]["threshold_sets"]["ge_0.5"]["set_exactly_equal"] = False
    assert b._exact_sets_rate(pf) == {"mode": "parity", "images_per_s": 7.5}
    pf["parity"]["visibility_set_equal"] = False
    assert b._exact_sets_rate(pf) is None
    # no parity leg (e.g. --no-cpu-baseline): the fields are null, not false
    assert b._sets_exact(None, "default") is None and b._sets_detail(None, "default") is None and b._exact_sets_rate(None) is None
