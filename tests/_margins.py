"""Measured error next to its bound: ``within(name, measured, bound)`` asserts and remembers the largest value seen per
name; the session writes them to gpurun_out/margins.json (tests/conftest.py), so that the bounds in the tests can be
stated - and kept - at about twice what is measured."""
MARGINS = {}


def within(name, measured, bound, detail=None):
    measured = float(measured)
    rec = MARGINS.setdefault(name, {"max_measured": 0.0, "bound": float(bound), "n": 0})
    rec["max_measured"] = max(rec["max_measured"], measured)
    rec["bound"] = float(bound)
    rec["n"] += 1
    assert measured <= bound, (name, measured, bound, detail)
