"""Flag comparison with the stated parity band (DESIGN.md "Parity band"), the adjudication of the
`oracle_uncertified` category by the f64 oracle, and the record of every report
(gpurun_out/parity_summary.json on the GPU box; the round's copy is committed under profiles/)."""
import json
import os

import numpy as np

BAND = {32: 2e-5, 64: 1e-9}   # |d - margin| band at margin 0.01
POSE_TOL = {32: 1e-6, 64: 1e-12}
UNCERT_RATE_MAX = 5e-3        # bound on the share of configurations excused as oracle_uncertified (f32)

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUMMARY = os.environ.get("GEAR_PARITY_SUMMARY", os.path.join(_ROOT, "gpurun_out", "parity_summary.json"))


def compare_flags(got, want, band, precision):
    """got: HIP flags; want: oracle flags (all-pairs mode); band: oracle's [n][2] diagnostics
    (band[:,0] = min |d - margin| over all pairs, band[:,1] = min (certified lower bound - margin) over
    the pairs the oracle reports free).
    Returns dict(total, mismatched, in_band, oracle_uncertified, outside) — `outside` must be 0."""
    got, want = np.asarray(got), np.asarray(want)
    mis = got != want
    eps = BAND[precision]
    in_band = mis & (band[:, 0] <= eps)
    # the oracle said "free" for a pair whose certified lower bound is below the margin
    uncert = mis & ~in_band & (band[:, 1] - eps < 0) & ((got & ~want & 3) != 0) & ((want & ~got) == 0)
    outside = mis & ~in_band & ~uncert
    return {"total": int(got.size), "mismatched": int(mis.sum()), "in_band": int(in_band.sum()),
            "oracle_uncertified": int(uncert.sum()), "outside": int(outside.sum()),
            "outside_idx": np.nonzero(outside)[0][:10].tolist(), "uncertified_idx": np.nonzero(uncert)[0].tolist()}


def certain(band, precision):
    """Configurations whose every pair decision is unambiguous BY THE ORACLE'S OWN EVIDENCE: no pair
    inside the stated band, and every pair it reports free certified free (lower bound above the margin).
    Per-link / per-pair / first-bad comparisons are made on these (no wider "safe" filter)."""
    eps = BAND[precision]
    return (band[:, 0] > eps) & (band[:, 1] > eps)


def adjudicate(rep, got, want, q, what, ora64):
    """The f32 oracle stops its GJK early on some pairs (absolute simplex thresholds) and then reports
    "free" with a certified interval that straddles the margin.  Each such configuration is re-evaluated
    by the f64 oracle on the same (f32-valued) input: it must either report the bits the GPU reported, or
    put the pair inside the f32 band.  Adds `adjudicated` / `unresolved` to the report; unresolved cases
    move to `outside`."""
    idx = np.asarray(rep["uncertified_idx"], np.int64)
    rep["adjudicated"], rep["unresolved"] = 0, 0
    if idx.size == 0:
        return rep
    w64, b64 = ora64.check(np.asarray(q)[idx].astype(np.float64), what, threads=16)
    extra = np.asarray(got)[idx] & ~np.asarray(want)[idx] & 3          # bits the GPU has and the f32 oracle lacks
    ok = ((extra & ~w64) == 0) | (b64[:, 0] <= BAND[32])
    rep["adjudicated"] = int(ok.sum())
    rep["unresolved"] = int((~ok).sum())
    rep["outside"] += rep["unresolved"]
    rep["outside_idx"] += idx[~ok][:10].tolist()
    return rep


def record(name, rep, **extra):
    """Adds one report to the parity summary file (merged by name; tests may run in any order)."""
    entry = {k: v for k, v in rep.items() if k not in ("outside_idx", "uncertified_idx")}
    entry.update(extra)
    try:
        os.makedirs(os.path.dirname(SUMMARY), exist_ok=True)
        data = {}
        if os.path.exists(SUMMARY):
            with open(SUMMARY) as fh:
                data = json.load(fh)
        data[name] = entry
        with open(SUMMARY, "w") as fh:
            json.dump(data, fh, indent=1, sort_keys=True)
    except OSError:
        pass  # a read-only tree must not fail a parity test
    print("parity", name, entry)


def check_flags(name, hip, ora, q, what, precision, ora64=None, threads=16):
    """HIP flags through the C ABI vs the oracle on the same input: zero mismatches outside the band;
    every oracle_uncertified case adjudicated by the f64 oracle; its rate bounded.  Returns (report, want)."""
    got = hip.check_batch(q, what)
    want, band = ora.check(q, what, threads=threads)
    rep = compare_flags(got, want, band, precision)
    if precision == 32 and ora64 is not None:
        adjudicate(rep, got, want, q, what, ora64)
    else:  # the f64 oracle always certifies its own answers; an f32 check without the f64 oracle may excuse nothing
        assert rep["oracle_uncertified"] == 0, rep
    rates = {k: round(float(((want & b) > 0).mean()), 4) for k, b in (("env", 1), ("self", 2), ("limit", 4))}
    record(name, rep, rates=rates)
    assert rep["outside"] == 0, rep
    assert rep["oracle_uncertified"] <= max(2, UNCERT_RATE_MAX * rep["total"]), rep
    return rep, want
