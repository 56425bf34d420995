"""Flag comparison against the oracle, in three named figures (DESIGN.md "Parity bands"):

  in_band      mismatches with some pair inside the STATED band |d - margin| <= 2e-5 (f32) / 1e-9 (f64);
  uncertified  f32 only, the SECOND stated band: the restated reference reports "free" for a pair whose own GJK
               run stopped (VoronoiSimplex::add_point refused a support point: absolute thresholds eps_tol = 1.19e-6
               on squared lengths and areas) with the margin inside its certified interval [lower, d].  Every such
               case is re-evaluated by the f64 oracle, which must side with the GPU; the per-case evidence (pair
               kinds, d32, lower32, GJK iterations and exit, d64, d64 - margin) is recorded;
  outside      everything else: must be ZERO.

Reports go to gpurun_out/parity_summary.json on the GPU box; the round's copy is committed under profiles/."""
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


MAX_CASES = 24   # per-case evidence kept per report


def explain_case(q_row, what, ora32, ora64):
    """The evidence for one uncertified configuration: every pair the f32 oracle reports free with the margin
    inside its certified interval [lower, d], with how its GJK run ended and what the f64 oracle makes of the
    same pair on the same (f32-valued) input."""
    m32 = ora32.margin
    r32 = [r for r in ora32.explain(q_row, what, window=0.0) if r["d"] >= m32 and r["lower"] < m32]
    all64 = {(r["self"], r["id_a"], r["id_b"]): r for r in
             ora64.explain(np.asarray(q_row, np.float64), what, window=float("inf"))}
    out = []
    for r in r32:
        o = all64.get((r["self"], r["id_a"], r["id_b"]))
        out.append({"pair": ("self" if r["self"] else "env") + f":{r['id_a']}:{r['id_b']}", "kinds": r["kinds"],
                    "d32": r["d"], "lower32": r["lower"], "gjk_iters32": r["iters"], "gjk_exit32": r["exit"],
                    "d64": None if o is None else o["d"],
                    "d64_minus_margin": None if o is None else o["d"] - ora64.margin,
                    "gjk_exit64": None if o is None else o["exit"]})
    return out


def adjudicate(rep, got, want, q, what, ora64, ora32=None):
    """The f32 oracle stops its GJK early on some pairs (absolute simplex thresholds) and then reports
    "free" with a certified interval that straddles the margin.  Each such configuration is re-evaluated
    by the f64 oracle on the same (f32-valued) input: it must either report the bits the GPU reported, or
    put the pair inside the f32 band.  Adds `adjudicated` / `unresolved` to the report; unresolved cases
    move to `outside`.  With the f32 oracle at hand the per-case evidence is recorded too (`cases`), and
    `max_margin_minus_d64`: how far below the margin the true distance of an excused pair was."""
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
    if ora32 is not None:
        cases, worst = [], 0.0
        for k, i in enumerate(idx[:MAX_CASES]):
            pairs = explain_case(np.asarray(q)[i], what, ora32, ora64)
            for p_ in pairs:
                if p_["d64_minus_margin"] is not None and p_["d64_minus_margin"] < 0:
                    worst = max(worst, -p_["d64_minus_margin"])
            cases.append({"config": int(i), "gpu": int(np.asarray(got)[i]), "oracle_f32": int(np.asarray(want)[i]),
                          "oracle_f64": int(w64[k]), "pairs": pairs})
        rep["cases"] = cases
        rep["max_margin_minus_d64"] = worst
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
        adjudicate(rep, got, want, q, what, ora64, ora32=ora)
    else:  # the f64 oracle always certifies its own answers; an f32 check without the f64 oracle may excuse nothing
        assert rep["oracle_uncertified"] == 0, rep
    rates = {k: round(float(((want & b) > 0).mean()), 4) for k, b in (("env", 1), ("self", 2), ("limit", 4))}
    record(name, rep, rates=rates)
    assert rep["outside"] == 0, rep
    assert rep["oracle_uncertified"] <= max(2, UNCERT_RATE_MAX * rep["total"]), rep
    return rep, want


_EXPLAINED = {}


def check_flags_fixture(name, hip, fx, what=None, got=None):
    """HIP flags through the C ABI vs the oracle answers COMMITTED under tests/golden/ (tests/oracle_fixtures.py):
    no oracle run on the GPU box except to collect the evidence of the handful of second-band cases.
    fx: oracle_fixtures.load(...).  `what` defaults to the fixture's predicates (path-selection bits may be OR-ed in).
    Returns (report, got)."""
    spec = fx["spec"]
    precision = spec["prec"]
    what = spec["what"] if what is None else what
    q, want, band = fx["q"], fx["flags"], fx["band"]
    if got is None:
        got = hip.check_batch(q, what)
    rep = compare_flags(got, want, band, precision)
    if precision == 32:
        idx = np.asarray(rep["uncertified_idx"], np.int64)
        rep["adjudicated"], rep["unresolved"] = 0, 0
        if idx.size:
            pos = np.searchsorted(fx["adj_idx"], idx)
            assert (fx["adj_idx"][pos] == idx).all(), "an uncertified case the fixture holds no f64 verdict for"
            w64, b64 = fx["adj_flags64"][pos], fx["adj_band64"][pos]
            extra = got[idx] & ~want[idx] & 3
            ok = ((extra & ~w64) == 0) | (b64 <= BAND[32])
            rep["adjudicated"], rep["unresolved"] = int(ok.sum()), int((~ok).sum())
            rep["outside"] += rep["unresolved"]
            rep["outside_idx"] += idx[~ok][:10].tolist()
            # the evidence of these few cases: one oracle evaluation each
            from oracle import binding as orc
            w = fx["w"]
            o32 = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 32)
            o64 = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 64)
            cases, worst = [], 0.0
            for k, i in enumerate(idx[:MAX_CASES]):
                key = (fx["name"], int(i))   # (the same case comes back on every evaluation path: explained once)
                if key not in _EXPLAINED:
                    _EXPLAINED[key] = explain_case(q[i], spec["what"], o32, o64)
                pairs = _EXPLAINED[key]
                worst = max([worst] + [-p_["d64_minus_margin"] for p_ in pairs if p_["d64_minus_margin"] is not None])
                cases.append({"config": int(i), "gpu": int(got[i]), "oracle_f32": int(want[i]), "oracle_f64": int(w64[k]),
                              "pairs": pairs})
            rep["cases"], rep["max_margin_minus_d64"] = cases, worst
    else:
        assert rep["oracle_uncertified"] == 0, rep
    rates = {k: round(float(((want & b) > 0).mean()), 4) for k, b in (("env", 1), ("self", 2), ("limit", 4))}
    record(name, rep, rates=rates, oracle="committed fixture tests/golden/oracle_%s.npz" % fx["name"])
    assert rep["outside"] == 0, rep
    assert rep["oracle_uncertified"] <= max(2, UNCERT_RATE_MAX * rep["total"]), rep
    return rep, got
