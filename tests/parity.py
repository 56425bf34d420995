"""Flag comparison with the stated parity band (DESIGN.md "Parity band")."""
import numpy as np

BAND = {32: 2e-5, 64: 1e-9}   # |d - margin| band at margin 0.01
POSE_TOL = {32: 1e-6, 64: 1e-12}


def compare_flags(got, want, band, precision):
    """got: HIP flags; want: oracle flags (all-pairs mode); band: oracle's [n][2] diagnostics.
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
            "outside_idx": np.nonzero(outside)[0][:10].tolist()}
