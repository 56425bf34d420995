"""The second parity band, demonstrated instead of inferred (DESIGN.md "Parity bands", tests/parity.py).

ncollide3d 0.23's VoronoiSimplex::add_point (restated in oracle/gear_oracle.hpp) refuses a new support point
with ABSOLUTE thresholds — eps_tol = 10 * f32::EPSILON = 1.19e-6 on a squared length (a point within 1.1 mm
of a simplex vertex), on a squared cross product (a triangle of less than ~5 cm^2 on shapes of this size) or on a
plane offset — and GJK then returns the distance of the simplex it has.  A ball's support point moves
continuously with the direction, so for a small ball near a box the f32 run ends that way after 2-9 iterations
with an upper bound d that can lie millimetres above the true distance: the reference then reports "free" although
its own certified interval [lower, d] contains the margin.  The f64 run (eps_tol = 2.2e-15) does not.  This file
constructs such pairs deliberately, on the CPU, and checks what the GPU test-suite relies on:
the f32 run's interval is honest (lower <= true distance <= d), the exit is the one named, and the product's
narrow phase (compiled for the host) sides with the f64 oracle.  The reference itself cannot be run here
(no Rust toolchain; the crates are not vendored), so the thresholds are [recalled]: parity unpinned."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from gear_amd import descriptors as D
from oracle import binding as orc
from tests import harness

MARGIN = 0.01
BAND32 = 2e-5


def near_margin_ball_box_pairs(n, seed):
    """Small balls 8.5-10 mm (true distance, up to the sideways slide) from a face or an edge of a small box."""
    rng = np.random.Generator(np.random.Philox(seed))
    for _ in range(n):
        r = rng.uniform(0.03, 0.12)
        h = rng.uniform(0.02, 0.15, 3)
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        p = rng.uniform(-1, 1, 3) * h
        ax = int(rng.integers(3))
        p[ax] = rng.choice([-1.0, 1.0]) * (h[ax] + r + rng.uniform(0.0085, 0.00999))
        if rng.random() < 0.5:
            ax2 = (ax + 1) % 3
            p[ax2] = rng.choice([-1.0, 1.0]) * (h[ax2] + rng.uniform(0, 0.05))
        yield D.ShapeSpec.ball(r, tuple(Rotation.from_quat(q).apply(p))), D.ShapeSpec.box(h, (0, 0, 0), tuple(q))


def test_f32_gjk_early_exit_reports_free_inside_its_own_interval():
    cases = []
    for ball, box in near_margin_ball_box_pairs(3000, seed=11):
        d32, lo32 = orc.distance_bounds(ball, box, 32)
        iters, exit32 = orc.last_gjk()
        d64 = orc.distance(ball, box, 64)
        if d64 < MARGIN - BAND32 and d32 >= float(np.float32(MARGIN)):      # f32 says free, the distance says colliding
            cases.append((ball, box, d32, lo32, d64, iters, exit32))
    assert len(cases) >= 20, "the early exit did not occur: the category would be inferred, not demonstrated"
    worst = 0.0
    for ball, box, d32, lo32, d64, iters, exit32 in cases:
        assert exit32 == "point_rejected" and 1 <= iters <= 12            # how the run ended
        assert lo32 < MARGIN <= d32                                        # the margin is inside the certified interval
        assert lo32 <= d64 * (1 + 1e-4) + 1e-6 and d64 <= d32 * (1 + 1e-4) + 1e-6    # ... which is honest
        # the product's narrow phase (closed form for ball-box) sides with the true distance in both precisions
        assert harness.pair_within(ball, box, MARGIN, 32) == 1
        assert harness.pair_within(ball, box, MARGIN, 64) == 1
        worst = max(worst, MARGIN - d64)
    print(f"{len(cases)} of 3000 constructed pairs: f32 reference free, true distance up to {worst:.2e} below the margin")
    assert worst > 5e-4      # far outside the first band: this category cannot be folded into |d - margin| <= 2e-5


def test_f64_gjk_does_not_exit_early_on_the_same_pairs():
    n = 0
    for ball, box in near_margin_ball_box_pairs(600, seed=11):
        d64, lo64 = orc.distance_bounds(ball, box, 64)
        if abs(d64 - MARGIN) > 1e-9:
            if d64 >= MARGIN:
                assert lo64 >= MARGIN - 1e-9, (d64, lo64)   # "free" is always certified by the run's own lower bound
            n += 1
    assert n > 500
