"""Analytic known answers for the oracle's restatement of ncollide3d::query::distance.

The reference holds no numeric distance fixtures (parity unpinned
beyond the three link-name lists), so the restated GJK / simplex / TriMesh code
is pinned here to closed-form geometry instead."""
import itertools
import math

import numpy as np
import pytest

from gear_amd.descriptors import ShapeSpec
from gear_amd.urdf import quat_from_rpy
from oracle import binding as orc

REL32 = 1.2e-3  # sqrt(10*eps_f32): GJK's relative stopping tolerance
REL64 = 5e-8


def brute_simplex(points):
    """closest point to the origin on conv(points) by active-set enumeration"""
    P = np.asarray(points, float)
    best, bestp = np.inf, None
    k = len(P)
    for r in range(1, k + 1):
        for S in itertools.combinations(range(k), r):
            Q = P[list(S)]
            if r == 1:
                lam = np.array([1.0])
            else:
                A = (Q[1:] - Q[0]).T
                x, *_ = np.linalg.lstsq(A, -Q[0], rcond=None)
                lam = np.concatenate([[1 - x.sum()], x])
            if (lam >= -1e-12).all():
                p = lam @ Q
                d = np.linalg.norm(p)
                if d < best:
                    best, bestp = d, p
    return bestp


@pytest.mark.parametrize("k", [2, 3, 4])
def test_simplex_projection_matches_bruteforce(k):
    rng = np.random.default_rng(k)
    for trial in range(400):
        P = rng.normal(size=(k, 3)) * rng.uniform(0.1, 2.0)
        if trial % 3 == 0:
            P += rng.normal(size=3) * 2.0  # origin mostly outside
        got = orc.project_simplex(P)
        want = brute_simplex(P)
        assert abs(np.linalg.norm(got) - np.linalg.norm(want)) < 1e-9, (P, got, want)


def test_ball_ball_closed_form():
    a = ShapeSpec.ball(0.3, t=(0, 0, 0))
    b = ShapeSpec.ball(0.2, t=(1, 0, 0))
    assert orc.distance(a, b, 64) == pytest.approx(0.5, abs=1e-15)
    assert orc.distance(a, ShapeSpec.ball(0.2, t=(0.4, 0, 0)), 64) == 0.0


@pytest.mark.parametrize("prec,rel", [(32, REL32), (64, REL64)])
def test_axis_aligned_boxes(prec, rel):
    a = ShapeSpec.box((0.5, 0.5, 0.5))
    # face-face, edge-edge, vertex-vertex separations
    for off, want in [((2.0, 0.0, 0.0), 1.0), ((2.0, 2.0, 0.0), math.sqrt(2.0)),
                      ((2.0, 2.0, 2.0), math.sqrt(3.0)), ((1.3, 0.2, -0.1), 0.3)]:
        b = ShapeSpec.box((0.5, 0.5, 0.5), t=off)
        d = orc.distance(a, b, prec)
        assert want <= d * (1 + 1e-6) and d <= want * (1 + rel) + 1e-7, (off, d, want)
    assert orc.distance(a, ShapeSpec.box((0.5, 0.5, 0.5), t=(0.9, 0.1, 0.0)), prec) == 0.0


@pytest.mark.parametrize("prec,rel", [(32, REL32), (64, REL64)])
def test_rotated_box_vertex_to_face(prec, rel):
    # cube rotated (shortest arc) so that its (1,1,1) vertex points down -x at the other box's face
    u = np.array([1.0, 1.0, 1.0]) / math.sqrt(3.0)
    v = np.array([-1.0, 0.0, 0.0])
    w = np.cross(u, v)
    q = np.array([w[0], w[1], w[2], 1.0 + u @ v])
    q = tuple(q / np.linalg.norm(q))
    h = 0.5
    a = ShapeSpec.box((h, h, h), t=(3.0, 0, 0), q=q)
    b = ShapeSpec.box((1.0, 1.0, 1.0))
    want = 3.0 - h * math.sqrt(3.0) - 1.0
    d = orc.distance(a, b, prec)
    assert want * (1 - 1e-6) <= d <= want * (1 + rel) + 1e-7


@pytest.mark.parametrize("prec,rel", [(32, REL32), (64, REL64)])
def test_ball_box(prec, rel):
    b = ShapeSpec.box((0.5, 1.0, 0.25))
    for c, r in [((2.0, 0, 0), 0.3), ((1.5, 2.0, 0), 0.5), ((1.0, 1.5, 0.75), 0.1)]:
        p = np.array(c)
        q = np.clip(p, [-0.5, -1.0, -0.25], [0.5, 1.0, 0.25])
        want = np.linalg.norm(p - q) - r
        d = orc.distance(ShapeSpec.ball(r, t=c), b, prec)
        assert want * (1 - 1e-5) - 1e-7 <= d <= want * (1 + rel) + 1e-7
    assert orc.distance(ShapeSpec.ball(0.3, t=(0.7, 0, 0)), b, prec) == 0.0


def test_cylinder_is_30gon_prism():
    v = orc.cylinder_vertices(0.02, 0.15, 64)
    assert v.shape == (60, 3)
    np.testing.assert_allclose(np.hypot(v[:, 0], v[:, 1]), 0.02, rtol=1e-12)
    np.testing.assert_allclose(np.abs(v[:, 2]), 0.075, rtol=1e-12)
    ang = np.arctan2(v[:30, 1], v[:30, 0]) % (2 * math.pi)
    np.testing.assert_allclose(ang, np.arange(30) * 2 * math.pi / 30, atol=1e-12)


@pytest.mark.parametrize("prec,rel", [(32, REL32), (64, REL64)])
def test_cylinder_vs_box_solid_distance(prec, rel):
    cyl = ShapeSpec.cylinder(0.1, 0.4)  # axis z, vertex 0 at angle 0 -> reaches x = 0.1
    # box to the +x side: closest feature is the prism vertex/edge at x = r
    d = orc.distance(cyl, ShapeSpec.box((0.1, 0.1, 0.1), t=(0.5, 0, 0)), prec)
    assert 0.3 * (1 - 1e-5) <= d <= 0.3 * (1 + rel) + 1e-7
    # box above the cap
    d = orc.distance(cyl, ShapeSpec.box((0.05, 0.05, 0.05), t=(0, 0, 0.45)), prec)
    assert 0.2 * (1 - 1e-5) <= d <= 0.2 * (1 + rel) + 1e-7
    # along the direction between two prism vertices the flat side is nearer than r
    a = math.pi / 30
    want = 0.5 * math.cos(0) - 0.1 * math.cos(a)  # ball centre on the mid-face direction
    c = (0.5 * math.cos(a), 0.5 * math.sin(a), 0.0)
    d = orc.distance(cyl, ShapeSpec.ball(0.05, t=c), prec)
    want = 0.5 - 0.1 * math.cos(a) - 0.05
    # round shapes: the f32 simplex's absolute degeneracy thresholds (eps_tol on squared
    # lengths) stop GJK a little before eps_rel is met -> allow 3e-3 in f32
    assert want * (1 - 1e-5) <= d <= want * (1 + (3e-3 if prec == 32 else rel)) + 1e-7


@pytest.mark.parametrize("prec", [32, 64])
def test_hollow_trimesh_semantics(prec):
    """A TriMesh is a surface: a shape strictly inside a closed mesh is at a positive
    distance (SURVEY §8a "hollow mesh"); the converse (triangle inside a solid) is 0."""
    cyl = ShapeSpec.cylinder(0.5, 2.0)
    inner = ShapeSpec.ball(0.1, t=(0, 0, 0))
    a = math.pi / 30
    want = 0.5 * math.cos(a) - 0.1  # nearest flat side of the 30-gon
    d = orc.distance(cyl, inner, prec)
    assert d == pytest.approx(want, rel=2e-3)
    box_in = ShapeSpec.box((0.1, 0.1, 0.1), t=(0, 0, 0.8))
    d = orc.distance(cyl, box_in, prec)
    assert d == pytest.approx(0.1, rel=2e-3)  # to the top cap at z = 1
    # touching the surface -> 0
    assert orc.distance(cyl, ShapeSpec.ball(0.1, t=(0.45, 0, 0)), prec) == 0.0
    # small cylinder inside a big box: triangles are inside a solid cuboid -> 0
    assert orc.distance(ShapeSpec.cylinder(0.05, 0.1), ShapeSpec.box((1, 1, 1)), prec) == 0.0
    # argument order must not matter
    assert orc.distance(inner, cyl, prec) == pytest.approx(want, rel=2e-3)


@pytest.mark.parametrize("prec,rel", [(32, REL32), (64, REL64)])
def test_mesh_mesh(prec, rel):
    a = ShapeSpec.cylinder(0.1, 0.4)
    b = ShapeSpec.cylinder(0.1, 0.4, t=(0, 0, 1.0))
    d = orc.distance(a, b, prec)
    assert 0.6 * (1 - 1e-5) <= d <= 0.6 * (1 + rel) + 1e-7
    assert orc.distance(a, ShapeSpec.cylinder(0.1, 0.4, t=(0.05, 0, 0.3)), prec) == 0.0


def test_random_box_pairs_f32_vs_f64_band():
    rng = np.random.default_rng(7)
    worst = 0.0
    for _ in range(300):
        ha, hb = rng.uniform(0.05, 0.4, 3), rng.uniform(0.05, 0.4, 3)
        ta, tb = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
        qa, qb = quat_from_rpy(*rng.uniform(-3, 3, 3)), quat_from_rpy(*rng.uniform(-3, 3, 3))
        a, b = ShapeSpec.box(ha, ta, qa), ShapeSpec.box(hb, tb, qb)
        d32, d64 = orc.distance(a, b, 32), orc.distance(a, b, 64)
        if d64 > 1e-4:
            assert d64 * (1 - 1e-5) - 1e-6 <= d32 <= d64 * (1 + REL32) + 1e-6
            worst = max(worst, d32 / d64 - 1)
        else:
            assert d32 < 1e-3
    assert worst < REL32
