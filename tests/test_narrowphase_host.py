"""The product's narrow phase (SAT / closed forms / GJK margin query / hollow-mesh rule),
compiled for the CPU by tests/harness.py, against the oracle's restated
ncollide3d::query::distance on random shape pairs.  Flags must agree except
inside the stated band around the margin (DESIGN.md "Parity band")."""
import math

import numpy as np
import pytest

from gear_amd.descriptors import ShapeSpec
from gear_amd.urdf import quat_from_rpy
from oracle import binding as orc
from tests import harness

MARGIN = 0.01
BAND = {32: 2e-5, 64: 1e-9}


def rand_convex_mesh(rng, scale):
    from scipy.spatial import ConvexHull
    pts = rng.normal(size=(int(rng.integers(8, 24)), 3))
    pts *= scale * rng.uniform(0.5, 1.0, 3) / np.linalg.norm(pts, axis=1, keepdims=True)
    hull = ConvexHull(pts)
    used = np.unique(hull.simplices)
    remap = {int(o): i for i, o in enumerate(used)}
    verts = pts[used]
    tris = [[remap[int(i)] for i in s] for s in hull.simplices]
    return verts.tolist(), tris


def rand_shape(rng, kind, spread):
    t = tuple(rng.uniform(-spread, spread, 3))
    q = quat_from_rpy(*rng.uniform(-math.pi, math.pi, 3))
    if kind == "box":
        return ShapeSpec.box(rng.uniform(0.03, 0.25, 3), t, q)
    if kind == "ball":
        return ShapeSpec.ball(rng.uniform(0.03, 0.2), t, q)
    if kind == "cyl":
        return ShapeSpec.cylinder(rng.uniform(0.02, 0.12), rng.uniform(0.05, 0.4), t, q)
    v, tr = rand_convex_mesh(rng, rng.uniform(0.05, 0.25))
    return ShapeSpec.mesh(v, tr, t, q)


PAIRS = [("box", "box"), ("box", "ball"), ("ball", "box"), ("ball", "ball"), ("cyl", "box"), ("box", "cyl"),
         ("cyl", "ball"), ("ball", "cyl"), ("cyl", "cyl"), ("mesh", "box"), ("mesh", "mesh"), ("ball", "mesh")]


@pytest.mark.parametrize("prec", [32, 64])
@pytest.mark.parametrize("ka,kb", PAIRS)
def test_random_pairs_match_oracle(ka, kb, prec):
    rng = np.random.default_rng(1000 + PAIRS.index((ka, kb)))
    n = 600 if "cyl" in (ka, kb) or "mesh" in (ka, kb) else 2000
    hits = near = uncert = bad = 0
    for _ in range(n):
        a = rand_shape(rng, ka, 0.22)
        b = rand_shape(rng, kb, 0.22)
        d, lo = orc.distance_bounds(a, b, prec)
        got = harness.pair_within(a, b, MARGIN, prec)
        assert got in (0, 1), got
        want = 1 if d < MARGIN else 0
        hits += want
        if got != want:
            if abs(d - MARGIN) <= BAND[prec]:
                near += 1
            elif want == 0 and lo - BAND[prec] < MARGIN:
                # the restated f32 GJK stopped (simplex degeneracy thresholds) before its own bounds
                # separated d from the margin: true distance in [lo, d] straddles it
                uncert += 1
            else:
                bad += 1
                print("MISMATCH", ka, kb, prec, d, lo, got, a, b)
    print(f"{ka}-{kb} f{prec}: n={n} hits={hits} in-band={near} oracle-uncertified={uncert} bad={bad}")
    assert bad == 0
    if prec == 64:
        assert uncert == 0
    assert 0.05 * n < hits < 0.98 * n  # the sample exercises both outcomes


@pytest.mark.parametrize("prec", [32, 64])
def test_near_margin_box_pairs(prec):
    """Pairs placed deliberately close to the margin: face-face, edge-edge and vertex-vertex gaps."""
    rng = np.random.default_rng(5)
    bad = 0
    for _ in range(400):
        ha, hb = rng.uniform(0.05, 0.3, 3), rng.uniform(0.05, 0.3, 3)
        sign = rng.choice([-1.0, 1.0], 3)
        mode = rng.integers(1, 4)  # number of axes along which the boxes are separated
        gap = MARGIN * rng.uniform(0.7, 1.3) / math.sqrt(mode)
        off = np.zeros(3)
        axes = rng.permutation(3)[:mode]
        for ax in range(3):
            if ax in axes:
                off[ax] = sign[ax] * (ha[ax] + hb[ax] + gap)
            else:
                off[ax] = rng.uniform(-0.5, 0.5) * min(ha[ax], hb[ax])
        a, b = ShapeSpec.box(ha), ShapeSpec.box(hb, tuple(off))
        true_d = gap * math.sqrt(mode)
        got = harness.pair_within(a, b, MARGIN, prec)
        d = orc.distance(a, b, prec)
        if abs(true_d - MARGIN) > BAND[prec]:
            assert got == (1 if true_d < MARGIN else 0), (true_d, got)
        if got != (1 if d < MARGIN else 0) and abs(d - MARGIN) > BAND[prec]:
            bad += 1
    assert bad == 0


@pytest.mark.parametrize("prec", [32, 64])
def test_hollow_mesh_rule(prec):
    cyl = ShapeSpec.cylinder(0.5, 2.0)
    assert harness.pair_within(cyl, ShapeSpec.ball(0.1), MARGIN, prec) == 0          # strictly inside: free
    assert harness.pair_within(ShapeSpec.ball(0.1), cyl, MARGIN, prec) == 0
    assert harness.pair_within(cyl, ShapeSpec.box((0.1, 0.1, 0.1), (0, 0, 0.8)), MARGIN, prec) == 0
    assert harness.pair_within(cyl, ShapeSpec.box((0.1, 0.1, 0.1), (0, 0, 0.895)), MARGIN, prec) == 1  # 5 mm from the cap
    assert harness.pair_within(cyl, ShapeSpec.ball(0.1, (0.45, 0, 0)), MARGIN, prec) == 1  # crosses the wall
    assert harness.pair_within(ShapeSpec.cylinder(0.05, 0.1), ShapeSpec.box((1, 1, 1)), MARGIN, prec) == 1  # solid box
    assert harness.pair_within(cyl, ShapeSpec.cylinder(0.05, 0.1), MARGIN, prec) == 0   # mesh inside mesh
    assert harness.pair_within(ShapeSpec.cylinder(0.05, 0.1), cyl, MARGIN, prec) == 0


def test_open_mesh_is_a_surface():
    """An open mesh (one face of a tetrahedron missing) is no longer refused: it is its triangles.  A small ball at
    the tetrahedron's centroid touches none of them — free — and one through a remaining face collides."""
    v = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)]
    open_tris = [(0, 1, 2), (0, 1, 3), (0, 2, 3)]
    mesh = ShapeSpec.mesh(v, open_tris)
    assert harness.pair_within(mesh, ShapeSpec.ball(0.1, (5, 5, 5)), MARGIN, 64) == 0
    assert harness.pair_within(mesh, ShapeSpec.ball(0.05, (0.25, 0.25, 0.25)), MARGIN, 64) == 0
    assert harness.pair_within(mesh, ShapeSpec.ball(0.05, (0.25, 0.25, 0.03)), MARGIN, 64) == 1
    assert harness.pair_within(mesh, ShapeSpec.ball(0.05, (0.4, 0.4, 0.4)), MARGIN, 64) == 0      # outside, beyond the missing face
    for c in ((0.25, 0.25, 0.25), (0.25, 0.25, 0.03), (0.4, 0.4, 0.4)):
        d, _ = orc.distance_bounds(mesh, ShapeSpec.ball(0.05, c), 64)
        assert harness.pair_within(mesh, ShapeSpec.ball(0.05, c), MARGIN, 64) == (1 if d < MARGIN else 0)
    bad = ShapeSpec.mesh(v, [(0, 1, 7)])
    assert harness.pair_within(bad, ShapeSpec.ball(0.1, (5, 5, 5)), MARGIN, 64) == -1             # GH_ERR_INVALID


@pytest.mark.parametrize("prec", [32, 64])
def test_gjk_iteration_cap_is_conservative(prec):
    """The reference turns GJK non-convergence into distance 0, i.e. colliding (ncollide3d
    distance_support_map_support_map; restated in oracle/gear_oracle.hpp distance_support_maps).  The
    product's margin query must leave through its iteration cap on the same side.  A build of the
    narrow phase whose cap is ONE iteration (tests/harness.py lib_capped) forces that exit: a pair the
    full query proves free after several iterations must come back "within" from the capped build, and
    no pair may ever go from "within" to "free"."""
    rng = np.random.default_rng(77)
    flipped = same = 0
    for k in range(400):
        a = rand_shape(rng, "cyl" if k % 2 else "mesh", 0.2)
        b = rand_shape(rng, ("box", "ball", "cyl")[k % 3], 0.2)
        full = harness.pair_within(a, b, MARGIN, prec)
        capped = harness.pair_within(a, b, MARGIN, prec, capped=True)
        assert full in (0, 1) and capped in (0, 1)
        assert capped >= full, (k, full, capped)       # the cap never frees a colliding pair
        if capped != full:
            flipped += 1
            # such a pair is genuinely free by the oracle, clear of the band: the flip is the cap, not the geometry
            assert orc.distance(a, b, prec) > MARGIN + BAND[prec]
        else:
            same += 1
    print(f"f{prec}: {flipped} free pairs reported within by the one-iteration build, {same} unchanged")
    assert flipped > 10 and same > 10
    # a crafted one: a ball off the side face of the prism, 5.5 mm beyond the margin, reachable only
    # through the exact stage (its bounding cube overlaps the prism's bounding box near the box corner)
    cyl = ShapeSpec.cylinder(0.1, 0.4)
    ball = ShapeSpec.ball(0.01, (0.125 * math.cos(math.radians(42)), 0.125 * math.sin(math.radians(42)), 0.0))
    assert orc.distance(cyl, ball, prec) > MARGIN + 1e-3
    assert harness.pair_within(cyl, ball, MARGIN, prec) == 0
    assert harness.pair_within(cyl, ball, MARGIN, prec, capped=True) == 1


# ---- general triangle meshes (not closed, not convex): one device piece per triangle ---------------------
def rand_soup(rng, scale, kind):
    """kind 'open': a convex hull with a third of its faces removed; 'star': a non-convex closed star (the hull's
    vertices pushed in and out alternately); 'sheet': a wavy open height field."""
    if kind == "sheet":
        n = 4
        xs = np.linspace(-scale, scale, n)
        verts = [[x, y, 0.3 * scale * math.sin(5 * x / scale) * math.cos(4 * y / scale)] for x in xs for y in xs]
        tris = []
        for i in range(n - 1):
            for j in range(n - 1):
                a, b, c, d = i * n + j, i * n + j + 1, (i + 1) * n + j, (i + 1) * n + j + 1
                tris += [[a, b, c], [b, d, c]]
        return verts, tris
    v, t = rand_convex_mesh(rng, scale)
    v = np.asarray(v)
    if kind == "open":
        keep = [tri for k, tri in enumerate(t) if k % 3 != 0]
        return v.tolist(), keep
    f = np.where(np.arange(len(v)) % 2 == 0, 1.0, 0.45)[:, None]
    return (v * f).tolist(), t


@pytest.mark.parametrize("prec", [32, 64])
@pytest.mark.parametrize("soup", ["open", "star", "sheet"])
@pytest.mark.parametrize("other", ["box", "ball", "cyl", "mesh", "soup"])
def test_general_meshes_match_oracle(soup, other, prec):
    """The reference's TriMesh is a triangle soup: distance = min over triangles (urdf.rs:54-71).  The product
    flattens a mesh that is not a closed convex polytope into one convex piece per triangle; the oracle walks the
    triangles of the whole mesh."""
    rng = np.random.default_rng(7000 + 10 * ["open", "star", "sheet"].index(soup) + ["box", "ball", "cyl", "mesh", "soup"].index(other))
    n = 150 if other in ("mesh", "soup", "cyl") else 300
    hits = near = uncert = bad = 0
    for k in range(n):
        v, t = rand_soup(rng, rng.uniform(0.08, 0.25), soup)
        a = ShapeSpec.mesh(v, t, tuple(rng.uniform(-0.2, 0.2, 3)), quat_from_rpy(*rng.uniform(-math.pi, math.pi, 3)))
        if other == "soup":
            v2, t2 = rand_soup(rng, rng.uniform(0.08, 0.2), ("open", "star", "sheet")[k % 3])
            b = ShapeSpec.mesh(v2, t2, tuple(rng.uniform(-0.2, 0.2, 3)), quat_from_rpy(*rng.uniform(-math.pi, math.pi, 3)))
        else:
            b = rand_shape(rng, other, 0.2)
        if k % 2:
            a, b = b, a
        d, lo = orc.distance_bounds(a, b, prec)
        got = harness.pair_within(a, b, MARGIN, prec)
        assert got in (0, 1), got
        want = 1 if d < MARGIN else 0
        hits += want
        if got != want:
            if abs(d - MARGIN) <= BAND[prec]:
                near += 1
            elif want == 0 and lo - BAND[prec] < MARGIN:
                uncert += 1
            else:
                bad += 1
                print("MISMATCH", soup, other, prec, d, lo, got)
    print(f"{soup}-{other} f{prec}: n={n} hits={hits} in-band={near} oracle-uncertified={uncert} bad={bad}")
    assert bad == 0
    if prec == 64:
        assert uncert == 0
    assert 0.03 * n < hits < 0.99 * n
