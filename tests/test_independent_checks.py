"""Checks that do NOT share code with the product (CPU, no GPU, nothing from csrc/ imported or re-used):

* the LiDAR projection the oracle takes from the product header csrc/nvbx_lidar_math.h (so HIP-vs-oracle parity compares that
  code with itself) is restated here with numpy's arcsin / arctan2 in float64 exactly as the reference's own script does
  (/root/reference/nvblox_ros/scripts/calculate_lidar_params.py:50-58: elevation = arcsin(z / r), azimuth = arctan2(y, x),
  equal angular bins) -- pixel bins must agree on >= 10^5 random points away from bin edges, including the azimuth wrap;
* the three generated marching-cubes tables are checked against brute-force sign topology: per case every triangle edge on a
  cube face is a boundary edge (used once), every interior edge is shared by exactly two triangles with opposite direction
  (closed, consistently oriented surface), the face boundary separates exactly the inside from the outside corners of that
  face, the normals point to the positive side; rules 0 and 1 are crack-free across every shared face (all 4096 two-cube
  sign configurations), rule 2 (the classic table's complement symmetry) is not -- the known defect.
"""
import itertools

import numpy as np
import pytest

import oracle

CORNERS = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)], float)
EDGES = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]


# ------------------------------------------------------------------------------------------------ LiDAR projection
def numpy_lidar_pixel(lidar, pts):
    """Pixel (row, col) of each point and its distance to the nearest bin edge (in pixels), float64, libm."""
    cols, rows, min_range, min_el, max_el = lidar
    r = np.linalg.norm(pts, axis=1)
    el = np.arcsin(np.clip(pts[:, 2] / r, -1.0, 1.0))
    az = np.arctan2(pts[:, 1], pts[:, 0])
    rpp_el = (max_el - min_el) / (rows - 1); rpp_az = 2.0 * np.pi / cols
    # beam (k, j) through the pixel centre (j + 0.5, k + 0.5): corner-referenced coordinates
    u = (az + np.pi) / rpp_az + 0.5
    v = (max_el - el) / rpp_el + 0.5
    u = np.where(u >= cols, u - cols, u)
    inside = (v >= 0) & (v < rows) & (r >= min_range)
    margin = np.minimum(np.abs(u - np.round(u)), np.abs(v - np.round(v)))
    return np.floor(v).astype(int), np.floor(u).astype(int), inside, margin


def test_lidar_projection_against_numpy_restatement():
    rng = np.random.default_rng(3)
    for lidar in [(1024, 64, 0.1, -np.deg2rad(22.5), np.deg2rad(22.5)), (512, 16, 0.5, -np.deg2rad(15.0), np.deg2rad(10.0)),
                  (2048, 128, 0.1, -np.deg2rad(45.0), np.deg2rad(45.0))]:
        n = 200000
        el = rng.uniform(lidar[3] - 0.08, lidar[4] + 0.08, n); az = rng.uniform(-np.pi, np.pi, n)     # the field of view + a rim outside it
        d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], 1)
        pts = (d * rng.uniform(0.2, 180.0, (n, 1))).astype(np.float32)
        # force a share of the points onto the azimuth seam (y ~ 0, x < 0) and the axes
        pts[:2000, 1] = rng.normal(0.0, 1e-3, 2000).astype(np.float32); pts[:2000, 0] = -np.abs(pts[:2000, 0])
        pts[2000:2200, 0] = 0.0; pts[2200:2400, 1] = 0.0
        row, col, inside, margin = numpy_lidar_pixel(lidar, pts.astype(np.float64))
        checked = 0
        for i in range(n):
            if margin[i] < 2e-3:             # the polynomial atan2 of the product differs from libm by < 2e-7 rad = ~1e-4 px here
                continue
            got = oracle.lidar_project(lidar, pts[i])
            if not inside[i]:
                # outside the vertical field of view (with margin): the product must reject it too
                if np.abs(np.arcsin(np.clip(pts[i, 2] / np.linalg.norm(pts[i]), -1, 1))) > max(abs(lidar[3]), abs(lidar[4])) + 0.01:
                    assert got is None
                continue
            assert got is not None, (lidar, pts[i])
            assert (int(np.floor(got[1])), int(np.floor(got[0]))) == (row[i], col[i]), (lidar, pts[i], got, row[i], col[i])
            checked += 1
        assert checked >= 100000, checked


def test_lidar_beam_through_pixel_centre_round_trip():
    """Independent of the projection code: the direction of beam (k, j) built with numpy sin / cos projects to (j + .5, k + .5)."""
    lidar = (1024, 64, 0.1, -np.deg2rad(22.5), np.deg2rad(22.5))
    cols, rows, _, min_el, max_el = lidar
    for k in (0, 1, 31, 62, 63):
        for j in (0, 1, 511, 512, 1023):
            el = max_el - k * (max_el - min_el) / (rows - 1); az = -np.pi + j * 2 * np.pi / cols
            p = 17.0 * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
            u, v = oracle.lidar_project(lidar, p.astype(np.float32))
            assert abs(u - (j + 0.5)) < 2e-3 and abs(v - (k + 0.5)) < 2e-3, (k, j, u, v)


# ------------------------------------------------------------------------------------------------ marching cubes
def load_table(name):
    rows = []
    for line in open(name):
        line = line.strip()
        if line.startswith("{"):
            rows.append([int(x) for x in line.strip("{},").split(",")])
    assert len(rows) == 256
    return [[tuple(r[3 * t:3 * t + 3]) for t in range(5) if r[3 * t] >= 0] for r in rows]


def tables():
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "isaac_ros_nvblox_amd", "csrc")
    return [load_table(os.path.join(d, n)) for n in ("mc_table.inc", "mc_table_r1.inc", "mc_table_r2.inc")]


def edge_faces(e):
    """Cube faces (axis, side) that contain lattice edge e."""
    a, b = CORNERS[EDGES[e][0]], CORNERS[EDGES[e][1]]
    return {(ax, int(a[ax])) for ax in range(3) if a[ax] == b[ax]}


def edge_mid(e):
    return 0.5 * (CORNERS[EDGES[e][0]] + CORNERS[EDGES[e][1]])


def face_segments(tris):
    """Directed triangle edges that lie in a cube face -> {face: [(e_from, e_to), ...]}; interior edges -> directed-use counter."""
    on_face, interior = {}, {}
    for t in tris:
        for k in range(3):
            a, b = t[k], t[(k + 1) % 3]
            common = edge_faces(a) & edge_faces(b)
            if common:
                assert len(common) == 1
                on_face.setdefault(next(iter(common)), []).append((a, b))
            else:
                interior[(a, b)] = interior.get((a, b), 0) + 1
    return on_face, interior


def test_marching_cubes_tables_against_brute_force_topology():
    for rule, table in enumerate(tables()):
        for case in range(256):
            inside = [(case >> i) & 1 for i in range(8)]
            tris = table[case]
            crossed = {e for e, (a, b) in enumerate(EDGES) if inside[a] != inside[b]}
            used = {e for t in tris for e in t}
            assert used == crossed, (rule, case)                       # a vertex on exactly the sign-changing edges
            on_face, interior = face_segments(tris)
            # closed, consistently oriented: every interior edge is used once in each direction
            for (a, b), n in interior.items():
                assert n == 1 and interior.get((b, a), 0) == 1, (rule, case, a, b)
            for face, segs_all in on_face.items():
                ax, side = face
                # (a fan triangulation may put a diagonal -- even a whole flat triangle -- into the face plane: a segment used in
                #  both directions is such a diagonal, not part of the surface's boundary on the face)
                segs = [sg for sg in segs_all if (sg[1], sg[0]) not in segs_all]
                fc = [i for i in range(8) if CORNERS[i][ax] == side]
                n_in = sum(inside[i] for i in fc)
                assert len(segs) == len(set(segs))                     # boundary edges are used once
                # the boundary on a face has one segment per pair of crossings: 1 (two crossings) or 2 (ambiguous face)
                n_cross = sum(1 for e in crossed if face in edge_faces(e))
                assert len(segs) == n_cross // 2 and n_in not in (0, 4), (rule, case, face)
            # orientation: normals point from the inside (negative) corners to the outside
            for t in tris:
                p = [edge_mid(e) for e in t]
                nrm = np.cross(p[1] - p[0], p[2] - p[0])
                if np.linalg.norm(nrm) < 1e-12:
                    continue
                c = sum(p) / 3.0
                # the nearest corner along -normal is inside, along +normal outside (test with the signed corner distances)
                sd = [(np.dot(CORNERS[i] - c, nrm), inside[i]) for i in range(8)]
                neg = [ins for s, ins in sd if s < -1e-9]; pos = [ins for s, ins in sd if s > 1e-9]
                assert (not neg or any(neg)) and (not pos or not all(pos)), (rule, case, t)


def shared_face_segments(table, case, ax, side):
    """Undirected boundary segments of `case` on face (ax, side), as frozensets of edge mid-points projected onto the face."""
    on_face, _ = face_segments(table[case])
    out = set()
    segs_all = on_face.get((ax, side), [])
    for a, b in [sg for sg in segs_all if (sg[1], sg[0]) not in segs_all]:
        pa, pb = np.delete(edge_mid(a), ax), np.delete(edge_mid(b), ax)
        out.add(frozenset([tuple(pa), tuple(pb)]))
    return out


def test_marching_cubes_rules_0_and_1_are_crack_free_rule_2_is_not():
    t = tables()
    cracks = [0, 0, 0]
    # two cubes sharing the x face: cube A corners + cube B corners = 12 lattice points, B's x=0 face = A's x=1 face
    a_face = [i for i in range(8) if CORNERS[i][0] == 1]; b_face = [i for i in range(8) if CORNERS[i][0] == 0]
    pair = {ia: ib for ia in a_face for ib in b_face if tuple(CORNERS[ia][1:]) == tuple(CORNERS[ib][1:])}
    for bits in itertools.product((0, 1), repeat=12):
        sa = list(bits[:8])
        sb = [0] * 8
        free = iter(bits[8:])
        for ib in range(8):
            if ib in b_face:
                ia = [k for k, v in pair.items() if v == ib][0]
                sb[ib] = sa[ia]
            else:
                sb[ib] = next(free)
        ca = sum(v << i for i, v in enumerate(sa)); cb = sum(v << i for i, v in enumerate(sb))
        for rule in range(3):
            if shared_face_segments(t[rule], ca, 0, 1) != shared_face_segments(t[rule], cb, 0, 0):
                cracks[rule] += 1
    assert cracks[0] == 0 and cracks[1] == 0, cracks
    assert cracks[2] > 0, cracks          # complement-symmetric tables disagree on ambiguous faces: the classic table's cracks


def test_oracle_and_product_tables_are_the_same_files():
    """(both trees are written by tools/gen_mc_table.py; the checks above are what makes the content trustworthy)"""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for n in ("mc_table.inc", "mc_table_r1.inc", "mc_table_r2.inc"):
        assert open(os.path.join(root, "oracle", n)).read() == open(os.path.join(root, "isaac_ros_nvblox_amd", "csrc", n)).read()


def test_asin_polynomial_of_the_lidar_model_is_the_fitted_one_and_accurate():
    """csrc/nvbx_lidar_math.h nvbx_asin_small: its coefficients are the ones tools/fit_asin.py produces (the header is not hand-edited),
    and evaluated the way the kernel does (float32, fused multiply-adds) they reproduce asin on |s| <= 0.5 to 5e-8."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fit_asin.py")], capture_output=True, text=True, check=True).stdout
    fitted = [float(v) for v in re.search(r"c0\.\.c4 = \[(.*)\]", out).group(1).split(",")]
    src = open(os.path.join(root, "isaac_ros_nvblox_amd", "csrc", "nvbx_lidar_math.h")).read()
    body = src[src.index("NVBX_HD float nvbx_asin_small"):src.index("NVBX_HD float nvbx_asin_small") + 700]
    in_header = [float(v) for v in re.findall(r"([0-9]\.[0-9]+e-[0-9]+)f", body)]
    assert len(in_header) == 5
    # header order: c4, c3, c2, c1, c0 (Horner from the highest power)
    assert np.allclose(np.float32(in_header[::-1]), np.float32(fitted), rtol=0, atol=0)
    c = np.float32(fitted)
    x = np.linspace(-0.5, 0.5, 400001).astype(np.float32)
    z = x * x
    fma = lambda a, b, cc: (a.astype(np.float64) * b.astype(np.float64) + np.float64(cc)).astype(np.float32)
    p = np.full_like(x, c[4])
    for k in (3, 2, 1, 0):
        p = fma(p, z, c[k])
    y = fma(p * z, x, x)
    assert np.abs(y.astype(np.float64) - np.arcsin(x.astype(np.float64))).max() < 5e-8


# ------------------------------------------------------------------------------------------------ LiDAR measurement model (taps, branches)
def _voxel_centres_near_rays(rng, lidar, img, n, voxel=0.1):
    """voxel-centre-like points: most of them where the map has them -- along the beams up to the measured range (+ a truncation band)
    with a lateral scatter of a few voxels -- and some anywhere in the field of view"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from isaac_ros_nvblox_amd import synthetic as S
    dirs = S.lidar_beam_dirs(lidar)
    rows, cols = img.shape
    k = rng.integers(0, rows, n); j = rng.integers(0, cols, n)
    rng_m = np.where(img[k, j] > 0, img[k, j], 60.0)
    t = rng.uniform(0.3, 1.0, n) * (rng_m + 0.4)
    p = dirs[k, j] * t[:, None] + rng.normal(0.0, 1.0, (n, 3)) * (voxel * np.array([0.3, 1.5, 4.0]))[rng.integers(0, 3, n)][:, None]
    p = (np.floor(p / voxel) + 0.5) * voxel                      # snapped to a voxel grid, like the integrator's inputs
    return p


def test_lidar_measurement_model_against_independent_numpy_restatement():
    """The oracle's lidar_sample (the code the HIP kernel is bit-compared with, sensor model from csrc/nvbx_lidar_math.h) against
    tests/lidar_independent.py (numpy float64, libm): the same rule (none / four-tap bilinear / nearest beam) and the same measured
    range on >= 10^5 voxel centres whose decisions do not hang on the last bits -- clean and noisy range images, three thresholds."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import lidar_independent as LI
    from isaac_ros_nvblox_amd import synthetic as S
    rng = np.random.default_rng(9)
    total = 0; seen = {0: 0, 1: 0, 2: 0}
    for lidar, extent, noise, thr in [(S.SPINNING_LIDAR, 150.0, 0.0, (2.0, 0.5)), (S.SPINNING_LIDAR, 150.0, 0.03, (2.0, 0.5)),
                                      ((256, 32, 0.1, -np.deg2rad(60.0), np.deg2rad(10.0)), 40.0, 0.02, (0.5, 1.5))]:
        sc = S.LidarScene(n_boxes=30, extent=extent)
        T = S.lidar_pose(5)
        img = S.render_lidar(sc, T, lidar, max_range=200.0)
        if noise:
            img = np.where(img > 0, img + rng.normal(0.0, noise, img.shape), 0.0).astype(np.float32)
            img[rng.random(img.shape) < 0.03] = 0.0
        p = oracle.default_params(voxel_size=0.1, lidar_max_integration_distance_m=200.0,
                                  lidar_linear_interpolation_max_allowable_difference_vox=thr[0], lidar_nearest_interpolation_max_allowable_dist_to_ray_vox=thr[1])
        pts = _voxel_centres_near_rays(rng, lidar, img, 250000).astype(np.float32)
        br, ds = oracle.lidar_sample_points(p, lidar, img, pts, 200.0)
        ref = LI.sample(lidar, img, pts.astype(np.float64), thr[0] * 0.1, thr[1] * 0.1, 200.0)
        robust = (ref["margin_px"] > 2e-3) & (ref["margin_m"] > 2e-4)
        assert robust.mean() > 0.9
        assert np.array_equal(br[robust], ref["branch"][robust]), np.nonzero(robust & (br != ref["branch"]))[0][:5]
        got = robust & (br > 0)
        assert np.abs(ds[got] - ref["ds"][got]).max() < 2e-4 * max(1.0, float(ds[got].max()) / 50.0)      # float32 ranges up to 200 m
        total += int(robust.sum())
        for b in (0, 1, 2):
            seen[b] += int((ref["branch"][robust] == b).sum())
    assert total >= 500000 and min(seen.values()) > 20000, (total, seen)


def test_view_calculation_against_float64_geometry(oracle_mod):
    """The blocks in view of a camera frame (oracle: float32 Amanatides-Woo with closed-form crossing parameters -- what the HIP view marking
    reproduces bit for bit, tests/test_gpu_parity.py) against tests/view_independent.py: float64 segment / grid-plane geometry, no stepping.
    Two-sided with a margin for rays that graze a block: crossed over more than 1e-3 of a block => in view; in view => within 1e-3 of a ray."""
    import helpers as H
    import view_independent as V
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    for k, (d, _, T) in enumerate(H.frames(3, cam, stride=23, color=False)):
        # (the last setting: 1 cm voxels = 8 cm blocks, i.e. rays of up to ~90 steps -- the regime of a 200 m LiDAR ray at 0.8 m blocks)
        for sub, maxd, vs in ((4, 7.0, 0.05), (1, 3.0, 0.05), (8, 7.0, 0.01)):
            p = H.copy_params(M.default_params(raycast_subsampling_factor=sub, max_integration_distance_m=maxd, voxel_size=vs), oracle_mod.OrcParams)
            o = oracle_mod.OracleMap(p)
            o.integrate_depth(d, T, cam)
            view = {tuple(int(v) for v in r) for r in np.asarray(o.last_view())}
            org, ends = V.camera_rays(d, T, cam, p.voxel_size, p.truncation_distance_vox, p.max_integration_distance_m, sub)
            bs = p.voxel_size * 8.0
            must = V.blocks_crossed(org, ends, bs, 1e-3)
            assert len(must) > 100 and must <= view, (k, sub, len(must), len(view), sorted(must - view)[:5])
            extra = sorted(view - V.blocks_crossed(org, ends, bs, 0.0))
            assert len(extra) <= len(view) // 50, (len(extra), len(view))          # (grazed blocks only: a few per frame)
            assert all(V.near_some_ray(extra, org, ends, bs, 1e-3)), (k, sub, extra[:5])


def test_closed_form_traversal_against_textbook_accumulation(oracle_mod):
    """The closed-form crossing parameters T_a(k) = fmaf(k, tdelta_a, tmax0_a) are this repository's own definition (kernel and checker changed in the
    same commit).  The checker keeps the textbook Amanatides-Woo accumulation tmax_a += tdelta_a behind a switch (oracle.set_traversal_accumulate):
    the two may differ only where two crossings tie within rounding.  Counted here on camera frames (rays of ~20 steps), on a fine grid (rays of ~90
    steps) and on a 200 m LiDAR scan (rays of up to ~450 steps, where the accumulated rounding error is largest): the symmetric difference of the
    two block sets stays below 1 % (camera) / 2 % (LiDAR) of the view, and every block only one of them names lies within 1e-3 of a block of some
    ray (float64 geometry, tests/view_independent.py)."""
    import helpers as H
    import view_independent as V
    from isaac_ros_nvblox_amd import mapper as M, synthetic as S
    cam = H.SMALL_CAM
    try:
        for k, (d, _, T) in enumerate(H.frames(2, cam, stride=31, color=False)):
            for sub, maxd, vs in ((4, 7.0, 0.05), (8, 7.0, 0.01)):
                p = H.copy_params(M.default_params(raycast_subsampling_factor=sub, max_integration_distance_m=maxd, voxel_size=vs), oracle_mod.OrcParams)
                views = []
                for acc in (0, 1):
                    oracle_mod.set_traversal_accumulate(acc)
                    o = oracle_mod.OracleMap(p); o.integrate_depth(d, T, cam)
                    views.append({tuple(int(v) for v in r) for r in np.asarray(o.last_view())})
                diff = views[0] ^ views[1]
                assert len(views[0]) > 100 and len(diff) <= max(2, len(views[0]) // 100), (k, sub, len(views[0]), len(diff))
                if diff:
                    org, ends = V.camera_rays(d, T, cam, p.voxel_size, p.truncation_distance_vox, p.max_integration_distance_m, sub)
                    assert all(V.near_some_ray(sorted(diff), org, ends, p.voxel_size * 8.0, 1e-3)), sorted(diff)[:5]
        # LiDAR, 200 m
        p = H.copy_params(M.default_params(voxel_size=0.1, lidar_max_integration_distance_m=200.0, raycast_subsampling_factor=2), oracle_mod.OrcParams)
        T = S.lidar_pose(0); img = S.render_lidar(S.LidarScene(), T, S.SPINNING_LIDAR, max_range=200.0)
        views = []
        for acc in (0, 1):
            oracle_mod.set_traversal_accumulate(acc)
            o = oracle_mod.OracleMap(p); o.integrate_lidar_depth(img, T, S.SPINNING_LIDAR)
            views.append({tuple(int(v) for v in r) for r in np.asarray(o.last_view())})
        diff = views[0] ^ views[1]
        assert len(views[0]) > 100000 and len(diff) <= len(views[0]) // 50, (len(views[0]), len(diff))
        print("closed form vs accumulation, 200 m LiDAR scan: %d of %d blocks differ" % (len(diff), len(views[0])))
    finally:
        oracle_mod.set_traversal_accumulate(0)


def test_tsdf_update_rule_against_an_independent_float64_model(oracle_mod):
    """The checker's projective TSDF update (what the HIP integrator is compared with bit for bit) against tests/tsdf_independent.py -- numpy float64, whole
    arrays, no shared code: all six WeightingFunctionType values in both formula sets, two frames from two poses (so the blend with a previous weight and
    the weight clamp are exercised), every voxel of every block the checker allocated whose decisions do not hang on the last bits."""
    import helpers as H
    import tsdf_independent as TI
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    fr = H.frames(2, cam, stride=9, color=False)
    total = 0
    for mode in range(6):
        for variant in (0, 1):
            pg = M.default_params(weighting_mode=mode, tsdf_weighting_variant=variant, max_weight=1.7)
            p = H.copy_params(pg, oracle_mod.OrcParams)
            o = oracle_mod.OracleMap(p)
            model = {}
            for d, _, T in fr:
                o.integrate_depth(d, T, cam)
                for idx in o.block_indices(oracle_mod.L_TSDF):
                    key = tuple(int(v) for v in idx)
                    pd_, pw_ = model.get(key, (np.zeros(512), np.zeros(512)))
                    nd, nw, upd, rob = TI.update_block(pd_, pw_, key, d, T, cam, p)
                    model[key] = (nd, nw)
                    model.setdefault(("robust", key), np.ones(512, bool))
                    model[("robust", key)] &= rob
            n_cmp = 0; n_upd = 0
            for idx in o.block_indices(oracle_mod.L_TSDF):
                key = tuple(int(v) for v in idx)
                b = o.get_block(oracle_mod.L_TSDF, idx)
                ed, ew = model[key]; rob = model[("robust", key)]
                n_cmp += int(rob.sum()); n_upd += int((ew[rob] > 0).sum())
                assert np.abs(b["distance"][rob] - ed[rob]).max(initial=0.0) <= 2e-5, (mode, variant, key)
                assert np.abs(b["weight"][rob] - ew[rob]).max(initial=0.0) <= 2e-5 * max(1.0, float(ew.max())), (mode, variant, key)
            assert n_cmp > 100000 and n_upd > 20000, (mode, variant, n_cmp, n_upd)
            total += n_cmp
    assert total > 1500000


@pytest.mark.parametrize("site_rule", [0, 1])
def test_esdf_column_marking_against_a_numpy_model(oracle_mod, site_rule):
    """The 2-D ESDF's marking rule -- per (x, y) column over the z band [esdf_slice_min_height, esdf_slice_max_height]: observed = any voxel with
    weight >= esdf_min_weight; inside = any observed voxel with distance <= 0; site = any observed voxel with |distance| <= max_site_distance (and, rule 0,
    distance <= 0) -- restated with whole-array numpy on a dense copy of the checker's TSDF layer, against the observed / inside / site flags of the
    checker's ESDF slice plane: after a first update, and after further frames + a second (incremental) update.  No code shared with either side."""
    import helpers as H
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    pg = M.default_params(esdf_site_rule=site_rule)
    p = H.copy_params(pg, oracle_mod.OrcParams)
    o = oracle_mod.OracleMap(p)
    vs = float(p.voxel_size)
    kz_min, kz_max, kz_out = int(np.floor(np.float32(p.esdf_slice_min_height) / np.float32(vs))), int(np.floor(np.float32(p.esdf_slice_max_height) / np.float32(vs))), int(np.floor(np.float32(p.esdf_slice_height) / np.float32(vs)))
    s_max = float(p.esdf_max_site_distance_vox) * vs
    fr = H.frames(6, cam, stride=9, color=False)

    def check(tag):
        ti = o.block_indices(oracle_mod.L_TSDF)
        lo = ti.min(0); hi = ti.max(0)
        nz = kz_max - kz_min + 1
        W, Hh = (hi[0] - lo[0] + 1) * 8, (hi[1] - lo[1] + 1) * 8
        d = np.zeros((W, Hh, nz), np.float32); w = np.zeros((W, Hh, nz), np.float32)
        for i in ti:
            b = o.get_block(oracle_mod.L_TSDF, i).reshape(8, 8, 8)          # [x][y][z]
            for vz in range(8):
                kz = int(i[2]) * 8 + vz
                if kz_min <= kz <= kz_max:
                    xs, ys = (i[0] - lo[0]) * 8, (i[1] - lo[1]) * 8
                    d[xs:xs + 8, ys:ys + 8, kz - kz_min] = b["distance"][:, :, vz]; w[xs:xs + 8, ys:ys + 8, kz - kz_min] = b["weight"][:, :, vz]
        obs_v = w >= np.float32(p.esdf_min_weight)
        in_v = obs_v & (d <= 0)
        site_v = obs_v & (np.abs(d) <= np.float32(s_max)) & ((d <= 0) if site_rule == 0 else True)
        obs, ins, site = obs_v.any(2), in_v.any(2), site_v.any(2)
        n = 0; n_site = 0
        bz_out, vz_out = kz_out // 8, kz_out % 8
        for i in o.block_indices(oracle_mod.L_ESDF):
            assert int(i[2]) == bz_out
            b = o.get_block(oracle_mod.L_ESDF, i).reshape(8, 8, 8)[:, :, vz_out]       # [x][y]
            xs, ys = (i[0] - lo[0]) * 8, (i[1] - lo[1]) * 8
            if xs < 0 or ys < 0 or xs + 8 > W or ys + 8 > Hh:
                assert not b["observed"].any(); continue
            assert np.array_equal(b["observed"].astype(bool), obs[xs:xs + 8, ys:ys + 8]), (tag, i)
            assert np.array_equal(b["is_inside"].astype(bool), ins[xs:xs + 8, ys:ys + 8]), (tag, i)
            assert np.array_equal(b["is_site"].astype(bool), site[xs:xs + 8, ys:ys + 8]), (tag, i)
            n += 64; n_site += int(b["is_site"].sum())
        # every column the model marks observed lies in an ESDF block
        have = np.zeros((W, Hh), bool)
        for i in o.block_indices(oracle_mod.L_ESDF):
            xs, ys = (i[0] - lo[0]) * 8, (i[1] - lo[1]) * 8
            if 0 <= xs < W and 0 <= ys < Hh:
                have[xs:xs + 8, ys:ys + 8] = True
        assert not (obs & ~have).any(), tag
        assert n > 3000 and n_site > 100, (tag, n, n_site)

    for d_, _, T in fr[:3]:
        o.integrate_depth(d_, T, cam)
    o.update_esdf(); check("first update")
    for d_, _, T in fr[3:]:
        o.integrate_depth(d_, T, cam)
    o.update_esdf(); check("incremental update")


def test_colour_voxel_rule_against_a_numpy_model(oracle_mod):
    """The colour integrator's per-voxel rule -- projection, occlusion test against the synthetic depth (bilinear with validity at 1/4 resolution,
    |synthetic - voxel depth| <= truncation distance), bilinear colour, weight-1 blend rounded to u8, weight clamp -- restated with numpy float64 against the
    checker's colour layer over two colour frames (first: from nothing; second: blended), on the blocks the checker selected (last_color_view) and, as a
    negative, on every other allocated block (must stay uncoloured by that frame).  The synthetic depth is taken from the checker (its sphere tracing has an
    analytic check of its own, tests/test_oracle_ground_truth.py); nothing else is shared."""
    import helpers as H
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM; fu, fv, cu, cv, w, h = cam
    pg = M.default_params(max_weight=1.6)
    p = H.copy_params(pg, oracle_mod.OrcParams)
    o = oracle_mod.OracleMap(p)
    vs = float(p.voxel_size); bs = 8 * vs; trunc = float(p.truncation_distance_vox) * vs; f = int(p.sphere_tracing_subsampling)
    fr = H.frames(3, cam, stride=7, color=True)
    for d, _, T in fr:
        o.integrate_depth(d, T, cam)
    lin = np.arange(512); vx, vy, vz = lin // 64, (lin // 8) % 8, lin % 8
    model = {}
    n_cmp = 0; n_col = 0; worst = 0
    for d, rgb, T in fr[:2]:
        o.integrate_color(rgb, T, cam)
        synth = o.synthetic_depth().astype(np.float64); srows, scols = synth.shape
        Tm = np.asarray(T, np.float64); R = Tm[:3, :3]; t = Tm[:3, 3]
        img = rgb.astype(np.float64); rows, cols = img.shape[:2]
        for idx in o.last_color_view():
            key = tuple(int(v) for v in idx)
            c0, w0, rob_acc = model.get(key, (np.zeros((512, 3)), np.zeros(512), np.ones(512, bool)))
            pl = np.stack([idx[0] * bs + vx * vs + vs / 2, idx[1] * bs + vy * vs + vs / 2, idx[2] * bs + vz * vs + vs / 2], 1)
            pc = (pl - t) @ R; z = pc[:, 2]; zs = np.where(z > 0, z, 1.0)
            u = fu * pc[:, 0] / zs + cu; v = fv * pc[:, 1] / zs + cv
            ok = (z > 0) & (u >= 0) & (v >= 0) & (u <= w) & (v <= h) & (z <= float(p.max_integration_distance_m))
            # synthetic depth, bilinear with validity at (u / f, v / f)
            us, vs_ = u / f - 0.5, v / f - 0.5
            x0 = np.floor(us).astype(np.int64); y0 = np.floor(vs_).astype(np.int64)
            inb = (x0 >= 0) & (y0 >= 0) & (x0 + 1 <= scols - 1) & (y0 + 1 <= srows - 1)
            xs = np.clip(x0, 0, scols - 2); ys = np.clip(y0, 0, srows - 2)
            s00, s10, s01, s11 = synth[ys, xs], synth[ys, xs + 1], synth[ys + 1, xs], synth[ys + 1, xs + 1]
            sval = (s00 > 0) & (s10 > 0) & (s01 > 0) & (s11 > 0)
            ax, ay = us - np.floor(us), vs_ - np.floor(vs_)
            sd = (1 - ay) * ((1 - ax) * s00 + ax * s10) + ay * ((1 - ax) * s01 + ax * s11)
            occl_ok = np.abs(sd - z) <= trunc
            # colour, bilinear at (u, v)
            uc, vc = u - 0.5, v - 0.5
            cx0 = np.floor(uc).astype(np.int64); cy0 = np.floor(vc).astype(np.int64)
            cin = (cx0 >= 0) & (cy0 >= 0) & (cx0 + 1 <= cols - 1) & (cy0 + 1 <= rows - 1)
            cxs = np.clip(cx0, 0, cols - 2); cys = np.clip(cy0, 0, rows - 2)
            bx, by = (uc - np.floor(uc))[:, None], (vc - np.floor(vc))[:, None]
            col = (1 - by) * ((1 - bx) * img[cys, cxs] + bx * img[cys, cxs + 1]) + by * ((1 - bx) * img[cys + 1, cxs] + bx * img[cys + 1, cxs + 1])
            upd = ok & inb & sval & occl_ok & cin
            blended = np.floor((c0 * (w0 / (w0 + 1))[:, None] + col * (1 / (w0 + 1))[:, None]) + 0.5).clip(0, 255)
            c1 = np.where(upd[:, None], blended, c0); w1 = np.where(upd, np.minimum(w0 + 1, float(p.max_weight)), w0)
            spread = np.maximum.reduce([s00, s10, s01, s11]) - np.minimum.reduce([s00, s10, s01, s11])
            rob = ((np.minimum.reduce([np.abs(u), np.abs(v), np.abs(u - w), np.abs(v - h)]) > 0.02) & (np.abs(z - float(p.max_integration_distance_m)) > 1e-3) &
                   (np.minimum(np.abs(us - np.round(us)), np.abs(vs_ - np.round(vs_))) > 0.01) & (np.minimum(np.abs(uc - np.round(uc)), np.abs(vc - np.round(vc))) > 0.01) &
                   (np.abs(np.abs(sd - z) - trunc) > 2e-3) & ((spread < 0.3) | ~sval) & (z > 0.05))
            model[key] = (c1, w1, rob_acc & rob)
    for idx in o.block_indices(oracle_mod.L_COLOR):
        key = tuple(int(v) for v in idx)
        b = o.get_block(oracle_mod.L_COLOR, idx)
        if key not in model:
            assert not (b["weight"] > 0).any(); continue
        c1, w1, rob = model[key]
        got = np.stack([b["r"], b["g"], b["b"]], 1).astype(np.int64)
        assert np.array_equal(b["weight"][rob].astype(np.float64), w1[rob]), key
        diff = np.abs(got[rob] - c1[rob]).max(initial=0)
        worst = max(worst, int(diff))
        n_cmp += int(rob.sum()); n_col += int((w1[rob] > 0).sum())
    assert worst <= 1, worst                      # (a blend that lands on x.5 in one arithmetic and just off it in the other: one grey level)
    assert n_cmp > 100000 and n_col > 15000, (n_cmp, n_col)


@pytest.mark.parametrize("exclude,kw", [(False, {}), (True, {}), (False, dict(tsdf_set_free_distance_on_decayed=1, tsdf_decayed_free_distance_vox=3.0)),
                                        (False, dict(decay_deallocate_decayed_blocks=0))], ids=["all", "exclude_last_view", "free_on_decay", "keep_blocks"])
def test_tsdf_decay_rule_against_a_numpy_model(oracle_mod, exclude, kw):
    """decayTsdf / decayTsdfExcludeLastView restated with numpy float32 on a copy of the checker's TSDF layer: weight <- weight * factor; a block none of whose
    weights reaches the threshold is deallocated (unless decay_integrator_deallocate_decayed_blocks is off); with tsdf_set_free_distance_on_decayed an
    OBSERVED voxel that falls below the threshold becomes free instead (distance = tsdf_decayed_free_distance_vox voxels, weight = the threshold); the last
    camera view's blocks are spared on request.  Four calls in a row, block sets and every voxel compared exactly after each."""
    import helpers as H
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    pg = M.default_params(tsdf_decay_factor=0.45, tsdf_decayed_weight_threshold=0.3, **kw)
    p = H.copy_params(pg, oracle_mod.OrcParams)
    o = oracle_mod.OracleMap(p)
    for d, _, T in H.frames(3, cam, stride=13, color=False):
        o.integrate_depth(d, T, cam)
    spared = {tuple(int(v) for v in r) for r in np.asarray(o.last_view())} if exclude else set()
    model = {tuple(int(v) for v in i): (o.get_block(oracle_mod.L_TSDF, i)["distance"].astype(np.float32).copy(), o.get_block(oracle_mod.L_TSDF, i)["weight"].astype(np.float32).copy())
             for i in o.block_indices(oracle_mod.L_TSDF)}
    f = np.float32(p.tsdf_decay_factor); thr = np.float32(p.tsdf_decayed_weight_threshold); free_d = np.float32(p.tsdf_decayed_free_distance_vox) * np.float32(p.voxel_size)
    n_dropped = 0; n_freed_vox = 0
    for call in range(4):
        o.decay_tsdf(exclude)
        for key in list(model):
            if key in spared:
                continue
            d, w0 = model[key]
            w = (w0 * f).astype(np.float32)
            below = w < thr
            alive = bool((~below).any()) or not int(p.decay_deallocate_decayed_blocks)
            if int(p.tsdf_set_free_distance_on_decayed):
                to_free = below & (w0 > 0)
                d = np.where(to_free, free_d, d).astype(np.float32); w = np.where(to_free, thr, w).astype(np.float32); n_freed_vox += int(to_free.sum())
            if alive:
                model[key] = (d, w)
            else:
                del model[key]; n_dropped += 1
        got = {tuple(int(v) for v in i) for i in o.block_indices(oracle_mod.L_TSDF)}
        assert got == set(model), (call, len(got), len(model))
        for key, (d, w) in model.items():
            b = o.get_block(oracle_mod.L_TSDF, np.array(key, np.int32))
            assert np.array_equal(b["weight"], w) and np.array_equal(b["distance"], d), (call, key)
    if kw.get("decay_deallocate_decayed_blocks", 1) and not kw.get("tsdf_set_free_distance_on_decayed"):
        assert n_dropped > 20, n_dropped                  # (weights <= 3 after three frames: 3 * 0.45^3 < 0.3 -- whole blocks go)
    if kw.get("tsdf_set_free_distance_on_decayed"):
        assert n_freed_vox > 1000
    if exclude:
        assert len(spared) > 50 and spared <= set(model)


def test_occupancy_update_and_invalid_depth_decay_against_the_independent_model(oracle_mod):
    """Two more [U] per-voxel rules against tests/tsdf_independent.py (numpy float64, no shared code): the occupancy mapper's log-odds update by region along
    the ray (free / occupied / unobserved, clamp +-10) over three frames, and the TSDF integrator's invalid-depth decay (weight *= factor where the voxel
    projects onto an invalid depth tap) on depth images with holes."""
    import helpers as H
    import tsdf_independent as TI
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    fr = H.frames(3, cam, stride=9, color=False)
    # occupancy
    p = H.copy_params(M.default_params(projective_layer_type=1), oracle_mod.OrcParams)
    o = oracle_mod.OracleMap(p); model = {}; rob_acc = {}
    for d, _, T in fr:
        o.integrate_depth(d, T, cam)
        for idx in np.asarray(o.last_view()):
            key = tuple(int(v) for v in idx)
            new, rob = TI.update_block_occupancy(model.get(key, np.zeros(512)), key, d, T, cam, p)
            model[key] = new; rob_acc[key] = rob_acc.get(key, np.ones(512, bool)) & rob
    n = 0; n_occ = 0
    for idx in o.block_indices(oracle_mod.L_TSDF):
        key = tuple(int(v) for v in idx); b = o.get_block(oracle_mod.L_TSDF, idx)
        rob = rob_acc[key]
        assert np.abs(b["distance"][rob] - model[key][rob]).max(initial=0.0) <= 2e-5, key
        n += int(rob.sum()); n_occ += int((model[key][rob] > 0).sum())
    assert n > 100000 and n_occ > 500, (n, n_occ)
    # invalid-depth decay
    rng = np.random.default_rng(4)
    p = H.copy_params(M.default_params(invalid_depth_decay_factor=0.8), oracle_mod.OrcParams)
    o = oracle_mod.OracleMap(p); model = {}; rob_acc = {}
    n_dec = 0
    for k, (d, _, T) in enumerate(fr):
        d = d.copy()
        if k > 0:
            for _ in range(12):
                r0, c0 = rng.integers(0, d.shape[0] - 12), rng.integers(0, d.shape[1] - 16)
                d[r0:r0 + 12, c0:c0 + 16] = 0.0                                      # holes: the voxels behind them lose confidence
        o.integrate_depth(d, T, cam)
        for idx in o.block_indices(oracle_mod.L_TSDF):
            key = tuple(int(v) for v in idx)
            pd_, pw_ = model.get(key, (np.zeros(512), np.zeros(512)))
            nd, nw, rob = TI.update_block_invalid_decay(pd_, pw_, key, d, T, cam, p)
            n_dec += int(((nw < pw_) & rob).sum())
            model[key] = (nd, nw); rob_acc[key] = rob_acc.get(key, np.ones(512, bool)) & rob
    n = 0
    for idx in o.block_indices(oracle_mod.L_TSDF):
        key = tuple(int(v) for v in idx); b = o.get_block(oracle_mod.L_TSDF, idx); rob = rob_acc[key]
        assert np.abs(b["distance"][rob] - model[key][0][rob]).max(initial=0.0) <= 2e-5, key
        assert np.abs(b["weight"][rob] - model[key][1][rob]).max(initial=0.0) <= 2e-5, key
        n += int(rob.sum())
    assert n > 100000 and n_dec > 2000, (n, n_dec)


def test_radius_clearing_against_a_numpy_model(oracle_mod):
    """clearOutsideRadius: a block goes iff the centre of its cube lies farther than the radius from the centre point -- numpy float32 on the block index list;
    the cleared-block list names exactly the projective blocks that went."""
    import helpers as H
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    p = H.copy_params(M.default_params(), oracle_mod.OrcParams)
    o = oracle_mod.OracleMap(p)
    for d, _, T in H.frames(4, cam, stride=11, color=False):
        o.integrate_depth(d, T, cam)
    o.update_esdf()
    before = o.block_indices(oracle_mod.L_TSDF)
    c = np.array([0.3, -0.2, 1.1], np.float32); r = np.float32(1.9); bs = np.float32(8) * np.float32(p.voxel_size)
    ctr = (before.astype(np.float32) * bs + bs * np.float32(0.5)) - c
    d2 = (ctr[:, 0] * ctr[:, 0] + ctr[:, 1] * ctr[:, 1]) + ctr[:, 2] * ctr[:, 2]
    keep = ~(d2 > r * r)
    o.take_cleared_blocks()
    o.clear_outside_radius(tuple(float(v) for v in c), float(r))
    after = {tuple(int(v) for v in i) for i in o.block_indices(oracle_mod.L_TSDF)}
    assert after == {tuple(int(v) for v in i) for i in before[keep]} and 20 < keep.sum() < len(before) - 20
    gone = {tuple(int(v) for v in i) for i in np.asarray(o.take_cleared_blocks()).reshape(-1, 3)}
    assert gone == {tuple(int(v) for v in i) for i in before[~keep]}


def test_mesh_rules_against_a_table_free_numpy_model(oracle_mod):
    """The mesh integrator's rules that do not depend on the triangle table (that one: the brute-force topology tests above), tests/mesh_independent.py:
    which cubes are meshed (eight corners present with weight >= mesh_min_weight, mixed signs), one welded vertex per crossed lattice edge bordering such a
    cube, at the float32 linear zero crossing, in ascending edge order; every triangle inside one meshed cube of its block and every meshed cube with a
    triangle; triangles facing the positive side; vertex colour = the nearer end voxel's (127 grey without one); normal = the first referencing triangle's."""
    import helpers as H
    import mesh_independent as MI
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    p = H.copy_params(M.default_params(), oracle_mod.OrcParams)
    o = oracle_mod.OracleMap(p)
    for d_, rgb, T in H.frames(5, cam, stride=9, color=True):
        o.integrate_depth(d_, T, cam); o.integrate_color(rgb, T, cam)
    o.update_mesh(full=True)
    vs = float(p.voxel_size)
    ti, lo, d, w, has, col, cw = MI.dense_layers(o, oracle_mod)
    nv = nt = n_col = 0; n_flip = 0
    for i in ti:
        mb = o.mesh_block(i)
        eids, pos, cols, active = MI.expected_block(i, lo, d, w, has, col, cw, vs, float(p.mesh_min_weight))
        if mb is None:
            assert len(eids) == 0; continue
        v, t, nrm, c = mb["vertices"], mb["triangles"], mb["normals"], mb["colors"]
        assert len(v) == len(eids), (tuple(i), len(v), len(eids))
        if len(v) == 0:
            assert len(t) == 0 and not active.any(); continue
        assert np.abs(v - pos).max() <= 1e-6, tuple(i)                         # same vertices, same order
        assert np.array_equal(c[:, :3], cols) and (c[:, 3] == 255).all(), tuple(i)
        cube, inside = MI.triangle_cubes(i, v, t, vs)
        assert inside.all() and cube.min() >= 0 and cube.max() <= 7, tuple(i)
        seen = np.zeros((8, 8, 8), bool); seen[tuple(cube.T)] = True
        assert np.array_equal(seen, active), tuple(i)
        # orientation: the face normal points the way the distance grows (mean gradient over the cube's corners)
        tp = v[t].astype(np.float64); fn = np.cross(tp[:, 1] - tp[:, 0], tp[:, 2] - tp[:, 0])
        o9 = (np.asarray(i) - lo) * 8
        D = d[tuple(slice(int(a), int(a) + 9) for a in o9)].astype(np.float64)
        g = np.stack([(D[1:, :-1, :-1] + D[1:, 1:, :-1] + D[1:, :-1, 1:] + D[1:, 1:, 1:]) - (D[:-1, :-1, :-1] + D[:-1, 1:, :-1] + D[:-1, :-1, 1:] + D[:-1, 1:, 1:]),
                      (D[:-1, 1:, :-1] + D[1:, 1:, :-1] + D[:-1, 1:, 1:] + D[1:, 1:, 1:]) - (D[:-1, :-1, :-1] + D[1:, :-1, :-1] + D[:-1, :-1, 1:] + D[1:, :-1, 1:]),
                      (D[:-1, :-1, 1:] + D[1:, :-1, 1:] + D[:-1, 1:, 1:] + D[1:, 1:, 1:]) - (D[:-1, :-1, :-1] + D[1:, :-1, :-1] + D[:-1, 1:, :-1] + D[1:, 1:, :-1])], -1)
        dots = (fn * g[tuple(cube.T)]).sum(1)
        n_flip += int((dots < 0).sum())
        # normal rule 0: the first triangle that references the vertex
        first = np.full(len(v), -1, np.int64)
        for k in range(len(t) - 1, -1, -1):
            first[t[k]] = k
        assert (first >= 0).all()
        n0 = np.cross(v[t[first, 1]] - v[t[first, 0]], v[t[first, 2]] - v[t[first, 0]]).astype(np.float64)
        ln = np.linalg.norm(n0, axis=1); good = ln > 1e-12
        assert np.abs(nrm[good] - (n0[good] / ln[good, None])).max(initial=0.0) <= 2e-4, tuple(i)
        nv += len(v); nt += len(t); n_col += int((cols != 127).any(1).sum())
    assert nv > 5000 and nt > 5000 and n_col > 1000, (nv, nt, n_col)
    assert n_flip <= 0.002 * nt, (n_flip, nt)          # (a saddle cube's mean gradient can disagree with one of its sheets; measured: 0 of 18 217)


def test_freespace_state_machine_against_a_numpy_model(oracle_mod):
    """The freespace layer's per-voxel state machine (first touch; occupied with / without the 6-neighbourhood; consecutive-occupancy duration with its
    forgiveness gap; reset; promotion to high-confidence freespace) as whole-array numpy over a dense copy of the TSDF layer: eight frames at irregular
    times from a moving camera with an object dropped into some of them, every voxel of every freespace block equal after every frame; then the dynamic
    mask of a frame against the numpy lookup of each pixel's 3-D point."""
    import helpers as H
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    kw = dict(projective_layer_type=2, min_duration_since_occupied_for_freespace_ms=250, max_unobserved_to_keep_consecutive_occupancy_ms=150,
              min_consecutive_occupancy_duration_for_reset_ms=300, check_neighborhood=1)
    p = H.copy_params(M.default_params(**kw), oracle_mod.OrcParams)
    o = oracle_mod.OracleMap(p)
    fr = H.frames(8, cam, stride=3, color=False)
    times = [0, 90, 210, 260, 400, 520, 640, 760]
    NB_ = 40; G = NB_ * 8                                                        # dense window of 40^3 blocks around the origin
    off = np.array([NB_ // 2] * 3) * 8
    init = np.zeros((G, G, G), bool); last = np.zeros((G, G, G), np.int64); dur = np.zeros((G, G, G), np.int64); hc = np.zeros((G, G, G), bool)
    thr = np.float32(p.max_tsdf_distance_for_occupancy_m)
    n_checked = n_occ = n_hc = n_reset = 0
    for k, ((d_, _, T), now) in enumerate(zip(fr, times)):
        d_ = d_.copy()
        if k >= 4:
            d_[40:90, 60:110] = np.minimum(d_[40:90, 60:110], 0.9)             # something standing in what was free space
        o.set_time_ms(now); o.integrate_depth(d_, T, cam)
        dist = np.zeros((G, G, G), np.float32); wgt = np.zeros((G, G, G), np.float32)
        ti = o.block_indices(oracle_mod.L_TSDF)
        assert ti.min() > -NB_ // 2 and ti.max() < NB_ // 2 - 1
        for i in ti:
            b = o.get_block(oracle_mod.L_TSDF, i).reshape(8, 8, 8); s = tuple(slice(int(a) * 8 + int(c), int(a) * 8 + int(c) + 8) for a, c in zip(i, off))
            dist[s] = b["distance"]; wgt[s] = b["weight"]
        view = np.zeros((G, G, G), bool)
        for i in np.asarray(o.last_view()).reshape(-1, 3):
            view[tuple(slice(int(a) * 8 + int(c), int(a) * 8 + int(c) + 8) for a, c in zip(i, off))] = True
        occ_self = (wgt > 0) & (dist < thr)
        occ = occ_self.copy()
        for ax in range(3):
            occ |= np.roll(occ_self, 1, ax) | np.roll(occ_self, -1, ax)         # (the window's rim is empty: nothing wraps)
        first = view & ~init
        init |= first; last[first] = now; dur[first] = 0; hc[first] = bool(p.initialize_to_high_confidence_freespace)
        obs = view & (wgt > 0)
        is_occ = obs & occ
        gap = now - last
        dur = np.where(is_occ, np.where(gap <= int(p.max_unobserved_to_keep_consecutive_occupancy_ms), dur + gap, 0), dur)
        last = np.where(is_occ, now, last)
        reset = is_occ & (dur >= int(p.min_consecutive_occupancy_duration_for_reset_ms))
        n_reset += int((reset & hc).sum())
        hc = np.where(reset, False, hc)
        hc = np.where(obs & ~is_occ & (now - last >= int(p.min_duration_since_occupied_for_freespace_ms)), True, hc)
        for i in o.block_indices(oracle_mod.L_FREESPACE):
            f = o.get_block(oracle_mod.L_FREESPACE, i).reshape(8, 8, 8); s = tuple(slice(int(a) * 8 + int(c), int(a) * 8 + int(c) + 8) for a, c in zip(i, off))
            assert init[s].all(), (k, tuple(i))
            assert np.array_equal(f["is_high_confidence_freespace"].astype(bool), hc[s]), (k, tuple(i))
            assert np.array_equal(f["last_occupied_timestamp_ms"], last[s]) and np.array_equal(f["consecutive_occupancy_duration_ms"], dur[s]), (k, tuple(i))
            n_checked += 512
        assert init.sum() == 512 * len(o.block_indices(oracle_mod.L_FREESPACE))
        n_occ += int(is_occ.sum()); n_hc = int(hc.sum())
    assert n_checked > 10 ** 6 and n_occ > 10 ** 4 and n_hc > 10 ** 4 and n_reset > 100, (n_checked, n_occ, n_hc, n_reset)
    # dynamic mask of one more frame with an object in known free space
    d_, _, T = fr[-1]; d_ = d_.copy(); d_[30:70, 40:100] = np.minimum(d_[30:70, 40:100], 0.8)
    mask = o.detect_dynamics(d_, T, cam, float(p.max_integration_distance_m))
    fu, fv, cu, cv, w_, h_ = cam
    rr, cc = np.mgrid[0:d_.shape[0], 0:d_.shape[1]]
    dd = d_.astype(np.float64)
    pc = np.stack([dd * ((cc + 0.5) - cu) / fu, dd * ((rr + 0.5) - cv) / fv, dd], -1)
    T64 = np.asarray(T, np.float64); pl = pc @ T64[:3, :3].T + T64[:3, 3]
    g = pl / float(p.voxel_size)
    gi = np.floor(g).astype(np.int64) + off
    robust = (np.abs(g - np.round(g)) > 1e-3).all(-1) & (d_ > 0) & (d_ <= float(p.max_integration_distance_m) - 1e-4)
    exp = hc[gi[..., 0], gi[..., 1], gi[..., 2]] & (d_ > 0)
    assert np.array_equal(mask.astype(bool)[robust], exp[robust]) and exp[robust].sum() > 500, int(exp[robust].sum())


def test_shape_clearing_against_a_numpy_model(oracle_mod):
    """clearTsdfInsideShapes: a voxel is reset (distance 0, weight 0) iff its centre lies inside one of the spheres / boxes; every other voxel is untouched --
    numpy float32 over the dense layer."""
    import helpers as H
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    p = H.copy_params(M.default_params(), oracle_mod.OrcParams)
    o = oracle_mod.OracleMap(p)
    for d, _, T in H.frames(3, cam, stride=11, color=False):
        o.integrate_depth(d, T, cam)
    ti = o.block_indices(oracle_mod.L_TSDF)
    before = {tuple(int(v) for v in i): o.get_block(oracle_mod.L_TSDF, i).copy() for i in ti}
    shapes = np.array([[0, 1.5, 1.0, 0.6, 0.8, 0, 0], [1, -3.1, -2.6, -0.1, -1.0, 0.0, 1.0], [0, 9.0, 9.0, 9.0, 0.2, 0, 0]], np.float32)
    n_reported = o.clear_tsdf_inside_shapes([("sphere", tuple(s[1:4]), s[4]) if s[0] == 0 else ("aabb", tuple(s[1:4]), tuple(s[4:7])) for s in shapes])
    vs = np.float32(p.voxel_size); bs = np.float32(8) * vs
    lin = np.arange(512); v3 = np.stack([lin // 64, (lin // 8) % 8, lin % 8], 1).astype(np.float32)
    n = 0
    for key, b0 in before.items():
        c = (np.asarray(key, np.float32) * bs)[None, :] + v3 * vs + vs * np.float32(0.5)
        inside = np.zeros(512, bool)
        for s in shapes:
            if s[0] == 0:
                dd = c - s[1:4]
                inside |= ((dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2]) <= s[4] * s[4]
            else:
                inside |= (c >= s[1:4]).all(1) & (c <= s[4:7]).all(1)
        b1 = o.get_block(oracle_mod.L_TSDF, np.asarray(key, np.int32))
        assert (b1["distance"][inside] == 0).all() and (b1["weight"][inside] == 0).all(), key
        assert np.array_equal(b1["distance"][~inside], b0["distance"][~inside]) and np.array_equal(b1["weight"][~inside], b0["weight"][~inside]), key
        n += int(inside.sum())
    assert n == n_reported and n > 5000, (n, n_reported)
