"""MultiMapper::ground_plane_estimator() (nvblox_node.cpp:1456,1474; parameters mapper_initialization.cpp:133-153).  [U] restated as TSDF zero
crossings in a height band + a RANSAC plane (csrc/ground.hip <-> oracle).  CPU: the oracle's candidates lie on the analytic floor and the plane
is the floor; the plane fitter against a numpy restatement of the same sampling sequence; GPU: candidates and plane bit-equal to the oracle's."""
import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S

CAM = H.SMALL_CAM


def _feed(m, n=6):
    for d, rgb, T in H.frames(n, CAM, color=False, stride=9, pitch_deg=-35.0):
        m.integrate_depth(d, T, CAM)


def _numpy_ransac(pts, thr, iters, seed):
    """independent restatement of the sampling sequence (LCG 1664525 / 1013904223, index = (state >> 8) mod n) in float32 numpy"""
    f = np.float32
    n = len(pts); state = np.uint64(seed); best = 0; plane = np.array([0, 0, 1, 0], f)
    for _ in range(iters):
        idx = []
        for _k in range(3):
            state = (state * np.uint64(1664525) + np.uint64(1013904223)) & np.uint64(0xFFFFFFFF)
            idx.append(int((int(state) >> 8) % n))
        if len(set(idx)) < 3:
            continue
        a, b, c = (pts[i].astype(f) for i in idx)
        u, v = b - a, c - a
        nrm = np.array([u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]], f)
        ln = np.sqrt(f(f(nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2]))
        if not ln > 1e-12:
            continue
        nrm = (nrm / ln).astype(f)
        if nrm[2] < 0:
            nrm = -nrm
        d = -f(f(nrm[0] * a[0] + nrm[1] * a[1]) + nrm[2] * a[2])
        dist = np.abs((pts[:, 0] * nrm[0] + pts[:, 1] * nrm[1]).astype(f) + pts[:, 2] * nrm[2] + d)
        inl = int((dist <= f(thr)).sum())
        if inl > best:
            best = inl; plane = np.array([nrm[0], nrm[1], nrm[2], d], f)
    return plane, best


def test_oracle_ground_candidates_and_plane_on_the_analytic_floor(oracle_mod):
    o = oracle_mod.OracleMap(oracle_mod.default_params(max_integration_distance_m=5.0))
    _feed(o)
    pts = o.tsdf_zero_crossings(-0.15, 0.15)
    assert len(pts) > 2000 and np.percentile(np.abs(pts[:, 2]), 95) < 0.03     # the room's floor is z = 0: crossings within half a voxel of it (a few
    #   candidates sit where the sphere, 0.1 m above the floor, meets the floor's band: RANSAC's business)
    assert np.array_equal(pts, pts[np.lexsort((pts[:, 2], pts[:, 1], pts[:, 0]))])
    allc = o.tsdf_zero_crossings(-10.0, 10.0)                            # without the band: also the box top / sphere upper half ...
    assert len(allc) > len(pts) and allc[:, 2].max() > 0.5
    plane, inl = oracle_mod.fit_plane_ransac(pts, 0.02, 100, seed=7)
    assert inl > 0.9 * len(pts) and abs(plane[2] - 1.0) < 1e-3 and abs(plane[3]) < 0.02
    # the sampling sequence / arithmetic against an independent numpy restatement (noisy points + outliers, so that triples differ in quality)
    rng = np.random.default_rng(0)
    noisy = np.concatenate([pts[::7] + rng.normal(0, 0.01, pts[::7].shape).astype(np.float32), rng.uniform(-2, 2, (60, 3)).astype(np.float32)])
    p1, n1 = oracle_mod.fit_plane_ransac(noisy, 0.02, 60, seed=3); p2, n2 = _numpy_ransac(noisy, 0.02, 60, 3)
    assert n1 == n2 and np.allclose(p1, p2, atol=1e-6)
    assert oracle_mod.fit_plane_ransac(pts[:2], 0.02, 10)[1] == 0        # fewer than three points: no plane


@pytest.mark.gpu
def test_ground_plane_parity(oracle_mod, hip_lib):
    from test_gpu_parity import make_pair
    M, g, o = make_pair(oracle_mod, max_integration_distance_m=5.0)
    _feed(g); _feed(o)
    for band in ((-0.15, 0.15), (-10.0, 10.0), (0.3, 1.2)):
        pg = g.tsdf_zero_crossings(*band); po = o.tsdf_zero_crossings(*band)
        assert pg.shape == po.shape and np.array_equal(pg, po), band
    pg = g.tsdf_zero_crossings(-0.15, 0.15)
    assert len(pg) > 2000
    rng = np.random.default_rng(1)
    noisy = np.concatenate([pg[::5] + rng.normal(0, 0.01, pg[::5].shape).astype(np.float32), rng.uniform(-2, 2, (80, 3)).astype(np.float32)])
    for seed in (1, 2, 9):
        a, na = g.fit_plane_ransac(noisy, 0.02, 80, seed); b, nb = oracle_mod.fit_plane_ransac(noisy, 0.02, 80, seed)
        assert na == nb and np.array_equal(a, b)
    # decay to nothing: no candidates, no plane
    g2 = M.Mapper(M.default_params(projective_layer_type=1), block_capacity=1 << 12)
    assert len(g2.tsdf_zero_crossings(-1.0, 1.0)) == 0


def _plane_params(M, plane, above, thick, **kw):
    p = M.default_params(esdf_use_ground_plane=1, slice_height_above_plane_m=above, slice_height_thickness_m=thick, **kw)
    for i in range(4):
        p.esdf_ground_plane[i] = float(plane[i])
    return p


@pytest.mark.gpu
def test_esdf_slice_follows_the_ground_plane(oracle_mod, hip_lib):
    """[U] ground-plane-relative 2-D slice (mapper_initialization.cpp:136,257-260: slice_height_above_plane_m / slice_height_thickness_m with
    multi_mapper.experimental_use_ground_plane_estimation): (1) over the horizontal plane z = 0 the band [above, above + thickness] IS the fixed band
    [esdf_slice_min_height, esdf_slice_max_height] of the same heights -- identical ESDF layer; (2) over a tilted plane every column looks at its own
    band: HIP == checker on every ESDF voxel over incremental updates with a decay in between, and the layer differs from the fixed-height one."""
    from isaac_ros_nvblox_amd import mapper as M
    from test_gpu_parity import compare_layer
    cam = H.SMALL_CAM
    fr = H.frames(10, cam, stride=9, color=False)
    fields = ("squared_distance_vox", "parent_direction", "is_inside", "observed", "is_site")
    fixed = M.Mapper(M.default_params(esdf_slice_min_height=0.09, esdf_slice_max_height=0.65), block_capacity=1 << 13)
    flat = M.Mapper(_plane_params(M, (0.0, 0.0, 1.0, 0.0), 0.09, 0.56), block_capacity=1 << 13)
    tilt = (0.06, -0.04, 1.0, -0.12)
    n = float(np.linalg.norm(tilt[:3])); tilt = tuple(v / n for v in tilt)
    pt = _plane_params(M, tilt, 0.05, 0.5, tsdf_decay_factor=0.6, tsdf_decayed_weight_threshold=0.3)
    gt = M.Mapper(pt, block_capacity=1 << 13); ot = oracle_mod.OracleMap(H.copy_params(pt, oracle_mod.OrcParams))
    for k, (d, _, T) in enumerate(fr):
        for m_ in (fixed, flat, gt, ot):
            m_.integrate_depth(d, T, cam)
        if k % 3 == 2:
            for m_ in (fixed, flat, gt, ot):
                m_.update_esdf()
        if k == 6:
            gt.decay_tsdf(True); ot.decay_tsdf(True)          # deallocations in plane mode: the columns of freed blocks are re-marked
    for m_ in (fixed, flat, gt, ot):
        m_.update_esdf()
    i0, i1 = fixed.block_indices(M.LAYER_ESDF), flat.block_indices(M.LAYER_ESDF)
    assert np.array_equal(i0, i1) and len(i0) > 30
    b0, _ = fixed.get_blocks(M.LAYER_ESDF, i0); b1, _ = flat.get_blocks(M.LAYER_ESDF, i0)
    for f in fields:
        assert np.array_equal(b0[f], b1[f]), f
    nblk, _ = compare_layer(M, gt, ot, M.LAYER_ESDF, oracle_mod.L_ESDF, fields_exact=fields)
    assert nblk > 30
    sg, _ = gt.esdf_slice_image(); so, _ = ot.esdf_slice_image(); sf, _ = fixed.esdf_slice_image()
    assert sg.shape == so.shape and np.array_equal(sg, so)
    assert sg.shape != sf.shape or not np.array_equal(sg, sf)               # the tilted band sees other voxels than the fixed one
