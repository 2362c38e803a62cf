"""Independent correctness bounds for the oracle (SURVEY.md 8c 'consequence'): analytic scene ground truth.

The reference pins none of this, so the oracle is checked against maths it cannot have been fitted to:
TSDF vs the analytic projective distance, ESDF vs a brute-force Euclidean distance transform of the site set,
marching cubes vs the analytic surfaces + mesh topology.
"""
import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S


@pytest.fixture(scope="module")
def mapped(oracle_mod):
    o = oracle_mod.OracleMap(oracle_mod.default_params())
    fr = H.frames(10, H.SMALL_CAM, color=True, stride=20)
    for d, rgb, T in fr:
        o.integrate_depth(d, T, H.SMALL_CAM)
        o.integrate_color(rgb, T, H.SMALL_CAM)
    o.update_esdf()
    o.update_mesh()
    return o, fr


def scene_sdf(p):
    """Euclidean signed distance to the synthetic scene (positive in free space)."""
    sc = S.Scene()
    d_room = np.minimum(p - sc.room_min, sc.room_max - p).min(axis=-1)
    d_sph = np.linalg.norm(p - sc.sphere_c, axis=-1) - sc.sphere_r
    q = np.maximum(sc.box_min - p, p - sc.box_max)
    d_box = np.linalg.norm(np.maximum(q, 0), axis=-1) + np.minimum(q.max(axis=-1), 0)
    return np.minimum(np.minimum(d_room, d_sph), d_box)


def test_tsdf_close_to_analytic_distance(mapped, oracle_mod):
    o, _ = mapped
    vs, trunc = 0.05, 0.2
    errs = []
    for idx in o.block_indices(oracle_mod.L_TSDF):
        b = o.get_block(oracle_mod.L_TSDF, idx).reshape(8, 8, 8)           # [x][y][z]
        gx, gy, gz = np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij")
        p = (np.stack([gx, gy, gz], -1) + idx * 8 + 0.5) * vs
        sel = (b["weight"] >= 1.0) & (np.abs(b["distance"]) < 0.5 * trunc)
        if sel.any():
            # projective distance >= euclidean distance; equal for head-on views. bound: within 2 voxels
            errs.append(np.abs(b["distance"][sel] - scene_sdf(p[sel])))
    errs = np.concatenate(errs)
    assert errs.size > 2000
    assert np.median(errs) < 0.5 * vs
    assert np.percentile(errs, 99) < 2.5 * vs


def test_esdf_equals_bruteforce_edt(mapped, oracle_mod):
    o, _ = mapped
    idx = o.block_indices(oracle_mod.L_ESDF)
    assert len(idx) > 10
    bx0, by0 = idx[:, 0].min(), idx[:, 1].min()
    W, Hh = (idx[:, 0].max() - bx0 + 1) * 8, (idx[:, 1].max() - by0 + 1) * 8
    site = np.zeros((Hh, W), bool); alloc = np.zeros((Hh, W), bool); sq = np.zeros((Hh, W), np.float32)
    par = np.zeros((Hh, W, 2), np.int32)
    vz = 1       # kz_out = floor(0.09/0.05) = 1
    for i in idx:
        b = o.get_block(oracle_mod.L_ESDF, i).reshape(8, 8, 8)[:, :, vz]   # [x][y]
        ys, xs = (i[1] - by0) * 8, (i[0] - bx0) * 8
        site[ys:ys + 8, xs:xs + 8] = b["is_site"].T.astype(bool)
        alloc[ys:ys + 8, xs:xs + 8] = True
        sq[ys:ys + 8, xs:xs + 8] = b["squared_distance_vox"].T
        par[ys:ys + 8, xs:xs + 8, 0] = b["parent_direction"][:, :, 0].T
        par[ys:ys + 8, xs:xs + 8, 1] = b["parent_direction"][:, :, 1].T
    sy, sx = np.nonzero(site)
    assert len(sy) > 50
    yy, xx = np.nonzero(alloc)
    d2 = (yy[:, None] - sy[None, :]) ** 2 + (xx[:, None] - sx[None, :]) ** 2
    best = d2.min(axis=1).astype(np.float32)
    max_sq = np.float32((np.float32(2.0) / np.float32(0.05)) ** 2)
    want = np.where(best <= max_sq, best, max_sq)
    assert np.array_equal(sq[yy, xx], want)
    # parent direction points at a site at exactly that distance
    within = best <= max_sq
    py, px = yy[within] + par[yy[within], xx[within], 1], xx[within] + par[yy[within], xx[within], 0]
    assert site[py, px].all()
    assert np.array_equal((par[yy[within], xx[within]] ** 2).sum(-1).astype(np.float32), best[within])


def test_slice_image_matches_esdf_layer(mapped, oracle_mod):
    o, _ = mapped
    img, aabb = o.esdf_slice_image(1000.0)
    assert img.shape[0] % 8 == 0 and img.shape[1] % 8 == 0
    known = img < 999.0
    assert known.mean() > 0.2
    assert img[known].max() <= 2.0 + 1e-4 and img[known].min() >= -2.0 - 1e-4
    # origin = aabb.min (DistanceMapSlice.msg:13-14); resolution = voxel size
    assert abs(aabb[3] - aabb[0] - img.shape[1] * 0.05) < 1e-4 and abs(aabb[4] - aabb[1] - img.shape[0] * 0.05) < 1e-4


def test_mesh_vertices_on_analytic_surface_and_topology(mapped, oracle_mod):
    o, _ = mapped
    nv = nt = 0
    errs = []
    for idx in o.block_indices(oracle_mod.L_TSDF):
        mb = o.mesh_block(idx)
        if mb is None or len(mb["vertices"]) == 0:
            continue
        v, t, n = mb["vertices"].astype(np.float64), mb["triangles"], mb["normals"]
        nv += len(v); nt += len(t)
        assert t.min() >= 0 and t.max() < len(v)
        errs.append(np.abs(scene_sdf(v)))
        assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-3) or (np.linalg.norm(n, axis=1) < 1e-6).any()
        # welded: no duplicate vertex positions inside a block
        assert len(np.unique(np.round(v / 1e-6).astype(np.int64), axis=0)) == len(v)
        # every vertex is referenced
        assert len(np.unique(t)) == len(v)
    assert nv > 3000 and nt > 3000
    # projective TSDF fusion leaves a few spurious crossings at silhouettes / partially observed voxels; the bulk of the
    # vertices must sit on the analytic surface (measured: median 0.1 mm, p90 1.2 mm, p99 0.1 m)
    errs = np.concatenate(errs)
    assert np.median(errs) < 0.1 * 0.05 and np.percentile(errs, 90) < 0.5 * 0.05 and np.percentile(errs, 98) < 2.5 * 0.05


def test_mc_table_is_watertight_on_a_sphere():
    """Generated 256-case table: closed, consistently oriented surface on a sphere SDF sampled on a lattice."""
    import re
    rows = [list(map(int, re.findall(r"-?\d+", l))) for l in open(H.__file__.replace("tests/helpers.py", "oracle/mc_table.inc")) if l.startswith("{")]
    assert len(rows) == 256
    corners = np.array([(0,0,0),(1,0,0),(1,1,0),(0,1,0),(0,0,1),(1,0,1),(1,1,1),(0,1,1)])
    edges = [(0,1),(1,2),(2,3),(3,0),(4,5),(5,6),(6,7),(7,4),(0,4),(1,5),(2,6),(3,7)]
    N = 12
    g = np.arange(N) - (N - 1) / 2.0
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    sdf = np.sqrt(X ** 2 + Y ** 2 + Z ** 2) - 3.7
    edge_count = {}
    out_ok = 0
    for x in range(N - 1):
        for y in range(N - 1):
            for z in range(N - 1):
                case = 0
                for c, (dx, dy, dz) in enumerate(corners):
                    if sdf[x + dx, y + dy, z + dz] < 0:
                        case |= 1 << c
                row = rows[case]
                for t in range(0, 15, 3):
                    if row[t] < 0:
                        break
                    vid = []
                    pts = []
                    for e in row[t:t + 3]:
                        a, b = corners[edges[e][0]] + (x, y, z), corners[edges[e][1]] + (x, y, z)
                        vid.append(tuple(sorted([tuple(a), tuple(b)])))
                        da, db = sdf[tuple(a)], sdf[tuple(b)]
                        pts.append(a + (b - a) * (da / (da - db)))
                    for k in range(3):
                        key = (vid[k], vid[(k + 1) % 3])
                        edge_count[key] = edge_count.get(key, 0) + 1
                    nrm = np.cross(pts[1] - pts[0], pts[2] - pts[0])
                    cen = (pts[0] + pts[1] + pts[2]) / 3.0 - (N - 1) / 2.0 + np.array([x, y, z]) * 0
                    cen = (pts[0] + pts[1] + pts[2]) / 3.0 - (N - 1) / 2.0
                    if np.dot(nrm, cen) > 0:
                        out_ok += 1
    # every directed edge appears exactly once and its reverse exactly once (closed, oriented 2-manifold)
    assert len(edge_count) > 500
    for (a, b), c in edge_count.items():
        assert c == 1 and edge_count.get((b, a), 0) == 1
    assert out_ok == len(edge_count) // 3      # all normals point to the positive (outside) side


def test_occupancy_oracle_against_analytic_scene(oracle_mod):
    """Occupancy restatement vs maths it cannot have been fitted to: after a few frames, voxels the oracle calls occupied
    (log-odds > 0) lie within the occupied half width (+ a voxel diagonal) of the analytic surface along the viewing ray,
    confidently free voxels lie in free space, and log-odds stay inside the clamp."""
    p = oracle_mod.default_params(projective_layer_type=1, free_region_occupancy_probability=0.3,
                                  occupied_region_occupancy_probability=0.9, unobserved_region_occupancy_probability=0.5,
                                  occupied_region_half_width_m=0.1, max_integration_distance_m=6.0)
    o = oracle_mod.OracleMap(p)
    for d, rgb, T in H.frames(4, H.SMALL_CAM, color=False, stride=5):
        o.integrate_depth(d, T, H.SMALL_CAM)
    occ_d, free_d = [], []
    gx, gy, gz = np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij")
    for idx in o.block_indices(oracle_mod.L_TSDF):
        lo = o.get_block(oracle_mod.L_TSDF, idx)["distance"].reshape(8, 8, 8)
        pts = (np.stack([gx, gy, gz], -1) + idx * 8 + 0.5) * 0.05
        sd = scene_sdf(pts)
        assert np.abs(lo).max() <= 10.0
        occ_d.append(np.abs(sd[lo > 1.0])); free_d.append(sd[lo < -2.0])
    occ_d = np.concatenate(occ_d); free_d = np.concatenate(free_d)
    assert occ_d.size > 2000 and free_d.size > 20000
    # projective distance >= Euclidean distance: an occupied voxel is at most half width (0.1) + voxel diagonal from the surface
    assert np.percentile(occ_d, 99) < 0.1 + 0.05 * np.sqrt(3.0) + 1e-3
    assert (free_d > 0.0).mean() > 0.995
    # ESDF from occupancy: sites exactly where the slice band holds an occupied voxel
    o.update_esdf()
    img, _ = o.esdf_slice_image(1000.0)
    assert (img <= 0.0).sum() > 20 and (img[img < 999.0] <= 2.0 + 1e-4).all()


def test_mask_split_oracle_identity_and_parallax(oracle_mod):
    d = np.full((60, 80), 2.0, np.float32); d[:5] = 0.0
    mask = np.zeros((60, 80), np.uint8); mask[20:40, 30:50] = 7
    cam = (40.0, 40.0, 39.5, 29.5, 80, 60)
    un, ma = oracle_mod.split_depth_by_mask(d, mask, np.eye(4, dtype=np.float32), cam, cam, 0.25)
    assert np.array_equal(ma > 0, (mask != 0) & (d > 0)) and np.array_equal(un > 0, (mask == 0) & (d > 0))
    assert np.array_equal(un[:5], d[:5]) and np.array_equal(ma[:5], np.full((5, 80), -1.0, np.float32))    # invalid depth stays invalid
    # mask camera 0.1 m to the right: at 2 m depth and fu = 40 the mask appears shifted by exactly 2 px in the depth image
    T = np.eye(4, dtype=np.float32); T[0, 3] = -0.1
    un2, ma2 = oracle_mod.split_depth_by_mask(d, mask, T, cam, cam, 0.25)
    want = np.zeros_like(mask); want[20:40, 32:52] = 1
    assert np.array_equal(ma2 > 0, (want != 0) & (d > 0))


def test_freespace_oracle_timeline(oracle_mod):
    """The freespace restatement on a hand-checkable timeline: a wall 2 m in front of a fixed camera.  Free voxels in front of it
    turn into high-confidence freespace exactly min_duration_since_occupied_for_freespace_ms after they were first seen; the
    wall's voxels never do, and their consecutive occupancy duration grows by the frame period."""
    cam = (40.0, 40.0, 39.5, 29.5, 80, 60)
    p = oracle_mod.default_params(projective_layer_type=2, max_integration_distance_m=4.0, min_duration_since_occupied_for_freespace_ms=250,
                                  max_unobserved_to_keep_consecutive_occupancy_ms=200, min_consecutive_occupancy_duration_for_reset_ms=600,
                                  check_neighborhood=0)
    o = oracle_mod.OracleMap(p)
    T = np.array([[0, 0, 1, 0], [-1, 0, 0, 0], [0, -1, 0, 1.0], [0, 0, 0, 1]], np.float32)      # camera z along +x of the layer frame
    depth = np.full((60, 80), 2.0, np.float32)

    def voxel(layer, gx, gy, gz):
        b = o.get_block(layer, np.array([gx >> 3, gy >> 3, gz >> 3], np.int32))
        return None if b is None else b[(gz & 7) + 8 * (gy & 7) + 64 * (gx & 7)]
    free_v, wall_v = (20, 0, 20), (40, 0, 20)            # 1.0 m and 2.0 m along +x at height 1 m (voxel 0.05 m)
    hist = []
    for k in range(9):
        o.set_time_ms(100 * k)
        o.integrate_depth(depth, T, cam)
        f, w = voxel(oracle_mod.L_FREESPACE, *free_v), voxel(oracle_mod.L_FREESPACE, *wall_v)
        hist.append((int(f["is_high_confidence_freespace"]), int(w["is_high_confidence_freespace"]), int(w["consecutive_occupancy_duration_ms"]), int(w["last_occupied_timestamp_ms"])))
    assert [h[0] for h in hist] == [0, 0, 0, 1, 1, 1, 1, 1, 1]          # 250 ms after t = 0 -> the frame at t = 300
    assert all(h[1] == 0 for h in hist)                                    # the wall is never freespace
    assert [h[2] for h in hist] == [0, 100, 200, 300, 400, 500, 600, 700, 800] and hist[-1][3] == 800
    # the mask of dynamic pixels: an object at 1 m in known freespace is dynamic, the wall is not
    obj = depth.copy(); obj[20:40, 30:50] = 1.0
    mask = o.detect_dynamics(obj, T, cam, 4.0)
    assert mask[20:40, 30:50].all() and mask.sum() == 20 * 20
    # standing still for more than the reset duration, its voxels stop being freespace
    for k in range(9, 18):
        o.set_time_ms(100 * k); o.integrate_depth(obj, T, cam)
    assert o.detect_dynamics(obj, T, cam, 4.0).sum() < 0.5 * 20 * 20


def test_remove_small_components_oracle_vs_scipy(oracle_mod):
    import scipy.ndimage as ndi
    rng = np.random.default_rng(11)
    for density in (0.2, 0.42, 0.6):
        mk = (rng.random((90, 130)) < density).astype(np.uint8)
        lab, n = ndi.label(mk, structure=np.ones((3, 3)))
        sizes = ndi.sum(mk, lab, index=np.arange(1, n + 1))
        for thr in (2, 7, 50):
            want = mk.copy(); want[np.isin(lab, np.nonzero(sizes < thr)[0] + 1)] = 0
            assert np.array_equal(oracle_mod.remove_small_components(mk, thr), want), (density, thr)


def test_esdf_3d_oracle_equals_bruteforce(oracle_mod):
    """The 3-D ESDF restatement (x / y / z passes with cut-off) is the exact Euclidean distance transform of its own sites."""
    o = oracle_mod.OracleMap(oracle_mod.default_params(esdf_mode=1, esdf_max_distance_m=0.5, max_integration_distance_m=3.0))
    cam = (40.0, 40.0, 39.5, 29.5, 80, 60)
    for d, rgb, T in H.frames(2, cam, color=False, stride=12):
        o.integrate_depth(d, T, cam)
    o.update_esdf()
    idx = o.block_indices(oracle_mod.L_ESDF)
    assert len(idx) > 100 and H.idx_set(idx) == H.idx_set(o.block_indices(oracle_mod.L_TSDF))
    lo = idx.min(0); dims = (idx.max(0) - lo + 1) * 8
    site = np.zeros(dims, bool); sq = np.full(dims, -1.0, np.float32)
    for i in idx:
        b = o.get_block(oracle_mod.L_ESDF, i).reshape(8, 8, 8); s = (i - lo) * 8
        site[s[0]:s[0] + 8, s[1]:s[1] + 8, s[2]:s[2] + 8] = b["is_site"].astype(bool)
        sq[s[0]:s[0] + 8, s[1]:s[1] + 8, s[2]:s[2] + 8] = b["squared_distance_vox"]
        assert (np.abs(b["parent_direction"]).max() <= 10)
        within = b["squared_distance_vox"] < 100.0
        assert np.array_equal((b["parent_direction"].astype(np.int64) ** 2).sum(-1)[within], b["squared_distance_vox"][within].astype(np.int64))
    pts = np.argwhere(site); assert len(pts) > 500
    sub = np.argwhere(sq >= 0.0)
    sub = sub[np.random.default_rng(0).choice(len(sub), 4000, replace=False)]
    best = np.full(len(sub), np.inf)
    for s0 in range(0, len(pts), 4000):
        best = np.minimum(best, ((sub[:, None, :] - pts[None, s0:s0 + 4000, :]) ** 2).sum(-1).min(1))
    want = np.where(best <= 100.0, best, 100.0).astype(np.float32)
    assert np.array_equal(sq[sub[:, 0], sub[:, 1], sub[:, 2]], want)


def test_colour_close_to_the_analytic_checker_pattern(mapped, oracle_mod):
    """The synthetic scene is painted with a checker of 12.5 cm cells per channel (r / g / b = 64 or 192 by the parity of the cell along x / y / z,
    isaac_ros_nvblox_amd/synthetic.py render): after ten
    colour frames a coloured voxel next to the surface must carry the colour of the checker cell the SURFACE POINT under it lies in -- wherever
    that point is well inside its cell (away from cell borders the bilinear colour tap and the view-to-view blend cannot mix the two greys).
    Independent of the colour integrator's arithmetic: the expectation is the scene's own colour function at the analytic closest surface point."""
    o, _ = mapped
    vs = 0.05
    sc = S.Scene()
    n = 0; bad = 0; n_dark = 0
    for idx in o.block_indices(oracle_mod.L_COLOR):
        c = o.get_block(oracle_mod.L_COLOR, idx).reshape(8, 8, 8)
        t = o.get_block(oracle_mod.L_TSDF, idx).reshape(8, 8, 8)
        gx, gy, gz = np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij")
        p = (np.stack([gx, gy, gz], -1) + idx * 8 + 0.5) * vs
        sel = (c["weight"] > 0.0) & (t["weight"] >= 1.0) & (np.abs(t["distance"]) < 0.6 * vs)
        if not sel.any():
            continue
        ps = p[sel]
        # the closest surface point by the analytic SDF's gradient (central differences in float64), then the checker cell it lies in
        eps = 1e-4
        g = np.stack([(scene_sdf(ps + np.eye(3)[a] * eps) - scene_sdf(ps - np.eye(3)[a] * eps)) / (2 * eps) for a in range(3)], -1)
        q = ps - g * scene_sdf(ps)[:, None]
        cell = 8.0 * q + 0.37
        inside = np.abs(cell - np.round(cell)).min(axis=-1) > 0.3          # > 0.3 cells = 3.75 cm from every cell border (a voxel is 5 cm, a pixel ~2 cm)
        smooth = np.abs(np.linalg.norm(g, axis=-1) - 1.0) < 1e-3            # (not on an edge / corner of the scene, where the closest point jumps)
        want = np.where((np.floor(cell).astype(np.int64) & 1) == 1, 192, 64)
        got = np.stack([c["r"][sel], c["g"][sel], c["b"][sel]], -1).astype(np.int64)
        k = inside & smooth
        n += int(k.sum()); n_dark += int((want[k] == 64).sum())
        bad += int((np.abs(got[k] - want[k]).max(axis=-1) > 40).sum())
    assert n > 1500 and 0.3 * 3 * n < n_dark < 0.7 * 3 * n, (n, n_dark)        # (the checker is per channel and axis: both greys are well represented)
    assert bad <= 0.03 * n, (bad, n)                  # (voxels seen only at grazing angles keep a tap from the neighbouring cell)


def test_synthetic_depth_of_the_sphere_tracer_close_to_the_analytic_depth(oracle_mod):
    """integrateColor's occlusion test sphere-traces the TSDF at a quarter of the resolution: where a ray hits, its depth must be the analytic
    depth of the scene along that pixel's ray (the renderer's own ray casting, float64) to within a voxel or so -- the TSDF is projective and
    the tracer stops within 0.1 voxel of the zero crossing of a nearest-voxel field.  Independent of the tracer's stepping."""
    cam = H.SMALL_CAM
    o = oracle_mod.OracleMap(oracle_mod.default_params())
    fr = H.frames(6, cam, color=True, stride=9)
    for d, rgb, T in fr:
        o.integrate_depth(d, T, cam)
    d, rgb, T = fr[-1]
    o.integrate_color(rgb, T, cam)
    synth = np.asarray(o.synthetic_depth())
    sub = 4
    assert synth.shape == (cam[5] // sub, cam[4] // sub)
    # the tracer's ray of synthetic pixel (r, c) goes through the centre of full-resolution pixel (r * sub, c * sub)
    true = d[::sub, ::sub][: synth.shape[0], : synth.shape[1]].astype(np.float64)
    hit = synth > 0.0
    assert hit.mean() > 0.6
    err = np.abs(synth[hit] - true[hit])
    assert np.median(err) < 0.5 * 0.05 and np.percentile(err, 95) < 2.0 * 0.05, (np.median(err), np.percentile(err, 95))
