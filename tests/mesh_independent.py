"""Table-free model of the mesh integrator's rules, for tests/test_independent_checks.py (numpy; shares no code with the checker or the product).

What a marching-cubes mesh of one block must be, whatever the triangle table: one vertex on every lattice edge whose end voxels differ in sign and that
borders a cube of the block with eight observed corners; the vertex sits at the linear zero crossing; vertices in ascending edge order; every triangle
lives inside one such cube; each such cube has at least one; the colour is the nearer end voxel's; (normal rule 0) the normal is the first referencing
triangle's."""
import numpy as np

AX = np.eye(3, dtype=np.int64)


def dense_layers(o, oracle_mod, with_color=True):
    ti = o.block_indices(oracle_mod.L_TSDF)
    lo = ti.min(0); hi = ti.max(0) + 1                       # (+1: the +x/+y/+z neighbours of the outermost blocks)
    shp = tuple(((hi - lo + 1) * 8).tolist())
    d = np.zeros(shp, np.float32); w = np.zeros(shp, np.float32); has = np.zeros(shp, bool)
    col = np.zeros(shp + (3,), np.uint8); cw = np.zeros(shp, np.float32)
    for i in ti:
        s = tuple(slice(int(a) * 8, int(a) * 8 + 8) for a in (i - lo))
        b = o.get_block(oracle_mod.L_TSDF, i).reshape(8, 8, 8)
        d[s] = b["distance"]; w[s] = b["weight"]; has[s] = True
        c = o.get_block(oracle_mod.L_COLOR, i) if with_color else None
        if c is not None:
            c = c.reshape(8, 8, 8); cw[s] = c["weight"]
            for k, ch in enumerate("rgb"):
                col[s + (k,)] = c[ch]
    return ti, lo, d, w, has, col, cw


def expected_block(i, lo, d, w, has, col, cw, vs, min_weight):
    """-> (edge ids ascending, positions f32 [n,3], colours u8 [n,3], active cubes bool [8,8,8])"""
    o = (np.asarray(i) - lo) * 8
    s9 = tuple(slice(int(a), int(a) + 9) for a in o)
    D, W, Hs, Cc, Cw = d[s9], w[s9], has[s9], col[s9], cw[s9]
    ok = Hs & (W >= np.float32(min_weight)); neg = D < 0
    cube_ok = np.ones((8, 8, 8), bool); any_neg = np.zeros((8, 8, 8), bool); all_neg = np.ones((8, 8, 8), bool)
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                sl = (slice(dx, dx + 8), slice(dy, dy + 8), slice(dz, dz + 8))
                cube_ok &= ok[sl]; any_neg |= neg[sl]; all_neg &= neg[sl]
    active = cube_ok & any_neg & ~all_neg
    eids, pos, cols = [], [], []
    bs = np.float32(8) * np.float32(vs)
    for axis in range(3):
        # lattice edges (l, axis), l[axis] <= 7; the cubes around it: l - {0,1} in the two other axes
        n_l = [9, 9, 9]; n_l[axis] = 8
        a_sl = [slice(0, 9)] * 3; a_sl[axis] = slice(0, 8); b_sl = list(a_sl); b_sl[axis] = slice(1, 9)
        crossed = neg[tuple(a_sl)] != neg[tuple(b_sl)]
        touched = np.zeros(n_l, bool)
        others = [a for a in range(3) if a != axis]
        for s0 in (0, 1):
            for s1 in (0, 1):
                dst = [slice(0, 8)] * 3; dst[others[0]] = slice(s0, s0 + 8); dst[others[1]] = slice(s1, s1 + 8)
                touched[tuple(dst)] |= active            # the cube at c borders the edges at c + {0,1} in the two other axes
        L = np.argwhere(crossed & touched)
        if len(L) == 0:
            continue
        la = tuple(L.T); lb = tuple((L + AX[axis]).T)
        da, db = D[la], D[lb]
        t = da / (da - db)
        p = (np.asarray(i, np.float32)[None, :] * bs + L.astype(np.float32) * np.float32(vs)) + np.float32(vs) * np.float32(0.5)
        p[:, axis] = p[:, axis] + t * np.float32(vs)
        near_a = t < np.float32(0.5)
        c = np.where(near_a[:, None], Cc[la], Cc[lb]); cwt = np.where(near_a, Cw[la], Cw[lb])
        c = np.where((cwt > 0)[:, None], c, np.uint8(127))
        eids.append(((L[:, 0] * 9 + L[:, 1]) * 9 + L[:, 2]) * 3 + axis); pos.append(p); cols.append(c)
    if not eids:
        return np.zeros(0, np.int64), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8), active
    eids = np.concatenate(eids); order = np.argsort(eids)
    return eids[order], np.concatenate(pos)[order], np.concatenate(cols)[order], active


def triangle_cubes(i, v, tri, vs):
    """the cube each triangle lies in (lattice units of block i), and whether all three vertices are on that cube"""
    bs = 8.0 * vs
    p = (v.astype(np.float64) - (np.asarray(i, np.float64) * bs + vs / 2)) / vs
    tp = p[tri]                                      # [nt, 3 vertices, 3]
    c = np.floor(tp.mean(1) + 1e-9).astype(np.int64)
    inside = ((tp >= c[:, None, :] - 1e-4) & (tp <= c[:, None, :] + 1 + 1e-4)).all((1, 2))
    return c, inside
