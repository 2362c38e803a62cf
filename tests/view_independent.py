"""Independent float64 model of the view calculation (ViewCalculator::getBlocksInImageViewRaycast as DESIGN.md 3 restates it): which blocks does
the segment sensor origin -> (depth + truncation distance, capped) of every sub-sampled pixel's ray pass through?  Pure geometry in numpy
float64 -- the segment's parameter values at the block-grid planes, sorted; the block at the midpoint of every interval -- no stepping, no
Amanatides-Woo, no float32: it shares no code and no arithmetic with oracle/nvblox_oracle.c raycast_blocks or csrc/tsdf.hip (whose closed-form
crossing parameters T_a(k) = fmaf(k, tdelta_a, tmax0_a) are a definition of THIS round, so a third opinion is due).

Float32 stepping and float64 geometry may disagree on blocks a ray only grazes (a corner within rounding error), so the comparison is two-sided
with a margin:  (1) every block a segment crosses over a length > `tol` blocks must be in the product's view;  (2) every block of the product's
view must lie within `tol` blocks of some ray's segment (max-norm)."""
import numpy as np


def camera_rays(depth, T_L_C, cam, voxel_size, trunc_vox, max_dist, subsample):
    """Segments [origin, end] (float64) of the sub-sampled rays with a valid depth: pixel (min(i f, rows - 1), min(j f, cols - 1)), through the
    pixel centre, end at z-depth min(depth + truncation, max integration distance)."""
    fu, fv, cu, cv, cols, rows = cam
    depth = np.asarray(depth, np.float64)
    T = np.asarray(T_L_C, np.float64)
    f = max(1, int(subsample))
    pr = np.minimum(np.arange(0, rows + f - 1, f), rows - 1); pc = np.minimum(np.arange(0, cols + f - 1, f), cols - 1)
    R, C = np.meshgrid(pr, pc, indexing="ij")
    d = depth[R, C]
    ok = d > 0.0
    de = d[ok] + trunc_vox * voxel_size
    if max_dist > 0.0:
        de = np.minimum(de, max_dist)
    rx = ((C[ok] + 0.5) - cu) / fu; ry = ((R[ok] + 0.5) - cv) / fv
    p_c = np.stack([de * rx, de * ry, de], 1)
    ends = p_c @ T[:3, :3].T + T[:3, 3]
    org = np.broadcast_to(T[:3, 3], ends.shape).copy()
    return org, ends


def blocks_crossed(org, ends, block_size, tol):
    """Set of block indices each segment crosses over a length of more than `tol` (in blocks, along the segment)."""
    out = set()
    o = org / block_size; e = ends / block_size
    for a, b in zip(o, e):
        d = b - a
        ts = [0.0, 1.0]
        for ax in range(3):
            if d[ax] == 0.0:
                continue
            lo, hi = sorted((a[ax], b[ax]))
            planes = np.arange(np.ceil(lo), np.floor(hi) + 1.0)
            ts.extend(((planes - a[ax]) / d[ax]).tolist())
        ts = np.unique(np.clip(np.asarray(ts), 0.0, 1.0))
        length = np.linalg.norm(d)
        seg = np.diff(ts) * length
        mid = a[None, :] + ((ts[:-1] + ts[1:]) * 0.5)[:, None] * d[None, :]
        idx = np.floor(mid).astype(np.int64)
        for k in np.nonzero(seg > tol)[0]:
            out.add(tuple(idx[k]))
    return out


def near_some_ray(blocks, org, ends, block_size, tol):
    """For each block index: does some segment pass within `tol` blocks (max-norm) of its cube?  Slab test against the cube grown by tol."""
    o = org / block_size; d = ends / block_size - o
    res = []
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.where(d != 0.0, 1.0 / d, np.inf)
        for b in blocks:
            lo = np.asarray(b, np.float64) - tol; hi = np.asarray(b, np.float64) + 1.0 + tol
            t1 = (lo - o) * inv; t2 = (hi - o) * inv
            par = d == 0.0                                  # parallel axes: inside the slab or never
            inside = (o >= lo) & (o <= hi)
            tn = np.where(par, np.where(inside, -np.inf, np.inf), np.minimum(t1, t2))
            tf = np.where(par, np.where(inside, np.inf, -np.inf), np.maximum(t1, t2))
            t_in = np.maximum(tn.max(1), 0.0); t_out = np.minimum(tf.min(1), 1.0)
            res.append(bool(np.any(t_in <= t_out)))
    return res
