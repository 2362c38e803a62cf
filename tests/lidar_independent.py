"""An INDEPENDENT restatement of the spinning-LiDAR measurement model -- projection, four-tap bilinear interpolation with its validity /
agreement test, nearest-beam fallback with its distance-to-ray test -- in numpy float64 with libm's arcsin / arctan2 / sin / cos.
Nothing here is shared with the product or with the oracle's C code (which takes the sensor model from csrc/nvbx_lidar_math.h for bit
parity): written from the model's description only --
  beam (k, j) passes through pixel centre (j + 0.5, k + 0.5); elevation = asin(z / r), azimuth = atan2(y, x), equal angular bins
  (/root/reference/nvblox_ros/scripts/calculate_lidar_params.py:50-58; conversions/pointcloud_conversions.cu:78-92);
  [U] interpolateLidarImage: bilinear over the four surrounding beams if all four are valid (> 0) and max - min <= max_diff; otherwise
  the nearest beam (the pixel the point falls into) if it is valid and the point lies within max_ray of that beam's ray; depth = range.
Every decision also reports its MARGIN (distance of the deciding quantity from its threshold), so that a comparison with a float32
implementation can leave out the points whose decision legitimately depends on the last bits."""
import numpy as np


def sample(lidar, img, pts, max_diff, max_ray, max_dist):
    """lidar = (cols, rows, min_range, min_el, max_el); img [rows, cols] ranges (<= 0 / nan = no return); pts [N, 3] float64, sensor frame.
    Returns dict: branch (0 none, 1 bilinear, 2 nearest), ds (measured range), r (the point's own range), margin_px, margin_m."""
    cols, rows, min_range, min_el, max_el = lidar
    cols, rows = int(cols), int(rows)
    img = np.asarray(img, np.float64)
    valid_img = np.isfinite(img) & (img > 0)
    p = np.asarray(pts, np.float64)
    n = len(p)
    r = np.sqrt((p * p).sum(1))
    with np.errstate(invalid="ignore", divide="ignore"):
        el = np.arcsin(np.clip(p[:, 2] / r, -1.0, 1.0))
    az = np.arctan2(p[:, 1], p[:, 0])
    rpp_el = (max_el - min_el) / (rows - 1); rpp_az = 2.0 * np.pi / cols
    u = (az + np.pi) / rpp_az + 0.5
    v = (max_el - el) / rpp_el + 0.5
    u = np.where(u >= cols, u - cols, u)
    big = 1e9
    margin_px = np.full(n, big); margin_m = np.full(n, big)
    ok = (r >= min_range) & (r > 1e-9) & (v >= 0) & (v < rows) & (u >= 0) & (u < cols)
    margin_m = np.minimum(margin_m, np.abs(r - min_range))
    margin_px = np.minimum(margin_px, np.minimum(np.abs(v), np.abs(v - rows)))
    if max_dist > 0:
        ok &= ~(r > max_dist); margin_m = np.minimum(margin_m, np.abs(r - max_dist))
    branch = np.zeros(n, np.int32); ds = np.zeros(n)
    # ---- bilinear over the four beams around the point (pixel-centre referenced coordinates)
    uc, vc = u - 0.5, v - 0.5
    x0 = np.floor(uc).astype(np.int64); y0 = np.floor(vc).astype(np.int64)
    margin_bins = np.minimum(np.minimum(np.abs(uc - np.round(uc)), np.abs(vc - np.round(vc))), np.minimum(np.abs(u - np.round(u)), np.abs(v - np.round(v))))
    margin_px = np.minimum(margin_px, margin_bins)
    inb = ok & (x0 >= 0) & (y0 >= 0) & (x0 + 1 <= cols - 1) & (y0 + 1 <= rows - 1)
    xs = np.clip(x0, 0, cols - 2); ys = np.clip(y0, 0, rows - 2)
    f00, f10, f01, f11 = img[ys, xs], img[ys, xs + 1], img[ys + 1, xs], img[ys + 1, xs + 1]
    allv = inb & valid_img[ys, xs] & valid_img[ys, xs + 1] & valid_img[ys + 1, xs] & valid_img[ys + 1, xs + 1]
    taps = np.stack([f00, f10, f01, f11], 1)
    spread = np.where(allv, np.nanmax(np.where(np.isfinite(taps), taps, -np.inf), 1) - np.nanmin(np.where(np.isfinite(taps), taps, np.inf), 1), np.inf)
    agree = allv & (spread <= max_diff)
    margin_m = np.where(allv, np.minimum(margin_m, np.abs(spread - max_diff)), margin_m)
    ax, ay = uc - x0, vc - y0
    with np.errstate(invalid="ignore"):
        bil = (1 - ay) * ((1 - ax) * f00 + ax * f10) + ay * ((1 - ax) * f01 + ax * f11)
    branch[agree] = 1; ds[agree] = bil[agree]
    # ---- nearest beam
    rest = ok & ~agree
    c = np.floor(u).astype(np.int64); rr = np.floor(v).astype(np.int64)
    inside = rest & (c >= 0) & (rr >= 0) & (c < cols) & (rr < rows)
    cs = np.clip(c, 0, cols - 1); rs = np.clip(rr, 0, rows - 1)
    d = img[rs, cs]
    nv = inside & valid_img[rs, cs]
    bel = max_el - rs * rpp_el; baz = -np.pi + cs * rpp_az
    dirs = np.stack([np.cos(bel) * np.cos(baz), np.cos(bel) * np.sin(baz), np.sin(bel)], 1)
    along = (p * dirs).sum(1)
    perp = np.sqrt(np.maximum(0.0, ((p - along[:, None] * dirs) ** 2).sum(1)))
    near = nv & (perp <= max_ray)
    margin_m = np.where(nv, np.minimum(margin_m, np.abs(perp - max_ray)), margin_m)
    branch[near] = 2; ds[near] = d[near]
    return dict(branch=branch, ds=ds, r=r, margin_px=margin_px, margin_m=margin_m)
