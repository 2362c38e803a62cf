#!/usr/bin/env python3
"""Coefficients of nvbx_asin_small (csrc/nvbx_lidar_math.h): asin(s) = s + s^3 P(s^2) on |s| <= 0.51, P of degree 4, fitted by
Lawson-reweighted least squares on Chebyshev nodes (a minimax approximation of the absolute error), then checked in float32
arithmetic with fused multiply-adds as the kernel evaluates it.  Prints the float32 coefficients c0..c4 and both error maxima."""
import numpy as np

ZMAX, DEG = 0.2601, 4
k = np.arange(2000)
z = (np.cos(np.pi * (k + 0.5) / 2000) * 0.5 + 0.5) * ZMAX
x = np.sqrt(z)
g = np.where(x > 1e-4, (np.arcsin(x) - x) / np.maximum(x, 1e-300) ** 3, 1 / 6 + 3 / 40 * z)
w = x ** 3
A = np.vander(z, DEG + 1, increasing=True)
lw = np.ones_like(z)
for _ in range(200):
    c = np.linalg.lstsq(A * (w * np.sqrt(lw))[:, None], g * w * np.sqrt(lw), rcond=None)[0]
    err = np.abs((A @ c - g) * w)
    lw = lw * (err / err.max() + 1e-3); lw /= lw.sum() / len(lw)
xs = np.linspace(0, 0.51, 2000001)
e64 = np.abs(xs + xs ** 3 * np.polyval(c[::-1], xs * xs) - np.arcsin(xs)).max()
c32 = c.astype(np.float32); x32 = xs.astype(np.float32); z32 = x32 * x32
fma = lambda a, b, cc: (a.astype(np.float64) * b.astype(np.float64) + np.float64(cc)).astype(np.float32)
acc = np.full_like(x32, c32[-1])
for cc in c32[-2::-1]:
    acc = fma(acc, z32, cc)
r = fma(acc * z32, x32, x32)
e32 = np.abs(r.astype(np.float64) - np.arcsin(x32.astype(np.float64))).max()
print("c0..c4 =", [float(v) for v in c32])
print("max |err| float64 %.3g, float32+fma %.3g" % (e64, e32))
