#!/usr/bin/env python3
"""CPU census (float64 geometry): the source bytes a bicubic mapping MUST read — distinct texels under the 4 x 4 footprints of
all output pixels, and distinct 128-byte lines holding them — to hold the PMC read bytes against (equirect ->
fisheye bicubic rotated reads 301 MB where 'the view covers ~53 % of a 268 MB source').  usage: compulsory_reads_census.py"""
import numpy as np

N, C = 4096, 4
deg = (30.0, -15.0, 5.0)


def rot_matrix(pan, pitch, roll):
    cp, sp = np.cos(pan), np.sin(pan)
    ct, st = np.cos(pitch), np.sin(pitch)
    cr, sr = np.cos(roll), np.sin(roll)
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    rx = np.array([[1, 0, 0], [0, ct, -st], [0, st, ct]])
    rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    return ry @ (rx @ rz)


R = rot_matrix(*[d * np.pi / 180 for d in deg])
touched = np.zeros((N, N), dtype=bool)
rows_per = 256
for y0 in range(0, N, rows_per):
    cy = (np.arange(y0, y0 + rows_per) + 0.5 - N * 0.5)[:, None] * np.ones((1, N))
    cx = (np.arange(N) + 0.5 - N * 0.5)[None, :] * np.ones((rows_per, 1))
    # equidistant_to_vec (src/reproject.cpp:171-186): fov pi, sensor 36
    r_px = np.sqrt(cx * cx + cy * cy)
    theta = r_px / N * 36.0 / (36.0 / np.pi)
    s = np.sin(theta) / r_px
    v = np.stack([s * cx, s * cy, np.cos(theta)])
    n = np.tensordot(R, v, axes=1)
    # vec_to_equirectangular (:259-271), full panorama
    th = -np.arctan2(-n[0], -n[2])
    ph = np.arcsin(n[1] / np.sqrt((n * n).sum(axis=0)))
    sx = ((th + np.pi) / (2 * np.pi) - 0.5) * N - 0.5 + N * 0.5
    sy = ((ph + np.pi / 2) / np.pi - 0.5) * N - 0.5 + N * 0.5
    ix, iy = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    for dy in (-1, 0, 1, 2):
        yy = np.clip(iy + dy, 0, N - 1)
        for dx in (-1, 0, 1, 2):
            touched[yy, (ix + dx) % N] = True
texels = touched.sum()
lines = touched.reshape(N, N // 8, 8).any(axis=2).sum()  # 128-byte lines of 8 RGBA texels
print(f"equirect -> fisheye(180) bicubic, rotation {deg}, {N}^2 RGBA:")
print(f"  texels under some footprint: {100.0 * texels / (N * N):.1f} % of the source = {texels * 16 / 1e6:.0f} MB")
print(f"  128-byte lines holding them: {100.0 * lines / (N * N // 8):.1f} % = {lines * 128 / 1e6:.0f} MB (the least the HBM interface can deliver)")
