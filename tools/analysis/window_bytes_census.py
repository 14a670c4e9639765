#!/usr/bin/env python3
"""CPU census (float64 geometry, no GPU): LDS-window bytes the in-view blocks of a rectilinear view rendered into a panorama
(BASELINE configs[3], 4096^2) would stage under different window shapes and buffer sizes — whole 16 x 16 blocks, half blocks
(16 x 8), passes (16 x 4) — against the 16 taps per pixel of the per-pixel path.  usage: window_bytes_census.py [focal] [size]"""
import sys
import numpy as np

focal = float(sys.argv[1]) if len(sys.argv) > 1 else 18.0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
W = H = N
x = (np.arange(W) + 0.5) - W * 0.5
y = (np.arange(H) + 0.5) - H * 0.5
lon = ((x / W) + 0.5) * 2 * np.pi - np.pi
lat = ((y / H) + 0.5) * np.pi - np.pi / 2
vx = np.sin(lon)[None, :] * np.ones((H, 1))
vz = -np.cos(lon)[None, :] * np.ones((H, 1))
vy = np.sin(lat)[:, None] * np.ones((1, W))
px = vx / -vz
py = vy / -vz
sx = px * N / 36.0 * focal - 0.5 + N * 0.5
sy = py * N / 36.0 * focal - 0.5 + N * 0.5
inside = (sx >= 1) & (sx < N - 2) & (sy >= 1) & (sy < N - 2)


def windows(rows):  # windows of 16-wide, `rows`-high pixel groups that lie in view whole: (slots, width)
    out = []
    for by in range(0, H, rows):
        ins = inside[by:by + rows].reshape(rows, W // 16, 16).all(axis=(0, 2))
        fx = np.floor(sx[by:by + rows]).reshape(rows, W // 16, 16)
        fy = np.floor(sy[by:by + rows]).reshape(rows, W // 16, 16)
        bw = fx.max(axis=(0, 2)) - fx.min(axis=(0, 2)) + 4
        bh = fy.max(axis=(0, 2)) - fy.min(axis=(0, 2)) + 4
        out.append(np.stack([bw[ins], bh[ins]], axis=1))
    return np.concatenate(out)


total_px = inside.sum()
print(f"focal {focal}: {100.0 * total_px / (W * H):.1f} % of the pixels in view")
res = {}
for name, rows in (("block 16x16", 16), ("half 16x8", 8), ("pass 16x4", 4)):
    w = windows(rows)
    slots = (w[:, 0].astype(np.int64) | 1) * w[:, 1]
    res[name] = (w, slots, rows)
    pct = np.percentile(slots, [10, 50, 90, 99])
    print(f"{name:12s} groups {len(slots):7d}  slots/pixel mean {slots.sum() / (len(slots) * 16 * rows):6.2f}  slots p10/50/90/99 {pct}  width p50/99 {np.percentile(w[:, 0], [50, 99])}")
for cap in (640, 1280, 1536, 2048, 2560, 3072, 4096):
    for maxw in (64, 128):
        line = f"cap {cap:5d} slots ({cap * 16 // 1024:3d} KiB) maxw {maxw:3d}:"
        for name in ("block 16x16", "half 16x8", "pass 16x4"):
            w, slots, rows = res[name]
            fits = (slots <= cap) & (w[:, 0] <= maxw)
            # pixels whose group fits stage the window, the rest gather 16 slots per pixel
            staged = slots[fits].sum() + (~fits).sum() * 16 * rows * 16
            line += f"  {name}: fits {100.0 * fits.mean():5.1f} %, {staged / (len(slots) * 16 * rows):6.2f} slots/px"
        print(line)

# the cascade: a block stages its whole window if that fits; else each half stages its own if that fits; else each pass; else
# the pass gathers 16 slots per pixel
print("cascade block -> half -> pass -> gather:")
fxa = np.floor(sx)
fya = np.floor(sy)


def ext(a, rows):  # (H / rows, W / 16): max and min over groups
    r = a.reshape(H // rows, rows, W // 16, 16)
    return r.max(axis=(1, 3)), r.min(axis=(1, 3))


def group_slots(rows):
    mx, mn = ext(fxa, rows)
    my, ny = ext(fya, rows)
    bw, bh = mx - mn + 4, my - ny + 4
    ins = inside.reshape(H // rows, rows, W // 16, 16).all(axis=(1, 3))
    return (bw.astype(np.int64) | 1) * bh.astype(np.int64), bw, ins


s16, w16, in16 = group_slots(16)
s8, w8, in8 = group_slots(8)
s4, w4, in4 = group_slots(4)
for cap in (1280, 1536, 2048, 2560):
    for maxw in (64, 128):
        ok16 = in16 & (s16 <= cap) & (w16 <= maxw)
        up16 = np.repeat(ok16, 2, axis=0)
        ok8 = in8 & ~up16 & (s8 <= cap) & (w8 <= maxw)
        up8 = np.repeat(up16 | ok8, 2, axis=0)
        ok4 = in4 & ~up8 & (s4 <= cap) & (w4 <= maxw)
        gather = in4 & ~up8 & ~ok4
        staged = s16[ok16].sum() + s8[ok8].sum() + s4[ok4].sum() + gather.sum() * 64 * 16
        px = in4.sum() * 64
        print(f"  cap {cap} maxw {maxw}: whole {100.0 * ok16.sum() * 256 / px:5.1f} % half {100.0 * ok8.sum() * 128 / px:5.1f} % pass {100.0 * ok4.sum() * 64 / px:5.1f} % "
              f"gather {100.0 * gather.sum() * 64 / px:5.1f} % of the in-view pixels -> {staged / px:6.2f} slots/px")
