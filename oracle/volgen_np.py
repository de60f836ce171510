"""TEST INFRASTRUCTURE (not shipped, never imported by the product): plain numpy /
Python-loop restatements of the reference's volume producers, written to follow the
Clojure source line by line so that the device kernels (csrc/rm_volgen.hip) can be
checked against them bit for bit.

Parity status: **unpinned** against the reference itself -- the Clojure host cannot run
in this image (no JVM) and the reference has no tests or fixtures for these functions.
The pin is the source text cited per function and hand-checked cases in
tests/test_volgen_oracle.py.
"""
import math

import numpy as np


def make_terrain(rx, ry, rz):
    """generators.clj:44-60, loops as written (second doseq overrides the first)."""
    vox = np.zeros(rx * ry * rz, dtype=np.uint8)
    rxy = rx * ry
    for z in range(4):                                   # :48  (doseq [z (range 4) y (range (int (* ry 0.666))) x (range rx)]
        for y in range(int(ry * 0.666)):
            for x in range(rx):
                vox[z * rxy + y * rx + x] = 64           # :49
                i2 = x * rxy + y * rx + (rx - z - 1)     # :50  (madd x rxy y rx (dec (- rx z)))
                if i2 < vox.size:                        # (the JVM would throw beyond the array: rz < rx)
                    vox[i2] = 64
    for z in range(rz):                                  # :51
        for x in range(rx):
            dx = 16 - (x % 32)                           # :52
            dz = 16 - (z % 32)                           # :53
            if dx * dx + dz * dz <= 121:                 # :54-55
                y = int(ry * (0.25 + 0.125 * (math.sin(z * 0.02) * math.cos(x * 0.03))))  # :56
                for yy in range(y + 1):                  # :57
                    vox[z * rxy + yy * rx + x] = 255     # :58
    return vox


def mesh_scale(vertices, res):
    """meshvoxel.clj:16-25 -> the scale function."""
    v = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
    p = v.min(axis=0)                                    # :18 gu/bounding-box -> [p size]
    size = v.max(axis=0) - p
    md = max(size)                                       # :19
    off = np.array([0.5 * res * (1.0 - s / md) for s in size])  # :20  (* 0.5 res (- 1.0 (/ % md)))
    s = res / md                                         # :21
    return lambda q: off + (np.asarray(q, dtype=np.float64) - p) * s  # :23


def voxelize(vertices, res):
    """meshvoxel.clj:61-71."""
    vox = np.zeros(res ** 3, dtype=np.uint8)
    rxy = res * res
    f = mesh_scale(vertices, res)
    for v in np.asarray(vertices, dtype=np.float64).reshape(-1, 3):
        x, y, z = (int(c) for c in f(v))                 # :67  (map int (scale-fn v)): truncation
        if 0 <= z < res and 0 <= y < res and 0 <= x < res:   # :68
            vox[z * rxy + y * res + x] = 255             # :69
    return vox


def voxelize_ks(vertices, res, ks):
    """meshvoxel.clj:47-59."""
    vox = np.zeros((res, res, res), dtype=np.uint8)
    f = mesh_scale(vertices, res)
    for v in np.asarray(vertices, dtype=np.float64).reshape(-1, 3):
        x, y, z = (int(c) for c in f(v))                 # :54
        vox[max(0, z - ks):min(res, z + ks + 1),         # :56-58 clipped ranges
            max(0, y - ks):min(res, y + ks + 1),
            max(0, x - ks):min(res, x + ks + 1)] = 255
    return vox.reshape(-1)


def make_heatmap(pixels, amp):
    """meshvoxel.clj:73-87; columns cut at res voxels (see include/raymarch_hip.h)."""
    px = np.asarray(pixels, dtype=np.uint32)
    res = px.shape[0]
    vox = np.zeros(res ** 3, dtype=np.uint8)
    rxy = res * res
    for y in range(res):                                 # :80
        for x in range(res):
            c = int(px[y, x]) & 255                      # :81
            h = (2 if c > 224 else max(2, c * float(amp))) if c > 0 else 0   # :82
            hh = 0
            while hh < h and hh < res:                   # :83 (range h)
                vox[y * rxy + hh * res + x] = 255        # :84
                hh += 1
    return vox


def scatter_uniform(seed, vertex, k):
    """Draw k of vertex `vertex`: the counter-based uniform in [0, 1) that stands in for the reference's unseeded
    (rand) -- SplitMix64 finaliser of seed + golden * (16 * vertex + k + 1), top 53 bits (csrc/rm_volgen.hip)."""
    m = (1 << 64) - 1
    z = (seed + 0x9E3779B97F4A7C15 * (vertex * 16 + k + 1)) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    z = z ^ (z >> 31)
    return (z >> 11) * 2.0 ** -53


def voxelize_scatter(vertices, res, seed=0):
    """meshvoxel.clj:25-43, loops as written; (rand) -> scatter_uniform in the reference's call order."""
    vox = np.zeros(res ** 3, dtype=np.uint8)
    rxy = res * res
    f = mesh_scale(vertices, res)
    for vi, v in enumerate(np.asarray(vertices, dtype=np.float64).reshape(-1, 3)):
        x, y, z = (int(c) for c in f(v))                                     # :32
        draw = 0

        def rand(n=1.0):
            nonlocal draw
            u = scatter_uniform(seed, vi, draw)
            draw += 1
            return n * u                                                      # (rand n) = (* n (Math/random))
        count = 1                                                             # :33  (range (if (< (rand) 0.25) (rand 5) 1)):
        if rand() < 0.25:                                                     #      the test draws first, then (rand 5);
            count = math.ceil(rand(5))                                        #      (range 3.7) = 0 1 2 3
        for i in range(count):
            dx = int(rand((i * res) / 10.0))                                  # :34  (int (rand (* (/ i 5) r2))), r2 = res/2
            xx = int((x - dx) - (res * -0.4))                                 # :35  (int (- x dx (* res -0.4)))
            zz = max(z - int((res * 0.5) * (0.125 * rand() + 0.125)), 0)      # :36
            yy = y + res * 0.4                                                # :37  (a double)
            for z3 in range(zz - 1, zz + 2):                                  # :38
                for k in range(3):
                    y3 = (yy - 1.0) + k                                       # :39  (range (dec y) (+ 2 y))
                    for x3 in range(xx - 1, xx + 2):                          # :40
                        if 0 <= z3 < res and 0 <= y3 < res and 0 <= x3 < res:  # :41
                            vox[int(y3) * rxy + z3 * res + x3] = 64           # :42
    return vox
