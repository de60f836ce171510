// rm_accel.hip -- derived acceleration structures of a resident volume.
//
// Not a reference feature: the reference fetches the raw byte grid for every
// fixed-step sample (renderer.cl:219-234, ~600 byte loads per pixel sample) and
// 27x7 bytes for a smooth normal (:190-203).  Results must stay bit-identical,
// so the structures only let the kernels SKIP work whose outcome is known:
//
//  dist8  (1 B / voxel)  0 for a cell the march would hit (v > isoVal), else
//         the Chebyshev distance (in cells, capped at 255) to the nearest cell
//         that is a hit OR lies outside the grid.  All cells closer than that
//         are in-bounds and empty, so the fixed-step samples that fall into
//         them neither hit nor leave the grid and need not be fetched
//         (rm_shade.hpp: exact multi-step advance).
//  oct8   (8 x 1 B / voxel, behind dist8 in the same buffer)  per sign combination of a walk
//         direction the edge of the largest empty in-grid cube AHEAD of the cell; same skip
//         rule with longer skips (see the section further down).
//  surf32 (4 B / voxel, meaningful where v > isoVal)  everything a hit needs:
//         bits 0-7 voxel value, 8-13 / 14-19 / 20-25 the x/y/z sums (+32) of the
//         smooth normal, 26-27 / 28-29 / 30-31 the central differences (+1) of
//         the flat normal -- 1 load instead of 189 (smooth) or 6 (flat).
//
// Both depend on isoVal (hit test `v > isoVal`, occupancy `v >= isoVal`,
// renderer.cl:222 vs :175) and are rebuilt when it or the volume changes.
#include <hip/hip_runtime.h>

#include "rm_kernels.h"

namespace {

struct Dim { int rx, ry, rz; };

__device__ __forceinline__ bool inb(const Dim& d, int x, int y, int z) {
  return x >= 0 && x < d.rx && y >= 0 && y < d.ry && z >= 0 && z < d.rz;
}

// pass X: distance along the row to the nearest hit cell or grid edge
__global__ __launch_bounds__(256) void dist_x_kernel(const uint8_t* __restrict__ vox, Dim d, int iso,
                                                     uint8_t* __restrict__ out) {
  const long long total = (long long)d.rx * d.ry * d.rz;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % d.rx);
    const uint8_t* row = vox + (i - x);
    int best = 0;
    if (row[x] <= iso) {
      best = min(255, min(x + 1, d.rx - x));  // grid edge counts as solid
      for (int r = 1; r < best; r++) {
        if (row[x - r] > iso || row[x + r] > iso) { best = r; break; }
      }
    }
    out[i] = (uint8_t)best;
  }
}

// passes Y and Z: out(c) = min_r max(r, min(in(c - r*stride), in(c + r*stride))), 0 beyond the edge
__global__ __launch_bounds__(256) void dist_axis_kernel(const uint8_t* __restrict__ in, Dim d, int axis,
                                                        uint8_t* __restrict__ out) {
  const long long total = (long long)d.rx * d.ry * d.rz;
  const long long stride = axis == 1 ? d.rx : (long long)d.rx * d.ry;
  const int len = axis == 1 ? d.ry : d.rz;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = axis == 1 ? (int)((i / d.rx) % d.ry) : (int)(i / ((long long)d.rx * d.ry));
    int best = in[i];
    for (int r = 1; r < best; r++) {
      const int lo = (c - r >= 0) ? in[i - r * stride] : 0;
      const int hi = (c + r < len) ? in[i + r * stride] : 0;
      best = min(best, max(r, min(lo, hi)));
    }
    out[i] = (uint8_t)best;
  }
}

__device__ __forceinline__ int occ(const uint8_t* __restrict__ vox, const Dim& d, int iso, int x,
                                   int y, int z) {
  if (!inb(d, x, y, z)) return 0;
  return vox[((long long)z * d.ry + y) * d.rx + x] >= iso ? 1 : 0;  // step(isoVal, v)
}
// g = (occ(+) - occ(-)) per axis; the reference's voxelNormal is -g
__device__ __forceinline__ void central(const uint8_t* __restrict__ vox, const Dim& d, int iso, int x,
                                        int y, int z, int& gx, int& gy, int& gz) {
  gx = occ(vox, d, iso, x + 1, y, z) - occ(vox, d, iso, x - 1, y, z);
  gy = occ(vox, d, iso, x, y + 1, z) - occ(vox, d, iso, x, y - 1, z);
  gz = occ(vox, d, iso, x, y, z + 1) - occ(vox, d, iso, x, y, z - 1);
}

// surf32 in two passes (round 6; one pass read 27 x 7 bytes per hit voxel: 0.33 of the 1.28 ms of a 256^3 build):
//   1. per cell one byte: occupied (v >= isoVal) ? 1 | the three central differences + 1 in two bits each : 0
//   2. per hit cell (v > isoVal) the sum over the 27 neighbours of those bytes -- 27 loads instead of 189
__global__ __launch_bounds__(256) void surf_code_kernel(const uint8_t* __restrict__ vox, Dim d, int iso,
                                                        uint8_t* __restrict__ code) {
  const long long total = (long long)d.rx * d.ry * d.rz;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    uint8_t c = 0;
    if (vox[i] >= iso) {
      const int x = (int)(i % d.rx), y = (int)((i / d.rx) % d.ry), z = (int)(i / ((long long)d.rx * d.ry));
      int gx, gy, gz;
      central(vox, d, iso, x, y, z, gx, gy, gz);
      c = (uint8_t)(1 | (gx + 1) << 1 | (gy + 1) << 3 | (gz + 1) << 5);
    }
    code[i] = c;
  }
}
__global__ __launch_bounds__(256) void surf_sum_kernel(const uint8_t* __restrict__ vox, const uint8_t* __restrict__ code,
                                                       Dim d, int iso, uint32_t* __restrict__ out) {
  const long long total = (long long)d.rx * d.ry * d.rz;
  const long long sy = d.rx, sz = (long long)d.rx * d.ry;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int v = vox[i];
    uint32_t w = (uint32_t)v;
    if (v > iso) {
      const int x = (int)(i % d.rx), y = (int)((i / d.rx) % d.ry), z = (int)(i / sz);
      const int own = code[i];  // (v > isoVal implies occupied)
      const int gx = ((own >> 1) & 3) - 1, gy = ((own >> 3) & 3) - 1, gz = ((own >> 5) & 3) - 1;
      int sx = 0, sy_ = 0, sz_ = 0;
      for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
          for (int dx = -1; dx <= 1; dx++) {
            if (!inb(d, x + dx, y + dy, z + dz)) continue;
            const int c = code[i + dz * sz + dy * sy + dx];
            if (c & 1) { sx -= ((c >> 1) & 3) - 1; sy_ -= ((c >> 3) & 3) - 1; sz_ -= ((c >> 5) & 3) - 1; }
          }
      w |= (uint32_t)(sx + 32) << 8 | (uint32_t)(sy_ + 32) << 14 | (uint32_t)(sz_ + 32) << 20 |
           (uint32_t)(gx + 1) << 26 | (uint32_t)(gy + 1) << 28 | (uint32_t)(gz + 1) << 30;
    }
    out[i] = w;
  }
}
// (the one-pass form: grids for which the caller has no scratch bytes)
__global__ __launch_bounds__(256) void surf_kernel(const uint8_t* __restrict__ vox, Dim d, int iso,
                                                   uint32_t* __restrict__ out) {
  const long long total = (long long)d.rx * d.ry * d.rz;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int v = vox[i];
    uint32_t w = (uint32_t)v;
    if (v > iso) {
      const int x = (int)(i % d.rx), y = (int)((i / d.rx) % d.ry), z = (int)(i / ((long long)d.rx * d.ry));
      int gx, gy, gz;
      central(vox, d, iso, x, y, z, gx, gy, gz);
      int sx = 0, sy = 0, sz = 0;
      for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
          for (int dx = -1; dx <= 1; dx++)
            if (occ(vox, d, iso, x + dx, y + dy, z + dz)) {
              int ax, ay, az;
              central(vox, d, iso, x + dx, y + dy, z + dz, ax, ay, az);
              sx -= ax; sy -= ay; sz -= az;
            }
      w |= (uint32_t)(sx + 32) << 8 | (uint32_t)(sy + 32) << 14 | (uint32_t)(sz + 32) << 20 |
           (uint32_t)(gx + 1) << 26 | (uint32_t)(gy + 1) << 28 | (uint32_t)(gz + 1) << 30;
    }
    out[i] = w;
  }
}

// ---- oct8: directional empty-cube sizes ------------------------------------------
// dist8 is limited by the nearest obstacle in ANY direction -- also the surface a
// shadow ray, AO probe or reflection has just left.  oct8[o][q] (o = sign bits of the
// walk direction, x | y<<1 | z<<2, bit set = negative) is the edge n of the largest
// cube of empty in-grid cells that has q as its corner and extends AHEAD of the walk:
// cells q + s*(i,j,k), 0 <= i,j,k < n.  A walk never moves against its direction
// signs, so the samples it may skip are exactly those of dist8's rule with d := n
// (n >= d always).  0 = hit cell, capped at 255.
//
// Built by dynamic programming: the cube of edge n at q is empty iff q is empty and the seven
// cubes of edge n-1 at q + s*(dx,dy,dz), (dx,dy,dz) in {0,1}^3 \\ 0, are (together with q they
// cover it), so  n(q) = hit(q) ? 0 : 1 + min over those seven neighbours AHEAD of n, with 0 for
// neighbours beyond the grid.  With u = per-axis distance to the grid face ahead, a cell depends
// on cells with smaller u only.  The grid is cut into 16^3-cell tiles in u-space; a workgroup
// sweeps one tile plane by plane (u_x + u_y + u_z = const: 46 steps of at most 256 independent
// cells, one per thread), and the tiles on a tile diagonal -- independent of each other -- form
// one launch, all eight octants in it (each in its own mirrored frame): 3 * R/16 launches.  The
// tile and the finished layer ahead of it are staged in LDS (17^3 bytes), the sweep runs there.  (Round 1 bisected on a summed-volume table: 64 uint32 reads
// per (cell, octant) and 4.3 GiB of scratch at 1024^3; one launch per cell plane -- 766 at 256^3 --
// is bound by launch latency: 12 ms.)
// index of cell (x, y, z) in a byte table: row-major, or 8x4x4-cell bricks of 128 B (x fastest
// inside and between bricks)
__device__ __forceinline__ long long tab_index(const Dim& d, int x, int y, int z, int bricked) {
  if (!bricked) return ((long long)z * d.ry + y) * d.rx + x;
  const long long nbx = (d.rx + 7) >> 3, nby = (d.ry + 3) >> 2;
  const long long brick = ((long long)(z >> 2) * nby + (y >> 2)) * nbx + (x >> 3);
  return (brick << 7) | (long long)(((z & 3) << 5) | ((y & 3) << 3) | (x & 7));
}
__global__ __launch_bounds__(256) void unbrick_kernel(const uint8_t* __restrict__ bricked, Dim d,
                                                      uint8_t* __restrict__ lin) {
  const long long total = (long long)d.rx * d.ry * d.rz;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % d.rx), y = (int)((i / d.rx) % d.ry), z = (int)(i / ((long long)d.rx * d.ry));
    lin[i] = bricked[tab_index(d, x, y, z, 1)];
  }
}

constexpr int kOctTile = 16;
constexpr int kOctLds = kOctTile + 1;  // + the layer of cells ahead of the tile (u - 1)
// LDS layout of a tile: PLANE-MAJOR.  The sweep handles one plane la + lb + lc = t per step, one cell per thread; with the
// tile stored as s[lc][lb][la] the threads of a wavefront read addresses 272-288 bytes apart -- four LDS banks for all
// of them -- and a step took ~700 cycles.  Stored as s[t][lb][la] (t = la + lb + lc, every plane a dense 17 x 17 array)
// the cells of a step are consecutive bytes, and the seven neighbours ahead sit at fixed offsets in the planes t-1 .. t-3.
constexpr int kOctPlane = kOctLds * kOctLds;            // bytes per plane
constexpr int kOctPlanes = 3 * kOctLds - 2;             // t + 3 for la, lb, lc in [-1, 15]
__device__ __forceinline__ int oct_lds_index(int la, int lb, int lc) {
  return (la + lb + lc + 3) * kOctPlane + (lb + 1) * kOctLds + (la + 1);
}
__global__ __launch_bounds__(256) void oct_tile_kernel(const uint8_t* __restrict__ vox, Dim d, int iso,
                                                       uint8_t* __restrict__ out8, int k, int a_lo, int nb, int nc,
                                                       long long table_bytes, int bricked) {
  // the tile and the layer of cells ahead of it, in u-space: s[lc + 1][lb + 1][la + 1]
  __shared__ uint8_t s[kOctPlanes * kOctPlane];
  const int o = blockIdx.y;                       // octant
  const bool nx = o & 1, ny = o & 2, nz = o & 4;  // bit set: walking towards the low face
  // tile (A, B, C) of this block on the tile diagonal A + B + C = k
  const int A = a_lo + (int)(blockIdx.x / nb), B = (int)(blockIdx.x % nb), C = k - A - B;
  if (C < 0 || C >= nc) return;
  const long long sy = d.rx, sz = (long long)d.rx * d.ry;
  uint8_t* __restrict__ tab = out8 + (long long)o * table_bytes;
  const int a0 = A * kOctTile, b0 = B * kOctTile, c0 = C * kOctTile;
  // fill: tile cells get 255 (empty, edge unknown) or 0 (hit, or behind the grid's far face);
  // the layer ahead gets the finished values of the neighbouring tiles, 0 beyond the grid.
  // (Round 6: a launch of this kernel is as long as ONE tile takes -- the tile diagonals are a chain of 3R/16 launches --
  //  and a tile took ~20 us, most of it in this fill: 20 global byte loads per thread, each issued after the previous one
  //  had arrived.  Now a thread fetches its row of 16 voxels with ONE 16-byte load and at most four halo bytes, all
  //  independent; rows of 16 cells are contiguous and aligned when the row length is a multiple of 16.)
  const bool rows16 = (d.rx & 15) == 0 && (reinterpret_cast<unsigned long long>(vox) & 15ull) == 0;  // (a borrowed volume may sit anywhere)
  if (rows16) {
    const int rb = threadIdx.x & (kOctTile - 1), rc = threadIdx.x >> 4;  // this thread's row of the tile
    const int b = b0 + rb, c = c0 + rc;
    uint4 q = make_uint4(0u, 0u, 0u, 0u);
    const bool row_ok = b < d.ry && c < d.rz;  // (a0 + 15 < rx: the row length is a multiple of the tile edge)
    const int y = ny ? b : d.ry - 1 - b, z = nz ? c : d.rz - 1 - c;
    const int xs = nx ? a0 : d.rx - kOctTile - a0;  // lowest x of the row
    if (row_ok) q = *reinterpret_cast<const uint4*>(vox + (long long)z * sz + (long long)y * sy + xs);
    // the halo: la = -1 (17 x 17 cells), lb = -1 with la >= 0 (16 x 17), lc = -1 with la, lb >= 0 (16 x 16): 817 cells
    uint8_t hv[4];
    int hi[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int h = (int)threadIdx.x + 256 * r;
      int la, lb, lc;
      if (h < 289) { la = -1; lb = h % 17 - 1; lc = h / 17 - 1; }
      else if (h < 561) { const int g = h - 289; la = g & 15; lb = -1; lc = g / 16 - 1; }
      else { const int g = h - 561; la = g & 15; lb = (g >> 4) & 15; lc = -1; }
      const int a = a0 + la, bb = b0 + lb, cc = c0 + lc;
      const bool ok = h < 817 && a >= 0 && bb >= 0 && cc >= 0 && a < d.rx && bb < d.ry && cc < d.rz;
      const int x = nx ? a : d.rx - 1 - a, yy = ny ? bb : d.ry - 1 - bb, zz = nz ? cc : d.rz - 1 - cc;
      hv[r] = ok ? tab[tab_index(d, x, yy, zz, bricked)] : (uint8_t)0;  // finished by an earlier launch
      hi[r] = h < 817 ? oct_lds_index(la, lb, lc) : -1;
    }
    const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int jx = 0; jx < kOctTile; jx++) {
      const int v = (int)((qw[jx >> 2] >> (8 * (jx & 3))) & 0xffu);
      s[oct_lds_index(nx ? jx : kOctTile - 1 - jx, rb, rc)] = (row_ok && v <= iso) ? 255 : 0;
    }
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (hi[r] >= 0) s[hi[r]] = hv[r];
  } else {
  for (int i = threadIdx.x; i < kOctLds * kOctLds * kOctLds; i += 256) {
    const int la = i % kOctLds - 1, lb = (i / kOctLds) % kOctLds - 1, lc = i / (kOctLds * kOctLds) - 1;
    const int a = a0 + la, b = b0 + lb, c = c0 + lc;  // u
    uint8_t v = 0;
    if (a >= 0 && b >= 0 && c >= 0 && a < d.rx && b < d.ry && c < d.rz) {
      const int x = nx ? a : d.rx - 1 - a, y = ny ? b : d.ry - 1 - b, z = nz ? c : d.rz - 1 - c;
      if (la < 0 || lb < 0 || lc < 0) v = tab[tab_index(d, x, y, z, bricked)];  // finished by an earlier launch
      else v = vox[(long long)z * sz + (long long)y * sy + x] <= iso ? 255 : 0;
    }
    s[oct_lds_index(la, lb, lc)] = v;
  }
  }
  __syncthreads();
  const int la = threadIdx.x & (kOctTile - 1), lb = threadIdx.x >> 4;
  for (int t = 0; t < 3 * kOctTile - 2; t++) {
    const int lc = t - la - lb;
    if (lc >= 0 && lc < kOctTile) {
      const int i = oct_lds_index(la, lb, lc);
      if (s[i]) {
        // one step back along an axis = one plane back (and one byte / one row back for la / lb)
        const int e1 = kOctPlane + 1, e2 = kOctPlane + kOctLds, e3 = kOctPlane;
        int m = s[i - e1];
        m = min(m, (int)s[i - e2]);
        m = min(m, (int)s[i - e3]);
        m = min(m, (int)s[i - e1 - e2]);
        m = min(m, (int)s[i - e1 - e3]);
        m = min(m, (int)s[i - e2 - e3]);
        m = min(m, (int)s[i - e1 - e2 - e3]);
        s[i] = (uint8_t)min(255, m + 1);
      }
    }
    __syncthreads();
  }
  if (rows16) {  // a thread stores its row: 16 bytes, contiguous in a row-major table, two runs of 8 in a bricked one
    const int rb = threadIdx.x & (kOctTile - 1), rc = threadIdx.x >> 4;
    const int b = b0 + rb, c = c0 + rc;
    if (b < d.ry && c < d.rz) {
      const int y = ny ? b : d.ry - 1 - b, z = nz ? c : d.rz - 1 - c;
      const int xs = nx ? a0 : d.rx - kOctTile - a0;
      uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int jx = 0; jx < kOctTile; jx++)  // byte jx of the run = cell x = xs + jx
        w[jx >> 2] |= (uint32_t)s[oct_lds_index(nx ? jx : kOctTile - 1 - jx, rb, rc)] << (8 * (jx & 3));
      if (!bricked) {
        *reinterpret_cast<uint4*>(tab + tab_index(d, xs, y, z, 0)) = make_uint4(w[0], w[1], w[2], w[3]);
      } else {
        *reinterpret_cast<uint2*>(tab + tab_index(d, xs, y, z, 1)) = make_uint2(w[0], w[1]);
        *reinterpret_cast<uint2*>(tab + tab_index(d, xs + 8, y, z, 1)) = make_uint2(w[2], w[3]);
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < kOctTile * kOctTile * kOctTile; i += 256) {
    const int la2 = i & (kOctTile - 1), lb2 = (i >> 4) & (kOctTile - 1), lc2 = i >> 8;
    const int a = a0 + la2, b = b0 + lb2, c = c0 + lc2;
    if (a < d.rx && b < d.ry && c < d.rz) {
      const int x = nx ? a : d.rx - 1 - a, y = ny ? b : d.ry - 1 - b, z = nz ? c : d.rz - 1 - c;
      tab[tab_index(d, x, y, z, bricked)] = s[oct_lds_index(la2, lb2, lc2)];
    }
  }
}

// dist8 from the eight directional tables: the nearest obstacle lies in one of the closed
// octants around the cell, so the Chebyshev distance is the smallest of the eight cube edges
__global__ __launch_bounds__(256) void dist_from_oct_kernel(const uint8_t* __restrict__ oct8, long long total,
                                                            uint8_t* __restrict__ dist) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int m = oct8[i];
    for (int o = 1; o < 8; o++) m = min(m, (int)oct8[(long long)o * total + i]);
    dist[i] = (uint8_t)m;
  }
}

// The benchmark volume on the device (reference generators.clj:18-42 fills it on one
// JVM thread, minutes for 512^3).  Same formula in binary64; cos/sin come from the
// device math library, so a voxel whose value sits within an ulp of a threshold may
// differ from a host-generated grid -- the grid is an INPUT of the render path, parity
// tests always feed both sides the same bytes.
__global__ __launch_bounds__(256) void gyroid_kernel(uint8_t* __restrict__ out, Dim d) {
  const long long total = (long long)d.rx * d.ry * d.rz;
  const double scl = 0.01 * (512.0 / (double)d.rx);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % d.rx), y = (int)((i / d.rx) % d.ry), z = (int)(i / ((long long)d.rx * d.ry));
    uint8_t b = 0;
    if ((z & 0x3f) >= 32) {
      const double X = (double)x * scl + 0.3875, Y = (double)y * scl + 0.0, Z = (double)z * scl + 0.0;
      const double v = fabs(cos(X) * sin(Z) + cos(Y) * sin(X) + cos(Z) * sin(Y)) - 1.0;
      if (fabs(0.2 - v) < 0.05) b = (x & 0x3f) < 32 ? 64 : 128;
      else if (v > 0.35) b = 255;
    }
    out[i] = b;
  }
}

}  // namespace

namespace rmk {

// Quality mode: the distance field re-laid as one float4 per cell = the four values of the
// cell's xy-face at its own layer, (x, y) (x+1, y) (x, y+1) (x+1, y+1), neighbours clamped at the grid's edge.
// A trilinear sample then takes TWO 16-byte loads (this layer, the next) instead of four 8-byte ones from four
// rows: the mode is bound by the L1's tag lookups, one per lane and load.  Built once per field.
__global__ __launch_bounds__(256) void sdf_quads_kernel(const float* __restrict__ g, float4* __restrict__ q, int rx, int ry, int rz) {
  const unsigned long long cells = (unsigned long long)rx * ry * rz;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < cells;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % rx), y = (int)((i / rx) % ry);
    const unsigned long long dx = x + 1 < rx ? 1 : 0, dy = y + 1 < ry ? (unsigned long long)rx : 0;
    q[i] = make_float4(g[i], g[i + dx], g[i + dy], g[i + dy + dx]);
  }
}
hipError_t launch_sdf_quads(hipStream_t st, const float* d_field, int rx, int ry, int rz, float* d_quads) {
  sdf_quads_kernel<<<4096, 256, 0, st>>>(d_field, reinterpret_cast<float4*>(d_quads), rx, ry, rz);
  return hipGetLastError();
}

hipError_t launch_gyroid(hipStream_t st, uint8_t* d_out, int rx, int ry, int rz) {
  const Dim d{rx, ry, rz};
  const long long total = (long long)rx * ry * rz;
  const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  gyroid_kernel<<<blocks, 256, 0, st>>>(d_out, d);
  return hipGetLastError();
}

// d_dist9: table 0 = dist8 (written here from the octants), tables 1..8 = the directional ones
long long bricked_bytes(int rx, int ry, int rz) {
  return (long long)((rx + 7) >> 3) * ((ry + 3) >> 2) * ((rz + 3) >> 2) * 128;
}
hipError_t launch_unbrick(hipStream_t st, const uint8_t* d_bricked, int rx, int ry, int rz, uint8_t* d_linear) {
  const long long total = (long long)rx * ry * rz;
  const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  unbrick_kernel<<<blocks, 256, 0, st>>>(d_bricked, Dim{rx, ry, rz}, d_linear);
  return hipGetLastError();
}

hipError_t build_octants(hipStream_t st, const uint8_t* d_vox, int rx, int ry, int rz, int iso,
                         uint8_t* d_dist9, bool bricked) {
  const Dim d{rx, ry, rz};
  const long long total = bricked ? bricked_bytes(rx, ry, rz) : (long long)rx * ry * rz;  // bytes per table
  uint8_t* oct = d_dist9 + total;
  const int na = (rx + kOctTile - 1) / kOctTile, nb = (ry + kOctTile - 1) / kOctTile,
            nc = (rz + kOctTile - 1) / kOctTile;
  for (int k = 0; k < na + nb + nc - 2; k++) {
    // tile columns A that hold a tile of this diagonal (B and C range over their whole extent)
    const int a_lo = max(0, k - (nb - 1) - (nc - 1)), a_hi = min(na - 1, k);
    if (a_hi < a_lo) continue;
    const dim3 grid((unsigned)((a_hi - a_lo + 1) * nb), 8u);
    oct_tile_kernel<<<grid, 256, 0, st>>>(d_vox, d, iso, oct, k, a_lo, nb, nc, total, bricked ? 1 : 0);
  }
  const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  dist_from_oct_kernel<<<blocks, 256, 0, st>>>(oct, total, d_dist9);
  return hipGetLastError();
}

hipError_t build_accel(hipStream_t st, const uint8_t* d_vox, int rx, int ry, int rz, int iso,
                       uint8_t* d_dist, uint8_t* d_tmp, uint32_t* d_surf, uint8_t* d_scratch) {
  const Dim d{rx, ry, rz};
  const long long total = (long long)rx * ry * rz;
  const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  if (d_dist) {  // dist8 alone by separable passes (when the directional tables are not built)
    dist_x_kernel<<<blocks, 256, 0, st>>>(d_vox, d, iso, d_dist);
    dist_axis_kernel<<<blocks, 256, 0, st>>>(d_dist, d, 1, d_tmp);
    dist_axis_kernel<<<blocks, 256, 0, st>>>(d_tmp, d, 2, d_dist);
  }
  if (d_scratch) {  // rx*ry*rz bytes nobody needs until this call has finished
    surf_code_kernel<<<blocks, 256, 0, st>>>(d_vox, d, iso, d_scratch);
    surf_sum_kernel<<<blocks, 256, 0, st>>>(d_vox, d_scratch, d, iso, d_surf);
  } else {
    surf_kernel<<<blocks, 256, 0, st>>>(d_vox, d, iso, d_surf);
  }
  return hipGetLastError();
}

}  // namespace rmk
