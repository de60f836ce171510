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
// Built from a summed-volume table of the hit mask (box sum == 0 <=> box empty) by
// bisection on n: 8 steps x 8 reads per (cell, octant).
__global__ __launch_bounds__(256) void sat_x_kernel(const uint8_t* __restrict__ vox, Dim d, int iso,
                                                    uint32_t* __restrict__ sat) {
  // sat has (rx+1, ry+1, rz+1) entries; entry (x,y,z) = number of hit cells in [0,x) x [0,y) x [0,z)
  const long long rows = (long long)(d.ry + 1) * (d.rz + 1);
  const long long sx = d.rx + 1;
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows;
       r += (long long)gridDim.x * blockDim.x) {
    const int y = (int)(r % (d.ry + 1)), z = (int)(r / (d.ry + 1));
    uint32_t* row = sat + r * sx;
    uint32_t acc = 0;
    row[0] = 0;
    if (y == 0 || z == 0) {
      for (int x = 1; x <= d.rx; x++) row[x] = 0;
      continue;
    }
    const uint8_t* src = vox + ((long long)(z - 1) * d.ry + (y - 1)) * d.rx;
    for (int x = 1; x <= d.rx; x++) {
      acc += src[x - 1] > iso ? 1u : 0u;
      row[x] = acc;
    }
  }
}
__global__ __launch_bounds__(256) void sat_axis_kernel(uint32_t* __restrict__ sat, Dim d, int axis) {
  const long long sx = d.rx + 1, sy = d.ry + 1, sz = d.rz + 1;
  const long long lines = axis == 1 ? sx * sz : sx * sy;
  for (long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x; l < lines;
       l += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(l % sx);
    const long long o = l / sx;  // z (axis 1) or y (axis 2)
    long long base, stride;
    int len;
    if (axis == 1) { base = (o * sy) * sx + x; stride = sx; len = (int)sy; }
    else { base = o * sx + x; stride = sx * sy; len = (int)sz; }
    uint32_t acc = 0;
    for (int k = 0; k < len; k++) {
      acc += sat[base + k * stride];
      sat[base + k * stride] = acc;
    }
  }
}
__device__ __forceinline__ uint32_t sat_box(const uint32_t* __restrict__ sat, const Dim& d, int x0,
                                            int x1, int y0, int y1, int z0, int z1) {
  // hit cells in [x0,x1) x [y0,y1) x [z0,z1)
  const long long sx = d.rx + 1, sxy = sx * (d.ry + 1);
#define RM_S(X, Y, Z) sat[(long long)(Z) * sxy + (long long)(Y) * sx + (X)]
  return RM_S(x1, y1, z1) - RM_S(x0, y1, z1) - RM_S(x1, y0, z1) - RM_S(x1, y1, z0) + RM_S(x0, y0, z1) +
         RM_S(x0, y1, z0) + RM_S(x1, y0, z0) - RM_S(x0, y0, z0);
#undef RM_S
}
__global__ __launch_bounds__(256) void oct_kernel(const uint32_t* __restrict__ sat, Dim d,
                                                  uint8_t* __restrict__ out8) {
  const long long total = (long long)d.rx * d.ry * d.rz;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total * 8;
       i += (long long)gridDim.x * blockDim.x) {
    const int o = (int)(i / total);
    const long long c = i % total;
    const int x = (int)(c % d.rx), y = (int)((c / d.rx) % d.ry), z = (int)(c / ((long long)d.rx * d.ry));
    const bool nx = o & 1, ny = o & 2, nz = o & 4;
    // room to the grid edge ahead, per axis (cells including q itself)
    int hi = min(255, min(nx ? x + 1 : d.rx - x, min(ny ? y + 1 : d.ry - y, nz ? z + 1 : d.rz - z)));
    int lo = 0;  // largest n known empty
    while (lo < hi) {
      const int n = (lo + hi + 1) >> 1;
      const int x0 = nx ? x - n + 1 : x, y0 = ny ? y - n + 1 : y, z0 = nz ? z - n + 1 : z;
      if (sat_box(sat, d, x0, x0 + n, y0, y0 + n, z0, z0 + n) == 0) lo = n;
      else hi = n - 1;
    }
    out8[i] = (uint8_t)lo;
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

hipError_t launch_gyroid(hipStream_t st, uint8_t* d_out, int rx, int ry, int rz) {
  const Dim d{rx, ry, rz};
  const long long total = (long long)rx * ry * rz;
  const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  gyroid_kernel<<<blocks, 256, 0, st>>>(d_out, d);
  return hipGetLastError();
}

size_t octant_scratch_bytes(int rx, int ry, int rz) { return (size_t)(rx + 1) * (ry + 1) * (rz + 1) * 4; }

hipError_t build_octants(hipStream_t st, const uint8_t* d_vox, int rx, int ry, int rz, int iso,
                         uint8_t* d_dist9, uint32_t* d_sat) {
  const Dim d{rx, ry, rz};
  const long long total = (long long)rx * ry * rz;
  auto blocks_for = [](long long n) { return (int)((n + 255) / 256 > 65536 ? 65536 : (n + 255) / 256); };
  sat_x_kernel<<<blocks_for((long long)(ry + 1) * (rz + 1)), 256, 0, st>>>(d_vox, d, iso, d_sat);
  sat_axis_kernel<<<blocks_for((long long)(rx + 1) * (rz + 1)), 256, 0, st>>>(d_sat, d, 1);
  sat_axis_kernel<<<blocks_for((long long)(rx + 1) * (ry + 1)), 256, 0, st>>>(d_sat, d, 2);
  oct_kernel<<<blocks_for(total * 8), 256, 0, st>>>(d_sat, d, d_dist9 + total);
  return hipGetLastError();
}

hipError_t build_accel(hipStream_t st, const uint8_t* d_vox, int rx, int ry, int rz, int iso,
                       uint8_t* d_dist, uint8_t* d_tmp, uint32_t* d_surf) {
  const Dim d{rx, ry, rz};
  const long long total = (long long)rx * ry * rz;
  const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  dist_x_kernel<<<blocks, 256, 0, st>>>(d_vox, d, iso, d_dist);
  dist_axis_kernel<<<blocks, 256, 0, st>>>(d_dist, d, 1, d_tmp);
  dist_axis_kernel<<<blocks, 256, 0, st>>>(d_tmp, d, 2, d_dist);
  surf_kernel<<<blocks, 256, 0, st>>>(d_vox, d, iso, d_surf);
  return hipGetLastError();
}

}  // namespace rmk
