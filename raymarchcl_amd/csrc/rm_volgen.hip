// rm_volgen.hip -- the volume producers either side of the render path, on the device:
// the reference fills its byte grids on one JVM thread (generators.clj, meshvoxel.clj;
// minutes for 512^3), these are HBM-bound fill / scatter kernels.
//
//   terrain_kernel   generators.clj:44-60   make-terrain
//   splat_kernel     meshvoxel.clj:16-25 (mesh-scale) + :47-59 (voxelize-ks) / :61-71 (voxelize)
//   heatmap_kernel   meshvoxel.clj:73-87    make-heatmap
//
// All arithmetic that decides a voxel is binary64 +,-,*,/ and int casts -- the same
// IEEE operations the JVM performs -- except terrain's sin/cos, which come from the
// device math library (a column whose height sits within an ulp of an integer may
// differ from a host-generated grid; the grid is an INPUT of the render path).
#include "rm_kernels.h"

namespace {

struct Dim3i { int rx, ry, rz; };

// Gather form of the reference's two write loops (the second overrides the first):
//   walls   (z < 4 | x >= rx-4 with the loop's x used as z), y < int(ry*0.666)  -> 64
//   columns (16 - x%32)^2 + (16 - z%32)^2 <= 121, y <= int(ry*(0.25 + 0.125*sin(z*0.02)*cos(x*0.03))) -> 255
__global__ __launch_bounds__(256) void terrain_kernel(uint8_t* __restrict__ out, Dim3i d) {
  const long long total = (long long)d.rx * d.ry * d.rz;
  const int ytop = (int)((double)d.ry * 0.666);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % d.rx), y = (int)((i / d.rx) % d.ry), z = (int)(i / ((long long)d.rx * d.ry));
    uint8_t b = 0;
    // (aset voxels (madd z rxy y rx x) 64): z in 0..3, x in 0..rx-1
    if (z < 4 && y < ytop) b = 64;
    // (aset voxels (madd x rxy y rx (dec (- rx z))) 64): slab index = the loop's x (0..rx-1),
    // column rx-1-z for z in 0..3
    if (x >= d.rx - 4 && z < d.rx && y < ytop) b = 64;
    const int dx = 16 - (x % 32), dz = 16 - (z % 32);
    if (dx * dx + dz * dz <= 121) {
      const int top = (int)((double)d.ry * (0.25 + 0.125 * (sin((double)z * 0.02) * cos((double)x * 0.03))));
      if (y <= top) b = 255;
    }
    out[i] = b;
  }
}

struct Splat {
  double px, py, pz;     // bounding-box minimum
  double ox, oy, oz;     // centring offset
  double s;              // res / largest extent
  int res, ks;           // ks < 0: single cell with bounds test (voxelize)
};

__device__ __forceinline__ int d2i(double v) {  // Clojure (int v) for in-range values: truncate
  return (int)v;
}

__global__ __launch_bounds__(256) void splat_kernel(uint8_t* __restrict__ out,
                                                    const double* __restrict__ xyz, long long n,
                                                    Splat sp) {
  const long long rxy = (long long)sp.res * sp.res;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    // (g/+ off (g/* (g/- v p) s))
    const int x = d2i(sp.ox + (xyz[3 * i + 0] - sp.px) * sp.s);
    const int y = d2i(sp.oy + (xyz[3 * i + 1] - sp.py) * sp.s);
    const int z = d2i(sp.oz + (xyz[3 * i + 2] - sp.pz) * sp.s);
    if (sp.ks < 0) {
      if (z >= 0 && z < sp.res && y >= 0 && y < sp.res && x >= 0 && x < sp.res)
        out[z * rxy + (long long)y * sp.res + x] = 255;
      continue;
    }
    const int z0 = max(0, z - sp.ks), z1 = min(sp.res, z + sp.ks + 1);
    const int y0 = max(0, y - sp.ks), y1 = min(sp.res, y + sp.ks + 1);
    const int x0 = max(0, x - sp.ks), x1 = min(sp.res, x + sp.ks + 1);
    for (int zz = z0; zz < z1; zz++)
      for (int yy = y0; yy < y1; yy++)
        for (int xx = x0; xx < x1; xx++) out[zz * rxy + (long long)yy * sp.res + xx] = 255;
  }
}

// voxelize-scatter (meshvoxel.clj:25-43) with its random draws SEEDED.  The reference calls (rand) -- Math/random,
// unseeded -- in a fixed order per vertex: one draw decides (p < 0.25) whether the vertex is smeared, a second how many
// copies (range (rand 5)), then per copy i one draw for the x shift (rand (* (/ i 5) r2)) and one for the z shift.  Here
// draw k of vertex v is a counter-based uniform u(seed, v, k) in [0,1) (SplitMix64 finaliser, 53 bits), so every vertex
// is independent of the others and the volume is a function of (vertices, res, seed).  Everything else is the
// reference's arithmetic: binary64, (int ..) truncation, the bounds tests on the (double) y range, the index
// y*res^2 + z*res + x (y and z swapped relative to the other voxelisers, :42), byte 64.
__device__ __forceinline__ double scatter_uniform(unsigned long long seed, long long vertex, int k) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)vertex * 16ull + (unsigned long long)k + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * 0x1p-53;
}
__global__ __launch_bounds__(256) void scatter_kernel(uint8_t* __restrict__ out, const double* __restrict__ xyz,
                                                      long long n, Splat sp, unsigned long long seed) {
  const long long rxy = (long long)sp.res * sp.res;
  const double dres = (double)sp.res;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n;
       v += (long long)gridDim.x * blockDim.x) {
    const int x0 = d2i(sp.ox + (xyz[3 * v + 0] - sp.px) * sp.s);
    const int y0 = d2i(sp.oy + (xyz[3 * v + 1] - sp.py) * sp.s);
    const int z0 = d2i(sp.oz + (xyz[3 * v + 2] - sp.pz) * sp.s);
    int copies = 1, draw = 1;                                   // (range (if (< (rand) 0.25) (rand 5) 1)): draws are
    if (scatter_uniform(seed, v, 0) < 0.25) {                   // numbered in the reference's call order, so the copy
      copies = (int)ceil(5.0 * scatter_uniform(seed, v, 1));    // count consumes draw 1 only when it is asked for
      draw = 2;
    }
    for (int i = 0; i < copies; i++) {
      const double span = (double)((long long)i * sp.res) / 10.0;   // (* (/ i 5) r2), r2 = res/2: the rational i*res/10
      const int dx = d2i(span * scatter_uniform(seed, v, draw + 2 * i));
      const int x = d2i((double)(x0 - dx) - dres * -0.4);          // (int (- x dx (* res -0.4)))
      const int back = d2i((dres * 0.5) * (0.125 * scatter_uniform(seed, v, draw + 2 * i + 1) + 0.125));
      const int z = max(z0 - back, 0);
      const double y = (double)y0 + dres * 0.4;                     // (+ y (* res 0.4)): a double from here on
      for (int zz = z - 1; zz < z + 2; zz++)
        for (int k = 0; k < 3; k++) {
          const double yy = (y - 1.0) + (double)k;                  // (range (dec y) (+ 2 y)): y-1, y, y+1
          for (int xx = x - 1; xx < x + 2; xx++)
            if (zz >= 0 && zz < sp.res && yy >= 0.0 && yy < dres && xx >= 0 && xx < sp.res)
              out[(long long)d2i(yy) * rxy + (long long)zz * sp.res + xx] = 64;
        }
    }
  }
}

// pixel (x, y) of a res x res image -> a column of ceil(h) voxels in slab y:
//   c = argb & 255;  h = c > 0 ? (c > 224 ? 2 : max(2, c*amp)) : 0;  voxels[y*rxy + hh*res + x] = -1
__global__ __launch_bounds__(256) void heatmap_kernel(uint8_t* __restrict__ out,
                                                      const uint32_t* __restrict__ argb, int res,
                                                      double amp) {
  const long long rxy = (long long)res * res;
  const long long total = rxy * res;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % res), hh = (int)((i / res) % res), y = (int)(i / rxy);
    const int c = (int)(argb[(long long)y * res + x] & 255u);
    double h = 0.0;
    if (c > 0) {
      h = 2.0;
      if (c <= 224) {
        const double v = (double)c * amp;
        if (v > 2.0) h = v;  // (max 2 v)
      }
    }
    out[i] = ((double)hh < h) ? 255 : 0;  // (range h): 0, 1, ... while < h
  }
}

inline int blocks_for(long long total) {
  const long long b = (total + 255) / 256;
  return (int)(b > 65536 ? 65536 : (b < 1 ? 1 : b));
}

}  // namespace

namespace rmk {

hipError_t launch_terrain(hipStream_t st, uint8_t* d_out, int rx, int ry, int rz) {
  terrain_kernel<<<blocks_for((long long)rx * ry * rz), 256, 0, st>>>(d_out, Dim3i{rx, ry, rz});
  return hipGetLastError();
}

hipError_t launch_splat(hipStream_t st, uint8_t* d_out, const double* d_xyz, long long n,
                        const double p[3], const double off[3], double s, int res, int ks) {
  hipError_t e = hipMemsetAsync(d_out, 0, (size_t)res * res * res, st);
  if (e != hipSuccess || n == 0) return e;
  const Splat sp{p[0], p[1], p[2], off[0], off[1], off[2], s, res, ks};
  splat_kernel<<<blocks_for(n), 256, 0, st>>>(d_out, d_xyz, n, sp);
  return hipGetLastError();
}

hipError_t launch_scatter(hipStream_t st, uint8_t* d_out, const double* d_xyz, long long n,
                          const double p[3], const double off[3], double s, int res, unsigned long long seed) {
  hipError_t e = hipMemsetAsync(d_out, 0, (size_t)res * res * res, st);
  if (e != hipSuccess || n == 0) return e;
  const Splat sp{p[0], p[1], p[2], off[0], off[1], off[2], s, res, 0};
  scatter_kernel<<<blocks_for(n), 256, 0, st>>>(d_out, d_xyz, n, sp, seed);
  return hipGetLastError();
}

hipError_t launch_heatmap(hipStream_t st, uint8_t* d_out, const uint32_t* d_argb, int res, double amp) {
  heatmap_kernel<<<blocks_for((long long)res * res * res), 256, 0, st>>>(d_out, d_argb, res, amp);
  return hipGetLastError();
}

}  // namespace rmk
