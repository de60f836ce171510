// rm_kernels.hip -- gfx950 kernels of the render path and their launchers.
//
//   render_pass_kernel   == one NDRange of the reference's RenderImage
//                           (renderer.cl:478-494): one lane per sample, the
//                           accumulator is blended in place.
//   tonemap_kernel       == TonemapImage (renderer.cl:448-454, 496-508).
//   prims_kernel         -- device side of rm_selftest_prims.
//
// Work decomposition: the image is cut into 8x8-pixel tiles (row-major tile
// order); a workgroup is ONE 64-lane wavefront (finest dispatch granularity) that owns a tile
// (single-pass kernels) or a 2x2-pixel block of a tile x 16 passes (frame kernel), so that the
// rays of a wave are coherent (their table fetches share cache lines).  Everything that is uniform across the launch (the
// 544-byte option record) is read through a uniform pointer, i.e. by scalar
// loads into SGPRs -- the reference's per-work-item private copy of the record
// is what costs it 560 B of scratch per lane on this chip.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <type_traits>

#include "rm_kernels.h"
#include "rm_shade.hpp"



#ifndef RM_SDF_WAVE
#define RM_SDF_WAVE 1  // quality mode: AO probes and soft-shadow marches of a wavefront's hits traced by all its lanes
#endif

namespace {

constexpr int kTile = 8;            // tile edge in pixels; 64 px == one wavefront
constexpr int kWavesPerBlock = 1;  // one wavefront per workgroup: finest dispatch granularity (measured best)

struct TileGeom {
  int tiles_x, tiles_total;
};
__host__ __device__ inline TileGeom tile_geom(int resx, int n) {
  const int rows = (n + resx - 1) / resx;
  TileGeom g;
  g.tiles_x = (resx + kTile - 1) / kTile;
  g.tiles_total = g.tiles_x * ((rows + kTile - 1) / kTile);
  return g;
}

// lane -> work-item id of the pixel it owns, or -1
__device__ __forceinline__ int lane_pixel(int tile, int lane, int resx, int tiles_x, int n,
                                          int id0, int id1) {
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int x = tx * kTile + (lane & (kTile - 1));
  const int y = ty * kTile + (lane >> 3);
  if (x >= resx) return -1;
  const long long id = (long long)y * resx + x;
  if (id >= n || id < id0 || id >= id1) return -1;
  return (int)id;
}

// TILE_MAJOR: the accumulator is stored tile by tile (slot*64 + lane, one
// contiguous 1 KiB store per wave) instead of at the work-item id; used by the
// device-resident pipeline and the multi-GPU partition, un-permuted by
// resolve_kernel.
// ARITH: the arithmetic contract (rmk::ArithOf, rm_math.hpp)
template <bool COUNT, bool TILE_MAJOR, bool ACCEL, int LAYOUT = 0, int ARITH = 0>
// (4 wavefronts per SIMD = 128 VGPRs: every lane of these kernels carries its own secondary rays; compiled for more
//  wavefronts some instantiations spill SGPRs, which the build's lint refuses)
__global__ __launch_bounds__(64 * kWavesPerBlock, 4) void render_pass_kernel(
    const uint8_t* __restrict__ vox, const uint8_t* __restrict__ dist8,
    const uint32_t* __restrict__ surf32, const float4* __restrict__ mc,
    const RmOpts* __restrict__ opts,
    float4* __restrict__ pixels, int n, int id0, int id1, int tile_first, int tile_stride,
    rmk::Counters* __restrict__ counters, unsigned long long oct_stride = 0, unsigned log2res = 0) {
  const int resx = opts->resolution[0];
  const TileGeom g = tile_geom(resx, n);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long slot = (long long)blockIdx.x * kWavesPerBlock + wave;
  const long long tile = tile_first + slot * tile_stride;
  if (tile >= g.tiles_total) return;
  const int id = lane_pixel((int)tile, lane, resx, g.tiles_x, n, id0, id1);
  rmk::Scene sc{vox, mc, opts, dist8, surf32, oct_stride};
  sc.log2res = log2res;
  using M = typename rmk::ArithOf<ARITH>::type;
  rmk::Tracer<COUNT, ACCEL, false, LAYOUT, M> tr(sc);
  if (id >= 0) {
    const rmk::v3 col = tr.shade(id);
    const float fb = opts->frameBlend;
    const long long at = TILE_MAJOR ? slot * 64 + lane : (long long)id;
    const float4 p = pixels[at];
    // mix(p, col, frameBlend): renderer.cl:492
    pixels[at] = make_float4(M::mix(p.x, col.x, fb), M::mix(p.y, col.y, fb), M::mix(p.z, col.z, fb), 1.0f);
  }
  if (COUNT) {
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(counters);
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&tr.cnt);
    for (int k = 0; k < (int)(sizeof(rmk::Counters) / 8); k++) {
      unsigned long long v = src[k];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0 && v) atomicAdd(dst + k, v);
    }
  }
}

// The frame kernel: ALL RenderImage passes of a frame for a partition of the image, blended in
// the reference's order, in one launch (renderer.cl:478-494 applied pass after pass,
// core.clj:81-90) -- and, for an unpartitioned image, TonemapImage as well (renderer.cl:496-508).
//
// Lane -> (pixel, pass).  With pp = 2^pp_log2 passes per wavefront a wavefront owns 64/pp pixels
// of one 8x8 tile and walks through the frame's passes pp at a time: lane l traces pixel l / pp
// in pass c0 + l % pp.  (pp > 1 needs records that differ in .time only, which is what the
// reference's host produces, core.clj:99-106.)  Lanes that trace the SAME pixel with different
// jitter follow almost the same control flow, which is what a 64-wide SIMT machine wants --
// neighbouring pixels of one pass diverge far more (sky / surface / reflection).
//
// Blend: the reference's accumulator update p <- mix(p, c, frameBlend) is an in-order
// recurrence over the passes (exponential, not a mean).  The pp colours of a pixel
// sit in pp neighbouring lanes: they go through LDS to the pixel's first lane, which applies
// the recurrence in pass order and carries p in registers from one group of passes to the
// next.  One float4 per pixel leaves the kernel; nothing is staged per pass.
struct FrameArgs {
  const uint8_t* __restrict__ vox;
  const uint8_t* __restrict__ dist8;
  const uint32_t* __restrict__ surf32;
  unsigned long long oct_stride;
  const float* __restrict__ sdf;
  const float4* __restrict__ mc_all;   // scatter table of the launch's first pass
  const RmOpts* __restrict__ opts_all; // record of the launch's first pass
  const RmOpts* __restrict__ opts0;    // record 0 of the frame (TonemapImage reads its gamma)
  float4* __restrict__ acc;            // tile-major partition accumulators, or the row-major image
  uint32_t* __restrict__ argb;         // row-major ARGB, or nullptr
  int n, resx, tile_first, tile_stride, tiles_per_part, pp_log2, passes, bpr;
  unsigned log2res;                    // table LAYOUT 2: edge of the cubic grid = 1 << log2res
  int accumulate;                      // 0: the accumulator starts at zero (first launch of a frame)
  int rows_desc;                       // XCD-aware order: tile rows dispatched bottom to top
  int xcd2d;                           // XCD-aware order in 2-D units (frame_block): n > 0 = an XCD renders n units of every pair of
                                       // tile rows, each 2 rows x 1/(8 n) of their width; 0 = whole tile rows per XCD
  int rows_real, full_groups, tail_share;  // XCD-aware order: tile rows of the launch, whole groups of 8 among them, blocks of the
                                       // last (< 8) rows per XCD (see frame_block)
  int band_r0, band_r1;                // ... and, when band_r1 > band_r0, the tile rows [band_r0, band_r1) FIRST: the rows whose
                                       // primary rays can meet the clip box (rm_api.hip volume_band), then the rows below them,
                                       // the rows above them -- sky in the reference's scenes -- last
  int row_major;                       // acc is indexed by work-item id instead of slot*64 + pixel
};

template <class M>
__device__ __forceinline__ uint32_t tonemap_argb(float px, float py, float pz, float g) {
  const float c[3] = {px, py, pz};
  uint32_t ch[3];
  for (int k = 0; k < 3; k++) {
    const float t = M::div(c[k], g + c[k]);  // renderer.cl:453
    const float v = t * t * 255.0f;
    ch[k] = (uint32_t)M::to_int(M::clamp(v, 0.0f, 255.0f));
  }
  return 0xff000000u | (ch[0] << 16) | (ch[1] << 8) | ch[2];
}

// hardware block -> logical block of the launch (tile slot << pp_log2 | sub-block), or -1 for the padding of the XCD-aware
// grid.  A permutation: every logical block of the launch is rendered by exactly one hardware block, whatever the order
// (tests/test_host_abi.py walks it on the host through rm_debug_block_order for many image shapes, partitions and orders).
template <class Args>
__host__ __device__ __forceinline__ long long logical_block(const Args& a, long long hw_block) {
  // XCD-aware order (bpr > 0): the dispatcher deals consecutive workgroups to the 8
  // XCDs round-robin, each with its own L2.  Hardware block b = 8*m + k is given the
  // logical block of tile row 8*(m / bpr) + k, so XCD k renders every 8th tile ROW:
  // its primary rays sweep an eighth of the volume's slabs instead of all of them,
  // while rows stay interleaved finely enough to balance the load.
  long long lb = hw_block;
  if (a.bpr > 0) {
    const int pp_log2 = a.pp_log2, pp = 1 << pp_log2;
    const long long m = lb >> 3, k = lb & 7;
    // The dispatcher's assignment of workgroups to XCDs is STATIC (workgroup i -> XCD i % 8), so an XCD's share of the
    // frame is fixed by this mapping.  Whole groups of 8 tile rows give every XCD one row each.  The last r < 8 rows used to
    // go to r of the XCDs as whole rows -- at 720 lines (90 tile rows) two XCDs rendered 12 rows and six 11, and the frame
    // took exactly as long as one of 96 rows (round 6: 3.97 ms for 712 .. 768 lines, 3.72 ms for 704).  Now the blocks of
    // those r rows are dealt to the eight XCDs in equal contiguous shares.
    const long long full = (long long)a.full_groups * a.bpr;
    long long row, col;  // row in dispatch order, block within the row
    if (a.xcd2d) {
      // 2-D UNITS (round 6): XCD k renders, of the row pair rp, the stripes (k - rp) mod 8 + 8 j (j < n) of the 8 n its width
      // is cut into -- two tile rows x 1/(8 n) row per unit, the two rows interleaved tile by tile.  Every XCD gets exactly n
      // units of every row pair (and an eighth of an odd last row): equal shares by construction, no padding; and the
      // wavefronts an XCD has in flight cover compact patches of the image (config 2, n = 1: 160 x 16 pixels instead of a
      // 448 x 8 strip), whose rays share more table lines in the XCD's L2: config 2 3.78 -> 3.69 ms.  n is chosen per
      // launch (frame_grid): config 2 is fastest at 1, config 3 at 2 (9.74 -> 9.22 ms), config 5 31.65 instead of 32.6 ms with
      // whole rows; units of 1 or 3 rows: slower.
      const long long nsu = a.xcd2d, sw = a.bpr / (8 * nsu), per_unit = 2 * sw, pairs = a.rows_real >> 1;
      const long long ul = m / per_unit, rp = ul / nsu, jj = ul - rp * nsu;
      if (rp < pairs) {
        const long long w = m - ul * per_unit, chunk = w >> pp_log2;
        row = 2 * rp + (chunk & 1);
        col = (((k - rp) & 7) + 8 * jj) * sw + ((chunk >> 1) << pp_log2) + (w & (pp - 1));
      } else {
        const long long w = m - pairs * nsu * per_unit;  // an odd last row: 1/8 of it per XCD
        if (w >= sw * nsu || !(a.rows_real & 1)) return -1;
        row = a.rows_real - 1;
        col = ((k - rp) & 7) * (sw * nsu) + w;
      }
    } else if (m < full) {
      row = (m / a.bpr) * 8 + k;
      col = m % a.bpr;
    } else {
      const long long j = k * (long long)a.tail_share + (m - full);
      if (m - full >= a.tail_share || j >= (long long)(a.rows_real - 8 * a.full_groups) * a.bpr) return -1;  // (padding of the shares)
      row = 8ll * a.full_groups + j / a.bpr;
      col = j % a.bpr;
    }
    // ... BOTTOM ROWS FIRST (round 5): a frame that is waited for ends with the tail of its last wavefronts, and in
    // the reference's scenes the top rows are the cheapest.  One blocking frame -0.5 %, a rank's share of an 8-way partition
    // -12 %, config 5 -3.5 %, config 3 +1.1 %, config 4 +0.2 % (RAYMARCH_ROW_ORDER=asc restores the old order).
    const long long rows = a.rows_real;
    // ... A BAND OF ROWS FIRST (round 6, opt-in: RAYMARCH_ROW_ORDER=band / RAYMARCH_ROW_BAND): those rows bottom to top, then
    // the rows below them, the rows above them last (rm_api.hip volume_band)
    long long at_row = a.rows_desc ? rows - 1 - row : row;
    const long long nb = (long long)a.band_r1 - a.band_r0;
    if (nb > 0) {
      const long long below = rows - a.band_r1;
      at_row = row < nb ? a.band_r1 - 1 - row : (row < nb + below ? rows - 1 - (row - nb) : a.band_r0 - 1 - (row - nb - below));
    }
    lb = at_row * a.bpr + col;
  }
  return lb;
}

// A launch holds at most the passes ONE wavefront holds (2^pp_log2): frames with more passes go out as several
// launches that continue each other's accumulator (rm_api.hip frame_on_device) -- nothing of a sample then lives
// across the body of another, and the kernel has no loop over groups of passes.
// ARITH: the arithmetic contract, rmk::ArithOf (rm_math.hpp): 0 = OpenCL CPU device arithmetic and casts, 1 = the same
// with the GPU lowering of the seed casts, 2 / 3 = ROCm's OpenCL library on this GPU, strict / default reference build
template <bool ACCEL, bool SDFM, int LAYOUT, int ARITH>
__device__ __forceinline__ void frame_block(const FrameArgs& a, long long hw_block, float* wave_lds) {
  using M = typename rmk::ArithOf<ARITH>::type;
  using Tr = rmk::Tracer<false, ACCEL, SDFM, LAYOUT, M>;
  const int pp_log2 = a.pp_log2;
  const int pp = 1 << pp_log2;              // passes per wavefront
  const int ppw = 64 >> pp_log2;            // pixels per wavefront
  const TileGeom g = tile_geom(a.resx, a.n);
  const int lane = threadIdx.x & 63;
  const long long lb = logical_block(a, hw_block);
  if (lb < 0) return;
  // wavefronts of a tile are consecutive: tile slot = w / pp, sub-block = w % pp
  const long long slot = lb >> pp_log2;
  const int sub = (int)(lb & (pp - 1));
  if (slot >= a.tiles_per_part) return;  // padding of the XCD-aware grid
  const long long tile = a.tile_first + slot * a.tile_stride;
  if (tile >= g.tiles_total) return;
  // Z-order inside the tile, so that any 2^k consecutive pixels form a compact block
  const int z = sub * ppw + (lane >> pp_log2);
  const int zx = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4);
  const int zy = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
  const int pix = zy * 8 + zx;  // 0..63 within the tile, row-major 8x8
  const int id = lane_pixel((int)tile, pix, a.resx, g.tiles_x, a.n, 0, a.n);
  if (id < 0) return;  // (all pp lanes of a pixel leave together)
  // (the colours of a group are exchanged through the area of the shared phases, whose posted
  //  values are dead once shade_wave() has returned)
  float* const blend_lds = wave_lds;
  const int pl = lane & (pp - 1);  // this lane's pass within a group
  const bool first = pl == 0;      // the lane that keeps its pixel's accumulator
  const long long at = a.row_major ? (long long)id : slot * 64 + pix;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (a.accumulate && first) {
    const float4 p = a.acc[at];
    px = p.x; py = p.y; pz = p.z;
  }
  for (int c0 = 0; c0 == 0; c0 += 1) {  // (one group: c0 = first pass of the group this wavefront holds = of the launch)
    const int pass = c0 + pl;
    const bool live = pass < a.passes;
    const RmOpts* __restrict__ opts = a.opts_all + c0;  // uniform (pp > 1: all of the group equal but .time)
    rmk::Scene sc{a.vox, a.mc_all + (size_t)c0 * RM_TABLE_ENTRIES, opts, a.dist8, a.surf32, a.oct_stride, a.sdf};
    sc.log2res = a.log2res;
    Tr tr(sc);
    if (pp > 1 && live) tr.set_pass(a.mc_all + (size_t)pass * RM_TABLE_ENTRIES, a.opts_all[pass].time);
    rmk::v3 col = rmk::V(0.f, 0.f, 0.f);
    // secondary rays shared by the wavefront (quality mode: while the record's AO probes fit the exchange area --
    // uniform; the accelerated kernels get frames with more probes through the single-pass kernels, rm_api.hip)
    const bool shared = ACCEL || (SDFM && RM_SDF_WAVE && opts->aoIter + 1 <= Tr::kWaveLdsRes);
    if (shared) col = tr.shade_wave(id, wave_lds, live);
    else if (live) col = tr.shade(id);
    // mix(p, col, frameBlend) in pass order: renderer.cl:492
    if (pp > 1) {
      blend_lds[lane] = col.x; blend_lds[64 + lane] = col.y; blend_lds[128 + lane] = col.z;
      __syncthreads();
      if (first) {
        const int cnt = min(pp, a.passes - c0);  // uniform
        for (int k = 0; k < cnt; k++) {
          const float fb = a.opts_all[c0 + k].frameBlend;
          px = M::mix(px, blend_lds[lane + k], fb);
          py = M::mix(py, blend_lds[64 + lane + k], fb);
          pz = M::mix(pz, blend_lds[128 + lane + k], fb);
        }
      }
      __syncthreads();
    } else {
      const float fb = opts->frameBlend;
      px = M::mix(px, col.x, fb);
      py = M::mix(py, col.y, fb);
      pz = M::mix(pz, col.z, fb);
    }
  }
  if (first) {
    a.acc[at] = make_float4(px, py, pz, 1.0f);
    if (a.argb) a.argb[id] = tonemap_argb<M>(px, py, pz, a.opts0->gamma);
  }
}

#ifdef RM_XCD_CLOCK
// DIAGNOSTIC BUILD ONLY (tools/ab_build.py clock=-DRM_XCD_CLOCK, tools/xcd_clock.py): when each XCD started and finished its share
// of the frame kernel's launches, on the 100 MHz constant clock.  [x]: first start, [16 + x]: last end, [32 + x]: sum of the
// wavefronts' lifetimes, [48 + x]: wavefronts (low 32 bits) and those whose block index mod 8 is NOT x (high 32 bits).
__device__ unsigned long long g_xcd_clock[64];
// (the first reading as asm that is neither volatile nor clobbers memory: behind wall_clock64() -- or any atomic, volatile asm
//  or store -- the compiler's pass that marks loads as not clobbered gives up, the 800 uniform loads of the kernel (the render
//  options) become vector loads and the allocator answers with 100 spilled VGPRs instead of 23: the first version of this
//  diagnostic ran 47 % slower than the product, this one 4 %)
__device__ __forceinline__ unsigned long long xcd_clock_now(unsigned after) {
  unsigned long long t;
  asm("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0) ; after %1" : "=s"(t) : "s"(after));
  return t;
}
__device__ __forceinline__ void xcd_clock_note(unsigned long long t0) {
  const unsigned long long t1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) {
    const unsigned x = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;  // HW_REG_XCC_ID[3:0]
    atomicMin(&g_xcd_clock[x], t0);
    atomicMax(&g_xcd_clock[16 + x], t1);
    atomicAdd(&g_xcd_clock[32 + x], t1 - t0);
    atomicAdd(&g_xcd_clock[48 + x], (blockIdx.x & 7) == x ? 1ull : (1ull << 32) + 1ull);
  }
}
#endif

template <bool ACCEL, int MINW, bool SDFM, int LAYOUT = 0, int ARITH = 0>
__global__ __launch_bounds__(64 * kWavesPerBlock, MINW) void render_frame_kernel(const FrameArgs a) {
  static_assert(kWavesPerBlock == 1, "the LDS area below belongs to one wavefront");
  __shared__ float wave_lds[(ACCEL || (SDFM && RM_SDF_WAVE)) ? rmk::Tracer<false, ACCEL, SDFM>::kWaveLdsFloats : 3 * 64];
#ifdef RM_XCD_CLOCK
  const unsigned long long t0 = xcd_clock_now(blockIdx.x);
  frame_block<ACCEL, SDFM, LAYOUT, ARITH>(a, blockIdx.x + (t0 == ~0ull ? 1 : 0), wave_lds);  // (pins the first reading to the start)
  xcd_clock_note(t0);
#else
  frame_block<ACCEL, SDFM, LAYOUT, ARITH>(a, blockIdx.x, wave_lds);
#endif
}

template <int ARITH>
__global__ __launch_bounds__(256) void tonemap_kernel(const float4* __restrict__ pixels,
                                                      const RmOpts* __restrict__ opts,
                                                      uint32_t* __restrict__ argb, int n) {
  const float g = opts->gamma;
  for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n;
       id += (long long)gridDim.x * blockDim.x) {
    const float4 p = pixels[id];
    argb[id] = tonemap_argb<typename rmk::ArithOf<ARITH>::type>(p.x, p.y, p.z, g);
  }
}

// Tile-major accumulators of `parts` interleaved partitions (partition r owns
// tiles r, r+parts, ...; each partition's buffer holds tiles_per_part tiles of
// 64 float4) -> row-major float4 pixels and/or tonemapped ARGB.
template <int ARITH>
__global__ __launch_bounds__(256) void resolve_kernel(const float4* __restrict__ tiles, int parts,
                                                      int tiles_per_part,
                                                      const RmOpts* __restrict__ opts,
                                                      float4* __restrict__ pixels,
                                                      uint32_t* __restrict__ argb, int n) {
  const int resx = opts->resolution[0];
  const int tiles_x = (resx + kTile - 1) / kTile;
  const float g = opts->gamma;
  for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n;
       id += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(id % resx), y = (int)(id / resx);
    const int tile = (y >> 3) * tiles_x + (x >> 3);
    const int lane = ((y & 7) << 3) | (x & 7);
    const long long at = ((long long)(tile % parts) * tiles_per_part + tile / parts) * 64 + lane;
    const float4 p = tiles[at];
    if (pixels) pixels[id] = p;
    if (argb) argb[id] = tonemap_argb<typename rmk::ArithOf<ARITH>::type>(p.x, p.y, p.z, g);
  }
}

// Tile-major ARGB words of `parts` interleaved partitions -> the row-major ARGB image (frames exchanged
// as tonemapped words: 4 bytes per pixel over the links instead of 16).
__global__ __launch_bounds__(256) void resolve_argb_kernel(const uint32_t* __restrict__ tiles, int parts, int tiles_per_part,
                                                           int resx, uint32_t* __restrict__ argb, int n) {
  const int tiles_x = (resx + kTile - 1) / kTile;
  for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n;
       id += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(id % resx), y = (int)(id / resx);
    const int tile = (y >> 3) * tiles_x + (x >> 3);
    const int lane = ((y & 7) << 3) | (x & 7);
    argb[id] = tiles[((long long)(tile % parts) * tiles_per_part + tile / parts) * 64 + lane];
  }
}

__global__ void prims_kernel(int op, const float* __restrict__ a, const float* __restrict__ b,
                             uint32_t* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = a[i];
  const float y = b ? b[i] : 0.0f;
  uint32_t r = 0;
  switch (op) {
    case 0: r = __float_as_uint(x / y); break;
    case 1: r = __float_as_uint(rmd::sqrt_rn(x)); break;
    case 2: r = __float_as_uint(rmd::exp_det(x)); break;
    case 3: r = __float_as_uint(rmd::exp2_det(x)); break;
    case 4: r = __float_as_uint(rmd::pow_det(x, y)); break;
    case 5: r = (uint32_t)rmd::f2i(x); break;
    case 6: r = rmd::f2u(x); break;
    case 7: r = (uint32_t)rmd::convert_int_sat(x); break;
    case 8: r = __float_as_uint(x * y + x); break;
    case 9: r = __float_as_uint(rmd::div_by(x, rmd::make_divisor(y))); break;
    default: break;
  }
  out[i] = r;
}

// Device side of rm_selftest_filter: the decisions of the slab-test filter and of the
// inside-the-box shortcut next to the exact slab test they stand in for.
__global__ void filter_check_kernel(const float* __restrict__ rays, const RmOpts* __restrict__ opts,
                                    uint32_t* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = rays + (size_t)i * 8;
  const rmk::v3 ro = rmk::V(r[0], r[1], r[2]), rd = rmk::V(r[3], r[4], r[5]);
  const float t = r[6], g = r[7];
  rmk::Scene sc{nullptr, nullptr, opts, nullptr, nullptr};
  rmk::Tracer<false, true> tr(sc);
  const auto flt = tr.make_filter(ro, rd);
  const rmk::v3 rpos = rmk::muladd(rd, t, ro);  // the position march() hands to the estimate
  const float t_in = rmk::box_entry_of<rmk::MathX86<0>>(*opts, rpos, rd);
  const RmOpts& o = *opts;
  const float m = 1e-4f;  // the margin of scene_distance's inside-the-box shortcut
  const bool inside = (rpos.x - o.voxelBoundsMin[0] > m) & (o.voxelBoundsMax[0] - rpos.x > m) &
                      (rpos.y - o.voxelBoundsMin[1] > m) & (o.voxelBoundsMax[1] - rpos.y > m) &
                      (rpos.z - o.voxelBoundsMin[2] > m) & (o.voxelBoundsMax[2] - rpos.z > m);
  uint32_t bits = 0;
  if (tr.surely_no_walk(flt, t, g)) bits |= 1u;
  if (t_in >= 0.0f && t_in < g) bits |= 4u;                    // renderer.cl:214: the estimate walks
  if (__float_as_uint(t_in) == 0u) bits |= 8u;                 // the slab test returned exactly +0
  if (inside) bits |= 16u;
  out[i] = bits;
}

}  // namespace

namespace rmk {

hipError_t launch_filter_check(hipStream_t st, const float* rays, const RmOpts* d_opts, uint32_t* out, int n) {
  if (n <= 0) return hipSuccess;
  filter_check_kernel<<<(n + 255) / 256, 256, 0, st>>>(rays, d_opts, out, n);
  return hipGetLastError();
}

int tiles_total(int resx, int n) { return tile_geom(resx, n).tiles_total; }

// ---- run-time -> compile-time: ONE switch per template axis, used by every launcher ----
template <int V> using ic = std::integral_constant<int, V>;
// the contract of a context (rm_api.hip contract_arith) -> ArithOf index
// (A/B builds: -DRM_ONLY_ARITH=a / -DRM_ONLY_LAYOUT=l instantiate the FRAME kernel for that contract / table layout
//  alone; -DRM_ONLY_FRAME leaves the single-pass kernels out; other launches fail with hipErrorInvalidValue)
template <bool FRAME, class F>
void with_arith(int arith, F&& fn) {
#ifdef RM_ONLY_ARITH
  if constexpr (FRAME) {
    if (arith == RM_ONLY_ARITH) fn(ic<RM_ONLY_ARITH>{});
    return;
  } else
#endif
  switch (arith) {
    case 3: fn(ic<3>{}); break;
    case 2: fn(ic<2>{}); break;
    case 1: fn(ic<1>{}); break;
    default: fn(ic<0>{}); break;
  }
}
template <class F>
void with_layout(int layout, F&& fn) {
#ifdef RM_ONLY_LAYOUT
  if (layout == RM_ONLY_LAYOUT) fn(ic<RM_ONLY_LAYOUT>{});
#else
  switch (layout) {
    case 1: fn(ic<1>{}); break;
    case 2: fn(ic<2>{}); break;
    case 3: fn(ic<3>{}); break;
    case 4: fn(ic<4>{}); break;
    case 5: fn(ic<5>{}); break;
    default: fn(ic<0>{}); break;
  }
#endif
}
// table layout of a volume's derived structures (walk_step); `frame`: layout 5 exists in the frame kernel only
int layout_of(const rmk::Accel& accel, bool frame) {
  if (accel.bricked) return accel.log2res == 9 ? 3 : (accel.log2res == 10 ? 4 : 1);
  if (frame && accel.log2res == 8 && accel.oct_stride) return 5;
  return accel.log2res ? 2 : 0;
}

// Does the frame kernel of this volume's table layout take records with ANY number of AO probes (rm_shade.hpp
// kChunkedAO: layouts 3, 4, 5)?  Otherwise frames whose records ask for more than RM_WAVE_AO_PROBES go pass by pass
// through the single-pass kernels (rm_api.hip frame_on_device).
bool frame_takes_any_ao(const rmk::Accel& accel) { return rmk::fixed_log2(layout_of(accel, true)) != 0; }

hipError_t launch_render_pass(hipStream_t st, const uint8_t* vox, Accel accel, const float* mc,
                              const RmOpts* d_opts, int resx, float* pixels, int n, int id0,
                              int id1, int tile_first, int tile_stride, bool tile_major,
                              Counters* d_counters, int arith) {
  const TileGeom g = tile_geom(resx, n);
  if (tile_stride < 1) tile_stride = 1;
  const long long my_tiles =
      tile_first >= g.tiles_total ? 0 : (g.tiles_total - tile_first + tile_stride - 1) / tile_stride;
  if (my_tiles == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((my_tiles + kWavesPerBlock - 1) / kWavesPerBlock);
  const float4* mc4 = reinterpret_cast<const float4*>(mc);
  float4* px4 = reinterpret_cast<float4*>(pixels);
  const dim3 grid(blocks), block(64 * kWavesPerBlock);
  const bool acc = accel.dist && accel.surf && !d_counters;  // (event counts are defined on the plain algorithm)
#ifdef RM_ONLY_FRAME  // (A/B screens of the frame kernel alone: no single-pass instantiations)
  return hipErrorInvalidValue;
#else
  with_arith<false>(arith, [&](auto A) {
    auto go = [&](auto C, auto T, auto AC, auto L) {
      render_pass_kernel<(decltype(C)::value != 0), (decltype(T)::value != 0), (decltype(AC)::value != 0), decltype(L)::value,
                         decltype(A)::value><<<grid, block, 0, st>>>(vox, accel.dist, accel.surf, mc4, d_opts, px4, n, id0, id1,
                                                                     tile_first, tile_stride, d_counters, accel.oct_stride,
                                                                     accel.log2res);
    };
    if (d_counters) go(ic<1>{}, ic<0>{}, ic<0>{}, ic<0>{});
    else if (!acc) { if (tile_major) go(ic<0>{}, ic<1>{}, ic<0>{}, ic<0>{}); else go(ic<0>{}, ic<0>{}, ic<0>{}, ic<0>{}); }
    else with_layout(layout_of(accel, false), [&](auto L) {
      if constexpr (decltype(L)::value != 5) {  // (layout 5: frame kernel only)
        if (tile_major) go(ic<0>{}, ic<1>{}, ic<1>{}, L); else go(ic<0>{}, ic<0>{}, ic<1>{}, L);
      }
    });
  });
  return hipGetLastError();
#endif
}

// log2 of the passes per wavefront (= per launch) for a run of `passes` records that differ in .time only:
// the largest k <= max_log2 whose last group leaves at most waste_pct % of the run's lane turns
// without a pass (k = 0 never leaves any).  Lanes without a pass still trace the other lanes'
// secondary rays: 25 passes as 16 + 9 measured 12 % faster than as 6 x 4 + 1.
int choose_pass_pack(int passes, int max_log2, int waste_pct) {
  for (int k = max_log2; k >= 1; k--) {
    const int pp = 1 << k;
    if ((double)(((passes + pp - 1) / pp) * pp) * 100.0 <= (100.0 + waste_pct) * (double)passes) return k;
  }
  return 0;
}

// grid of a frame launch: one wavefront per workgroup; *bpr_out = blocks per tile row of the XCD-aware order (0: plain)
static long long frame_grid(const FrameLaunch& f, int* bpr_out, int* pp_log2_out, int* rows_out = nullptr) {
  const TileGeom g = tile_geom(f.resx, f.n);
  const int tile_stride = f.tile_stride < 1 ? 1 : f.tile_stride;
  const long long my_tiles =
      f.tile_first >= g.tiles_total ? 0 : (g.tiles_total - f.tile_first + tile_stride - 1) / tile_stride;
  const int pp_log2 = f.pp_log2 < 0 ? 0 : (f.pp_log2 > 6 ? 6 : f.pp_log2);
  if (pp_log2_out) *pp_log2_out = pp_log2;
  if (bpr_out) *bpr_out = 0;
  if (my_tiles == 0 || f.passes <= 0) return 0;
  long long blocks = my_tiles << pp_log2;
  // blocks per tile row, for the XCD-aware order.  A partition (first, stride) whose
  // stride divides the row length owns tiles_x/stride tiles of every row -- columns of
  // tiles -- so its local slots still form rows and the same order applies.
  if (f.xcd_rows && g.tiles_x % tile_stride == 0 && f.tile_first < tile_stride) {
    const int bpr = (int)((long long)(g.tiles_x / tile_stride) << pp_log2);
    const long long rows = (blocks + bpr - 1) / bpr;
    // whole groups of 8 rows (one row per XCD) + the blocks of the last rows in eight equal shares (frame_block)
    const long long groups = rows / 8, tail_blocks = (rows % 8) * bpr, share = (tail_blocks + 7) / 8;
    blocks = groups * 8 * bpr + 8 * share;
    if (bpr_out) *bpr_out = bpr;
    if (rows_out) { rows_out[0] = (int)rows; rows_out[1] = (int)groups; rows_out[2] = (int)share; rows_out[3] = 0; }
    // 2-D units: stripes must hold whole tiles.  Unit width 1/(8 nsu) of a row; auto (xcd_2d < 0): the narrowest unit that
    // still holds unit_min_waves wavefronts -- by default half of what an XCD has in flight (7 per SIMD x 4 x 32 CUs = 896).
    // Measured: config 2 (640 wavefronts per unit at nsu = 1, 320 at 2) is fastest at 1, config 3 (960 / 480) at 2 (-4.6 %
    // against 1; 3 / 5 / 6: +2 / -0.4 / +0.2 %), config 4 (1920 / 960 / 480) within 0.4 % for 1 .. 6.  Narrow units balance the
    // XCDs better, wide ones share more lines in an XCD's L2; a volume whose tables no cache holds (1024^3: 9 GiB) has only
    // the first to gain and asks for 128: config 5 at nsu = 6 (160 wavefronts) 31.2 ms against 32.0-32.4 at 2
    {
      const int tpr = g.tiles_x / tile_stride;
      int nsu = f.xcd_2d;
      if (nsu < 0) {
        nsu = 0;
        const long long least = f.unit_min_waves > 0 ? f.unit_min_waves : 448;
        for (int c = 1; c <= 8; ++c)
          if (tpr % (8 * c) == 0 && (c == 1 || 2ll * (tpr / (8 * c)) * (1ll << pp_log2) >= least)) nsu = c;
      }
      if (nsu > 0 && tpr % (8 * nsu) == 0) {
        blocks = rows * bpr;
        if (rows_out) rows_out[3] = nsu;
      }
    }
  }
  return blocks;
}

// the part of a launch's arguments that decides which hardware block renders what (logical_block); -> blocks of the grid
static long long frame_order_args(const FrameLaunch& f, FrameArgs& a) {
  const TileGeom g = tile_geom(f.resx, f.n);
  const int tile_stride = f.tile_stride < 1 ? 1 : f.tile_stride;
  int bpr = 0, pp_log2 = 0;
  int rows3[4] = {0, 0, 0, 0};
  const long long blocks = frame_grid(f, &bpr, &pp_log2, rows3);
  a.rows_real = rows3[0]; a.full_groups = rows3[1]; a.tail_share = rows3[2]; a.xcd2d = rows3[3];
  a.n = f.n; a.resx = f.resx; a.tile_first = f.tile_first; a.tile_stride = tile_stride;
  a.tiles_per_part = tiles_per_part(g.tiles_total, tile_stride); a.pp_log2 = pp_log2; a.passes = f.passes; a.bpr = bpr;
  a.rows_desc = f.rows_desc ? 1 : 0;
  a.band_r0 = a.band_r1 = 0;
  if (bpr > 0 && f.rows_desc && f.band_hi > f.band_lo) {  // image-height fractions -> rows of this launch's grid
    const long long rows_real = a.rows_real;
    long long r0 = (long long)(f.band_lo * (double)rows_real), r1 = (long long)(f.band_hi * (double)rows_real) + 1;
    r0 = r0 < 0 ? 0 : r0;
    r1 = r1 > rows_real ? rows_real : r1;
    if (r1 > r0) { a.band_r0 = (int)r0; a.band_r1 = (int)r1; }
  }
  return blocks;
}

// rm_debug_block_order: what every hardware block of that launch would render -- out[b] = tile << 8 | sub-block, or -1 for a
// block that leaves at once (padding).  Host arithmetic only: the same logical_block() the kernel compiles.
long long debug_block_order(const FrameLaunch& f, long long* out, long long cap) {
  FrameArgs a{};
  const long long blocks = frame_order_args(f, a);
  const TileGeom g = tile_geom(f.resx, f.n);
  for (long long b = 0; b < blocks && b < cap; ++b) {
    const long long lb = logical_block(a, b);
    out[b] = -1;
    if (lb < 0) continue;
    const long long slot = lb >> a.pp_log2, tile = a.tile_first + slot * a.tile_stride;
    if (slot >= a.tiles_per_part || tile >= g.tiles_total) continue;
    out[b] = (tile << 8) | (lb & ((1 << a.pp_log2) - 1));
  }
  return blocks;
}

hipError_t launch_render_frame(hipStream_t st, const FrameLaunch& f) {
  FrameArgs a;
  const long long blocks = frame_order_args(f, a);
  if (blocks == 0) return hipSuccess;
  const int pp_log2 = a.pp_log2;
  a.vox = f.vox;
  a.dist8 = f.accel.dist;
  a.surf32 = f.accel.surf;
  a.oct_stride = f.accel.oct_stride;
  a.sdf = f.sdf;
  a.mc_all = reinterpret_cast<const float4*>(f.mc_all);
  a.opts_all = f.opts_all;
  a.opts0 = f.opts0;
  a.acc = reinterpret_cast<float4*>(f.acc);
  a.argb = f.argb;
  a.accumulate = f.accumulate ? 1 : 0;
  a.row_major = f.row_major ? 1 : 0;
  const dim3 grid((unsigned)blocks), block(64 * kWavesPerBlock);
  if (f.passes > (1 << pp_log2)) return hipErrorInvalidValue;  // (the caller splits a run into such launches)
// (7 wavefronts per SIMD = 72 VGPRs measured best: 6 +3.5 %, 8 +2.5 %, 5 +14 %; -DRM_FRAME_MINW=n re-measures)
#ifndef RM_FRAME_MINW
#define RM_FRAME_MINW 7
#endif
#ifndef RM_SDF_MINW
#define RM_SDF_MINW 5  // (quality mode: 4 / 5 / 6 waves per SIMD measured 9.21 / 9.06 / 9.09 ms)
#endif
  a.log2res = f.accel.log2res;
  const bool acc = f.accel.dist && f.accel.surf;
  if (f.sdf) {  // (quality mode: its own algorithm, CPU-device arithmetic only)
    if (f.arith == 1) render_frame_kernel<false, RM_SDF_MINW, true, 0, 1><<<grid, block, 0, st>>>(a);
    else render_frame_kernel<false, RM_SDF_MINW, true, 0, 0><<<grid, block, 0, st>>>(a);
  } else {
    with_arith<true>(f.arith, [&](auto A) {
      if (!acc) render_frame_kernel<false, 3, false, 0, decltype(A)::value><<<grid, block, 0, st>>>(a);
      else with_layout(layout_of(f.accel, true), [&](auto L) {
        render_frame_kernel<true, RM_FRAME_MINW, false, decltype(L)::value, decltype(A)::value><<<grid, block, 0, st>>>(a);
      });
    });
  }
  return hipGetLastError();
}

hipError_t launch_tonemap(hipStream_t st, const float* pixels, const RmOpts* d_opts, uint32_t* argb,
                          int n, int arith) {
  if (n <= 0) return hipSuccess;
  int blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  with_arith<false>(arith >= 2 ? arith : 0, [&](auto A) {  // (no seeds in TonemapImage: one CPU-device instantiation)
    tonemap_kernel<decltype(A)::value><<<blocks, 256, 0, st>>>(reinterpret_cast<const float4*>(pixels), d_opts, argb, n);
  });
  return hipGetLastError();
}

hipError_t launch_resolve(hipStream_t st, const float* tiles, int parts, int tiles_per_part,
                          const RmOpts* d_opts0, float* pixels, uint32_t* argb, int n, int arith) {
  if (n <= 0) return hipSuccess;
  int blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  with_arith<false>(arith >= 2 ? arith : 0, [&](auto A) {
    resolve_kernel<decltype(A)::value><<<blocks, 256, 0, st>>>(reinterpret_cast<const float4*>(tiles), parts, tiles_per_part,
                                                               d_opts0, reinterpret_cast<float4*>(pixels), argb, n);
  });
  return hipGetLastError();
}

hipError_t launch_resolve_argb(hipStream_t st, const uint32_t* tiles, int parts, int tiles_per_part, int resx,
                               uint32_t* argb, int n) {
  if (n <= 0) return hipSuccess;
  int blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  resolve_argb_kernel<<<blocks, 256, 0, st>>>(tiles, parts, tiles_per_part, resx, argb, n);
  return hipGetLastError();
}

hipError_t launch_prims(hipStream_t st, int op, const float* a, const float* b, uint32_t* out,
                        int n) {
  if (n <= 0) return hipSuccess;
  prims_kernel<<<(n + 255) / 256, 256, 0, st>>>(op, a, b, out, n);
  return hipGetLastError();
}

}  // namespace rmk

#ifdef RM_XCD_CLOCK
extern "C" int rm_debug_xcd_clock(unsigned long long* out64, int reset) {
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess && out64) e = hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_xcd_clock), 64 * sizeof(unsigned long long));
  if (e == hipSuccess && reset) {
    unsigned long long init[64];
    for (int i = 0; i < 64; ++i) init[i] = i < 16 ? ~0ull : 0ull;
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_xcd_clock), init, sizeof(init));
  }
  return (int)e;
}
#endif
