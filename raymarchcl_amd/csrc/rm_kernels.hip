// rm_kernels.hip -- gfx950 kernels of the render path and their launchers.
//
//   render_pass_kernel   == one NDRange of the reference's RenderImage
//                           (renderer.cl:478-494): one lane per sample, the
//                           accumulator is blended in place.
//   tonemap_kernel       == TonemapImage (renderer.cl:448-454, 496-508).
//   prims_kernel         -- device side of rm_selftest_prims.
//
// Work decomposition: the image is cut into 8x8-pixel tiles (row-major tile
// order); one 64-lane wavefront owns one tile so that the rays of a wave are
// spatially coherent (their voxel fetches share cache lines), four tiles per
// 256-thread workgroup.  Everything that is uniform across the launch (the
// 544-byte option record) is read through a uniform pointer, i.e. by scalar
// loads into SGPRs -- the reference's per-work-item private copy of the record
// is what costs it 560 B of scratch per lane on this chip.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "rm_kernels.h"
#include "rm_shade.hpp"
#include "rm_wave.hpp"

#if defined(RM_WORK_STATS) || defined(RM_PHASE_CLOCK)
// debug build only (hipcc -DRM_WORK_STATS): what render_samples_kernel executes, summed
// over all lanes: samples, outer marches, their turns, filtered turns, voxel walks,
// dist8 fetches, samples advanced, AO loops.  rmk::dump_work_stats() prints and resets.
__device__ unsigned long long g_work_stats[64];
#endif

#ifndef RM_WAVE_SHARE
#define RM_WAVE_SHARE 1  // AO probes and shadow rays of a wavefront's hits dealt to all its lanes
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
template <bool COUNT, bool TILE_MAJOR, bool ACCEL>
__global__ __launch_bounds__(64 * kWavesPerBlock) void render_pass_kernel(
    const uint8_t* __restrict__ vox, const uint8_t* __restrict__ dist8,
    const uint32_t* __restrict__ surf32, const float4* __restrict__ mc,
    const RmOpts* __restrict__ opts,
    float4* __restrict__ pixels, int n, int id0, int id1, int tile_first, int tile_stride,
    rmk::Counters* __restrict__ counters) {
  const int resx = opts->resolution[0];
  const TileGeom g = tile_geom(resx, n);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long slot = (long long)blockIdx.x * kWavesPerBlock + wave;
  const long long tile = tile_first + slot * tile_stride;
  if (tile >= g.tiles_total) return;
  const int id = lane_pixel((int)tile, lane, resx, g.tiles_x, n, id0, id1);
  rmk::Scene sc{vox, mc, opts, dist8, surf32};
  rmk::Tracer<COUNT, ACCEL> tr(sc);
  if (id >= 0) {
    const rmk::v3 col = tr.shade(id);
    const float fb = opts->frameBlend;
    const long long at = TILE_MAJOR ? slot * 64 + lane : (long long)id;
    const float4 p = pixels[at];
    // mix(p, col, frameBlend): renderer.cl:492
    pixels[at] = make_float4(p.x + (col.x - p.x) * fb, p.y + (col.y - p.y) * fb,
                             p.z + (col.z - p.z) * fb, 1.0f);
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

// All passes of a frame in ONE launch: blockIdx.y = pass, blockIdx.x covers the
// partition's tiles.  Samples of different passes are independent until the
// blend, so each lane writes its colour*exposure to a staging slot
// [pass][local tile][lane] and blend_kernel applies the reference's in-order
// recurrence afterwards.  Compared with one launch per pass this removes the
// per-pass tail (waves of the next pass fill the CUs while expensive tiles of
// the previous one finish) and costs the same 32 B per sample the reference
// spends on its read-modify-write of the accumulator.
// Lane -> (pixel, pass).  With pp = 1 a wavefront is one 8x8 tile of one pass.
// With pp = 2^k > 1 (possible when the passes' records differ only in .time) it
// is a (64/pp)-pixel block of the tile times pp consecutive passes: lanes that
// trace the SAME pixel with different jitter follow almost the same control flow,
// which is what a 64-wide SIMT machine wants -- neighbouring pixels of one pass
// diverge far more (sky / surface / reflection) than passes of one pixel.
template <bool ACCEL, int MINW, bool SDFM = false>
__global__ __launch_bounds__(64 * kWavesPerBlock, MINW) void render_samples_kernel(
    const uint8_t* __restrict__ vox, const uint8_t* __restrict__ dist8,
    const uint32_t* __restrict__ surf32, const float4* __restrict__ mc_all,
    const RmOpts* __restrict__ opts_all, float4* __restrict__ staging, int n, int tile_first,
    int tile_stride, int tiles_per_part, int pp_log2, int bpr, unsigned long long oct_stride,
    const float* __restrict__ sdf = nullptr) {
  const int pp = 1 << pp_log2;              // passes per wavefront
  const int ppw = 64 >> pp_log2;            // pixels per wavefront
  const int pass0 = blockIdx.y * pp;
  const RmOpts* __restrict__ opts = opts_all + pass0;  // uniform record (pp > 1: all equal but .time)
  const int resx = opts->resolution[0];
  const TileGeom g = tile_geom(resx, n);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // XCD-aware order (bpr > 0): the dispatcher deals consecutive workgroups to the 8
  // XCDs round-robin, each with its own L2.  Hardware block b = 8*m + k is given the
  // logical block of tile row 8*(m / bpr) + k, so XCD k renders every 8th tile ROW:
  // its primary rays sweep an eighth of the volume's slabs instead of all of them,
  // while rows stay interleaved finely enough to balance the load.
  long long lb = blockIdx.x;
  if (bpr > 0) {
    const long long m = lb >> 3, k = lb & 7;
    lb = ((m / bpr) * 8 + k) * bpr + (m % bpr);
  }
  // wavefronts of a tile are consecutive: tile slot = w / pp, sub-block = w % pp
  const long long w = lb * kWavesPerBlock + wave;
  const long long slot = w >> pp_log2;
  const int sub = (int)(w & (pp - 1));
  const long long tile = tile_first + slot * tile_stride;
  if (tile >= g.tiles_total) return;
  const int pass = pass0 + (lane & (pp - 1));
  // Z-order inside the tile, so that any 2^k consecutive pixels form a compact block
  const int z = sub * ppw + (lane >> pp_log2);
  const int zx = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4);
  const int zy = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
  const int pix = zy * 8 + zx;  // 0..63 within the tile, row-major 8x8
  const int id = lane_pixel((int)tile, pix, resx, g.tiles_x, n, 0, n);
  if (id < 0) return;
  rmk::Scene sc{vox, mc_all + (size_t)pass0 * RM_TABLE_ENTRIES, opts, dist8, surf32, oct_stride, sdf};
  rmk::Tracer<false, ACCEL, SDFM> tr(sc);
  if (pp > 1) tr.set_pass(mc_all + (size_t)pass * RM_TABLE_ENTRIES, opts_all[pass].time);
#if RM_WAVE_SHARE
  static_assert(kWavesPerBlock == 1, "the shared phases use one LDS block per workgroup");
  __shared__ float wave_lds[ACCEL ? rmk::Tracer<false, ACCEL, SDFM>::kWaveLdsFloats : 1];
  const rmk::v3 col = ACCEL ? tr.shade_wave(id, wave_lds) : tr.shade(id);
#else
  const rmk::v3 col = tr.shade(id);
#endif
  staging[((long long)pass * tiles_per_part + slot) * 64 + pix] = make_float4(col.x, col.y, col.z, 1.0f);
#ifdef RM_PHASE_CLOCK
  for (int k = 0; k < 5; k++)
    if (tr.ws_clk[k]) atomicAdd(&g_work_stats[32 + k], tr.ws_clk[k]);
  if (tr.ws_now()) atomicAdd(&g_work_stats[37], 1ull);
#endif
#ifdef RM_WORK_STATS
  {
    atomicAdd(&g_work_stats[43], (unsigned long long)tr.ws_redo);
    for (int k = 0; k < 3; k++) {
      atomicAdd(&g_work_stats[48 + k], (unsigned long long)tr.ws_k_est[k]);
      atomicAdd(&g_work_stats[52 + k], (unsigned long long)tr.ws_k_filt[k]);
    }
    for (int k = 0; k < 4; k++) atomicAdd(&g_work_stats[44 + k], (unsigned long long)tr.ws_k_nohit[k]);
    atomicAdd(&g_work_stats[38], (unsigned long long)(tr.ws_k_one[0] + tr.ws_k_one[1]));
    atomicAdd(&g_work_stats[39], (unsigned long long)tr.ws_k_one[2]);
    atomicAdd(&g_work_stats[40], (unsigned long long)tr.ws_pairs);
    atomicAdd(&g_work_stats[41], (unsigned long long)tr.ws_pairs_back);
    atomicAdd(&g_work_stats[42], (unsigned long long)tr.ws_pairs_dark);
    unsigned int v[32] = {1u, tr.ws_rays, tr.ws_iters, tr.ws_filtered, tr.ws_walks, tr.ws_lookups,
                          tr.ws_steps, tr.ws_probes, tr.wv_walk, tr.wv_filt, tr.wv_est};
    for (int k = 0; k < 4; k++) {
      v[11 + k] = tr.ws_k_walks[k];
      v[15 + k] = tr.ws_k_fetch[k];
      v[19 + k] = tr.ws_k_slots[k];
    }
    for (int k = 0; k < 6; k++) v[23 + k] = tr.ws_dhist[k];
    v[29] = tr.ws_adds_hit; v[30] = tr.ws_adds_nohit; v[31] = tr.ws_adds_lazy;
    for (int k = 0; k < 32; k++) atomicAdd(&g_work_stats[k], (unsigned long long)v[k]);
  }
#endif
}

// The two halves of a sample (Tracer::trace_chain / shade_from_hits): same grid
// and tile order as render_samples_kernel, hit records in HBM in between.
template <int MINW>
__global__ __launch_bounds__(64 * kWavesPerBlock, MINW) void trace_chain_kernel(
    const uint8_t* __restrict__ vox, const uint8_t* __restrict__ dist8,
    const uint32_t* __restrict__ surf32, const float4* __restrict__ mc_all,
    const RmOpts* __restrict__ opts_all, float4* __restrict__ hits, int n, int tile_first,
    int tile_stride, int tiles_per_part) {
  const int pass = blockIdx.y;
  const RmOpts* __restrict__ opts = opts_all + pass;
  const int resx = opts->resolution[0];
  const TileGeom g = tile_geom(resx, n);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long slot = (long long)blockIdx.x * kWavesPerBlock + wave;
  const long long tile = tile_first + slot * tile_stride;
  if (tile >= g.tiles_total) return;
  const int id = lane_pixel((int)tile, lane, resx, g.tiles_x, n, 0, n);
  if (id < 0) return;
  rmk::Scene sc{vox, mc_all + (size_t)pass * RM_TABLE_ENTRIES, opts, dist8, surf32};
  rmk::Tracer<false, true> tr(sc);
  const size_t samples = (size_t)gridDim.y * tiles_per_part * 64;
  tr.trace_chain(id, hits, samples, ((size_t)pass * tiles_per_part + slot) * 64 + lane);
}

template <int MINW>
__global__ __launch_bounds__(64 * kWavesPerBlock, MINW) void light_kernel(
    const uint8_t* __restrict__ vox, const uint8_t* __restrict__ dist8,
    const uint32_t* __restrict__ surf32, const float4* __restrict__ mc_all,
    const RmOpts* __restrict__ opts_all, const float4* __restrict__ hits,
    float4* __restrict__ staging, int n, int tile_first, int tile_stride, int tiles_per_part) {
  const int pass = blockIdx.y;
  const RmOpts* __restrict__ opts = opts_all + pass;
  const int resx = opts->resolution[0];
  const TileGeom g = tile_geom(resx, n);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long slot = (long long)blockIdx.x * kWavesPerBlock + wave;
  const long long tile = tile_first + slot * tile_stride;
  if (tile >= g.tiles_total) return;
  const int id = lane_pixel((int)tile, lane, resx, g.tiles_x, n, 0, n);
  if (id < 0) return;
  rmk::Scene sc{vox, mc_all + (size_t)pass * RM_TABLE_ENTRIES, opts, dist8, surf32};
  rmk::Tracer<false, true> tr(sc);
  const size_t samples = (size_t)gridDim.y * tiles_per_part * 64;
  const size_t sidx = ((size_t)pass * tiles_per_part + slot) * 64 + lane;
  const rmk::v3 col = tr.shade_from_hits(id, hits, samples, sidx);
  staging[sidx] = make_float4(col.x, col.y, col.z, 1.0f);
}

// Three-phase form (Tracer::trace_chain / point_rays / shade_from_rays).  All three use
// the pass-packed lane mapping of render_samples_kernel; phase 2 has a z dimension
// over the shading levels (0 = primary hit, k = reflection k).
struct LaneMap { int id, pass; long long sidx; };
__device__ __forceinline__ LaneMap lane_map(const RmOpts* opts_all, int n, int tile_first, int tile_stride,
                                            int tiles_per_part, int pp_log2) {
  LaneMap r;
  r.id = -1;
  const int pp = 1 << pp_log2, ppw = 64 >> pp_log2;
  const int pass0 = blockIdx.y * pp;
  const int resx = opts_all[pass0].resolution[0];
  const TileGeom g = tile_geom(resx, n);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long w = (long long)blockIdx.x * kWavesPerBlock + wave;
  const long long slot = w >> pp_log2;
  const int sub = (int)(w & (pp - 1));
  const long long tile = tile_first + slot * tile_stride;
  r.pass = pass0 + (lane & (pp - 1));
  r.sidx = 0;
  if (tile >= g.tiles_total) return r;
  const int z = sub * ppw + (lane >> pp_log2);
  const int zx = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4);
  const int zy = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
  const int pix = zy * 8 + zx;
  r.id = lane_pixel((int)tile, pix, resx, g.tiles_x, n, 0, n);
  r.sidx = ((long long)r.pass * tiles_per_part + slot) * 64 + pix;
  return r;
}

template <int PHASE>
__global__ __launch_bounds__(64 * kWavesPerBlock, 8) void phase_kernel(
    const uint8_t* __restrict__ vox, const uint8_t* __restrict__ dist8,
    const uint32_t* __restrict__ surf32, const float4* __restrict__ mc_all,
    const RmOpts* __restrict__ opts_all, float4* __restrict__ hits, float2* __restrict__ rays,
    float4* __restrict__ staging, int n, int tile_first, int tile_stride, int tiles_per_part,
    int pp_log2, int iter) {
  const LaneMap lm = lane_map(opts_all, n, tile_first, tile_stride, tiles_per_part, pp_log2);
  if (lm.id < 0) return;
  const int pass0 = blockIdx.y << pp_log2;
  const RmOpts* __restrict__ opts = opts_all + pass0;
  rmk::Scene sc{vox, mc_all + (size_t)pass0 * RM_TABLE_ENTRIES, opts, dist8, surf32};
  rmk::Tracer<false, true> tr(sc);
  if (pp_log2 > 0) tr.set_pass(mc_all + (size_t)lm.pass * RM_TABLE_ENTRIES, opts_all[lm.pass].time);
  const size_t samples = (size_t)iter * tiles_per_part * 64;
  if (PHASE == 1) {
    tr.trace_chain(lm.id, hits, samples, (size_t)lm.sidx);
  } else if (PHASE == 2) {
    const int level = blockIdx.z;
    const float4* h = hits + ((size_t)level * samples + lm.sidx) * 2;
    const float4 ha = h[0], hb = h[1];
    const int obj = __float_as_int(hb.w);
    rmk::v3 nrm = rmk::V(hb.x, hb.y, hb.z);
    if (level == 0) {
      if (ha.w >= opts->maxDist) return;  // primary miss
      const rmk::Material m = rmk::material_of(*opts, obj);
      const float k = 1.0f / (m.smoothness * 200.0f + 5.0f);
      nrm = rmk::mads(tr.sample_mc_normal(lm.id), k, nrm);  // renderer.cl:420
    } else if (obj < 0) {
      return;  // no such bounce / it left the scene
    }
    const auto s = tr.sample_seeds(lm.id);
    rays[(size_t)level * samples + lm.sidx] = tr.point_rays(s, rmk::V(ha.x, ha.y, ha.z), nrm);
  } else {
    const rmk::v3 col = tr.shade_from_rays(lm.id, hits, rays, samples, (size_t)lm.sidx);
    staging[lm.sidx] = make_float4(col.x, col.y, col.z, 1.0f);
  }
}

// Persistent, wave-scheduled renderer (rm_wave.hpp): each wavefront pulls tiles
// from a queue, keeps a pool of (pixel, pass) samples of its tile, and
// alternates between one shared march loop and short per-lane continuations.
constexpr int kMarchBudget = 2;  // lookups per lane between two ballots

// MINW = waves per SIMD the register allocator must leave room for (2: no spills,
// ~200 VGPRs; 4: 128 VGPRs with the cold lane state spilled to scratch).
template <int MINW>
__global__ __launch_bounds__(64 * kWavesPerBlock, MINW) void render_wave_kernel(rmk::WaveArgs a) {
  rmk::WaveTracer T(a);
  rmk::WaveLane L{};
  L.st = rmk::S_IDLE;
  L.marching = false;
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  const int pool = 64 * a.iter;
  const TileGeom g = tile_geom(a.resx, a.n);
  int slot = 0, next = pool;  // wave-uniform: current local tile, next sample of its pool
  bool exhausted = false;
  for (;;) {
    // ---- hand samples to idle lanes
    unsigned long long need = __ballot(L.st == rmk::S_IDLE);
    while (need != 0ull && !exhausted) {
      if (next >= pool) {
        unsigned int t = 0;
        if (lane == 0) t = atomicAdd(a.queue, 1u);
        t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
        if (t >= (unsigned int)a.my_tiles) { exhausted = true; break; }
        slot = (int)t;
        next = 0;
      }
      if (L.st == rmk::S_IDLE) {
        const int s = next + __popcll(need & below);
        if (s < pool) {
          const int pass = s >> 6, pix = s & 63;
          const long long tile = a.tile_first + (long long)slot * a.tile_stride;
          const int id = lane_pixel((int)tile, pix, a.resx, g.tiles_x, a.n, 0, a.n);
          if (id >= 0) {
            L.id = id;
            L.pass = pass;
            L.out_idx = (pass * a.tiles_per_part + slot) * 64 + pix;
            L.st = rmk::S_NEW;
          }
        }
      }
      next = min(pool, next + __popcll(need));
      need = __ballot(L.st == rmk::S_IDLE);
    }
    // ---- vote: run whichever phase has more lanes ready for it
    const bool wants_step = !L.marching && L.st != rmk::S_IDLE;
    const unsigned long long mc = __ballot(wants_step);
    const unsigned long long mm = __ballot(L.marching);
    if ((mc | mm) == 0ull) {
      if (exhausted) break;
      continue;
    }
    if (__popcll(mc) * a.wait_lanes >= __popcll(mm) * 16) {
      if (wants_step) T.advance(L);  // one continuation step
    } else {
      if (L.marching) T.march_some(L, kMarchBudget);  // the shared march loop
    }
  }
}

// In-order accumulation of the staged pass colours: p <- mix(p, c_i, frameBlend_i)
// for i = 0..iter-1 starting from 0 (renderer.cl:492 applied pass after pass,
// core.clj:81-90).  Lanes without a pixel hold garbage that nobody reads.
__global__ __launch_bounds__(256) void blend_kernel(const float4* __restrict__ staging,
                                                    const RmOpts* __restrict__ opts_all, int iter,
                                                    long long count, float4* __restrict__ tiles) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (long long)gridDim.x * blockDim.x) {
    float px = 0.f, py = 0.f, pz = 0.f;
    for (int k = 0; k < iter; k++) {
      const float fb = opts_all[k].frameBlend;
      const float4 c = staging[(long long)k * count + i];
      px = px + (c.x - px) * fb;
      py = py + (c.y - py) * fb;
      pz = pz + (c.z - pz) * fb;
    }
    tiles[i] = make_float4(px, py, pz, 1.0f);
  }
}

__global__ __launch_bounds__(256) void tonemap_kernel(const float4* __restrict__ pixels,
                                                      const RmOpts* __restrict__ opts,
                                                      uint32_t* __restrict__ argb, int n) {
  const float g = opts->gamma;
  for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n;
       id += (long long)gridDim.x * blockDim.x) {
    const float4 p = pixels[id];
    const float c[3] = {p.x, p.y, p.z};
    uint32_t ch[3];
    for (int k = 0; k < 3; k++) {
      const float t = c[k] / (g + c[k]);
      const float v = t * t * 255.0f;
      ch[k] = (uint32_t)rmd::f2i(rmd::clamp_cl(v, 0.0f, 255.0f));
    }
    argb[id] = 0xff000000u | (ch[0] << 16) | (ch[1] << 8) | ch[2];
  }
}

// Tile-major accumulators of `parts` interleaved partitions (partition r owns
// tiles r, r+parts, ...; each partition's buffer holds tiles_per_part tiles of
// 64 float4) -> row-major float4 pixels and/or tonemapped ARGB.
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
    if (argb) {
      const float c[3] = {p.x, p.y, p.z};
      uint32_t ch[3];
      for (int k = 0; k < 3; k++) {
        const float t = c[k] / (g + c[k]);
        const float v = t * t * 255.0f;
        ch[k] = (uint32_t)rmd::f2i(rmd::clamp_cl(v, 0.0f, 255.0f));
      }
      argb[id] = 0xff000000u | (ch[0] << 16) | (ch[1] << 8) | ch[2];
    }
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

}  // namespace

namespace rmk {

int tiles_total(int resx, int n) { return tile_geom(resx, n).tiles_total; }

void dump_work_stats() {
#ifdef RM_WORK_STATS
  unsigned long long h[64] = {0};
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_work_stats), sizeof h);
  const double n = h[0] ? (double)h[0] : 1.0;
  fprintf(stderr, "[work stats] samples=%llu per sample: marches=%.2f turns=%.2f filtered=%.2f walks=%.2f "
                  "fetches=%.2f steps_advanced=%.2f ao_loops=%.2f\n",
          h[0], h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n, h[6] / n, h[7] / n);
  fprintf(stderr, "[work stats] lane utilisation of the loops: walk %.1f%% (fetches / lane-slots), "
                  "filtered turns %.1f%%, estimate turns %.1f%%\n",
          100.0 * h[5] / (h[8] ? h[8] : 1), 100.0 * h[2] / (h[9] ? h[9] : 1),
          100.0 * (h[2] - h[3]) / (h[10] ? h[10] : 1));
  static const char* kind[4] = {"primary march", "reflection march", "shadow march", "AO probe"};
  for (int k = 0; k < 4; k++)
    fprintf(stderr, "[work stats]   %-16s %.2f walks/sample, %.1f fetches/walk, %.1f%% of the walk lane-slots at "
                    "%.1f%% utilisation\n",
            kind[k], h[11 + k] / n, (double)h[15 + k] / (h[11 + k] ? h[11 + k] : 1),
            100.0 * h[19 + k] / (h[8] ? h[8] : 1), 100.0 * h[15 + k] / (h[19 + k] ? h[19 + k] : 1));
  fprintf(stderr, "[work stats] fetched dist8 values: hit %.1f%%, 1: %.1f%%, 2: %.1f%%, 3: %.1f%%, 4-7: %.1f%%, "
                  "8+: %.1f%%\n",
          100.0 * h[23] / h[5], 100.0 * h[24] / h[5], 100.0 * h[25] / h[5], 100.0 * h[26] / h[5],
          100.0 * h[27] / h[5], 100.0 * h[28] / h[5]);
  fprintf(stderr, "[work stats] samples advanced per sample: %.1f in walks that hit, %.1f in walks that do not "
                  "(%.1f of them after the walk's last fetch with value <= 1)\n",
          h[29] / n, h[30] / n, h[31] / n);
  fprintf(stderr, "[work stats] (hit, light) pairs per sample %.2f: %.1f%% face away from the light, %.1f%% of all have "
                  "no specular term either; last turn repeated for its normal in %.3f marches per sample\n",
          h[40] / n, 100.0 * h[41] / (h[40] ? h[40] : 1), 100.0 * h[42] / (h[40] ? h[40] : 1), h[43] / n);
  fprintf(stderr, "[work stats] walks without a hit per sample: primary %.2f, reflection %.2f, shadow %.2f, AO %.2f; "
                  "ended by their first fetch: primary+reflection %.2f, shadow %.2f\n",
          h[44] / n, h[45] / n, h[46] / n, h[47] / n, h[38] / n, h[39] / n);
  fprintf(stderr, "[work stats] turns per sample (estimate + filtered): primary %.2f + %.2f, reflection %.2f + %.2f, "
                  "shadow %.2f + %.2f\n",
          h[48] / n, h[52] / n, h[49] / n, h[53] / n, h[50] / n, h[54] / n);
#endif
#ifdef RM_PHASE_CLOCK
  {
    unsigned long long h[64] = {0};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_work_stats), sizeof h);
    const double waves = h[37] ? (double)h[37] : 1.0;
    fprintf(stderr, "[phase clock] wave time by phase (shader clock ticks per wave): primary march %.0f, reflection "
                    "marches %.0f, AO phases %.0f, shadow phases %.0f, shading arithmetic %.0f\n",
            h[32] / waves, h[33] / waves, h[34] / waves, h[35] / waves, h[36] / waves);
  }
#endif
#if defined(RM_WORK_STATS) || defined(RM_PHASE_CLOCK)
  unsigned long long z[64] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_work_stats), z, sizeof z);
#endif
}

hipError_t launch_render_pass(hipStream_t st, const uint8_t* vox, Accel accel, const float* mc,
                              const RmOpts* d_opts, int resx, float* pixels, int n, int id0,
                              int id1, int tile_first, int tile_stride, bool tile_major,
                              Counters* d_counters) {
  const TileGeom g = tile_geom(resx, n);
  if (tile_stride < 1) tile_stride = 1;
  const long long my_tiles =
      tile_first >= g.tiles_total ? 0 : (g.tiles_total - tile_first + tile_stride - 1) / tile_stride;
  if (my_tiles == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((my_tiles + kWavesPerBlock - 1) / kWavesPerBlock);
  const float4* mc4 = reinterpret_cast<const float4*>(mc);
  float4* px4 = reinterpret_cast<float4*>(pixels);
  const dim3 grid(blocks), block(64 * kWavesPerBlock);
  const bool acc = accel.dist && accel.surf;
#define RM_LAUNCH(C, T, A)                                                                      \
  render_pass_kernel<C, T, A><<<grid, block, 0, st>>>(vox, accel.dist, accel.surf, mc4, d_opts, \
                                                      px4, n, id0, id1, tile_first, tile_stride, \
                                                      d_counters)
  if (d_counters) RM_LAUNCH(true, false, false);
  else if (tile_major && acc) RM_LAUNCH(false, true, true);
  else if (tile_major) RM_LAUNCH(false, true, false);
  else if (acc) RM_LAUNCH(false, false, true);
  else RM_LAUNCH(false, false, false);
#undef RM_LAUNCH
  return hipGetLastError();
}

hipError_t launch_render_samples(hipStream_t st, const uint8_t* vox, Accel accel, const float* mc_all,
                                 const RmOpts* d_opts_all, int resx, int iter, float* staging, int n,
                                 int tile_first, int tile_stride, int min_waves, int pp_log2,
                                 bool xcd_rows) {
  const TileGeom g = tile_geom(resx, n);
  if (tile_stride < 1) tile_stride = 1;
  const int tpp = tiles_per_part(g.tiles_total, tile_stride);
  const long long my_tiles =
      tile_first >= g.tiles_total ? 0 : (g.tiles_total - tile_first + tile_stride - 1) / tile_stride;
  if (my_tiles == 0 || iter <= 0) return hipSuccess;
  while (pp_log2 > 0 && (iter % (1 << pp_log2)) != 0) pp_log2--;  // pass groups must tile `iter`
  const long long waves = my_tiles << pp_log2;
  long long blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
  // blocks per tile row, for the XCD-aware order.  A partition (first, stride) whose
  // stride divides the row length owns tiles_x/stride tiles of every row -- columns of
  // tiles -- so its local slots still form rows and the same order applies.
  int bpr = 0;
  if (xcd_rows && g.tiles_x % tile_stride == 0 && tile_first < tile_stride &&
      ((long long)(g.tiles_x / tile_stride) << pp_log2) % kWavesPerBlock == 0) {
    bpr = (int)(((long long)(g.tiles_x / tile_stride) << pp_log2) / kWavesPerBlock);
    const long long rows = (blocks + bpr - 1) / bpr;
    blocks = ((rows + 7) / 8) * 8 * bpr;  // pad to groups of 8 rows; surplus blocks exit at once
  }
  const dim3 grid((unsigned)blocks, (unsigned)(iter >> pp_log2));
  const dim3 block(64 * kWavesPerBlock);
  const float4* mc4 = reinterpret_cast<const float4*>(mc_all);
  float4* st4 = reinterpret_cast<float4*>(staging);
  if (accel.dist && accel.surf)
    switch (min_waves) {
      case 4: render_samples_kernel<true, 4><<<grid, block, 0, st>>>(vox, accel.dist, accel.surf, mc4, d_opts_all, st4, n, tile_first, tile_stride, tpp, pp_log2, bpr, accel.oct_stride); break;
      case 5: render_samples_kernel<true, 5><<<grid, block, 0, st>>>(vox, accel.dist, accel.surf, mc4, d_opts_all, st4, n, tile_first, tile_stride, tpp, pp_log2, bpr, accel.oct_stride); break;
      case 6: render_samples_kernel<true, 6><<<grid, block, 0, st>>>(vox, accel.dist, accel.surf, mc4, d_opts_all, st4, n, tile_first, tile_stride, tpp, pp_log2, bpr, accel.oct_stride); break;
      case 7: render_samples_kernel<true, 7><<<grid, block, 0, st>>>(vox, accel.dist, accel.surf, mc4, d_opts_all, st4, n, tile_first, tile_stride, tpp, pp_log2, bpr, accel.oct_stride); break;
      case 8: render_samples_kernel<true, 8><<<grid, block, 0, st>>>(vox, accel.dist, accel.surf, mc4, d_opts_all, st4, n, tile_first, tile_stride, tpp, pp_log2, bpr, accel.oct_stride); break;
      default: render_samples_kernel<true, 3><<<grid, block, 0, st>>>(vox, accel.dist, accel.surf, mc4, d_opts_all, st4, n, tile_first, tile_stride, tpp, pp_log2, bpr, accel.oct_stride); break;
    }
  else
    render_samples_kernel<false, 3><<<grid, block, 0, st>>>(vox, nullptr, nullptr, mc4, d_opts_all, st4,
                                                         n, tile_first, tile_stride, tpp, pp_log2, bpr, 0ull);
  return hipGetLastError();
}

// QUALITY MODE (not reference-equivalent): same grid and lane layout, distance field instead
// of the byte grid
hipError_t launch_render_sdf(hipStream_t st, const float* d_sdf, const float* mc_all,
                             const RmOpts* d_opts_all, int resx, int iter, float* staging, int n,
                             int tile_first, int tile_stride, int pp_log2) {
  const TileGeom g = tile_geom(resx, n);
  if (tile_stride < 1) tile_stride = 1;
  const int tpp = tiles_per_part(g.tiles_total, tile_stride);
  const long long my_tiles =
      tile_first >= g.tiles_total ? 0 : (g.tiles_total - tile_first + tile_stride - 1) / tile_stride;
  if (my_tiles == 0 || iter <= 0) return hipSuccess;
  while (pp_log2 > 0 && (iter % (1 << pp_log2)) != 0) pp_log2--;
  const long long blocks = ((my_tiles << pp_log2) + kWavesPerBlock - 1) / kWavesPerBlock;
  const dim3 grid((unsigned)blocks, (unsigned)(iter >> pp_log2));
  render_samples_kernel<false, 4, true><<<grid, dim3(64 * kWavesPerBlock), 0, st>>>(
      nullptr, nullptr, nullptr, reinterpret_cast<const float4*>(mc_all), d_opts_all,
      reinterpret_cast<float4*>(staging), n, tile_first, tile_stride, tpp, pp_log2, 0, 0ull, d_sdf);
  return hipGetLastError();
}

hipError_t launch_render_split(hipStream_t st, const uint8_t* vox, Accel accel, const float* mc_all,
                               const RmOpts* d_opts_all, int resx, int iter, float* staging,
                               float* hits, int n, int tile_first, int tile_stride, int waves_trace,
                               int waves_light) {
  const TileGeom g = tile_geom(resx, n);
  if (tile_stride < 1) tile_stride = 1;
  const int tpp = tiles_per_part(g.tiles_total, tile_stride);
  const long long my_tiles =
      tile_first >= g.tiles_total ? 0 : (g.tiles_total - tile_first + tile_stride - 1) / tile_stride;
  if (my_tiles == 0 || iter <= 0) return hipSuccess;
  const dim3 grid((unsigned)((my_tiles + kWavesPerBlock - 1) / kWavesPerBlock), (unsigned)iter);
  const dim3 block(64 * kWavesPerBlock);
  const float4* mc4 = reinterpret_cast<const float4*>(mc_all);
  float4* st4 = reinterpret_cast<float4*>(staging);
  float4* h4 = reinterpret_cast<float4*>(hits);
#define RM_T(W) trace_chain_kernel<W><<<grid, block, 0, st>>>(vox, accel.dist, accel.surf, mc4, d_opts_all, h4, n, tile_first, tile_stride, tpp)
#define RM_L(W) light_kernel<W><<<grid, block, 0, st>>>(vox, accel.dist, accel.surf, mc4, d_opts_all, h4, st4, n, tile_first, tile_stride, tpp)
  switch (waves_trace) {
    case 4: RM_T(4); break;
    case 6: RM_T(6); break;
    default: RM_T(8); break;
  }
  switch (waves_light) {
    case 4: RM_L(4); break;
    case 5: RM_L(5); break;
    case 6: RM_L(6); break;
    case 7: RM_L(7); break;
    default: RM_L(8); break;
  }
#undef RM_T
#undef RM_L
  return hipGetLastError();
}

hipError_t launch_render_phases(hipStream_t st, const uint8_t* vox, Accel accel, const float* mc_all,
                                const RmOpts* d_opts_all, int resx, int iter, int levels,
                                float* staging, void* work, int n, int tile_first, int tile_stride,
                                int pp_log2) {
  const TileGeom g = tile_geom(resx, n);
  if (tile_stride < 1) tile_stride = 1;
  const int tpp = tiles_per_part(g.tiles_total, tile_stride);
  const long long my_tiles =
      tile_first >= g.tiles_total ? 0 : (g.tiles_total - tile_first + tile_stride - 1) / tile_stride;
  if (my_tiles == 0 || iter <= 0) return hipSuccess;
  while (pp_log2 > 0 && (iter % (1 << pp_log2)) != 0) pp_log2--;
  const long long waves = my_tiles << pp_log2;
  const unsigned bx = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
  const unsigned by = (unsigned)(iter >> pp_log2);
  const dim3 block(64 * kWavesPerBlock);
  const size_t samples = (size_t)iter * tpp * 64;
  float4* hits = static_cast<float4*>(work);
  float2* rays = reinterpret_cast<float2*>(hits + samples * levels * 2);
  const float4* mc4 = reinterpret_cast<const float4*>(mc_all);
  float4* st4 = reinterpret_cast<float4*>(staging);
  phase_kernel<1><<<dim3(bx, by, 1), block, 0, st>>>(vox, accel.dist, accel.surf, mc4, d_opts_all, hits, rays,
                                                      st4, n, tile_first, tile_stride, tpp, pp_log2, iter);
  phase_kernel<2><<<dim3(bx, by, (unsigned)levels), block, 0, st>>>(vox, accel.dist, accel.surf, mc4,
                                                                     d_opts_all, hits, rays, st4, n,
                                                                     tile_first, tile_stride, tpp,
                                                                     pp_log2, iter);
  phase_kernel<3><<<dim3(bx, by, 1), block, 0, st>>>(vox, accel.dist, accel.surf, mc4, d_opts_all, hits, rays,
                                                      st4, n, tile_first, tile_stride, tpp, pp_log2, iter);
  return hipGetLastError();
}
size_t phases_workspace_bytes(size_t samples, int levels) { return samples * levels * (32 + 8) + 256; }

hipError_t launch_render_wave(hipStream_t st, const uint8_t* vox, Accel accel, const float* mc_all,
                              const RmOpts* d_opts_all, int resx, int iter, float* staging, int n,
                              int tile_first, int tile_stride, unsigned int* d_queue, int blocks,
                              int min_waves, int wait_lanes) {
  const TileGeom g = tile_geom(resx, n);
  if (tile_stride < 1) tile_stride = 1;
  WaveArgs a;
  a.vox = vox;
  a.dist8 = accel.dist;
  a.surf32 = accel.surf;
  a.mc_all = reinterpret_cast<const float4*>(mc_all);
  a.opts_all = d_opts_all;
  a.staging = reinterpret_cast<float4*>(staging);
  a.queue = d_queue;
  a.n = n;
  a.iter = iter;
  a.tile_first = tile_first;
  a.tile_stride = tile_stride;
  a.tiles_per_part = tiles_per_part(g.tiles_total, tile_stride);
  a.my_tiles = tile_first >= g.tiles_total
                   ? 0
                   : (g.tiles_total - tile_first + tile_stride - 1) / tile_stride;
  a.resx = resx;
  a.wait_lanes = wait_lanes;
  if (a.my_tiles == 0 || iter <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(d_queue, 0, sizeof(unsigned int), st);
  if (e != hipSuccess) return e;
  const int waves_needed = (a.my_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > waves_needed) blocks = waves_needed;
  switch (min_waves) {
    case 2: render_wave_kernel<2><<<blocks, 64 * kWavesPerBlock, 0, st>>>(a); break;
    case 3: render_wave_kernel<3><<<blocks, 64 * kWavesPerBlock, 0, st>>>(a); break;
    case 5: render_wave_kernel<5><<<blocks, 64 * kWavesPerBlock, 0, st>>>(a); break;
    default: render_wave_kernel<4><<<blocks, 64 * kWavesPerBlock, 0, st>>>(a); break;
  }
  return hipGetLastError();
}

int wave_kernel_blocks_per_cu(int min_waves) {
  int nb = 0;
  hipError_t e;
  switch (min_waves) {
    case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, render_wave_kernel<2>, 64 * kWavesPerBlock, 0); break;
    case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, render_wave_kernel<3>, 64 * kWavesPerBlock, 0); break;
    case 5: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, render_wave_kernel<5>, 64 * kWavesPerBlock, 0); break;
    default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, render_wave_kernel<4>, 64 * kWavesPerBlock, 0); break;
  }
  if (e != hipSuccess || nb < 1) nb = 1;
  return nb;
}

hipError_t launch_blend(hipStream_t st, const float* staging, const RmOpts* d_opts_all, int iter,
                        long long count, float* tiles) {
  if (count <= 0) return hipSuccess;
  long long blocks = (count + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  blend_kernel<<<(unsigned)blocks, 256, 0, st>>>(reinterpret_cast<const float4*>(staging), d_opts_all,
                                                 iter, count, reinterpret_cast<float4*>(tiles));
  return hipGetLastError();
}

hipError_t launch_tonemap(hipStream_t st, const float* pixels, const RmOpts* d_opts, uint32_t* argb,
                          int n) {
  if (n <= 0) return hipSuccess;
  int blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  tonemap_kernel<<<blocks, 256, 0, st>>>(reinterpret_cast<const float4*>(pixels), d_opts, argb, n);
  return hipGetLastError();
}

hipError_t launch_resolve(hipStream_t st, const float* tiles, int parts, int tiles_per_part,
                          const RmOpts* d_opts0, float* pixels, uint32_t* argb, int n) {
  if (n <= 0) return hipSuccess;
  int blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  resolve_kernel<<<blocks, 256, 0, st>>>(reinterpret_cast<const float4*>(tiles), parts,
                                         tiles_per_part, d_opts0, reinterpret_cast<float4*>(pixels),
                                         argb, n);
  return hipGetLastError();
}

hipError_t launch_prims(hipStream_t st, int op, const float* a, const float* b, uint32_t* out,
                        int n) {
  if (n <= 0) return hipSuccess;
  prims_kernel<<<(n + 255) / 256, 256, 0, st>>>(op, a, b, out, n);
  return hipGetLastError();
}

}  // namespace rmk
