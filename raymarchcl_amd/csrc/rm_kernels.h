// rm_kernels.h -- host-side launchers of the gfx950 kernels (rm_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "rm_opts.h"

namespace rmk {
struct Counters;

inline int tiles_per_part(int tiles, int parts) { return (tiles + parts - 1) / parts; }
// number of 8x8 tiles covering work-items 0..n-1 of an image `resx` wide
int tiles_total(int resx, int n);

struct Accel {  // nullptrs = not available: the kernels then run the plain fixed-step march
  const uint8_t* dist = nullptr;
  const uint32_t* surf = nullptr;
  // > 0: `dist` is followed by 8 directional tables (rm_accel.hip oct8), each this many bytes
  unsigned long long oct_stride = 0;
  bool bricked = false;             // dist / oct tables in 8x4x4-cell bricks of 128 B (oct_stride = bricked table bytes)
  unsigned log2res = 0;             // > 0: cubic grid of edge 1 << log2res, tables below 4 GiB: row-major (walk_step LAYOUT 2) or the bricks of the 512^3 grid (LAYOUT 3); 10 with bricks: the 1024^3 grid (LAYOUT 4)
};
// bytes of one byte table in the bricked layout
long long bricked_bytes(int rx, int ry, int rz);
// bricked table -> row-major (test hooks)
hipError_t launch_unbrick(hipStream_t st, const uint8_t* d_bricked, int rx, int ry, int rz, uint8_t* d_linear);
hipError_t launch_render_pass(hipStream_t st, const uint8_t* d_vox, Accel accel, const float* d_mc,
                              const RmOpts* d_opts, int resx, float* d_pixels, int n, int id0,
                              int id1, int tile_first, int tile_stride, bool tile_major,
                              Counters* d_counters, int arith = 0);
// One launch of the frame kernel (rm_kernels.hip render_frame_kernel): `passes` consecutive
// RenderImage passes (at most 2^pp_log2: what one wavefront holds) over partition (tile_first, tile_stride)
// of the image, blended in order into `acc`.  pp_log2 > 0: a wavefront holds 2^pp_log2 passes of
// 64/2^pp_log2 pixels -- only valid when the records are identical except .time.
struct FrameLaunch {
  const uint8_t* vox = nullptr;
  Accel accel;
  const float* sdf = nullptr;      // quality mode: the distance field as float4 xy-faces (launch_sdf_quads) instead of vox / accel
  const float* mc_all = nullptr;   // table of the first pass of this launch (device)
  const RmOpts* opts_all = nullptr;  // record of the first pass of this launch (device)
  const RmOpts* opts0 = nullptr;     // record 0 of the frame (device): tonemap parameters
  float* acc = nullptr;            // float4 accumulators: tile-major [tiles_per_part][64], or the row-major image
  uint32_t* argb = nullptr;        // row_major only, nullable: TonemapImage output, written with the last pass
  int resx = 0, n = 0, passes = 0, tile_first = 0, tile_stride = 1;
  int pp_log2 = 0;
  bool xcd_rows = true, accumulate = false, row_major = false;
  int unit_min_waves = 0;          // with xcd_2d < 0: wavefronts the narrowest 2-D unit must still hold (0: 448)
  int xcd_2d = -1;                 // with xcd_rows: 2-D units per XCD: 2 tile rows x 1/(8 n) of the row, n units per row pair (n = 1, 2, 4, 8); 0 = whole rows; -1 = chosen by the launcher
  bool rows_desc = true;           // with xcd_rows: tile rows dispatched bottom to top (the top rows -- sky -- make the shortest tail)
  // with rows_desc: the part of the image height (fractions, 0 = top) whose rows are dispatched FIRST -- the rows that can see
  // the clip box (rm_api.hip volume_band); band_hi <= band_lo: none
  double band_lo = 0.0, band_hi = 0.0;
  // arithmetic contract (rm_math.hpp ArithOf): 0 = OpenCL CPU device, 1 = the same with the GPU lowering of
  // the seed casts (rm_set_seed_cast), 2 / 3 = ROCm's OpenCL library on this GPU as the strict / default
  // build of the reference uses it (rm_set_contract).  Every launcher below takes the same number.
  int arith = 0;
};
hipError_t launch_render_frame(hipStream_t st, const FrameLaunch& f);
// host arithmetic: out[b] = tile << 8 | sub-block that hardware block b of that launch renders, -1 = padding; -> blocks of the grid
long long debug_block_order(const FrameLaunch& f, long long* out, long long cap);
// true: the frame kernel of this volume's table layout takes records with any number of AO probes (chunked exchange)
bool frame_takes_any_ao(const Accel& accel);

int choose_pass_pack(int passes, int max_log2, int waste_pct = 60);
// tiles a partition of `parts` owns at most: ceil(tiles_total / parts)
hipError_t launch_resolve(hipStream_t st, const float* d_tiles, int parts, int tiles_per_part,
                          const RmOpts* d_opts0, float* d_pixels, uint32_t* d_argb, int n,
                          int arith = 0);
hipError_t launch_resolve_argb(hipStream_t st, const uint32_t* d_argb_tiles, int parts, int tiles_per_part, int resx,
                               uint32_t* d_argb, int n);
hipError_t launch_tonemap(hipStream_t st, const float* d_pixels, const RmOpts* d_opts,
                          uint32_t* d_argb, int n, int arith = 0);
// surf32 of a resident volume for hit threshold `iso` (rm_accel.hip) and -- when d_dist is not
// null -- dist8 alone by separable passes (d_tmp: scratch of the volume's size)
// d_scratch (nullable): rx*ry*rz bytes that may be overwritten -- surf32 is then built in two passes (27 instead of 189 loads per hit voxel)
hipError_t build_accel(hipStream_t st, const uint8_t* d_vox, int rx, int ry, int rz, int iso,
                       uint8_t* d_dist, uint8_t* d_tmp, uint32_t* d_surf, uint8_t* d_scratch = nullptr);
// the 8 directional tables (d_dist9 = 9 * volume bytes: tables 1..8) and dist8 derived from
// them (table 0); no scratch
hipError_t build_octants(hipStream_t st, const uint8_t* d_vox, int rx, int ry, int rz, int iso,
                         uint8_t* d_dist9, bool bricked = false);
hipError_t launch_gyroid(hipStream_t st, uint8_t* d_out, int rx, int ry, int rz);
// quality mode: float field -> one float4 per cell (the cell's xy-face at its layer), rm_accel.hip
hipError_t launch_sdf_quads(hipStream_t st, const float* d_field, int rx, int ry, int rz, float* d_quads);
// rm_volgen.hip: the other volume producers of the reference, on the device
hipError_t launch_terrain(hipStream_t st, uint8_t* d_out, int rx, int ry, int rz);
hipError_t launch_splat(hipStream_t st, uint8_t* d_out, const double* d_xyz, long long n,
                        const double p[3], const double off[3], double s, int res, int ks);
hipError_t launch_scatter(hipStream_t st, uint8_t* d_out, const double* d_xyz, long long n,
                          const double p[3], const double off[3], double s, int res, unsigned long long seed);
hipError_t launch_heatmap(hipStream_t st, uint8_t* d_out, const uint32_t* d_argb, int res, double amp);
hipError_t launch_filter_check(hipStream_t st, const float* d_rays, const RmOpts* d_opts, uint32_t* d_out, int n);
hipError_t launch_prims(hipStream_t st, int op, const float* a, const float* b, uint32_t* out,
                        int n);
}  // namespace rmk
