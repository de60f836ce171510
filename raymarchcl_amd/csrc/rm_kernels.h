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
void dump_work_stats();  // no-op unless built with -DRM_WORK_STATS

#ifndef RM_BRICKS
#define RM_BRICKS 0  // see rm_shade.hpp
#endif

struct Accel {  // nullptrs = not available: the kernels then run the plain fixed-step march
  const uint8_t* dist = nullptr;
  const uint32_t* surf = nullptr;
  // > 0: `dist` is followed by 8 directional tables (rm_accel.hip oct8), each this many bytes
  unsigned long long oct_stride = 0;
};
hipError_t launch_render_pass(hipStream_t st, const uint8_t* d_vox, Accel accel, const float* d_mc,
                              const RmOpts* d_opts, int resx, float* d_pixels, int n, int id0,
                              int id1, int tile_first, int tile_stride, bool tile_major,
                              Counters* d_counters);
// all `iter` passes of a partition in one launch -> staging [iter][tiles_per_part*64] float4.
// pp_log2 > 0: a wavefront holds 2^pp_log2 passes of 64/2^pp_log2 pixels -- only valid when the
// `iter` records are identical except .time (reduced automatically until it divides iter).
hipError_t launch_render_samples(hipStream_t st, const uint8_t* d_vox, Accel accel,
                                 const float* d_mc_all, const RmOpts* d_opts_all, int resx, int iter,
                                 float* d_staging, int n, int tile_first, int tile_stride,
                                 int min_waves, int pp_log2, bool xcd_rows);
// the same in two launches (march chain -> hit records in d_hits -> lighting);
// d_hits: (1 + reflectIter) * iter * tiles_per_part * 64 * 32 bytes
hipError_t launch_render_split(hipStream_t st, const uint8_t* d_vox, Accel accel, const float* d_mc_all,
                               const RmOpts* d_opts_all, int resx, int iter, float* d_staging,
                               float* d_hits, int n, int tile_first, int tile_stride, int waves_trace,
                               int waves_light);
// the same in three launches: march chain -> hit records; the rays of every shaded point
// (AO loop + shadow marches) -> one float + light bits; shading arithmetic -> staging.
// d_work: phases_workspace_bytes(iter * tiles_per_part * 64, levels) bytes.
size_t phases_workspace_bytes(size_t samples, int levels);
hipError_t launch_render_phases(hipStream_t st, const uint8_t* d_vox, Accel accel, const float* d_mc_all,
                                const RmOpts* d_opts_all, int resx, int iter, int levels,
                                float* d_staging, void* d_work, int n, int tile_first,
                                int tile_stride, int pp_log2);
// the same, by the persistent wave-scheduled kernel (needs the accel structures and
// option records that differ only in .time); d_queue: one device uint32 of scratch
hipError_t launch_render_wave(hipStream_t st, const uint8_t* d_vox, Accel accel, const float* d_mc_all,
                              const RmOpts* d_opts_all, int resx, int iter, float* d_staging, int n,
                              int tile_first, int tile_stride, unsigned int* d_queue, int blocks,
                              int min_waves, int wait_lanes);
int wave_kernel_blocks_per_cu(int min_waves);
// staging -> tile-major accumulators (in-order frame blend)
hipError_t launch_blend(hipStream_t st, const float* d_staging, const RmOpts* d_opts_all, int iter,
                        long long count, float* d_tiles);
// tiles a partition of `parts` owns at most: ceil(tiles_total / parts)
hipError_t launch_resolve(hipStream_t st, const float* d_tiles, int parts, int tiles_per_part,
                          const RmOpts* d_opts0, float* d_pixels, uint32_t* d_argb, int n);
hipError_t launch_tonemap(hipStream_t st, const float* d_pixels, const RmOpts* d_opts,
                          uint32_t* d_argb, int n);
// dist8 / surf32 of a resident volume for hit threshold `iso` (rm_accel.hip);
// d_tmp is scratch of the volume's size.
hipError_t build_accel(hipStream_t st, const uint8_t* d_vox, int rx, int ry, int rz, int iso,
                       uint8_t* d_dist, uint8_t* d_tmp, uint32_t* d_surf);
// the 8 directional tables behind dist8 (d_dist9 = 9 * volume bytes, table 0 = dist8 itself);
// d_sat is scratch of (rx+1)(ry+1)(rz+1) uint32
hipError_t build_octants(hipStream_t st, const uint8_t* d_vox, int rx, int ry, int rz, int iso,
                         uint8_t* d_dist9, uint32_t* d_sat);
// RM_BRICKS layout of a byte table (8x4x4-cell bricks of 128 B); to_bricks = false converts back
long long bricked_bytes(int rx, int ry, int rz);
hipError_t launch_brick(hipStream_t st, uint8_t* d_lin, int rx, int ry, int rz, uint8_t* d_bricked,
                        bool to_bricks);
// quality mode: render_samples_kernel over a float distance field
hipError_t launch_render_sdf(hipStream_t st, const float* d_sdf, const float* d_mc_all,
                             const RmOpts* d_opts_all, int resx, int iter, float* d_staging, int n,
                             int tile_first, int tile_stride, int pp_log2);
hipError_t launch_gyroid(hipStream_t st, uint8_t* d_out, int rx, int ry, int rz);
// rm_volgen.hip: the other volume producers of the reference, on the device
hipError_t launch_terrain(hipStream_t st, uint8_t* d_out, int rx, int ry, int rz);
hipError_t launch_splat(hipStream_t st, uint8_t* d_out, const double* d_xyz, long long n,
                        const double p[3], const double off[3], double s, int res, int ks);
hipError_t launch_heatmap(hipStream_t st, uint8_t* d_out, const uint32_t* d_argb, int res, double amp);
hipError_t launch_prims(hipStream_t st, int op, const float* a, const float* b, uint32_t* out,
                        int n);
}  // namespace rmk
