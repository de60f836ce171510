// rm_stream.h -- launcher of the stream (task-queue) form of the render path (rm_stream.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "rm_kernels.h"

namespace rmk {

struct StreamLaunch {
  const uint8_t* vox;
  Accel accel;           // required
  const float* mc;       // tables of this batch's passes
  const RmOpts* opts;    // records of this batch's passes (identical except .time)
  float* staging;        // [passes][count] float4, this batch's slice
  void* workspace;       // >= stream_workspace_bytes(passes*count, levels, num_lights)
  int n, resx, passes, count, tile_first, tile_stride;
  int levels;            // 1 + reflectIter
  int num_lights;
  int queue_blocks;      // grid of the queue-draining kernels
};

// limits of the task encoding; outside them the caller uses render_samples_kernel
constexpr int kStreamMaxAoIter = 7;      // aoIter + 1 probes stored per point
constexpr int kStreamMaxReflect = 6;     // bounce index travels in 3 bits
constexpr int kStreamMaxSamples = 1 << 27;

size_t stream_workspace_bytes(int samples, int levels, int num_lights);
hipError_t launch_stream_batch(hipStream_t st, const StreamLaunch& L);

}  // namespace rmk
