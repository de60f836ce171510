// rm_api.hip -- the C ABI declared in include/raymarch_hip.h.
//
// Replaces what thi.ng.simplecl does for the reference host (context, queue,
// buffers, the compiled pipeline of core.clj:76-97): device buffers live in an
// rm_ctx, every call validates its arguments, converts HIP errors into return
// codes + a thread-local message, and never throws across the boundary.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/raymarch_hip.h"
#include "rm_kernels.h"
#include "rm_shade.hpp"
#include "rm_stream.h"

static_assert(sizeof(rm_counters) == sizeof(rmk::Counters), "counter structs must match");
static_assert(RM_OPTS_BYTES == RM_OPTS_SIZE, "option record size");

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      return fail(RM_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                  __FILE__, __LINE__);                                                 \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace

struct rm_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  const uint8_t* d_vox = nullptr;  // owned (vox_buf) or borrowed
  DevBuf vox_buf, mc_buf, opts_buf, pix_buf, argb_buf, tile_buf, cnt_buf, prim_a, prim_b, prim_o;
  DevBuf dist_buf, tmp_buf, surf_buf, stage_buf, queue_buf, work_buf, gen_buf, sat_buf, lin_buf, sdf_buf;
  int sdf_rx = 0, sdf_ry = 0, sdf_rz = 0;  // quality mode: resident float distance field
  bool sdf_frame = false;                  // frame_on_device renders the distance field
  bool use_octants = false;
  unsigned long long oct_stride = 0;
  int stream_mode = 0;                 // RAYMARCH_KERNEL=straight (default) | stream | wave
  long long batch_samples = 8 << 20;   // RAYMARCH_BATCH_SAMPLES: samples per stream batch
  int phase_mode = 0;      // RAYMARCH_KERNEL=phases: chain / point rays / shading as three launches
  int split_mode = 0;      // RAYMARCH_KERNEL=split: march chain and lighting as two launches
  int split_tw = 8, split_lw = 8;  // RAYMARCH_SPLIT_WAVES=t,l
  bool xcd_rows = true;    // RAYMARCH_XCD_ROWS=0: plain block order
  int pass_pack = 4;       // RAYMARCH_PASS_PACK (0..6): log2 of the passes one wavefront holds
  int straight_waves = 7;  // RAYMARCH_STRAIGHT_WAVES (3..8): waves/SIMD the register budget of
                           // render_samples_kernel leaves room for (8 = 64 VGPRs + scratch spills)
  int wave_mode = 0;    // RAYMARCH_KERNEL=wave -> persistent wave-scheduled kernel (experimental)
  int min_waves = 4;    // RAYMARCH_WAVES=2..5: register budget of the wave kernel
  int wave_blocks = 0;  // persistent grid size
  int wait_lanes = 16;  // RAYMARCH_WAIT_LANES: continuation/march vote weight in 1/16 (16 = plain majority)
  int num_cus = 0;  // rm_accel.hip structures of the resident volume
  int accel_iso = -1;                  // isoVal they were built for, -1 = stale
  bool use_accel = true;               // RAYMARCH_NO_ACCEL=1 -> plain fixed-step march (A/B)
  std::vector<int> dev_iso;            // isoVal per record, noted by rm_check_device_opts
  std::vector<unsigned char> dev_same; // record i == record i-1 except .time
  std::vector<RmOpts> dev_recs;        // host copy of the checked records
  const void* dev_iso_src = nullptr;
  int rx = 0, ry = 0, rz = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  int launches = 0;
};

namespace {

int check_ctx(rm_ctx* c) {
  if (!c) return fail(RM_EINVAL, "rm_ctx is NULL");
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail(RM_EDEVICE, "hipSetDevice(%d): %s", c->device, hipGetErrorString(e));
  return RM_OK;
}

// The record's own fields must be usable before a kernel trusts them.
int check_opts(rm_ctx* c, const void* opts544, int n) {
  RmOpts o;
  memcpy(&o, opts544, sizeof o);
  if (o.resolution[0] <= 0 || o.resolution[1] <= 0)
    return fail(RM_EINVAL, "TRenderOpts.resolution = (%d,%d)", o.resolution[0], o.resolution[1]);
  if (o.voxelRes[0] != c->rx || o.voxelRes[1] != c->ry || o.voxelRes[2] != c->rz ||
      o.voxelRes[3] != c->rx * c->ry)
    return fail(RM_EINVAL, "TRenderOpts.voxelRes = (%d,%d,%d,%d) does not match the volume %dx%dx%d",
                o.voxelRes[0], o.voxelRes[1], o.voxelRes[2], o.voxelRes[3], c->rx, c->ry, c->rz);
  if (o.numLights > 4) return fail(RM_EINVAL, "TRenderOpts.numLights = %d (max 4)", (int)o.numLights);
  if (n < 0) return fail(RM_EINVAL, "n = %d", n);
  return RM_OK;
}

// Build (or reuse) dist8/surf32 for the hit threshold of this launch.
int ensure_accel(rm_ctx* c, int iso, rmk::Accel* out) {
  *out = rmk::Accel{};
  if (!c->use_accel) return RM_OK;
  // walk_step indexes with 24-bit multiplies: fall back to the plain march otherwise
  if ((long long)c->ry * c->rz >= (1 << 24) || c->rx >= (1 << 24)) return RM_OK;
  const size_t vox = (size_t)c->rx * c->ry * c->rz;
  if (c->accel_iso != iso) {
    // directional tables behind dist8 (measured -10 % frame time at 256^3, -12 % at 512^3 with
    // 8 % fill); table offsets are 64-bit, nine 1024^3 tables span 9 GiB
    const bool oct = c->use_octants;
    const int tables = oct ? 9 : 1;
#if RM_BRICKS
    // tables are built row-major in lin_buf, then re-laid in 8x4x4 bricks
    const size_t bb = (size_t)rmk::bricked_bytes(c->rx, c->ry, c->rz);
    HIP_TRY(c->lin_buf.reserve(vox * tables));
    HIP_TRY(c->dist_buf.reserve(bb * tables));
    uint8_t* lin = static_cast<uint8_t*>(c->lin_buf.p);
#else
    HIP_TRY(c->dist_buf.reserve(vox * tables));
    uint8_t* lin = static_cast<uint8_t*>(c->dist_buf.p);
#endif
    HIP_TRY(c->tmp_buf.reserve(vox));
    HIP_TRY(c->surf_buf.reserve(vox * 4));
    HIP_TRY(rmk::build_accel(c->stream, c->d_vox, c->rx, c->ry, c->rz, iso, lin,
                             static_cast<uint8_t*>(c->tmp_buf.p), static_cast<uint32_t*>(c->surf_buf.p)));
    c->oct_stride = 0;
    if (oct) {
      const size_t sat = (size_t)(c->rx + 1) * (c->ry + 1) * (c->rz + 1) * 4;
      HIP_TRY(c->sat_buf.reserve(sat));
      HIP_TRY(rmk::build_octants(c->stream, c->d_vox, c->rx, c->ry, c->rz, iso, lin,
                                 static_cast<uint32_t*>(c->sat_buf.p)));
      c->oct_stride = vox;
    }
#if RM_BRICKS
    for (int t = 0; t < tables; t++)
      HIP_TRY(rmk::launch_brick(c->stream, lin + (size_t)t * vox, c->rx, c->ry, c->rz,
                                static_cast<uint8_t*>(c->dist_buf.p) + (size_t)t * bb, true));
    if (oct) c->oct_stride = bb;
#endif
    c->accel_iso = iso;
  }
  out->oct_stride = c->oct_stride;
  out->dist = static_cast<const uint8_t*>(c->dist_buf.p);
  out->surf = static_cast<const uint32_t*>(c->surf_buf.p);
  return RM_OK;
}

// same[i] = 1 when record i equals record i-1 in every byte except .time
void records_same_as_prev(const void* opts_array, int iter, std::vector<unsigned char>* same) {
  same->assign(iter, 0);
  const char* base = static_cast<const char*>(opts_array);
  const size_t t0 = offsetof(RmOpts, time), t1 = t0 + sizeof(float);
  for (int i = 1; i < iter; i++) {
    const char* a = base + (size_t)(i - 1) * RM_OPTS_BYTES;
    const char* b = base + (size_t)i * RM_OPTS_BYTES;
    (*same)[i] = memcmp(a, b, t0) == 0 && memcmp(a + t1, b + t1, RM_OPTS_BYTES - t1) == 0;
  }
}

int render_pass_host(rm_ctx* c, const float* mc, const void* opts544, float* pixels, int n, int id0,
                     int id1, rm_counters* counters) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!mc || !opts544 || (!pixels && n > 0)) return fail(RM_EINVAL, "NULL buffer");
  if (!c->d_vox) return fail(RM_ESTATE, "rm_set_volume has not been called");
  rc = check_opts(c, opts544, n);
  if (rc) return rc;
  if (n == 0) return RM_OK;
  const size_t pix_bytes = (size_t)n * 16;
  HIP_TRY(c->mc_buf.reserve(RM_TABLE_FLOATS * 4));
  HIP_TRY(c->opts_buf.reserve(RM_OPTS_BYTES));
  HIP_TRY(c->pix_buf.reserve(pix_bytes));
  HIP_TRY(hipMemcpyAsync(c->mc_buf.p, mc, RM_TABLE_FLOATS * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->opts_buf.p, opts544, RM_OPTS_BYTES, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->pix_buf.p, pixels, pix_bytes, hipMemcpyHostToDevice, c->stream));
  rmk::Counters* d_cnt = nullptr;
  if (counters) {
    HIP_TRY(c->cnt_buf.reserve(sizeof(rmk::Counters)));
    HIP_TRY(hipMemsetAsync(c->cnt_buf.p, 0, sizeof(rmk::Counters), c->stream));
    d_cnt = static_cast<rmk::Counters*>(c->cnt_buf.p);
  }
  RmOpts o;
  memcpy(&o, opts544, sizeof o);
  rmk::Accel accel;
  rc = ensure_accel(c, o.isoVal, &accel);
  if (rc) return rc;
  HIP_TRY(rmk::launch_render_pass(c->stream, c->d_vox, accel, static_cast<const float*>(c->mc_buf.p),
                                  static_cast<const RmOpts*>(c->opts_buf.p), o.resolution[0],
                                  static_cast<float*>(c->pix_buf.p), n, id0, id1, 0, 1, false, d_cnt));
  HIP_TRY(hipMemcpyAsync(pixels, c->pix_buf.p, pix_bytes, hipMemcpyDeviceToHost, c->stream));
  rm_counters got{};
  if (counters)
    HIP_TRY(hipMemcpyAsync(&got, c->cnt_buf.p, sizeof got, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (counters) {
    uint64_t* dst = reinterpret_cast<uint64_t*>(counters);
    const uint64_t* src = reinterpret_cast<const uint64_t*>(&got);
    for (size_t k = 0; k < sizeof got / 8; k++) dst[k] += src[k];
  }
  return RM_OK;
}

// Records i0..i1-1 may share a wave-kernel launch when they are byte-identical
// apart from .time (what core.clj:99-106 produces).  `uniform[i]` is filled by
// the callers that have host copies of the records (1 = same as record i-1).
int frame_on_device(rm_ctx* c, const RmOpts* d_opts, const float* d_mc, int resx, int iter, int n,
                    int tile_first, int tile_stride, float* d_tiles, const int* iso_per_pass,
                    const unsigned char* same_as_prev, const RmOpts* host_recs) {
  const int tpp = rmk::tiles_per_part(rmk::tiles_total(resx, n), tile_stride);
  const long long count = (long long)tpp * 64;
  // passes that share a hit threshold share the derived structures and go out as
  // one launch (the reference's pipeline always has one isoVal: core.clj:49)
  HIP_TRY(c->stage_buf.reserve((size_t)iter * count * 16));
  float* staging = static_cast<float*>(c->stage_buf.p);
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  int launches = 0;
  for (int i0 = 0; i0 < iter;) {
    int i1 = i0 + 1;
    const bool wave = c->wave_mode && c->use_accel;
    const bool need_same = wave || c->pass_pack > 0 || c->phase_mode;  // launches whose lanes share one record
    while (i1 < iter && iso_per_pass[i1] == iso_per_pass[i0] && (!need_same || same_as_prev[i1])) i1++;
    if (c->sdf_frame) {  // quality mode: no derived structures
      HIP_TRY(rmk::launch_render_sdf(c->stream, static_cast<const float*>(c->sdf_buf.p),
                                     d_mc + (size_t)i0 * RM_TABLE_FLOATS, d_opts + i0, resx, i1 - i0,
                                     staging + (size_t)i0 * count * 4, n, tile_first, tile_stride,
                                     c->pass_pack));
      launches++;
      i0 = i1;
      continue;
    }
    rmk::Accel accel;
    int rc = ensure_accel(c, iso_per_pass[i0], &accel);
    if (rc) return rc;
    const RmOpts& h0 = host_recs[i0];
    const bool stream = c->stream_mode && c->use_accel && h0.aoIter >= 0 &&
                        h0.aoIter <= rmk::kStreamMaxAoIter && h0.reflectIter <= rmk::kStreamMaxReflect;
    if (stream) {
      // split the run further so that records differ only in .time, then into batches
      int j0 = i0;
      while (j0 < i1) {
        int j1 = j0 + 1;
        while (j1 < i1 && same_as_prev[j1]) j1++;
        const int levels = 1 + (host_recs[j0].reflectIter > 0 ? host_recs[j0].reflectIter : 0);
        const int nl = host_recs[j0].numLights;
        long long per = c->batch_samples / count;
        if (per < 1) per = 1;
        for (int b0 = j0; b0 < j1;) {
          int b1 = (int)((long long)b0 + per < j1 ? b0 + per : j1);
          while ((long long)(b1 - b0) * count > rmk::kStreamMaxSamples && b1 > b0 + 1) b1--;
          rmk::StreamLaunch sl;
          sl.vox = c->d_vox;
          sl.accel = accel;
          sl.mc = d_mc + (size_t)b0 * RM_TABLE_FLOATS;
          sl.opts = d_opts + b0;
          sl.staging = staging + (size_t)b0 * count * 4;
          sl.n = n; sl.resx = resx; sl.passes = b1 - b0; sl.count = (int)count;
          sl.tile_first = tile_first; sl.tile_stride = tile_stride;
          sl.levels = levels; sl.num_lights = nl < 1 ? 1 : nl;
          sl.queue_blocks = c->num_cus * 8;
          HIP_TRY(c->work_buf.reserve(rmk::stream_workspace_bytes(sl.passes * sl.count, levels, sl.num_lights)));
          sl.workspace = c->work_buf.p;
          HIP_TRY(rmk::launch_stream_batch(c->stream, sl));
          launches++;
          b0 = b1;
        }
        j0 = j1;
      }
      i0 = i1;
      continue;
    }
    if (c->phase_mode && c->use_accel) {
      const int levels = 1 + (h0.reflectIter > 0 ? h0.reflectIter : 0);
      HIP_TRY(c->work_buf.reserve(rmk::phases_workspace_bytes((size_t)(i1 - i0) * count, levels)));
      HIP_TRY(rmk::launch_render_phases(c->stream, c->d_vox, accel, d_mc + (size_t)i0 * RM_TABLE_FLOATS,
                                        d_opts + i0, resx, i1 - i0, levels,
                                        staging + (size_t)i0 * count * 4, c->work_buf.p, n, tile_first,
                                        tile_stride, c->pass_pack));
      launches++;
      i0 = i1;
      continue;
    }
    if (c->split_mode && c->use_accel) {
      const int levels = 1 + (h0.reflectIter > 0 ? h0.reflectIter : 0);
      HIP_TRY(c->work_buf.reserve((size_t)levels * (i1 - i0) * count * 32));
      HIP_TRY(rmk::launch_render_split(c->stream, c->d_vox, accel, d_mc + (size_t)i0 * RM_TABLE_FLOATS,
                                       d_opts + i0, resx, i1 - i0, staging + (size_t)i0 * count * 4,
                                       static_cast<float*>(c->work_buf.p), n, tile_first, tile_stride,
                                       c->split_tw, c->split_lw));
      launches++;
      i0 = i1;
      continue;
    }
    if (wave) {
      HIP_TRY(c->queue_buf.reserve(64));
      HIP_TRY(rmk::launch_render_wave(c->stream, c->d_vox, accel, d_mc + (size_t)i0 * RM_TABLE_FLOATS,
                                      d_opts + i0, resx, i1 - i0, staging + (size_t)i0 * count * 4, n,
                                      tile_first, tile_stride,
                                      static_cast<unsigned int*>(c->queue_buf.p), c->wave_blocks,
                                      c->min_waves, c->wait_lanes));
    } else {
      HIP_TRY(rmk::launch_render_samples(c->stream, c->d_vox, accel,
                                         d_mc + (size_t)i0 * RM_TABLE_FLOATS, d_opts + i0, resx,
                                         i1 - i0, staging + (size_t)i0 * count * 4, n, tile_first,
                                         tile_stride, c->straight_waves, c->pass_pack, c->xcd_rows));
    }
    launches++;
    i0 = i1;
  }
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(rmk::launch_blend(c->stream, staging, d_opts, iter, count, d_tiles));
  c->timed = true;
  c->launches = launches;
  return RM_OK;
}

}  // namespace

extern "C" {

// used by rm_host.cpp (host-only translation unit) to report through rm_last_error()
int rm_host_fail_(int code, const char* msg) { return fail(code, "%s", msg); }

const char* rm_last_error(void) { return g_err; }
int rm_abi_version(void) { return 1; }

int rm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int rm_create(int device_id, rm_ctx** out) {
  if (!out) return fail(RM_EINVAL, "out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(RM_EDEVICE, "no HIP device available (%s); this library has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= n) return fail(RM_EINVAL, "device_id %d of %d", device_id, n);
  HIP_TRY(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(RM_EDEVICE, "device %d is %s; this library is built for gfx950 only", device_id,
                prop.gcnArchName);
  rm_ctx* c = new (std::nothrow) rm_ctx();
  if (!c) return fail(RM_EDEVICE, "out of host memory");
  c->device = device_id;
  e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&c->ev0);
  if (e == hipSuccess) e = hipEventCreate(&c->ev1);
  if (e != hipSuccess) {
    rm_destroy(c);
    return fail(RM_EDEVICE, "stream/event creation: %s", hipGetErrorString(e));
  }
  c->stream = c->own_stream;
  const char* na = getenv("RAYMARCH_NO_ACCEL");
  c->use_accel = !(na && na[0] == '1');
  const char* oc = getenv("RAYMARCH_OCTANTS");
  c->use_octants = !(oc && oc[0] == '0');  // directional tables: on unless RAYMARCH_OCTANTS=0
  const char* km = getenv("RAYMARCH_KERNEL");
  c->wave_mode = km && strcmp(km, "wave") == 0;  // experimental, slower: see DESIGN.md
  c->stream_mode = km && strcmp(km, "stream") == 0;  // experimental task-queue pipeline
  c->split_mode = km && strcmp(km, "split") == 0;
  c->phase_mode = km && strcmp(km, "phases") == 0;
  const char* spw = getenv("RAYMARCH_SPLIT_WAVES");
  if (spw) sscanf(spw, "%d,%d", &c->split_tw, &c->split_lw);
  const char* xr = getenv("RAYMARCH_XCD_ROWS");
  if (xr) c->xcd_rows = xr[0] != '0';
  const char* pk = getenv("RAYMARCH_PASS_PACK");
  if (pk && atoi(pk) >= 0 && atoi(pk) <= 6) c->pass_pack = atoi(pk);
  const char* sw = getenv("RAYMARCH_STRAIGHT_WAVES");
  if (sw && atoi(sw) >= 3 && atoi(sw) <= 8) c->straight_waves = atoi(sw);
  const char* bs = getenv("RAYMARCH_BATCH_SAMPLES");
  if (bs && atoll(bs) > 0) c->batch_samples = atoll(bs);
  const char* mw = getenv("RAYMARCH_WAVES");
  if (mw && atoi(mw) >= 2 && atoi(mw) <= 5) c->min_waves = atoi(mw);
  c->num_cus = prop.multiProcessorCount;
  c->wave_blocks = c->num_cus * rmk::wave_kernel_blocks_per_cu(c->min_waves);
  const char* wl = getenv("RAYMARCH_WAIT_LANES");
  if (wl && atoi(wl) > 0) c->wait_lanes = atoi(wl);
  const char* wb = getenv("RAYMARCH_WAVE_BLOCKS");
  if (wb && atoi(wb) > 0) c->wave_blocks = atoi(wb);
  *out = c;
  return RM_OK;
}

void rm_destroy(rm_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  rmk::dump_work_stats();
  if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
  DevBuf* bufs[] = {&c->vox_buf, &c->mc_buf, &c->opts_buf, &c->pix_buf, &c->argb_buf, &c->tile_buf, &c->dist_buf, &c->tmp_buf, &c->surf_buf, &c->stage_buf, &c->queue_buf, &c->work_buf,
                    &c->cnt_buf, &c->prim_a, &c->prim_b, &c->prim_o, &c->gen_buf, &c->sat_buf, &c->lin_buf,
                    &c->sdf_buf};
  for (DevBuf* b : bufs) b->release();
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int rm_set_stream(rm_ctx* c, void* hip_stream) {
  int rc = check_ctx(c);
  if (rc) return rc;
  c->stream = hip_stream == RM_OWN_STREAM ? c->own_stream : static_cast<hipStream_t>(hip_stream);
  return RM_OK;
}

int rm_synchronize(rm_ctx* c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RM_OK;
}

static int check_res(int rx, int ry, int rz) {
  if (rx <= 0 || ry <= 0 || rz <= 0) return fail(RM_EINVAL, "volume resolution %dx%dx%d", rx, ry, rz);
  // the kernels index with 32-bit ints like the reference (renderer.cl:167)
  if ((long long)rx * ry * rz > 0x7fffffffLL)
    return fail(RM_EINVAL, "volume %dx%dx%d exceeds the 2^31-1 voxel index range", rx, ry, rz);
  return RM_OK;
}

int rm_set_volume(rm_ctx* c, const uint8_t* voxels, int rx, int ry, int rz) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!voxels) return fail(RM_EINVAL, "voxels is NULL");
  rc = check_res(rx, ry, rz);
  if (rc) return rc;
  const size_t bytes = (size_t)rx * ry * rz;
  HIP_TRY(c->vox_buf.reserve(bytes));
  HIP_TRY(hipMemcpyAsync(c->vox_buf.p, voxels, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->d_vox = static_cast<const uint8_t*>(c->vox_buf.p);
  c->rx = rx; c->ry = ry; c->rz = rz;
  c->accel_iso = -1;
  return RM_OK;
}

int rm_set_volume_device(rm_ctx* c, const void* d_voxels, int rx, int ry, int rz) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!d_voxels) return fail(RM_EINVAL, "d_voxels is NULL");
  rc = check_res(rx, ry, rz);
  if (rc) return rc;
  c->d_vox = static_cast<const uint8_t*>(d_voxels);
  c->rx = rx; c->ry = ry; c->rz = rz;
  c->accel_iso = -1;
  return RM_OK;
}

int rm_make_gyroid_volume(rm_ctx* c, int rx, int ry, int rz, uint8_t* voxels_out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  rc = check_res(rx, ry, rz);
  if (rc) return rc;
  const size_t bytes = (size_t)rx * ry * rz;
  HIP_TRY(c->vox_buf.reserve(bytes));
  HIP_TRY(rmk::launch_gyroid(c->stream, static_cast<uint8_t*>(c->vox_buf.p), rx, ry, rz));
  if (voxels_out)
    HIP_TRY(hipMemcpyAsync(voxels_out, c->vox_buf.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->d_vox = static_cast<const uint8_t*>(c->vox_buf.p);
  c->rx = rx; c->ry = ry; c->rz = rz;
  c->accel_iso = -1;
  return RM_OK;
}

// shared tail of the device-side volume producers: optional copy back, make resident
static int adopt_generated(rm_ctx* c, int rx, int ry, int rz, uint8_t* voxels_out) {
  const size_t bytes = (size_t)rx * ry * rz;
  if (voxels_out)
    HIP_TRY(hipMemcpyAsync(voxels_out, c->vox_buf.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->d_vox = static_cast<const uint8_t*>(c->vox_buf.p);
  c->rx = rx; c->ry = ry; c->rz = rz;
  c->accel_iso = -1;
  return RM_OK;
}

int rm_make_terrain_volume(rm_ctx* c, int rx, int ry, int rz, uint8_t* voxels_out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  rc = check_res(rx, ry, rz);
  if (rc) return rc;
  if (rx < 4 || rz < 4) return fail(RM_EINVAL, "terrain needs rx, rz >= 4 (walls are 4 voxels thick)");
  HIP_TRY(c->vox_buf.reserve((size_t)rx * ry * rz));
  HIP_TRY(rmk::launch_terrain(c->stream, static_cast<uint8_t*>(c->vox_buf.p), rx, ry, rz));
  return adopt_generated(c, rx, ry, rz, voxels_out);
}

int rm_voxelize_vertices(rm_ctx* c, const double* xyz, long long n_vertices, int res, int ks,
                         uint8_t* voxels_out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  rc = check_res(res, res, res);
  if (rc) return rc;
  if (n_vertices < 0 || (!xyz && n_vertices > 0)) return fail(RM_EINVAL, "bad vertex array");
  if (ks > res) ks = res;
  // mesh-scale (meshvoxel.clj:16-25): bounding box, largest extent, centring offsets --
  // a min/max pass over data that arrives from the host anyway
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (long long i = 0; i < n_vertices; i++)
    for (int k = 0; k < 3; k++) {
      const double v = xyz[3 * i + k];
      if (v != v) return fail(RM_EINVAL, "vertex %lld is NaN", i);
      if (i == 0 || v < lo[k]) lo[k] = v;
      if (i == 0 || v > hi[k]) hi[k] = v;
    }
  const double sx = hi[0] - lo[0], sy = hi[1] - lo[1], sz = hi[2] - lo[2];
  const double md = sx > sy ? (sx > sz ? sx : sz) : (sy > sz ? sy : sz);
  if (n_vertices > 0 && !(md > 0.0)) return fail(RM_EINVAL, "degenerate mesh: zero extent");
  const double size[3] = {sx, sy, sz};
  double off[3] = {0, 0, 0};
  for (int k = 0; k < 3; k++) off[k] = (0.5 * (double)res) * (1.0 - size[k] / md);
  const double s = n_vertices > 0 ? (double)res / md : 1.0;
  const size_t bytes = (size_t)res * res * res;
  HIP_TRY(c->vox_buf.reserve(bytes));
  if (n_vertices > 0) {
    HIP_TRY(c->gen_buf.reserve((size_t)n_vertices * 24));
    HIP_TRY(hipMemcpyAsync(c->gen_buf.p, xyz, (size_t)n_vertices * 24, hipMemcpyHostToDevice, c->stream));
  }
  HIP_TRY(rmk::launch_splat(c->stream, static_cast<uint8_t*>(c->vox_buf.p),
                            static_cast<const double*>(c->gen_buf.p), n_vertices, lo, off, s, res, ks));
  return adopt_generated(c, res, res, res, voxels_out);
}

int rm_make_heatmap_volume(rm_ctx* c, const uint32_t* argb, int res, double amp, uint8_t* voxels_out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  rc = check_res(res, res, res);
  if (rc) return rc;
  if (!argb) return fail(RM_EINVAL, "argb is NULL");
  if (amp != amp) return fail(RM_EINVAL, "amp is NaN");
  HIP_TRY(c->vox_buf.reserve((size_t)res * res * res));
  HIP_TRY(c->gen_buf.reserve((size_t)res * res * 4));
  HIP_TRY(hipMemcpyAsync(c->gen_buf.p, argb, (size_t)res * res * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(rmk::launch_heatmap(c->stream, static_cast<uint8_t*>(c->vox_buf.p),
                              static_cast<const uint32_t*>(c->gen_buf.p), res, amp));
  return adopt_generated(c, res, res, res, voxels_out);
}

int rm_render_image(rm_ctx* c, const float* mc, const void* opts544, float* pixels, int n) {
  return render_pass_host(c, mc, opts544, pixels, n, 0, n, nullptr);
}
int rm_render_image_range(rm_ctx* c, const float* mc, const void* opts544, float* pixels, int n,
                          int id0, int id1) {
  if (id0 < 0 || id1 < id0) return fail(RM_EINVAL, "id range [%d,%d)", id0, id1);
  return render_pass_host(c, mc, opts544, pixels, n, id0, id1, nullptr);
}
int rm_render_image_counted(rm_ctx* c, const float* mc, const void* opts544, float* pixels, int n,
                            rm_counters* out) {
  if (!out) return fail(RM_EINVAL, "out is NULL");
  return render_pass_host(c, mc, opts544, pixels, n, 0, n, out);
}

int rm_tonemap_image(rm_ctx* c, const float* pixels, const void* opts544, uint32_t* argb, int n) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (n < 0) return fail(RM_EINVAL, "n = %d", n);
  if (!opts544 || ((!pixels || !argb) && n > 0)) return fail(RM_EINVAL, "NULL buffer");
  if (n == 0) return RM_OK;
  HIP_TRY(c->opts_buf.reserve(RM_OPTS_BYTES));
  HIP_TRY(c->pix_buf.reserve((size_t)n * 16));
  HIP_TRY(c->argb_buf.reserve((size_t)n * 4));
  HIP_TRY(hipMemcpyAsync(c->opts_buf.p, opts544, RM_OPTS_BYTES, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->pix_buf.p, pixels, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(rmk::launch_tonemap(c->stream, static_cast<const float*>(c->pix_buf.p),
                              static_cast<const RmOpts*>(c->opts_buf.p),
                              static_cast<uint32_t*>(c->argb_buf.p), n));
  HIP_TRY(hipMemcpyAsync(argb, c->argb_buf.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RM_OK;
}

int rm_render_frame(rm_ctx* c, const void* opts_array, const float* mc_array, int iter, int n,
                    float* pixels_out, uint32_t* argb_out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!opts_array || !mc_array) return fail(RM_EINVAL, "NULL buffer");
  if (iter <= 0) return fail(RM_EINVAL, "iter = %d", iter);
  if (!c->d_vox) return fail(RM_ESTATE, "rm_set_volume has not been called");
  for (int i = 0; i < iter; i++) {
    rc = check_opts(c, static_cast<const char*>(opts_array) + (size_t)i * RM_OPTS_BYTES, n);
    if (rc) return rc;
  }
  if (n == 0) return RM_OK;
  HIP_TRY(c->opts_buf.reserve((size_t)iter * RM_OPTS_BYTES));
  HIP_TRY(c->mc_buf.reserve((size_t)iter * RM_TABLE_FLOATS * 4));
  HIP_TRY(c->pix_buf.reserve((size_t)n * 16));
  if (argb_out) HIP_TRY(c->argb_buf.reserve((size_t)n * 4));
  HIP_TRY(hipMemcpyAsync(c->opts_buf.p, opts_array, (size_t)iter * RM_OPTS_BYTES,
                         hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->mc_buf.p, mc_array, (size_t)iter * RM_TABLE_FLOATS * 4,
                         hipMemcpyHostToDevice, c->stream));
  RmOpts o0;
  memcpy(&o0, opts_array, sizeof o0);
  const int tiles = rmk::tiles_total(o0.resolution[0], n);
  HIP_TRY(c->tile_buf.reserve((size_t)tiles * 64 * 16));
  std::vector<unsigned char> same;
  records_same_as_prev(opts_array, iter, &same);
  std::vector<RmOpts> recs(iter);
  memcpy(recs.data(), opts_array, (size_t)iter * RM_OPTS_BYTES);
  std::vector<int> isos(iter);
  for (int i = 0; i < iter; i++)
    isos[i] = static_cast<const uint8_t*>(opts_array)[(size_t)i * RM_OPTS_BYTES + offsetof(RmOpts, isoVal)];
  rc = frame_on_device(c, static_cast<const RmOpts*>(c->opts_buf.p),
                       static_cast<const float*>(c->mc_buf.p), o0.resolution[0], iter, n, 0, 1,
                       static_cast<float*>(c->tile_buf.p), isos.data(), same.data(), recs.data());
  if (rc) return rc;
  HIP_TRY(rmk::launch_resolve(c->stream, static_cast<const float*>(c->tile_buf.p), 1, tiles,
                              static_cast<const RmOpts*>(c->opts_buf.p),
                              pixels_out ? static_cast<float*>(c->pix_buf.p) : nullptr,
                              argb_out ? static_cast<uint32_t*>(c->argb_buf.p) : nullptr, n));
  if (pixels_out)
    HIP_TRY(hipMemcpyAsync(pixels_out, c->pix_buf.p, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
  if (argb_out)
    HIP_TRY(hipMemcpyAsync(argb_out, c->argb_buf.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RM_OK;
}

int rm_set_sdf_volume(rm_ctx* c, const float* sdf, int rx, int ry, int rz) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!sdf) return fail(RM_EINVAL, "sdf is NULL");
  rc = check_res(rx, ry, rz);
  if (rc) return rc;
  if (rx < 2 || ry < 2 || rz < 2) return fail(RM_EINVAL, "a distance field needs at least 2 cells per axis");
  const size_t bytes = (size_t)rx * ry * rz * 4;
  HIP_TRY(c->sdf_buf.reserve(bytes));
  HIP_TRY(hipMemcpyAsync(c->sdf_buf.p, sdf, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->sdf_rx = rx; c->sdf_ry = ry; c->sdf_rz = rz;
  return RM_OK;
}

int rm_render_sdf_frame(rm_ctx* c, const void* opts_array, const float* mc_array, int iter, int n,
                        float* pixels_out, uint32_t* argb_out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!opts_array || !mc_array) return fail(RM_EINVAL, "NULL buffer");
  if (iter <= 0 || n < 0) return fail(RM_EINVAL, "iter = %d, n = %d", iter, n);
  if (!c->sdf_rx) return fail(RM_ESTATE, "rm_set_sdf_volume has not been called");
  std::vector<RmOpts> recs(iter);
  memcpy(recs.data(), opts_array, (size_t)iter * RM_OPTS_BYTES);
  for (int i = 0; i < iter; i++) {
    const RmOpts& o = recs[i];
    if (o.resolution[0] <= 0 || o.resolution[1] <= 0)
      return fail(RM_EINVAL, "TRenderOpts.resolution = (%d,%d)", o.resolution[0], o.resolution[1]);
    if (o.voxelRes[0] != c->sdf_rx || o.voxelRes[1] != c->sdf_ry || o.voxelRes[2] != c->sdf_rz)
      return fail(RM_EINVAL, "TRenderOpts.voxelRes = (%d,%d,%d) does not match the distance field %dx%dx%d",
                  o.voxelRes[0], o.voxelRes[1], o.voxelRes[2], c->sdf_rx, c->sdf_ry, c->sdf_rz);
    if (o.numLights > 4) return fail(RM_EINVAL, "TRenderOpts.numLights = %d (max 4)", (int)o.numLights);
  }
  if (n == 0) return RM_OK;
  HIP_TRY(c->opts_buf.reserve((size_t)iter * RM_OPTS_BYTES));
  HIP_TRY(c->mc_buf.reserve((size_t)iter * RM_TABLE_FLOATS * 4));
  HIP_TRY(c->pix_buf.reserve((size_t)n * 16));
  if (argb_out) HIP_TRY(c->argb_buf.reserve((size_t)n * 4));
  HIP_TRY(hipMemcpyAsync(c->opts_buf.p, opts_array, (size_t)iter * RM_OPTS_BYTES,
                         hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->mc_buf.p, mc_array, (size_t)iter * RM_TABLE_FLOATS * 4,
                         hipMemcpyHostToDevice, c->stream));
  const int resx = recs[0].resolution[0];
  const int tiles = rmk::tiles_total(resx, n);
  HIP_TRY(c->tile_buf.reserve((size_t)tiles * 64 * 16));
  std::vector<unsigned char> same;
  records_same_as_prev(opts_array, iter, &same);
  std::vector<int> isos(iter, 0);
  c->sdf_frame = true;
  rc = frame_on_device(c, static_cast<const RmOpts*>(c->opts_buf.p),
                       static_cast<const float*>(c->mc_buf.p), resx, iter, n, 0, 1,
                       static_cast<float*>(c->tile_buf.p), isos.data(), same.data(), recs.data());
  c->sdf_frame = false;
  if (rc) return rc;
  HIP_TRY(rmk::launch_resolve(c->stream, static_cast<const float*>(c->tile_buf.p), 1, tiles,
                              static_cast<const RmOpts*>(c->opts_buf.p),
                              pixels_out ? static_cast<float*>(c->pix_buf.p) : nullptr,
                              argb_out ? static_cast<uint32_t*>(c->argb_buf.p) : nullptr, n));
  if (pixels_out)
    HIP_TRY(hipMemcpyAsync(pixels_out, c->pix_buf.p, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
  if (argb_out)
    HIP_TRY(hipMemcpyAsync(argb_out, c->argb_buf.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RM_OK;
}

int rm_tiles_per_part(int resx, int n, int parts) {
  if (resx <= 0 || n < 0 || parts < 1) return fail(RM_EINVAL, "rm_tiles_per_part(%d,%d,%d)", resx, n, parts);
  return rmk::tiles_per_part(rmk::tiles_total(resx, n), parts);
}

int rm_check_device_opts(rm_ctx* c, const void* d_opts, int iter, int n, int width) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!d_opts || iter <= 0) return fail(RM_EINVAL, "d_opts NULL or iter = %d", iter);
  if (!c->d_vox) return fail(RM_ESTATE, "rm_set_volume has not been called");
  std::vector<RmOpts> recs(iter);
  HIP_TRY(hipMemcpyAsync(recs.data(), d_opts, (size_t)iter * RM_OPTS_BYTES, hipMemcpyDeviceToHost,
                         c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->dev_iso.clear();
  c->dev_iso_src = nullptr;
  for (int i = 0; i < iter; i++) {
    rc = check_opts(c, &recs[i], n);
    if (rc) return rc;
    if (recs[i].resolution[0] != width)
      return fail(RM_EINVAL, "record %d: resolution.x = %d but width = %d", i, recs[i].resolution[0], width);
    c->dev_iso.push_back(recs[i].isoVal);
  }
  records_same_as_prev(recs.data(), iter, &c->dev_same);
  c->dev_recs = recs;
  c->dev_iso_src = d_opts;
  // build the derived structures of the (first) hit threshold now, not inside the first frame
  rmk::Accel accel;
  rc = ensure_accel(c, recs[0].isoVal, &accel);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RM_OK;
}

int rm_frame_device(rm_ctx* c, const void* d_opts, const float* d_mc, int iter, int n, int width,
                    int tile_first, int tile_stride, float* d_tiles) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!d_opts || !d_mc || !d_tiles) return fail(RM_EINVAL, "NULL device buffer");
  if (iter <= 0 || n <= 0 || width <= 0) return fail(RM_EINVAL, "iter = %d, n = %d, width = %d", iter, n, width);
  if (tile_stride < 1 || tile_first < 0 || tile_first >= tile_stride)
    return fail(RM_EINVAL, "tile partition (%d,%d)", tile_first, tile_stride);
  if (!c->d_vox) return fail(RM_ESTATE, "rm_set_volume has not been called");
  if (c->dev_iso_src != d_opts || (int)c->dev_iso.size() != iter)
    return fail(RM_ESTATE, "rm_check_device_opts(d_opts, iter=%d, ...) must validate the records first", iter);
  return frame_on_device(c, static_cast<const RmOpts*>(d_opts), d_mc, width, iter, n, tile_first,
                         tile_stride, d_tiles, c->dev_iso.data(), c->dev_same.data(),
                         c->dev_recs.data());
}

int rm_resolve_device(rm_ctx* c, const float* d_tiles_all, int parts, const void* d_opts, int n,
                      int width, float* d_pixels, uint32_t* d_argb) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!d_tiles_all || !d_opts) return fail(RM_EINVAL, "NULL device buffer");
  if (parts < 1 || n <= 0 || width <= 0) return fail(RM_EINVAL, "parts = %d, n = %d, width = %d", parts, n, width);
  const int tpp = rmk::tiles_per_part(rmk::tiles_total(width, n), parts);
  HIP_TRY(rmk::launch_resolve(c->stream, d_tiles_all, parts, tpp, static_cast<const RmOpts*>(d_opts),
                              d_pixels, d_argb, n));
  return RM_OK;
}

int rm_last_frame_timing(rm_ctx* c, float* ms, int* launches) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->timed) return fail(RM_ESTATE, "no frame has been rendered");
  HIP_TRY(hipEventSynchronize(c->ev1));
  float t = 0.f;
  HIP_TRY(hipEventElapsedTime(&t, c->ev0, c->ev1));
  if (ms) *ms = t;
  if (launches) *launches = c->launches;
  return RM_OK;
}

int rm_debug_get_accel(rm_ctx* c, int iso, uint8_t* dist_out, uint32_t* surf_out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->d_vox) return fail(RM_ESTATE, "rm_set_volume has not been called");
  if (iso < 0 || iso > 255) return fail(RM_EINVAL, "iso = %d", iso);
  if (!c->use_accel) return fail(RM_ESTATE, "acceleration structures are disabled (RAYMARCH_NO_ACCEL)");
  rmk::Accel accel;
  rc = ensure_accel(c, iso, &accel);
  if (rc) return rc;
  const size_t vox = (size_t)c->rx * c->ry * c->rz;
#if RM_BRICKS
  if (dist_out) {  // what the kernels read, converted back to row-major
    HIP_TRY(rmk::launch_brick(c->stream, static_cast<uint8_t*>(c->tmp_buf.p), c->rx, c->ry, c->rz,
                              const_cast<uint8_t*>(accel.dist), false));
    HIP_TRY(hipMemcpyAsync(dist_out, c->tmp_buf.p, vox, hipMemcpyDeviceToHost, c->stream));
  }
#else
  if (dist_out) HIP_TRY(hipMemcpyAsync(dist_out, accel.dist, vox, hipMemcpyDeviceToHost, c->stream));
#endif
  if (surf_out) HIP_TRY(hipMemcpyAsync(surf_out, accel.surf, vox * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RM_OK;
}

int rm_debug_get_octants(rm_ctx* c, int iso, uint8_t* oct_out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->d_vox) return fail(RM_ESTATE, "rm_set_volume has not been called");
  if (iso < 0 || iso > 255 || !oct_out) return fail(RM_EINVAL, "bad argument");
  rmk::Accel accel;
  rc = ensure_accel(c, iso, &accel);
  if (rc) return rc;
  if (!accel.dist || !accel.oct_stride)
    return fail(RM_ESTATE, "directional tables are not built (disabled, or volume too large)");
  const size_t vox = (size_t)c->rx * c->ry * c->rz;
#if RM_BRICKS
  for (int t = 0; t < 8; t++) {
    HIP_TRY(rmk::launch_brick(c->stream, static_cast<uint8_t*>(c->tmp_buf.p), c->rx, c->ry, c->rz,
                              const_cast<uint8_t*>(accel.dist) + (size_t)(t + 1) * accel.oct_stride, false));
    HIP_TRY(hipMemcpyAsync(oct_out + (size_t)t * vox, c->tmp_buf.p, vox, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
#else
  HIP_TRY(hipMemcpyAsync(oct_out, accel.dist + vox, vox * 8, hipMemcpyDeviceToHost, c->stream));
#endif
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RM_OK;
}

int rm_selftest_prims(rm_ctx* c, int op, const float* a, const float* b, uint32_t* out, int n) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!a || !out || n < 0 || op < 0 || op > 9) return fail(RM_EINVAL, "bad argument");
  if (n == 0) return RM_OK;
  const size_t bytes = (size_t)n * 4;
  HIP_TRY(c->prim_a.reserve(bytes));
  HIP_TRY(c->prim_o.reserve(bytes));
  HIP_TRY(hipMemcpyAsync(c->prim_a.p, a, bytes, hipMemcpyHostToDevice, c->stream));
  if (b) {
    HIP_TRY(c->prim_b.reserve(bytes));
    HIP_TRY(hipMemcpyAsync(c->prim_b.p, b, bytes, hipMemcpyHostToDevice, c->stream));
  }
  HIP_TRY(rmk::launch_prims(c->stream, op, static_cast<const float*>(c->prim_a.p),
                            b ? static_cast<const float*>(c->prim_b.p) : nullptr,
                            static_cast<uint32_t*>(c->prim_o.p), n));
  HIP_TRY(hipMemcpyAsync(out, c->prim_o.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RM_OK;
}

}  // extern "C"
