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
#include <algorithm>
#include <cmath>
#include <atomic>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/raymarch_hip.h"
#include "rm_kernels.h"
#include "rm_shade.hpp"

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

// The resident byte volume and everything derived from it (rm_accel.hip).  Contexts of one
// device may share it (rm_share_volume): frames in flight on several streams then read ONE
// set of tables -- 208 MiB at 256^3 -- instead of one per context.
struct Volume {
  std::mutex mu;
  int device = 0;
  const uint8_t* d_vox = nullptr;  // owned (vox_buf) or borrowed
  DevBuf vox_buf, dist_buf, tmp_buf, surf_buf;
  int rx = 0, ry = 0, rz = 0;
  int accel_iso = -1;              // isoVal the tables were built for, -1 = stale
  unsigned long long oct_stride = 0;
  bool bricked = false;            // dist8 / oct8 stored in 8x4x4-cell bricks (volumes beyond the caches)
  unsigned long long generation = 0;  // bumped whenever the bytes (may) have changed
  double accel_build_ms = 0.0;     // wall time of the last table build (reported by bench.py)
  ~Volume() {
    (void)hipSetDevice(device);
    vox_buf.release(); dist_buf.release(); tmp_buf.release(); surf_buf.release();
  }
};

}  // namespace

struct rm_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::shared_ptr<Volume> vol;
  // rm_stage_volume_device: the NEXT volume, whose derived tables are built on prep_stream while the resident one renders
  std::shared_ptr<Volume> staged;
  hipStream_t prep_stream = nullptr;
  hipEvent_t ev_staged = nullptr;     // prep_stream: the staged volume's tables are complete
  hipEvent_t ev_back_free = nullptr;  // stream: every frame that read the volume retired by the last commit has been enqueued before it
  hipEvent_t ev_s0 = nullptr, ev_s1 = nullptr;  // timed: the staged build
  bool staged_ready = false, back_free_pending = false, staged_timing = false;
  DevBuf mc_buf, opts_buf, pix_buf, argb_buf, tile_buf, atile_buf, cnt_buf, prim_a, prim_b, prim_o, gen_buf, sdf_buf, sdfq_buf;
  int sdf_rx = 0, sdf_ry = 0, sdf_rz = 0;  // quality mode: resident float distance field
  bool use_octants = true;   // RAYMARCH_OCTANTS=0: dist8 only (A/B)
  bool xcd_rows = true;      // RAYMARCH_XCD_ROWS=0: plain block order
  int xcd_2d_forced = -1;    // RAYMARCH_XCD_2D=0: whole tile rows per XCD (rounds 1-5); 1/2/4/8: 2-D units of 1/8 .. 1/64 row; default: chosen per launch
  bool rows_desc = true;     // RAYMARCH_ROW_ORDER=asc: tile rows top to bottom (rounds 2-4); default bottom to top
  bool rows_band = false;    // RAYMARCH_ROW_ORDER=band: the rows where the clip box covers most of the width first (volume_band,
                             // round 6).  Opt-in: no band exists at BASELINE's camera (the box fills the view), and at three
                             // farther cameras it measured -1.9 % / -0.3 % / +3.5 % (profiles/r06_experiments.txt)
  double band_fixed_lo = 0.0, band_fixed_hi = 0.0;  // RAYMARCH_ROW_BAND=lo,hi (fractions of the image height): that band instead
  int pass_pack = 4;         // RAYMARCH_PASS_PACK (0..6): log2 of the passes one wavefront holds at most.  Default 4 =
                             // 4 pixels x 16 passes, measured best for full groups (64 passes as 4 x 16 / 2 x 32 / 1 x 64:
                             // 136.0 / 137.3 / 140.3 ms) ...
  bool pass_pack_auto = true;  // ... except that a run of 20..31 passes goes out as ONE launch of 2 pixels x 32 slots instead of
                             // 16 + a partial group (25 passes: config 5 -2.7 %); off when RAYMARCH_PASS_PACK is given
  int cus = 256;             // compute units of the device (thin launches below)
  bool thin_wide = true;     // RAYMARCH_THIN=0: no widening of thin launches (A/B)
  int pack_waste = 60;       // RAYMARCH_PACK_WASTE: % of lane turns a partial last group may leave without a
                             // pass (their lanes still trace other lanes' secondary rays: 25 passes as
                             // 16 + 9 measured 12 % faster than as 6 x 4 + 1)
  bool use_accel = true;     // RAYMARCH_NO_ACCEL=1 -> plain fixed-step march (A/B)
  int bricks = -1;           // RAYMARCH_BRICKS=0/1: never / always store the tables in bricks (default: by size)
  bool pow2_tables = true;   // RAYMARCH_POW2=0: generic table indexing also for cubic power-of-two grids (A/B)
  int seed_cast = 0;         // rm_set_seed_cast: RM_SEED_CAST_X86 (default) / RM_SEED_CAST_GPU
  int contract = RM_CONTRACT_GFX950_DEFAULT;  // (library default, ABI 4) rm_set_contract: RM_CONTRACT_GFX950_STRICT / RM_CONTRACT_GFX950_DEFAULT / RM_CONTRACT_CPU_DEVICE
  // records validated by rm_check_device_opts
  std::vector<RmOpts> dev_recs;
  std::vector<unsigned char> dev_same;  // record i == record i-1 except .time
  const void* dev_src = nullptr;
  int dev_iter = 0, dev_n = 0, dev_width = 0;
  unsigned long long dev_generation = 0;
  // ev0 / ev1 bracket the render kernels of a frame; they point into a ring of pairs so that a caller can read the
  // device times of the last kTimingRing frames AFTER a timed loop (rm_frame_timing_history)
  static constexpr int kTimingRing = 32;
  hipEvent_t ev_ring[2 * kTimingRing] = {};
  int ring_launches[kTimingRing] = {};
  unsigned long long frame_seq = 0;  // frames rendered through frame_on_device
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr, ev_b0 = nullptr, ev_b1 = nullptr, ev_in = nullptr,
             ev_resolved = nullptr;
  bool resolved_once = false;
  // multi-device contexts: the device inputs last replicated to the other devices
  const void* repl_opts = nullptr;
  const void* repl_mc = nullptr;
  int repl_iter = 0;
  bool timed = false;
  int last_frame_world = 1;  // devices whose events belong to the last frame (rm_last_frame_breakdown)
  int launches = 0;
  // rm_pin_host_buffer: caller buffers page-locked for the host-buffer entry points
  std::vector<const void*> host_bufs;
  // rm_create_multi: the other devices of a multi-device context (this one is rank 0)
  std::vector<rm_ctx*> peers;
  rm_ctx* parent = nullptr;
};

// The context's arithmetic contract as the kernels are instantiated on it (rmk::ArithOf, rm_math.hpp) -- the ONE place
// that maps rm_set_contract / rm_set_seed_cast to a kernel family.  sdf: the quality mode has CPU-device arithmetic only.
static int contract_arith(const rm_ctx* c, bool sdf = false) {
  if (!sdf && c->contract == RM_CONTRACT_GFX950_DEFAULT) return 3;
  if (!sdf && c->contract == RM_CONTRACT_GFX950_STRICT) return 2;
  return c->seed_cast ? 1 : 0;
}

namespace {

int check_ctx(rm_ctx* c) {
  if (!c) return fail(RM_EINVAL, "rm_ctx is NULL");
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail(RM_EDEVICE, "hipSetDevice(%d): %s", c->device, hipGetErrorString(e));
  return RM_OK;
}

bool have_volume(rm_ctx* c) { return c->vol && c->vol->d_vox; }

// The record's own fields must be usable before a kernel trusts them.
int check_opts(rm_ctx* c, const void* opts544, int n) {
  RmOpts o;
  memcpy(&o, opts544, sizeof o);
  const Volume& v = *c->vol;
  if (o.resolution[0] <= 0 || o.resolution[1] <= 0)
    return fail(RM_EINVAL, "TRenderOpts.resolution = (%d,%d)", o.resolution[0], o.resolution[1]);
  if (o.voxelRes[0] != v.rx || o.voxelRes[1] != v.ry || o.voxelRes[2] != v.rz ||
      o.voxelRes[3] != v.rx * v.ry)
    return fail(RM_EINVAL, "TRenderOpts.voxelRes = (%d,%d,%d,%d) does not match the volume %dx%dx%d",
                o.voxelRes[0], o.voxelRes[1], o.voxelRes[2], o.voxelRes[3], v.rx, v.ry, v.rz);
  if (o.numLights > 4) return fail(RM_EINVAL, "TRenderOpts.numLights = %d (max 4)", (int)o.numLights);
  if (n < 0) return fail(RM_EINVAL, "n = %d", n);
  return RM_OK;
}
// all records of a frame: each valid, all with record 0's image width (the accumulator layout,
// the tile geometry and the work-item -> pixel map are one per frame)
int check_frame_opts(rm_ctx* c, const RmOpts* recs, int iter, int n, int width) {
  for (int i = 0; i < iter; i++) {
    int rc = check_opts(c, &recs[i], n);
    if (rc) return rc;
    if (recs[i].resolution[0] != width)
      return fail(RM_EINVAL, "record %d: resolution.x = %d but the frame is %d wide", i, recs[i].resolution[0], width);
  }
  return RM_OK;
}

// can this volume have derived tables at all (walk_step's index arithmetic)?
static bool tables_possible(const rm_ctx* c, const Volume& v) {
  if (!c->use_accel) return false;
  // walk_step indexes with 24-bit multiplies: fall back to the plain march otherwise
  if ((long long)v.ry * v.rz >= (1 << 24) || v.rx >= (1 << 24)) return false;
  return (size_t)v.rx * v.ry * v.rz < ((size_t)1 << 31);  // the kernels hold a cell index in an int
}

// Enqueue the build of dist8 / oct8 / surf32 of `v` for hit threshold `iso` on stream `st`, bracketed by the timed events
// t0 / t1.  No host synchronisation: the caller decides who may see the tables when (ensure_accel waits; a staged
// volume -- rm_stage_volume_device -- hands an event to the stream that will render it).
static int enqueue_tables(rm_ctx* c, Volume& v, int iso, hipStream_t st, hipEvent_t t0, hipEvent_t t1) {
  const size_t vox = (size_t)v.rx * v.ry * v.rz;
  // directional tables behind dist8 (measured -10 % frame time at 256^3, -12 % at 512^3 with
  // 8 % fill); table offsets are 64-bit, nine 1024^3 tables span 9 GiB
  const bool oct = c->use_octants;
  const int tables = oct ? 9 : 1;
  // Row-major tables have the cheapest index arithmetic and win while the Infinity Cache
  // catches most misses (bricks: +1..3 % at 256^3); beyond it misses go to HBM and locality wins: 1024^3
  // -6.6 % (layout 1), 512^3 -2.5 % with the brick number formed by shifts (layout 3, round 4; with layout 1's
  // multiplies and 64-bit offsets it had been a draw)
  const bool cube512 = v.rx == 512 && v.ry == 512 && v.rz == 512 && c->pow2_tables;
  const bool bricked = oct &&
                       (c->bricks >= 0 ? c->bricks == 1 : (vox * 13 > ((size_t)4 << 30) || cube512));
  const size_t tbytes = bricked ? (size_t)rmk::bricked_bytes(v.rx, v.ry, v.rz) : vox;
  HIP_TRY(v.dist_buf.reserve(tbytes * tables));
  uint8_t* lin = static_cast<uint8_t*>(v.dist_buf.p);
  HIP_TRY(v.surf_buf.reserve(vox * 4));
  HIP_TRY(hipEventRecord(t0, st));
  v.oct_stride = 0;
  if (oct) {
    // (scratch of the two-pass surf32: the region of table 0, which dist_from_oct_kernel writes last)
    HIP_TRY(rmk::build_accel(st, v.d_vox, v.rx, v.ry, v.rz, iso, nullptr, nullptr,
                             static_cast<uint32_t*>(v.surf_buf.p), lin));
    HIP_TRY(rmk::build_octants(st, v.d_vox, v.rx, v.ry, v.rz, iso, lin, bricked));
    v.oct_stride = tbytes;
    v.bricked = bricked;
  } else {
    v.bricked = false;
    HIP_TRY(v.tmp_buf.reserve(vox));
    HIP_TRY(rmk::build_accel(st, v.d_vox, v.rx, v.ry, v.rz, iso, lin,
                             static_cast<uint8_t*>(v.tmp_buf.p), static_cast<uint32_t*>(v.surf_buf.p),
                             static_cast<uint8_t*>(v.tmp_buf.p)));  // (tmp is free again after the distance passes)
  }
  HIP_TRY(hipEventRecord(t1, st));
  return RM_OK;
}

// Build (or reuse) dist8 / oct8 / surf32 for the hit threshold of this launch.
int ensure_accel(rm_ctx* c, int iso, rmk::Accel* out) {
  *out = rmk::Accel{};
  Volume& v = *c->vol;
  if (!tables_possible(c, v)) return RM_OK;
  const size_t vox = (size_t)v.rx * v.ry * v.rz;
  std::lock_guard<std::mutex> lock(v.mu);
  if (v.accel_iso != iso) {
    int rc = enqueue_tables(c, v, iso, c->stream, c->ev_b0, c->ev_b1);
    if (rc) return rc;
    // contexts that share the volume run on other streams: the tables are complete before
    // anybody else can see accel_iso
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev_b0, c->ev_b1);
    v.accel_build_ms = ms;
    v.accel_iso = iso;
  }
  out->oct_stride = v.oct_stride;
  out->bricked = v.bricked;
  // cubic power-of-two grid whose tables stay below 4 GiB: shift-or cell index, 32-bit buffer offsets
  // (bricked tables: only on the 512^3 grid with octants, walk_step LAYOUT 3 has that edge compiled in)
  if ((!v.bricked || ((v.rx == 512 || v.rx == 1024) && v.oct_stride)) && v.rx == v.ry && v.ry == v.rz && (v.rx & (v.rx - 1)) == 0 && v.rx >= 2 &&
      (vox * (v.oct_stride ? 9 : 1) < ((size_t)1 << 32) || (v.bricked && v.rx == 1024)) && c->pow2_tables) {
    unsigned k = 0;
    while ((1 << k) < v.rx) k++;
    out->log2res = k;
  }
  out->dist = static_cast<const uint8_t*>(v.dist_buf.p);
  out->surf = static_cast<const uint32_t*>(v.surf_buf.p);
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
  if (!have_volume(c)) return fail(RM_ESTATE, "rm_set_volume has not been called");
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
  HIP_TRY(rmk::launch_render_pass(c->stream, c->vol->d_vox, accel, static_cast<const float*>(c->mc_buf.p),
                                  static_cast<const RmOpts*>(c->opts_buf.p), o.resolution[0],
                                  static_cast<float*>(c->pix_buf.p), n, id0, id1, 0, 1, false, d_cnt,
                                  contract_arith(c)));
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

// The band of tile rows that holds most of a frame's work: the rows in which the clip box of the volume covers at least
// half as much of the image's width as in the row where it covers most (fractions of the image height, 0 = top row;
// *lo = *hi = 0: no band).  Coverage per tile row = extent of the box's projection along that row: the twelve edges of
// [voxelBoundsMin, voxelBoundsMax] in the view space of record `o`'s camera -- the inverse of the reference's
// cameraRayLookat (renderer.cl:456-465: direction = right * x + up * y + forward with x, y linear in the pixel position)
// --, clipped against a near plane, projected, and cut by the row's line (the projection of a convex box is the hull of its
// projected edges).  A HEURISTIC for scheduling only -- the frame kernel dispatches these rows first, the rows below them
// next, the rows above them (sky in the reference's scenes) last --: a poor band costs time, never pixels.
static void volume_band(const RmOpts& o, double* lo, double* hi) {
  *lo = *hi = 0.0;
  const double e[3] = {o.eyePos[0], o.eyePos[1], o.eyePos[2]};
  double f[3] = {o.targetPos[0] - e[0], o.targetPos[1] - e[1], o.targetPos[2] - e[2]};
  const double fl = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
  const int resy = o.resolution[1];
  if (!(fl > 1e-9) || !(o.fov > 1e-6f) || !(o.invAspect > 1e-6f) || resy < 16 || resy > (1 << 16)) return;
  for (double& v : f) v /= fl;
  double r[3] = {f[1] * o.up[2] - f[2] * o.up[1], f[2] * o.up[0] - f[0] * o.up[2], f[0] * o.up[1] - f[1] * o.up[0]};
  const double rl = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (!(rl > 1e-9)) return;
  for (double& v : r) v /= rl;
  const double u[3] = {r[1] * f[2] - r[2] * f[1], r[2] * f[0] - r[0] * f[2], r[0] * f[1] - r[1] * f[0]};
  double cv[8][3];  // view-space corners: (right, up, forward) components
  for (int k = 0; k < 8; k++) {
    const double p[3] = {(k & 1 ? o.voxelBoundsMax[0] : o.voxelBoundsMin[0]) - e[0],
                         (k & 2 ? o.voxelBoundsMax[1] : o.voxelBoundsMin[1]) - e[1],
                         (k & 4 ? o.voxelBoundsMax[2] : o.voxelBoundsMin[2]) - e[2]};
    cv[k][0] = p[0] * r[0] + p[1] * r[1] + p[2] * r[2];
    cv[k][1] = p[0] * u[0] + p[1] * u[1] + p[2] * u[2];
    cv[k][2] = p[0] * f[0] + p[1] * f[1] + p[2] * f[2];
  }
  // the edges in front of the near plane, projected: (x, y) of the view vector at both ends
  const double cmin = 0.02;
  double seg[12][4];
  int nseg = 0;
  for (int k = 0; k < 8; k++)
    for (int ax = 0; ax < 3; ax++) {
      if (k & (1 << ax)) continue;
      const double* A = cv[k];
      const double* B = cv[k | (1 << ax)];
      double t0 = 0.0, t1 = 1.0;
      const double ga = A[2] - cmin, gb = B[2] - cmin;
      if (ga < 0.0 && gb < 0.0) continue;
      if (ga < 0.0) t0 = ga / (ga - gb);
      else if (gb < 0.0) t1 = ga / (ga - gb);
      for (int q = 0; q < 2; q++) {
        const double t = q ? t1 : t0;
        const double c = std::max(A[2] + t * (B[2] - A[2]), cmin);
        seg[nseg][2 * q] = (A[0] + t * (B[0] - A[0])) / c;
        seg[nseg][2 * q + 1] = (A[1] + t * (B[1] - A[1])) / c;
      }
      nseg++;
    }
  if (!nseg) return;
  const int rows = (resy + 7) / 8;
  const double hx = 0.5 * (double)o.fov;  // |x| of the view vector at the image's sides
  std::vector<double> cov((size_t)rows, 0.0);
  double best = 0.0;
  for (int j = 0; j < rows; j++) {
    const double py = std::min((double)j * 8.0 + 4.0, (double)resy - 0.5);
    const double Y = -(double)o.invAspect * (py / (double)resy * (double)o.fov - 0.5 * (double)o.fov);  // renderer.cl:461-463
    double x0 = 1e30, x1 = -1e30;
    for (int q = 0; q < nseg; q++) {
      const double ya = seg[q][1] - Y, yb = seg[q][3] - Y;
      if (ya * yb > 0.0 || ya == yb) continue;
      const double x = seg[q][0] + ya / (ya - yb) * (seg[q][2] - seg[q][0]);
      x0 = std::min(x0, x);
      x1 = std::max(x1, x);
    }
    if (x1 >= x0) cov[(size_t)j] = std::max(0.0, std::min(x1, hx) - std::max(x0, -hx)) / (2.0 * hx);
    best = std::max(best, cov[(size_t)j]);
  }
  if (!(best > 0.0)) return;  // the box is nowhere in the image: no row walks the tables, any order will do
  int r0 = -1, r1 = -1;
  for (int j = 0; j < rows; j++)
    if (cov[(size_t)j] >= 0.5 * best) {
      if (r0 < 0) r0 = j;
      r1 = j + 1;
    }
  if (r0 <= 0 && r1 >= rows) return;  // every row is about as heavy: plain bottom to top
  *lo = (double)r0 / (double)rows;
  *hi = (double)r1 / (double)rows;
}

// What a frame writes: tile-major accumulators of a partition (the multi-GPU exchange unit),
// or -- unpartitioned -- the row-major float4 image and, with the last pass, the ARGB image.
struct FrameOut {
  float* acc = nullptr;
  uint32_t* argb = nullptr;
  bool row_major = false;
  int tile_first = 0, tile_stride = 1;
};

// The pipeline of core.clj:76-97 on resident inputs: accumulator from zero, `iter` passes in
// order.  Consecutive passes whose records are identical apart from .time (what
// core.clj:99-106 produces) and share a hit threshold go out pass-packed, as many per launch
// of the frame kernel as one wavefront holds (16, or 32 slots for a run of 20..31; 16 passes are ONE launch: the whole frame of BASELINE's
// headline configuration is ONE launch); more passes, or a record that differs otherwise,
// start a new launch, which continues from the accumulator the previous one left (launches
// of a stream are ordered).
int frame_on_device(rm_ctx* c, const RmOpts* d_opts, const float* d_mc, int resx, int iter, int n,
                    const FrameOut& out, const unsigned char* same_as_prev, const RmOpts* host_recs,
                    bool sdf_frame) {
  const int ring_slot = (int)(c->frame_seq % rm_ctx::kTimingRing);
  c->ev0 = c->ev_ring[2 * ring_slot];
  c->ev1 = c->ev_ring[2 * ring_slot + 1];
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  int launches = 0, run_end = 0, pp_log2 = 0;
  for (int i0 = 0; i0 < iter;) {
    int i1 = i0 + 1;
    if (run_end <= i0) {  // a new run of records that differ in .time only: its pass packing
      run_end = i1;
      while (run_end < iter && same_as_prev[run_end]) run_end++;
      const int run = run_end - i0;
      // (the same waste rule as choose_pass_pack: 32 slots only if at most pack_waste % of them stay without a pass)
      const bool one_of_32 = c->pass_pack_auto && run > 16 && run < 32 && 32.0 * 100.0 <= (100.0 + c->pack_waste) * run;
      pp_log2 = one_of_32 ? 5 : rmk::choose_pass_pack(run, c->pass_pack, c->pack_waste);
      // THIN LAUNCHES (round 6): few tiles x few passes leave most of the chip's 28 wavefront slots per CU empty, and a frame
      // then takes as long as its slowest wavefront.  Below 16 wavefronts per CU the pixels of a wavefront are halved (its
      // other lanes hold no pass; they still trace the secondary rays of the lanes that do): 256 x 256 x 1 of a 256^3 volume
      // 0.251 -> 0.162 ms, config 1 1.09 -> 0.85, 640 x 360 x 1 0.275 -> 0.217, 320 x 180 x 4 0.266 -> 0.227; 1280 x 720 x 1 and
      // anything larger keeps its packing (r06_experiments.txt section 14).  Pixels do not depend on it.
      if (c->pass_pack_auto && c->thin_wide) {
        const long long tiles = rmk::tiles_per_part(rmk::tiles_total(resx, n), out.tile_stride);
        while (pp_log2 < 4 && pp_log2 < c->pass_pack && (tiles << pp_log2) < 16ll * c->cus) pp_log2++;
      }
    }
    i1 = std::min(run_end, i0 + (1 << pp_log2));  // one launch = what one wavefront holds
    const size_t acc_bytes = out.row_major ? (size_t)n * 16
                                           : (size_t)rmk::tiles_per_part(rmk::tiles_total(resx, n), out.tile_stride) * 64 * 16;
    // A record that asks for more AO probes than a wavefront's exchange area holds results for (8: the
    // reference's default is aoIter = 5 -> 6 probes): the frame kernels of the table layouts with the grid edge
    // compiled in (256^3, 512^3, 1024^3 volumes: BASELINE's) take the probes in chunks of 8 (rm_shade.hpp
    // occlusion_wave); on the generic layouts such a record goes through the single-pass kernel, one launch per pass
    // (each lane traces its own secondary rays there) -- their frame kernels carry no second AO path for it.
    bool single_pass = false;
    rmk::Accel accel;
    if (!sdf_frame && host_recs[i0].aoIter + 1 > RM_WAVE_AO_PROBES) {
      int rc = ensure_accel(c, host_recs[i0].isoVal, &accel);
      if (rc) return rc;
      single_pass = !rmk::frame_takes_any_ao(accel);
    }
    if (single_pass) {
      if (i0 == 0) HIP_TRY(hipMemsetAsync(out.acc, 0, acc_bytes, c->stream));
      HIP_TRY(rmk::launch_render_pass(c->stream, c->vol->d_vox, accel, d_mc + (size_t)i0 * RM_TABLE_FLOATS, d_opts + i0,
                                      resx, out.acc, n, 0, n, out.tile_first, out.tile_stride, !out.row_major, nullptr,
                                      contract_arith(c)));
      launches++;
      i0 = i0 + 1;
      if (i0 == iter && out.argb && out.row_major)
        HIP_TRY(rmk::launch_tonemap(c->stream, out.acc, d_opts, out.argb, n, contract_arith(c)));
      continue;
    }
    rmk::FrameLaunch f;
    if (sdf_frame) {  // quality mode: no derived structures
      f.sdf = static_cast<const float*>(c->sdfq_buf.p);
    } else {
      int rc = ensure_accel(c, host_recs[i0].isoVal, &f.accel);
      if (rc) return rc;
      f.vox = c->vol->d_vox;
    }
    f.mc_all = d_mc + (size_t)i0 * RM_TABLE_FLOATS;
    f.opts_all = d_opts + i0;
    f.opts0 = d_opts;
    f.acc = out.acc;
    f.argb = (i1 == iter && out.row_major) ? out.argb : nullptr;
    f.resx = resx; f.n = n; f.passes = i1 - i0;
    f.tile_first = out.tile_first; f.tile_stride = out.tile_stride;
    f.pp_log2 = pp_log2;
    f.xcd_rows = c->xcd_rows;
    f.xcd_2d = c->xcd_2d_forced;  // (-1: the launcher picks the unit width, rm_kernels.hip frame_grid)
    // (tables that no cache holds -- beyond 4 GiB, 1024^3 -- gain nothing from wide units: the narrowest that fill two CUs)
    if (!sdf_frame && (size_t)c->vol->rx * c->vol->ry * c->vol->rz * 9 >= ((size_t)4 << 30)) f.unit_min_waves = 128;
    f.rows_desc = c->rows_desc;
    if (c->band_fixed_hi > c->band_fixed_lo) { f.band_lo = c->band_fixed_lo; f.band_hi = c->band_fixed_hi; }
    else if (c->rows_band && !sdf_frame) volume_band(host_recs[0], &f.band_lo, &f.band_hi);
    f.accumulate = i0 > 0;
    f.row_major = out.row_major;
    f.arith = contract_arith(c, sdf_frame);
    HIP_TRY(rmk::launch_render_frame(c->stream, f));
    launches++;
    i0 = i1;
  }
  // a partition's TonemapImage words (rm_frame_device_argb): per pixel, so the order of the accumulators does not matter
  if (out.argb && !out.row_major)
    HIP_TRY(rmk::launch_tonemap(c->stream, out.acc, d_opts, out.argb,
                                rmk::tiles_per_part(rmk::tiles_total(resx, n), out.tile_stride) * 64,
                                contract_arith(c, sdf_frame)));
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  c->timed = true;
  c->launches = launches;
  c->ring_launches[ring_slot] = launches;
  c->frame_seq++;
  if (!c->parent) c->last_frame_world = 1;  // (frame_multi_device raises it once its peers have taken part)
  return RM_OK;
}

// fresh, unshared volume state for a context (rm_set_volume* detach from a shared one)
int new_volume(rm_ctx* c) {
  Volume* v = new (std::nothrow) Volume();
  if (!v) return fail(RM_EDEVICE, "out of host memory");
  v->device = c->device;
  static std::atomic<unsigned long long> next_generation{1};
  v->generation = next_generation.fetch_add(1);
  c->vol.reset(v);
  return RM_OK;
}
void bump_generation(Volume& v) {
  v.accel_iso = -1;
  v.generation += (1ull << 32);
}

}  // namespace

extern "C" {

// used by rm_host.cpp (host-only translation unit) to report through rm_last_error()
int rm_host_fail_(int code, const char* msg) { return fail(code, "%s", msg); }

const char* rm_last_error(void) { return g_err; }
int rm_abi_version(void) { return 4; }  // 3: rm_ctx defaults to RM_CONTRACT_GFX950 (strict); 4: to RM_CONTRACT_GFX950_DEFAULT

int rm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

static int create_one(int device_id, rm_ctx** out) {
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
  c->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  for (int k = 0; k < 2 * rm_ctx::kTimingRing && e == hipSuccess; k++) e = hipEventCreate(&c->ev_ring[k]);
  c->ev0 = c->ev_ring[0];
  c->ev1 = c->ev_ring[1];
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_resolved);  // (timed: rm_last_frame_breakdown)
  if (e == hipSuccess) e = hipEventCreate(&c->ev_b0);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_b1);
  if (e == hipSuccess) {
    // the staged build is a chain of ~50 short dependent launches: at the frame kernel's priority each of them queues
    // behind a machine full of frame wavefronts and the chain ends 2 ms AFTER the frame (measured: no overlap at all);
    // at the highest priority its workgroups take the slots that retiring frame wavefronts free
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    const char* pp = getenv("RAYMARCH_PREP_PRIORITY");  // (A/B: 0 = the frame kernel's priority)
    e = hipStreamCreateWithPriority(&c->prep_stream, hipStreamNonBlocking, (pp && pp[0] == '0') ? 0 : hi);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_staged, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_back_free, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_s0);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_s1);
  if (e != hipSuccess) {
    rm_destroy(c);
    return fail(RM_EDEVICE, "stream/event creation: %s", hipGetErrorString(e));
  }
  c->stream = c->own_stream;
  if (new_volume(c) != RM_OK) {
    rm_destroy(c);
    return RM_EDEVICE;
  }
  // A/B switches (defaults are the measured best)
  const char* na = getenv("RAYMARCH_NO_ACCEL");
  c->use_accel = !(na && na[0] == '1');
  const char* oc = getenv("RAYMARCH_OCTANTS");
  c->use_octants = !(oc && oc[0] == '0');
  const char* xr = getenv("RAYMARCH_XCD_ROWS");
  if (xr) c->xcd_rows = xr[0] != '0';
  if (const char* x2 = getenv("RAYMARCH_XCD_2D")) c->xcd_2d_forced = (x2[0] >= '0' && x2[0] <= '8') ? x2[0] - '0' : 1;
  const char* ro = getenv("RAYMARCH_ROW_ORDER");
  if (ro) c->rows_desc = !(ro[0] == 'a');
  if (ro) c->rows_band = ro[0] == 'b';  // "desc" (default) / "band" / "asc"
  if (const char* rb = getenv("RAYMARCH_ROW_BAND")) (void)sscanf(rb, "%lf,%lf", &c->band_fixed_lo, &c->band_fixed_hi);
  const char* pk = getenv("RAYMARCH_PASS_PACK");
  if (pk && atoi(pk) >= 0 && atoi(pk) <= 6) { c->pass_pack = atoi(pk); c->pass_pack_auto = false; }
  const char* bk = getenv("RAYMARCH_BRICKS");
  if (bk && (bk[0] == '0' || bk[0] == '1')) c->bricks = bk[0] - '0';
  if (const char* tw = getenv("RAYMARCH_THIN")) c->thin_wide = tw[0] != '0';
  const char* pw = getenv("RAYMARCH_PACK_WASTE");
  if (pw && atoi(pw) >= 0 && atoi(pw) <= 6400) c->pack_waste = atoi(pw);
  const char* p2 = getenv("RAYMARCH_POW2");
  if (p2) c->pow2_tables = p2[0] != '0';
  *out = c;
  return RM_OK;
}

int rm_create(int device_id, rm_ctx** out) {
  if (!out) return fail(RM_EINVAL, "out is NULL");
  return create_one(device_id, out);
}

int rm_create_multi(const int* device_ids, int n_devices, rm_ctx** out) {
  if (!out) return fail(RM_EINVAL, "out is NULL");
  *out = nullptr;
  if (!device_ids || n_devices < 1 || n_devices > 64)
    return fail(RM_EINVAL, "rm_create_multi: %d devices", n_devices);
  rm_ctx* root = nullptr;
  int rc = create_one(device_ids[0], &root);
  if (rc) return rc;
  for (int r = 1; r < n_devices; r++) {
    rm_ctx* p = nullptr;
    rc = create_one(device_ids[r], &p);
    if (rc) {
      rm_destroy(root);
      return rc;
    }
    p->parent = root;
    root->peers.push_back(p);
    // direct xGMI access between the root and this device (the tile gather is a peer copy);
    // harmless when unavailable or when both ranks sit on one device: the copy still works
    if (device_ids[r] != device_ids[0]) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, device_ids[0], device_ids[r]) == hipSuccess && can) {
        (void)hipSetDevice(device_ids[0]);
        (void)hipDeviceEnablePeerAccess(device_ids[r], 0);
        (void)hipSetDevice(device_ids[r]);
        (void)hipDeviceEnablePeerAccess(device_ids[0], 0);
      }
      (void)hipGetLastError();
    }
  }
  (void)hipSetDevice(device_ids[0]);
  *out = root;
  return RM_OK;
}

int rm_num_devices(rm_ctx* c) { return c ? 1 + (int)c->peers.size() : 0; }

void rm_destroy(rm_ctx* c) {
  if (!c) return;
  for (rm_ctx* p : c->peers) {
    p->parent = nullptr;
    rm_destroy(p);
  }
  c->peers.clear();
  (void)hipSetDevice(c->device);
  if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
  for (const void* h : c->host_bufs) (void)hipHostUnregister(const_cast<void*>(h));
  c->host_bufs.clear();
  DevBuf* bufs[] = {&c->mc_buf, &c->opts_buf, &c->pix_buf, &c->argb_buf, &c->tile_buf, &c->cnt_buf,
                    &c->prim_a, &c->prim_b, &c->prim_o, &c->gen_buf, &c->sdf_buf, &c->sdfq_buf, &c->atile_buf};
  for (DevBuf* b : bufs) b->release();
  c->vol.reset();
  for (hipEvent_t ev : c->ev_ring)
    if (ev) (void)hipEventDestroy(ev);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_resolved) (void)hipEventDestroy(c->ev_resolved);
  if (c->ev_b0) (void)hipEventDestroy(c->ev_b0);
  if (c->ev_b1) (void)hipEventDestroy(c->ev_b1);
  if (c->prep_stream) { (void)hipStreamSynchronize(c->prep_stream); (void)hipStreamDestroy(c->prep_stream); }
  c->staged.reset();
  for (hipEvent_t ev : {c->ev_staged, c->ev_back_free, c->ev_s0, c->ev_s1})
    if (ev) (void)hipEventDestroy(ev);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int rm_set_stream(rm_ctx* c, void* hip_stream) {
  int rc = check_ctx(c);
  if (rc) return rc;
  c->stream = hip_stream == RM_OWN_STREAM ? c->own_stream : static_cast<hipStream_t>(hip_stream);
  return RM_OK;
}

int rm_set_seed_cast(rm_ctx* c, int mode) {
  if (!c) return fail(RM_EINVAL, "rm_ctx is NULL");
  if (mode != RM_SEED_CAST_X86 && mode != RM_SEED_CAST_GPU) return fail(RM_EINVAL, "unknown seed cast mode %d", mode);
  c->seed_cast = mode;
  for (rm_ctx* p : c->peers) p->seed_cast = mode;
  return RM_OK;
}

int rm_set_contract(rm_ctx* c, int contract) {
  if (!c) return fail(RM_EINVAL, "rm_ctx is NULL");
  if (contract != RM_CONTRACT_CPU_DEVICE && contract != RM_CONTRACT_GFX950_STRICT && contract != RM_CONTRACT_GFX950_DEFAULT)
    return fail(RM_EINVAL, "unknown arithmetic contract %d", contract);
  c->contract = contract;
  for (rm_ctx* p : c->peers) p->contract = contract;
  return RM_OK;
}

int rm_pin_host_buffer(rm_ctx* c, const void* p, size_t bytes) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!p || !bytes) return fail(RM_EINVAL, "rm_pin_host_buffer: NULL or empty buffer");
  for (const void* h : c->host_bufs)
    if (h == p) return RM_OK;
  HIP_TRY(hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault));
  c->host_bufs.push_back(p);
  return RM_OK;
}
int rm_unpin_host_buffer(rm_ctx* c, const void* p) {
  int rc = check_ctx(c);
  if (rc) return rc;
  for (size_t i = 0; i < c->host_bufs.size(); i++)
    if (c->host_bufs[i] == p) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      (void)hipHostUnregister(const_cast<void*>(p));
      c->host_bufs.erase(c->host_bufs.begin() + (long)i);
      return RM_OK;
    }
  return fail(RM_EINVAL, "rm_unpin_host_buffer: not a buffer rm_pin_host_buffer registered");
}

int rm_synchronize(rm_ctx* c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  for (rm_ctx* p : c->peers) {
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
  }
  HIP_TRY(hipSetDevice(c->device));
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

// A new resident volume invalidates everything that was derived from or validated against
// the old one: the context gets fresh volume state (it no longer shares anybody's tables)
// and forgets the device records rm_check_device_opts had accepted.
static int begin_new_volume(rm_ctx* c, bool keep_buffer) {
  c->dev_src = nullptr;
  c->dev_recs.clear();
  if (keep_buffer && c->vol && c->vol.use_count() == 1) {
    bump_generation(*c->vol);
    return RM_OK;
  }
  return new_volume(c);
}

// replicate the root's resident bytes on the other devices of a multi-device context
static int broadcast_volume(rm_ctx* c) {
  if (c->peers.empty()) return RM_OK;
  const Volume& v = *c->vol;
  const size_t bytes = (size_t)v.rx * v.ry * v.rz;
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (rm_ctx* p : c->peers) {
    HIP_TRY(hipSetDevice(p->device));
    int rc = begin_new_volume(p, true);
    if (rc) return rc;
    Volume& pv = *p->vol;
    pv.d_vox = nullptr;  // published again only once the copy has succeeded
    pv.rx = pv.ry = pv.rz = 0;
    HIP_TRY(pv.vox_buf.reserve(bytes));
    HIP_TRY(hipMemcpyPeerAsync(pv.vox_buf.p, p->device, v.d_vox, c->device, bytes, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    pv.d_vox = static_cast<const uint8_t*>(pv.vox_buf.p);
    pv.rx = v.rx; pv.ry = v.ry; pv.rz = v.rz;
  }
  HIP_TRY(hipSetDevice(c->device));
  return RM_OK;
}

int rm_set_volume(rm_ctx* c, const uint8_t* voxels, int rx, int ry, int rz) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!voxels) return fail(RM_EINVAL, "voxels is NULL");
  rc = check_res(rx, ry, rz);
  if (rc) return rc;
  const size_t bytes = (size_t)rx * ry * rz;
  rc = begin_new_volume(c, true);
  if (rc) return rc;
  Volume& v = *c->vol;
  // (a failed allocation or copy must not leave the old pointer behind: reserve() frees first)
  v.d_vox = nullptr;
  v.rx = v.ry = v.rz = 0;
  HIP_TRY(v.vox_buf.reserve(bytes));
  HIP_TRY(hipMemcpyAsync(v.vox_buf.p, voxels, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  v.d_vox = static_cast<const uint8_t*>(v.vox_buf.p);
  v.rx = rx; v.ry = ry; v.rz = rz;
  return broadcast_volume(c);
}

int rm_set_volume_device(rm_ctx* c, const void* d_voxels, int rx, int ry, int rz) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!d_voxels) return fail(RM_EINVAL, "d_voxels is NULL");
  rc = check_res(rx, ry, rz);
  if (rc) return rc;
  rc = begin_new_volume(c, true);
  if (rc) return rc;
  Volume& v = *c->vol;
  v.d_vox = static_cast<const uint8_t*>(d_voxels);
  v.rx = rx; v.ry = ry; v.rz = rz;
  return broadcast_volume(c);
}

// ---- animated volumes: the next volume's tables are built while the resident one renders (meshvoxel.clj:85-89
// make-heatmap-anim feeds core.clj:181-213 a new volume per frame; built inside the frame they cost 2.25 ms per 256^3
// volume on top of a 4 ms frame).  Two volume states per context: `vol` (resident, rendered from) and `staged`.
static int stage_volume(rm_ctx* c, const void* voxels, bool on_host, int rx, int ry, int rz, int iso_val) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!voxels) return fail(RM_EINVAL, "voxels is NULL");
  if (iso_val < 0 || iso_val > 255) return fail(RM_EINVAL, "iso_val = %d", iso_val);
  rc = check_res(rx, ry, rz);
  if (rc) return rc;
  if (!c->peers.empty() || c->parent)
    return fail(RM_ESTATE, "rm_stage_volume*: not on a multi-device context (rm_set_volume* replicates there)");
  // the slot: the volume the last commit retired, if nobody else still renders from it
  if (!c->staged || c->staged.use_count() != 1) {
    Volume* v = new (std::nothrow) Volume();
    if (!v) return fail(RM_EDEVICE, "out of host memory");
    v->device = c->device;
    c->staged.reset(v);
  }
  c->staged_ready = false;
  Volume& v = *c->staged;
  std::lock_guard<std::mutex> lock(v.mu);
  static std::atomic<unsigned long long> next_staged_generation{1ull << 48};
  v.generation = next_staged_generation.fetch_add(1);
  v.accel_iso = -1;
  v.d_vox = nullptr;
  v.rx = v.ry = v.rz = 0;
  // frames that still read the retired volume (its bytes, if it owns them, and its tables) were enqueued on the
  // context's stream before ev_back_free
  if (c->back_free_pending) HIP_TRY(hipStreamWaitEvent(c->prep_stream, c->ev_back_free, 0));
  c->back_free_pending = false;
  if (on_host) {  // the bytes go into the slot's own buffer; the call returns once they have been taken
    const size_t bytes = (size_t)rx * ry * rz;
    HIP_TRY(v.vox_buf.reserve(bytes));
    HIP_TRY(hipMemcpyAsync(v.vox_buf.p, voxels, bytes, hipMemcpyHostToDevice, c->prep_stream));
    HIP_TRY(hipStreamSynchronize(c->prep_stream));
    v.d_vox = static_cast<const uint8_t*>(v.vox_buf.p);
  } else {
    v.d_vox = static_cast<const uint8_t*>(voxels);
  }
  v.rx = rx; v.ry = ry; v.rz = rz;
  if (tables_possible(c, v)) {
    rc = enqueue_tables(c, v, iso_val, c->prep_stream, c->ev_s0, c->ev_s1);
    if (rc) { v.d_vox = nullptr; return rc; }
    v.accel_iso = iso_val;  // (visible to the context's stream behind ev_staged only: rm_commit_staged_volume)
    c->staged_timing = true;
  }
  HIP_TRY(hipEventRecord(c->ev_staged, c->prep_stream));
  c->staged_ready = true;
  return RM_OK;
}
int rm_stage_volume_device(rm_ctx* c, const void* d_voxels, int rx, int ry, int rz, int iso_val) {
  return stage_volume(c, d_voxels, false, rx, ry, rz, iso_val);
}
int rm_stage_volume(rm_ctx* c, const uint8_t* voxels, int rx, int ry, int rz, int iso_val) {
  return stage_volume(c, voxels, true, rx, ry, rz, iso_val);
}

int rm_commit_staged_volume(rm_ctx* c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->staged || !c->staged_ready) return fail(RM_ESTATE, "rm_commit_staged_volume: no volume has been staged");
  // everything enqueued so far may read the volume that retires now ...
  HIP_TRY(hipEventRecord(c->ev_back_free, c->stream));
  c->back_free_pending = true;
  // ... and everything enqueued from here on sees the staged tables complete
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_staged, 0));
  const bool same_shape = c->vol && c->vol->rx == c->staged->rx && c->vol->ry == c->staged->ry && c->vol->rz == c->staged->rz;
  const bool validated = same_shape && c->dev_src && c->dev_generation == c->vol->generation;
  std::swap(c->vol, c->staged);
  c->staged_ready = false;
  if (validated) {
    // what rm_check_device_opts checked (resolution, voxelRes against the volume, numLights) does not depend on the
    // volume's bytes: the records stay accepted for a volume of the same shape
    c->dev_generation = c->vol->generation;
  } else {
    c->dev_src = nullptr;
    c->dev_recs.clear();
  }
  return RM_OK;
}

int rm_invalidate_volume(rm_ctx* c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!have_volume(c)) return fail(RM_ESTATE, "rm_set_volume has not been called");
  {
    std::lock_guard<std::mutex> lock(c->vol->mu);
    bump_generation(*c->vol);
  }
  c->dev_src = nullptr;
  return broadcast_volume(c);
}

int rm_share_volume(rm_ctx* dst, rm_ctx* src) {
  int rc = check_ctx(dst);
  if (rc) return rc;
  if (!src || !have_volume(src)) return fail(RM_ESTATE, "rm_share_volume: the source context has no volume");
  if (src->device != dst->device)
    return fail(RM_EINVAL, "rm_share_volume: contexts are on devices %d and %d", src->device, dst->device);
  if (!dst->peers.empty() || dst->parent) return fail(RM_EINVAL, "rm_share_volume: multi-device context");
  // (a volume committed from the staging slot: its tables are ordered behind src's stream only -- complete them for everybody)
  if (src->ev_staged) HIP_TRY(hipEventSynchronize(src->ev_staged));
  dst->vol = src->vol;
  dst->dev_src = nullptr;
  dst->dev_recs.clear();
  return RM_OK;
}

// shared tail of the device-side volume producers: optional copy back, make resident
static int adopt_generated(rm_ctx* c, int rx, int ry, int rz, uint8_t* voxels_out) {
  Volume& v = *c->vol;
  const size_t bytes = (size_t)rx * ry * rz;
  if (voxels_out)
    HIP_TRY(hipMemcpyAsync(voxels_out, v.vox_buf.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  v.d_vox = static_cast<const uint8_t*>(v.vox_buf.p);
  v.rx = rx; v.ry = ry; v.rz = rz;
  return broadcast_volume(c);
}
// shared head: fresh volume state with room for the bytes
static int begin_generated(rm_ctx* c, size_t bytes) {
  int rc = begin_new_volume(c, true);
  if (rc) return rc;
  c->vol->d_vox = nullptr;
  HIP_TRY(c->vol->vox_buf.reserve(bytes));
  return RM_OK;
}

int rm_make_gyroid_volume(rm_ctx* c, int rx, int ry, int rz, uint8_t* voxels_out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  rc = check_res(rx, ry, rz);
  if (rc) return rc;
  rc = begin_generated(c, (size_t)rx * ry * rz);
  if (rc) return rc;
  HIP_TRY(rmk::launch_gyroid(c->stream, static_cast<uint8_t*>(c->vol->vox_buf.p), rx, ry, rz));
  return adopt_generated(c, rx, ry, rz, voxels_out);
}

int rm_make_terrain_volume(rm_ctx* c, int rx, int ry, int rz, uint8_t* voxels_out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  rc = check_res(rx, ry, rz);
  if (rc) return rc;
  if (rx < 4 || rz < 4) return fail(RM_EINVAL, "terrain needs rx, rz >= 4 (walls are 4 voxels thick)");
  rc = begin_generated(c, (size_t)rx * ry * rz);
  if (rc) return rc;
  HIP_TRY(rmk::launch_terrain(c->stream, static_cast<uint8_t*>(c->vol->vox_buf.p), rx, ry, rz));
  return adopt_generated(c, rx, ry, rz, voxels_out);
}

// ks >= 0 voxelize-ks, ks == -1 voxelize, scatter: voxelize-scatter with seeded draws
static int voxelize_any(rm_ctx* c, const double* xyz, long long n_vertices, int res, int ks, bool scatter,
                        unsigned long long seed, uint8_t* voxels_out) {
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
  rc = begin_generated(c, (size_t)res * res * res);
  if (rc) return rc;
  if (n_vertices > 0) {
    HIP_TRY(c->gen_buf.reserve((size_t)n_vertices * 24));
    HIP_TRY(hipMemcpyAsync(c->gen_buf.p, xyz, (size_t)n_vertices * 24, hipMemcpyHostToDevice, c->stream));
  }
  if (scatter)
    HIP_TRY(rmk::launch_scatter(c->stream, static_cast<uint8_t*>(c->vol->vox_buf.p),
                                static_cast<const double*>(c->gen_buf.p), n_vertices, lo, off, s, res, seed));
  else
    HIP_TRY(rmk::launch_splat(c->stream, static_cast<uint8_t*>(c->vol->vox_buf.p),
                              static_cast<const double*>(c->gen_buf.p), n_vertices, lo, off, s, res, ks));
  return adopt_generated(c, res, res, res, voxels_out);
}

int rm_voxelize_vertices(rm_ctx* c, const double* xyz, long long n_vertices, int res, int ks,
                         uint8_t* voxels_out) {
  return voxelize_any(c, xyz, n_vertices, res, ks, false, 0ull, voxels_out);
}

int rm_voxelize_scatter(rm_ctx* c, const double* xyz, long long n_vertices, int res, unsigned long long seed,
                        uint8_t* voxels_out) {
  return voxelize_any(c, xyz, n_vertices, res, -1, true, seed, voxels_out);
}

int rm_make_heatmap_volume(rm_ctx* c, const uint32_t* argb, int res, double amp, uint8_t* voxels_out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  rc = check_res(res, res, res);
  if (rc) return rc;
  if (!argb) return fail(RM_EINVAL, "argb is NULL");
  if (amp != amp) return fail(RM_EINVAL, "amp is NaN");
  rc = begin_generated(c, (size_t)res * res * res);
  if (rc) return rc;
  HIP_TRY(c->gen_buf.reserve((size_t)res * res * 4));
  HIP_TRY(hipMemcpyAsync(c->gen_buf.p, argb, (size_t)res * res * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(rmk::launch_heatmap(c->stream, static_cast<uint8_t*>(c->vol->vox_buf.p),
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
                              static_cast<uint32_t*>(c->argb_buf.p), n, contract_arith(c)));
  HIP_TRY(hipMemcpyAsync(argb, c->argb_buf.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RM_OK;
}

// upload the frame's records and tables to one device (asynchronous on its stream)
static int upload_frame_inputs(rm_ctx* c, const void* opts_array, const float* mc_array, int iter) {
  HIP_TRY(c->opts_buf.reserve((size_t)iter * RM_OPTS_BYTES));
  HIP_TRY(c->mc_buf.reserve((size_t)iter * RM_TABLE_FLOATS * 4));
  HIP_TRY(hipMemcpyAsync(c->opts_buf.p, opts_array, (size_t)iter * RM_OPTS_BYTES,
                         hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->mc_buf.p, mc_array, (size_t)iter * RM_TABLE_FLOATS * 4,
                         hipMemcpyHostToDevice, c->stream));
  return RM_OK;
}

// One frame over the devices of a multi-device context, inputs and outputs in the ROOT's
// device memory: device r renders the image tiles r, r+N, ... (interleaved: cost per tile is
// very uneven) into tile-major accumulators, the root collects them with one peer copy per
// device (xGMI, all links into the root concurrently) and un-permutes + tonemaps.
// Asynchronous: everything is ordered by events, the host only enqueues.
//   replicate: copy the records and tables from the root to the other devices first (they
//   keep them in their own opts_buf / mc_buf; a caller whose inputs did not change skips it)
static int frame_multi_device(rm_ctx* c, const RmOpts* d_opts, const float* d_mc, bool replicate, int iter, int n,
                              const RmOpts* recs, const unsigned char* same, float* d_pixels, uint32_t* d_argb,
                              bool sdf) {
  const int world = 1 + (int)c->peers.size();
  const int resx = recs[0].resolution[0];
  const int tpp = rmk::tiles_per_part(rmk::tiles_total(resx, n), world);
  const size_t part_bytes = (size_t)tpp * 64 * 16;
  // A caller that wants the ARGB image only gets it exchanged as tonemapped words: every device tonemaps its
  // own tiles in the frame kernel (TonemapImage is per pixel, renderer.cl:496-508), 4 bytes per pixel cross
  // the links instead of 16, and the root only un-permutes.  (Quality-mode frames keep the float exchange:
  // their ARGB comes from the root's resolve.)
  const bool words = !d_pixels && d_argb && !sdf;
  const size_t xfer_bytes = words ? part_bytes / 4 : part_bytes;
  HIP_TRY(c->tile_buf.reserve(words ? part_bytes : part_bytes * world));
  if (words) HIP_TRY(c->atile_buf.reserve(xfer_bytes * world));
  const size_t opts_bytes = (size_t)iter * RM_OPTS_BYTES, mc_bytes = (size_t)iter * RM_TABLE_FLOATS * 4;
  if (replicate) HIP_TRY(hipEventRecord(c->ev_in, c->stream));  // the root's inputs are complete here
  for (int r = 0; r < world; r++) {
    rm_ctx* d = r == 0 ? c : c->peers[r - 1];
    HIP_TRY(hipSetDevice(d->device));
    const RmOpts* my_opts = d_opts;
    const float* my_mc = d_mc;
    if (r > 0) {
      if (replicate) {
        HIP_TRY(d->opts_buf.reserve(opts_bytes));
        HIP_TRY(d->mc_buf.reserve(mc_bytes));
        HIP_TRY(hipStreamWaitEvent(d->stream, c->ev_in, 0));
        HIP_TRY(hipMemcpyPeerAsync(d->opts_buf.p, d->device, d_opts, c->device, opts_bytes, d->stream));
        HIP_TRY(hipMemcpyPeerAsync(d->mc_buf.p, d->device, d_mc, c->device, mc_bytes, d->stream));
      }
      my_opts = static_cast<const RmOpts*>(d->opts_buf.p);
      my_mc = static_cast<const float*>(d->mc_buf.p);
      HIP_TRY(d->tile_buf.reserve(part_bytes));
      if (words) HIP_TRY(d->atile_buf.reserve(xfer_bytes));
    }
    FrameOut out;
    out.acc = static_cast<float*>(r == 0 ? c->tile_buf.p : d->tile_buf.p);
    if (words) out.argb = static_cast<uint32_t*>(d->atile_buf.p);  // (the root's partition is part 0 of its gather buffer)
    out.tile_first = r;
    out.tile_stride = world;
    int rc = frame_on_device(d, my_opts, my_mc, resx, iter, n, out, same, recs, sdf);
    if (rc) return rc;
    if (r > 0) {
      // the root's gather buffer is free once the previous frame's resolve has read it
      if (c->resolved_once) HIP_TRY(hipStreamWaitEvent(d->stream, c->ev_resolved, 0));
      if (words)
        HIP_TRY(hipMemcpyPeerAsync(static_cast<char*>(c->atile_buf.p) + xfer_bytes * r, c->device, d->atile_buf.p,
                                   d->device, xfer_bytes, d->stream));
      else
        HIP_TRY(hipMemcpyPeerAsync(static_cast<char*>(c->tile_buf.p) + part_bytes * r, c->device, d->tile_buf.p,
                                   d->device, part_bytes, d->stream));
      HIP_TRY(hipEventRecord(d->ev_done, d->stream));
    }
  }
  HIP_TRY(hipSetDevice(c->device));
  for (rm_ctx* p : c->peers) HIP_TRY(hipStreamWaitEvent(c->stream, p->ev_done, 0));
  if (words)
    HIP_TRY(rmk::launch_resolve_argb(c->stream, static_cast<const uint32_t*>(c->atile_buf.p), world, tpp, resx, d_argb, n));
  else
    HIP_TRY(rmk::launch_resolve(c->stream, static_cast<const float*>(c->tile_buf.p), world, tpp, d_opts, d_pixels,
                                d_argb, n, contract_arith(c, sdf)));
  HIP_TRY(hipEventRecord(c->ev_resolved, c->stream));
  c->resolved_once = true;
  c->last_frame_world = world;
  return RM_OK;
}

// rm_render_frame (host buffers) over the devices of a multi-device context: the records and
// tables cross PCIe once, to the root; the other devices take them from there over xGMI
static int render_frame_multi(rm_ctx* c, const void* opts_array, const float* mc_array, int iter, int n,
                              const RmOpts* recs, const unsigned char* same, float* pixels_out,
                              uint32_t* argb_out, bool sdf) {
  int rc = upload_frame_inputs(c, opts_array, mc_array, iter);
  if (rc) return rc;
  // the peers' record / table buffers are overwritten with THIS call's inputs below: a later
  // rm_frame_device_full with the pointers it used before must replicate again
  c->repl_opts = nullptr;
  c->repl_mc = nullptr;
  if (pixels_out) HIP_TRY(c->pix_buf.reserve((size_t)n * 16));
  if (argb_out) HIP_TRY(c->argb_buf.reserve((size_t)n * 4));
  return frame_multi_device(c, static_cast<const RmOpts*>(c->opts_buf.p), static_cast<const float*>(c->mc_buf.p), true,
                            iter, n, recs, same, pixels_out ? static_cast<float*>(c->pix_buf.p) : nullptr,
                            argb_out ? static_cast<uint32_t*>(c->argb_buf.p) : nullptr, sdf);
}

static int render_frame_host(rm_ctx* c, const void* opts_array, const float* mc_array, int iter, int n,
                             float* pixels_out, uint32_t* argb_out, bool sdf) {
  std::vector<RmOpts> recs(iter);
  memcpy(recs.data(), opts_array, (size_t)iter * RM_OPTS_BYTES);
  const int resx = recs[0].resolution[0];
  std::vector<unsigned char> same;
  records_same_as_prev(opts_array, iter, &same);
  if (!c->peers.empty() && !sdf) {
    int rc = render_frame_multi(c, opts_array, mc_array, iter, n, recs.data(), same.data(), pixels_out,
                                argb_out, sdf);
    if (rc) return rc;
  } else {
    int rc = upload_frame_inputs(c, opts_array, mc_array, iter);
    if (rc) return rc;
    // one device: the frame kernel keeps the image row-major and tonemaps with the last pass
    HIP_TRY(c->pix_buf.reserve((size_t)n * 16));
    if (argb_out) HIP_TRY(c->argb_buf.reserve((size_t)n * 4));
    FrameOut out;
    out.acc = static_cast<float*>(c->pix_buf.p);
    out.argb = argb_out ? static_cast<uint32_t*>(c->argb_buf.p) : nullptr;
    out.row_major = true;
    rc = frame_on_device(c, static_cast<const RmOpts*>(c->opts_buf.p), static_cast<const float*>(c->mc_buf.p),
                         resx, iter, n, out, same.data(), recs.data(), sdf);
    if (rc) return rc;
  }
  if (pixels_out)
    HIP_TRY(hipMemcpyAsync(pixels_out, c->pix_buf.p, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
  if (argb_out)
    HIP_TRY(hipMemcpyAsync(argb_out, c->argb_buf.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RM_OK;
}

int rm_render_frame(rm_ctx* c, const void* opts_array, const float* mc_array, int iter, int n,
                    float* pixels_out, uint32_t* argb_out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!opts_array || !mc_array) return fail(RM_EINVAL, "NULL buffer");
  if (iter <= 0) return fail(RM_EINVAL, "iter = %d", iter);
  if (!have_volume(c)) return fail(RM_ESTATE, "rm_set_volume has not been called");
  {
    std::vector<RmOpts> recs(iter);
    memcpy(recs.data(), opts_array, (size_t)iter * RM_OPTS_BYTES);
    rc = check_frame_opts(c, recs.data(), iter, n, recs[0].resolution[0]);
    if (rc) return rc;
  }
  if (n == 0) return RM_OK;
  return render_frame_host(c, opts_array, mc_array, iter, n, pixels_out, argb_out, false);
}

int rm_set_sdf_volume(rm_ctx* c, const float* sdf, int rx, int ry, int rz) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!sdf) return fail(RM_EINVAL, "sdf is NULL");
  rc = check_res(rx, ry, rz);
  if (rc) return rc;
  if (rx < 2 || ry < 2 || rz < 2) return fail(RM_EINVAL, "a distance field needs at least 2 cells per axis");
  if (rx > 4096 || ry > 4096 || rz > 4096 || (unsigned long long)rx * ry * rz >= (1ull << 32))
    return fail(RM_EINVAL, "distance field %dx%dx%d: at most 4096 cells per axis and fewer than 2^32 cells", rx, ry, rz);
  const size_t bytes = (size_t)rx * ry * rz * 4;
  // what the kernel samples is one float4 per cell (4x the field, built once here); the scalar field is only staged
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  const size_t have = free_b + c->sdf_buf.cap + c->sdfq_buf.cap;
  if (bytes * 5 > have)
    return fail(RM_EDEVICE, "distance field %dx%dx%d needs %.1f GiB of device memory (20 B per cell while it is built, 16 after), "
                "%.1f GiB are free", rx, ry, rz, (double)(bytes * 5) / (1 << 30), (double)have / (1 << 30));
  HIP_TRY(c->sdf_buf.reserve(bytes));
  HIP_TRY(hipMemcpyAsync(c->sdf_buf.p, sdf, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c->sdfq_buf.reserve(bytes * 4));
  HIP_TRY(rmk::launch_sdf_quads(c->stream, static_cast<const float*>(c->sdf_buf.p), rx, ry, rz, static_cast<float*>(c->sdfq_buf.p)));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->sdf_buf.release();  // nothing reads the scalar field after the quads are built
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
    if (o.resolution[0] != recs[0].resolution[0])
      return fail(RM_EINVAL, "record %d: resolution.x = %d but the frame is %d wide", i, o.resolution[0],
                  recs[0].resolution[0]);
    if (o.voxelRes[0] != c->sdf_rx || o.voxelRes[1] != c->sdf_ry || o.voxelRes[2] != c->sdf_rz)
      return fail(RM_EINVAL, "TRenderOpts.voxelRes = (%d,%d,%d) does not match the distance field %dx%dx%d",
                  o.voxelRes[0], o.voxelRes[1], o.voxelRes[2], c->sdf_rx, c->sdf_ry, c->sdf_rz);
    if (o.numLights > 4) return fail(RM_EINVAL, "TRenderOpts.numLights = %d (max 4)", (int)o.numLights);
  }
  if (n == 0) return RM_OK;
  return render_frame_host(c, opts_array, mc_array, iter, n, pixels_out, argb_out, true);
}

int rm_tiles_per_part(int resx, int n, int parts) {
  if (resx <= 0 || n < 0 || parts < 1) return fail(RM_EINVAL, "rm_tiles_per_part(%d,%d,%d)", resx, n, parts);
  return rmk::tiles_per_part(rmk::tiles_total(resx, n), parts);
}

int rm_check_device_opts(rm_ctx* c, const void* d_opts, int iter, int n, int width) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!d_opts || iter <= 0) return fail(RM_EINVAL, "d_opts NULL or iter = %d", iter);
  if (n <= 0 || width <= 0) return fail(RM_EINVAL, "n = %d, width = %d", n, width);
  if (!have_volume(c)) return fail(RM_ESTATE, "rm_set_volume has not been called");
  std::vector<RmOpts> recs(iter);
  HIP_TRY(hipMemcpyAsync(recs.data(), d_opts, (size_t)iter * RM_OPTS_BYTES, hipMemcpyDeviceToHost,
                         c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->dev_src = nullptr;
  c->repl_opts = nullptr;  // (multi-device: the next frame replicates records and tables again)
  rc = check_frame_opts(c, recs.data(), iter, n, width);
  if (rc) return rc;
  records_same_as_prev(recs.data(), iter, &c->dev_same);
  c->dev_recs = recs;
  c->dev_iter = iter;
  c->dev_n = n;
  c->dev_width = width;
  c->dev_generation = c->vol->generation;
  // build the derived structures of the (first) hit threshold now, not inside the first frame
  rmk::Accel accel;
  rc = ensure_accel(c, recs[0].isoVal, &accel);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->dev_src = d_opts;
  return RM_OK;
}

// what rm_check_device_opts accepted is still what this call describes
static int check_validated(rm_ctx* c, const void* d_opts, int iter, int n, int width) {
  if (!have_volume(c)) return fail(RM_ESTATE, "rm_set_volume has not been called");
  if (c->dev_src != d_opts || c->dev_iter != iter || c->dev_n != n || c->dev_width != width ||
      c->dev_generation != c->vol->generation)
    return fail(RM_ESTATE,
                "rm_check_device_opts(d_opts, iter=%d, n=%d, width=%d) must validate the records against the "
                "resident volume first (and again after either changes)", iter, n, width);
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
  rc = check_validated(c, d_opts, iter, n, width);
  if (rc) return rc;
  FrameOut out;
  out.acc = d_tiles;
  out.tile_first = tile_first;
  out.tile_stride = tile_stride;
  return frame_on_device(c, static_cast<const RmOpts*>(d_opts), d_mc, width, iter, n, out, c->dev_same.data(),
                         c->dev_recs.data(), false);
}

int rm_frame_device_argb(rm_ctx* c, const void* d_opts, const float* d_mc, int iter, int n, int width,
                         int tile_first, int tile_stride, float* d_tiles, uint32_t* d_argb_tiles) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!d_opts || !d_mc || !d_tiles || !d_argb_tiles) return fail(RM_EINVAL, "NULL device buffer");
  if (iter <= 0 || n <= 0 || width <= 0) return fail(RM_EINVAL, "iter = %d, n = %d, width = %d", iter, n, width);
  if (tile_stride < 1 || tile_first < 0 || tile_first >= tile_stride)
    return fail(RM_EINVAL, "tile partition (%d,%d)", tile_first, tile_stride);
  rc = check_validated(c, d_opts, iter, n, width);
  if (rc) return rc;
  FrameOut out;
  out.acc = d_tiles;
  out.argb = d_argb_tiles;
  out.tile_first = tile_first;
  out.tile_stride = tile_stride;
  return frame_on_device(c, static_cast<const RmOpts*>(d_opts), d_mc, width, iter, n, out, c->dev_same.data(),
                         c->dev_recs.data(), false);
}

int rm_resolve_device_argb(rm_ctx* c, const uint32_t* d_argb_tiles_all, int parts, int n, int width, uint32_t* d_argb) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!d_argb_tiles_all || !d_argb) return fail(RM_EINVAL, "NULL device buffer");
  if (parts < 1 || n <= 0 || width <= 0) return fail(RM_EINVAL, "parts = %d, n = %d, width = %d", parts, n, width);
  const int tpp = rmk::tiles_per_part(rmk::tiles_total(width, n), parts);
  HIP_TRY(rmk::launch_resolve_argb(c->stream, d_argb_tiles_all, parts, tpp, width, d_argb, n));
  return RM_OK;
}

int rm_frame_device_full(rm_ctx* c, const void* d_opts, const float* d_mc, int iter, int n, int width,
                         float* d_pixels, uint32_t* d_argb) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!d_opts || !d_mc || (!d_pixels && !d_argb)) return fail(RM_EINVAL, "NULL device buffer");
  if (iter <= 0 || n <= 0 || width <= 0) return fail(RM_EINVAL, "iter = %d, n = %d, width = %d", iter, n, width);
  rc = check_validated(c, d_opts, iter, n, width);
  if (rc) return rc;
  if (!c->peers.empty()) {
    // the frame tiled over all devices of the context; the records and tables (in the root's
    // memory) go to the other devices once per validation (rm_check_device_opts) / per new d_mc
    const bool repl = c->repl_opts != d_opts || c->repl_mc != d_mc || c->repl_iter != iter;
    rc = frame_multi_device(c, static_cast<const RmOpts*>(d_opts), d_mc, repl, iter, n, c->dev_recs.data(),
                            c->dev_same.data(), d_pixels, d_argb, false);
    if (rc == RM_OK) { c->repl_opts = d_opts; c->repl_mc = d_mc; c->repl_iter = iter; }
    return rc;
  }
  FrameOut out;
  if (!d_pixels) HIP_TRY(c->pix_buf.reserve((size_t)n * 16));
  out.acc = d_pixels ? d_pixels : static_cast<float*>(c->pix_buf.p);
  out.argb = d_argb;
  out.row_major = true;
  return frame_on_device(c, static_cast<const RmOpts*>(d_opts), d_mc, width, iter, n, out, c->dev_same.data(),
                         c->dev_recs.data(), false);
}

int rm_resolve_device(rm_ctx* c, const float* d_tiles_all, int parts, const void* d_opts, int n,
                      int width, float* d_pixels, uint32_t* d_argb) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!d_tiles_all || !d_opts) return fail(RM_EINVAL, "NULL device buffer");
  if (parts < 1 || n <= 0 || width <= 0) return fail(RM_EINVAL, "parts = %d, n = %d, width = %d", parts, n, width);
  const int tpp = rmk::tiles_per_part(rmk::tiles_total(width, n), parts);
  HIP_TRY(rmk::launch_resolve(c->stream, d_tiles_all, parts, tpp, static_cast<const RmOpts*>(d_opts),
                              d_pixels, d_argb, n, contract_arith(c)));
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

int rm_frame_timing_history(rm_ctx* c, float* ms, int* launches, int max_frames, int* count) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (max_frames < 0 || (max_frames > 0 && !ms)) return fail(RM_EINVAL, "max_frames = %d, ms = %p", max_frames, (void*)ms);
  const unsigned long long have = std::min<unsigned long long>(c->frame_seq, (unsigned long long)rm_ctx::kTimingRing);
  const int k = (int)std::min<unsigned long long>(have, (unsigned long long)max_frames);
  for (int i = 0; i < k; i++) {  // oldest of the k first
    const int slot = (int)((c->frame_seq - (unsigned long long)k + (unsigned long long)i) % rm_ctx::kTimingRing);
    HIP_TRY(hipEventSynchronize(c->ev_ring[2 * slot + 1]));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, c->ev_ring[2 * slot], c->ev_ring[2 * slot + 1]));
    ms[i] = t;
    if (launches) launches[i] = c->ring_launches[slot];
  }
  if (count) *count = k;
  return RM_OK;
}

int rm_last_frame_breakdown(rm_ctx* c, float* share_ms, int max_devices, float* frame_ms) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->timed) return fail(RM_ESTATE, "no frame has been rendered");
  // (devices that took part in the LAST frame: a quality-mode frame or an empty one runs on the root alone)
  const int world = c->last_frame_world;
  if (world > 1 && !c->resolved_once) return fail(RM_ESTATE, "no multi-device frame has been rendered");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventSynchronize(world > 1 ? c->ev_resolved : c->ev1));
  for (int r = 0; r < world && r < max_devices; r++) {
    rm_ctx* d = r == 0 ? c : c->peers[r - 1];
    float t = 0.f;
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipEventSynchronize(d->ev1));
    HIP_TRY(hipEventElapsedTime(&t, d->ev0, d->ev1));
    if (share_ms) share_ms[r] = t;
  }
  for (int r = world; share_ms && r < 1 + (int)c->peers.size() && r < max_devices; r++) share_ms[r] = 0.f;  // took no part
  HIP_TRY(hipSetDevice(c->device));
  if (frame_ms) {
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, c->ev0, world > 1 ? c->ev_resolved : c->ev1));
    *frame_ms = t;
  }
  return RM_OK;
}

int rm_last_table_build_ms(rm_ctx* c, float* ms) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!have_volume(c) || c->vol->accel_iso < 0) return fail(RM_ESTATE, "no derived tables have been built");
  if (c->staged_timing) {  // the last build was a staged one: its events are read here, not where it was enqueued
    HIP_TRY(hipEventSynchronize(c->ev_s1));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, c->ev_s0, c->ev_s1));
    (c->staged_ready ? c->staged : c->vol)->accel_build_ms = t;
    c->staged_timing = false;
  }
  if (ms) *ms = (float)c->vol->accel_build_ms;
  return RM_OK;
}

int rm_debug_get_accel(rm_ctx* c, int iso, uint8_t* dist_out, uint32_t* surf_out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!have_volume(c)) return fail(RM_ESTATE, "rm_set_volume has not been called");
  if (iso < 0 || iso > 255) return fail(RM_EINVAL, "iso = %d", iso);
  if (!c->use_accel) return fail(RM_ESTATE, "acceleration structures are disabled (RAYMARCH_NO_ACCEL)");
  rmk::Accel accel;
  rc = ensure_accel(c, iso, &accel);
  if (rc) return rc;
  if (!accel.dist) return fail(RM_ESTATE, "derived tables are not built for this volume size");
  const size_t vox = (size_t)c->vol->rx * c->vol->ry * c->vol->rz;
  if (dist_out && accel.bricked) {  // what the kernels read, converted back to row-major
    HIP_TRY(c->vol->tmp_buf.reserve(vox));
    HIP_TRY(rmk::launch_unbrick(c->stream, accel.dist, c->vol->rx, c->vol->ry, c->vol->rz,
                                static_cast<uint8_t*>(c->vol->tmp_buf.p)));
    HIP_TRY(hipMemcpyAsync(dist_out, c->vol->tmp_buf.p, vox, hipMemcpyDeviceToHost, c->stream));
  } else if (dist_out)
    HIP_TRY(hipMemcpyAsync(dist_out, accel.dist, vox, hipMemcpyDeviceToHost, c->stream));
  if (surf_out) HIP_TRY(hipMemcpyAsync(surf_out, accel.surf, vox * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RM_OK;
}

int rm_debug_volume_band(const void* opts, double* lo, double* hi) {
  if (!opts || !lo || !hi) return fail(RM_EINVAL, "null argument");
  RmOpts o;
  memcpy(&o, opts, sizeof(o));
  volume_band(o, lo, hi);
  return RM_OK;
}

long long rm_debug_block_order(int resx, int n, int passes, int tile_first, int tile_stride, int xcd_rows, int xcd_2d, int rows_desc,
                               double band_lo, double band_hi, long long* out, long long cap) {
  if (resx <= 0 || n <= 0 || passes <= 0 || passes > 64 || tile_stride < 1 || tile_first < 0 || (cap > 0 && !out))
    return fail(RM_EINVAL, "bad argument");
  rmk::FrameLaunch f;
  f.resx = resx; f.n = n; f.passes = passes; f.tile_first = tile_first; f.tile_stride = tile_stride;
  f.pp_log2 = 0;
  while ((1 << f.pp_log2) < passes) f.pp_log2++;  // (one launch = what one wavefront holds)
  f.xcd_rows = xcd_rows != 0; f.xcd_2d = xcd_2d; f.rows_desc = rows_desc != 0;
  f.band_lo = band_lo; f.band_hi = band_hi;
  return rmk::debug_block_order(f, out, cap);
}

int rm_debug_get_octants(rm_ctx* c, int iso, uint8_t* oct_out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!have_volume(c)) return fail(RM_ESTATE, "rm_set_volume has not been called");
  if (iso < 0 || iso > 255 || !oct_out) return fail(RM_EINVAL, "bad argument");
  rmk::Accel accel;
  rc = ensure_accel(c, iso, &accel);
  if (rc) return rc;
  if (!accel.dist || !accel.oct_stride)
    return fail(RM_ESTATE, "directional tables are not built (disabled, or volume too large)");
  const size_t vox = (size_t)c->vol->rx * c->vol->ry * c->vol->rz;
  if (accel.bricked) {
    HIP_TRY(c->vol->tmp_buf.reserve(vox));
    for (int t = 0; t < 8; t++) {
      HIP_TRY(rmk::launch_unbrick(c->stream, accel.dist + (size_t)(t + 1) * accel.oct_stride, c->vol->rx,
                                  c->vol->ry, c->vol->rz, static_cast<uint8_t*>(c->vol->tmp_buf.p)));
      HIP_TRY(hipMemcpyAsync(oct_out + (size_t)t * vox, c->vol->tmp_buf.p, vox, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
  } else {
    HIP_TRY(hipMemcpyAsync(oct_out, accel.dist + vox, vox * 8, hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RM_OK;
}

int rm_selftest_filter(rm_ctx* c, const void* opts544, const float* rays, int n, uint32_t* out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!opts544 || !rays || !out || n < 0) return fail(RM_EINVAL, "bad argument");
  if (n == 0) return RM_OK;
  HIP_TRY(c->opts_buf.reserve(RM_OPTS_BYTES));
  HIP_TRY(c->prim_a.reserve((size_t)n * 32));
  HIP_TRY(c->prim_o.reserve((size_t)n * 4));
  HIP_TRY(hipMemcpyAsync(c->opts_buf.p, opts544, RM_OPTS_BYTES, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->prim_a.p, rays, (size_t)n * 32, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(rmk::launch_filter_check(c->stream, static_cast<const float*>(c->prim_a.p),
                                   static_cast<const RmOpts*>(c->opts_buf.p), static_cast<uint32_t*>(c->prim_o.p), n));
  HIP_TRY(hipMemcpyAsync(out, c->prim_o.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
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
