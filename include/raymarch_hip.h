/*
 * raymarch_hip.h -- C ABI of libraymarch_hip.so: the MI355X (gfx950) drop-in
 * for the host<->device boundary of thi-ng/raymarchcl's render path.
 *
 * In the reference the Clojure host hands five buffers and a declarative step
 * list to thi.ng.simplecl (core.clj:76-97, 119-148), which issues
 * clEnqueueWriteBuffer / clEnqueueNDRangeKernel(RenderImage | TonemapImage) /
 * clEnqueueReadBuffer.  Each entry point below names the reference interface
 * it replaces.  Conventions:
 *   - plain pointers and sizes only; host buffers are borrowed for the call;
 *   - `opts544` is the reference's TRenderOpts record exactly as
 *     thi.ng/structgen encodes it (renderer.cl:35-78; 544 bytes, little-endian,
 *     OpenCL alignment) -- `opts_array` is `iter` of them back to back;
 *   - `mc` is one scatter table: 0x4000 float4 (generators.clj:8-16);
 *     `mc_array` is `iter` tables back to back;
 *   - `pixels` is the float4 accumulator p-buf (core.clj:144), `argb` the
 *     packed 0xAARRGGBB q-buf (core.clj:145);
 *   - every function returns RM_OK (0) or a negative RM_E* code; the message
 *     is available from rm_last_error() (thread local).  Nothing throws.
 *   - one rm_ctx is used by one thread at a time (the reference is a
 *     single-threaded REPL with one in-order queue); contexts are independent.
 *   - there is no CPU fallback: without a gfx950 device rm_create fails.
 */
#ifndef RAYMARCH_HIP_H
#define RAYMARCH_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RM_OK 0
#define RM_EINVAL (-1)   /* bad argument */
#define RM_EDEVICE (-2)  /* HIP runtime / device error */
#define RM_ESTATE (-3)   /* call order (e.g. render before rm_set_volume) */

#define RM_OPTS_BYTES 544
#define RM_TABLE_FLOATS (0x4000 * 4)

typedef struct rm_ctx rm_ctx;

/* Event counters of the render kernels (filled only by the *_counted calls);
 * vox_reads and mc_reads define the algorithmic bytes of a frame. */
typedef struct rm_counters {
  uint64_t vox_reads;    /* in-bounds voxel byte loads the algorithm performs */
  uint64_t mc_reads;     /* scatter table float4 loads */
  uint64_t rays;         /* outer marches (primary + shadow + reflection) */
  uint64_t dts_calls;    /* distance estimates */
  uint64_t march_steps;  /* fixed-step samples */
  uint64_t ao_calls;
  uint64_t primary_hits;
  uint64_t oob_material; /* material index outside the record: undefined in the reference */
} rm_counters;

const char* rm_last_error(void);
/* 4.  (3 -> 4, round 5: a context that never calls rm_set_contract renders RM_CONTRACT_GFX950_DEFAULT instead of
 * RM_CONTRACT_GFX950_STRICT; RM_CONTRACT_GFX950_DEFAULT is new; no entry point changed its signature.) */
int rm_abi_version(void);
/* number of visible HIP devices (0 when there is none / no driver) */
int rm_device_count(void);

/* cl/select-platform + max-device + make-context + init-state with the
 * compiled program (core.clj:121-128).  device_id: HIP ordinal. */
int rm_create(int device_id, rm_ctx** out);
/* The same for a frame spread over several devices of one node (not a reference feature:
 * the reference drives one device, core.clj:122).  The returned context is used like any
 * other; rm_set_volume* and the volume producers replicate the volume on every device,
 * rm_render_frame renders image tiles r, r+N, ... on device_ids[r] and collects the tile
 * accumulators on device_ids[0] with peer copies over xGMI before it un-permutes and tonemaps
 * there.  Every other entry point (single passes, the device-resident calls, the quality
 * mode) runs on device_ids[0] alone.  Ids may repeat (several ranks on one device: a
 * rehearsal of the partition logic on a single-GPU machine). */
int rm_create_multi(const int* device_ids, int n_devices, rm_ctx** out);
int rm_num_devices(rm_ctx* ctx);
void rm_destroy(rm_ctx* ctx);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream) instead of
 * the context's own non-blocking stream.  NULL is a valid handle: the legacy
 * default stream (what torch uses unless told otherwise); RM_OWN_STREAM
 * restores the internal one.  Work the caller enqueues around the calls
 * (copies, RCCL collectives) is ordered with the kernels only if it uses the
 * same stream. */
#define RM_OWN_STREAM ((void*)(intptr_t)-1)
int rm_set_stream(rm_ctx* ctx, void* hip_stream);
int rm_synchronize(rm_ctx* ctx);
/* Page-lock a LONG-LIVED caller buffer (hipHostRegister) that will be handed to the host-buffer
 * entry points frame after frame -- a JNI caller's direct NIO buffers (core.clj:137-145: the
 * reference allocates its buffers once in init-renderer).  The buffer must stay allocated until
 * rm_unpin_host_buffer or rm_destroy. */
int rm_pin_host_buffer(rm_ctx* ctx, const void* p, size_t bytes);
int rm_unpin_host_buffer(rm_ctx* ctx, const void* p);
/* The reference source casts float seed expressions to uint (renderer.cl:267, 334, 471, 472);
 * for negative values (about half of all ambient-occlusion seeds) the cast is undefined and
 * OpenCL devices lower it differently.  RM_SEED_CAST_X86 (default): as an OpenCL CPU device on
 * x86-64 does (64-bit truncate, low 32 bits: wraps) -- BASELINE config 1's device, and what
 * the CPU oracle evaluates.  RM_SEED_CAST_GPU: as GPU devices do (gfx950 v_cvt_u32_f32:
 * saturate, negatives -> 0) -- what the reference kernel compiled for this chip evaluates
 * (oracle/Makefile ref_gfx950, tools/pin_gfx950.py).  Applies to every later render call of the
 * context (all devices of a multi-device context). */
#define RM_SEED_CAST_X86 0
#define RM_SEED_CAST_GPU 1
int rm_set_seed_cast(rm_ctx* ctx, int mode);
/* The arithmetic contract: WHICH build of the reference, on which OpenCL device, the kernels reproduce bit for bit.
 * The reference source leaves the value of its 21 math built-ins (mad, mix, dot, normalize, length, min, max,
 * clamp, exp, exp2, pow, ...), of its (int)/(uint) casts, of `/` and of every a*b+c it spells inside one
 * expression (which its compiler may contract into an fma) to the device and options it is built with
 * (core.clj:122,128).
 *   RM_CONTRACT_GFX950_DEFAULT (the default since ABI 4): this GPU, the reference as ROCm's OpenCL compiler builds
 *     it with NO options -- the built-ins ARE ROCm's OpenCL built-in library (opencl.bc / ocml, linked into the
 *     kernels by the symbols the reference kernel links against), casts as gfx950 lowers them, clang's default
 *     -ffp-contract=on fusing the 15 a*b+c expressions of renderer.cl (:244, :253, :260, :267, :272, :334, :368,
 *     :372, :373, :434, :459, :463), `/` at OpenCL's 2.5 ulp as the gfx950 back end expands it (frexp / v_rcp_f32 /
 *     ldexp).  Checked bit for bit, on the GPU, against oracle/_ref/renderer_gfx950_default.hsaco (the unmodified
 *     renderer.cl) and its recorded outputs tests/golden/gfx950_default/: every fixture through every kernel, all
 *     five BASELINE configurations as whole frames (tests/test_gpu_device_contract.py).  The reference's OWN build
 *     options are -cl-fast-relaxed-math -cl-mad-enable (core.clj:128): fast-math lets the compiler re-associate, so
 *     no hand-written kernel can promise its bits, but the `default` build -- and therefore this contract --
 *     agrees with that build within 1e-4 relative on 100.0000 % of the pixels of EVERY BASELINE configuration at
 *     full size (c1-c5; max 3.5e-7, 85-87 % bit-equal) and of all fixture scenes (tests/test_gpu_pin_gfx950.py against
 *     the live code object or its recording tests/golden/gfx950_fast/; profiles/r06_pin_gfx950.txt).
 *   RM_CONTRACT_GFX950_STRICT (= RM_CONTRACT_GFX950, the default of ABI 3): the same built with -ffp-contract=off
 *     -cl-fp32-correctly-rounded-divide-sqrt (oracle/_ref/renderer_gfx950_strict.hsaco, tests/golden/gfx950_strict/).
 *     Bit-exact against that build; against the reference's own build only ~60 % of the headline frame's pixels
 *     are within 1e-4 (50 % at c3, 37 % at c5, 9 % at c4: one flipped hit/miss decision in any blended pass moves a pixel) -- the
 *     contraction-off build is the outlier among the reference's builds, which is why it is no longer the default.
 *   RM_CONTRACT_CPU_DEVICE: an OpenCL CPU device on x86-64 (BASELINE config 1's device) -- built-ins
 *     as the OpenCL 1.2 specification defines them operation by operation, x86-64 cast lowering
 *     (seed casts per rm_set_seed_cast).  Checked bit for bit against the CPU oracle (oracle/).
 * Applies to every later render / tonemap / resolve call of the context (all its devices),
 * including the counting variant rm_render_image_counted (the plain reference algorithm under the
 * context's contract).  The quality mode (rm_render_sdf_frame) always renders with the CPU-device
 * arithmetic; note that rm_tonemap_image / rm_resolve_device applied to ITS accumulators by the
 * caller follow the context's contract, while the ARGB words rm_render_sdf_frame itself returns
 * are tonemapped with the CPU-device arithmetic. */
#define RM_CONTRACT_CPU_DEVICE 0
#define RM_CONTRACT_GFX950_STRICT 1
#define RM_CONTRACT_GFX950 RM_CONTRACT_GFX950_STRICT /* name of ABI 3 */
#define RM_CONTRACT_GFX950_DEFAULT 2
int rm_set_contract(rm_ctx* ctx, int contract);

/* v-buf: vio/load-volume wraps the bytes into a read-only buffer that the
 * pipeline's first step writes to the device (io.clj:29-33, core.clj:81,146).
 * The volume is copied into HBM and stays resident until replaced. */
int rm_set_volume(rm_ctx* ctx, const uint8_t* voxels, int rx, int ry, int rz);
/* Same, but the bytes already live in device memory owned by the caller
 * (borrowed until the next rm_set_volume* / rm_destroy). */
int rm_set_volume_device(rm_ctx* ctx, const void* d_voxels, int rx, int ry, int rz);
/* The tables the kernels derive from the volume (csrc/rm_accel.hip) are cached per
 * (volume, isoVal).  After modifying a borrowed device volume IN PLACE (the reference's
 * heat-map animation rewrites its v-buf every frame) call this before the next frame:
 * the tables are rebuilt and records accepted by rm_check_device_opts must be checked again. */
/* ANIMATED VOLUMES -- the reference's heat-map animation (meshvoxel.clj:85-89 make-heatmap-anim -> core.clj:181-213)
 * renders a NEW volume every frame; the tables derived from a volume (1.1 ms at 256^3, 5.4 ms at 512^3, 31 ms at
 * 1024^3) would be built inside every frame, with a host wait.  rm_stage_volume_device enqueues their build for the
 * NEXT volume (borrowed device bytes, as rm_set_volume_device; iso_val = the isoVal its frames will use) on a stream
 * of the library's own and returns at once; rm_commit_staged_volume makes that volume the resident one for everything
 * the context's stream is given from then on (no host wait: the stream waits for the tables).  Loop: frame k ->
 * stage(volume k+1) -> wait for frame k, use its pixels -> commit -> frame k+1.  What this hides: the build runs while
 * the GPU is otherwise IDLE -- the reference's loop encodes a PNG on the host per frame (core.clj:203-208) --, and the
 * host never waits for it.  What it does not: beside a frame kernel that fills the chip the build's workgroups (4
 * wavefronts, 14 KB LDS) find no room -- the frame's one-wavefront workgroups refill every slot that frees, stream
 * priority does not change that -- and the chain ends after the frame (measured at 256^3, blocking frames: period
 * 5.00 ms staged, 5.15 ms serial, 3.7 ms the frame alone; bench.py `animated_volume`).  The volume a commit retires becomes the next staging slot; it is
 * overwritten only after the frames that read it (the library orders that itself).  A commit keeps what
 * rm_check_device_opts accepted when the new volume has the old one's resolution.  Single-device contexts only;
 * contexts that shared the old volume keep rendering it.  Pixels are those of rm_set_volume_device + the same frame. */
int rm_stage_volume_device(rm_ctx* ctx, const void* d_voxels, int rx, int ry, int rz, int iso_val);
/* The same for volume bytes in HOST memory (what a JNI caller has: a direct ByteBuffer, or the output of
 * rm_make_heatmap_volume): copied into a buffer the library owns; returns once the bytes have been taken -- the copy,
 * not the build, is waited for. */
int rm_stage_volume(rm_ctx* ctx, const uint8_t* voxels, int rx, int ry, int rz, int iso_val);
int rm_commit_staged_volume(rm_ctx* ctx);

int rm_invalidate_volume(rm_ctx* ctx);
/* Let `dst` use the resident volume of `src` (same device) AND the tables derived from it,
 * instead of holding its own: contexts that render the same scene on different streams
 * (frames in flight) then share one set of tables in HBM and in the caches.  The contexts
 * must not render with different isoVal at the same time; either one detaches by setting
 * a volume of its own. */
int rm_share_volume(rm_ctx* dst, rm_ctx* src);

/* gen/make-gyroid-volume (generators.clj:27-42) evaluated on the device: fills the
 * context's resident volume (as rm_set_volume would) and, if voxels_out is not
 * NULL, copies the rx*ry*rz bytes back.  The reference generates this grid on the
 * host in minutes for 512^3; here it takes milliseconds. */
int rm_make_gyroid_volume(rm_ctx* ctx, int rx, int ry, int rz, uint8_t* voxels_out);
/* gen/make-terrain (generators.clj:44-60) on the device, same conventions: two
 * 4-voxel walls of byte 64 below y = int(ry*0.666) and sine-modulated columns of byte
 * 255 on a 32-voxel grid.  Needs rx, rz >= 4 (and, like the reference, rz >= rx for
 * the second wall to be complete). */
int rm_make_terrain_volume(rm_ctx* ctx, int rx, int ry, int rz, uint8_t* voxels_out);
/* Mesh vertex splatting (meshvoxel.clj): the vertices (n_vertices x 3 binary64, xyz
 * interleaved) are scaled into a res^3 grid by mesh-scale (:16-25: bounding box,
 * largest extent -> res, smaller extents centred) and every vertex sets byte 255 in
 *   ks <  0: its own cell, if inside the grid            (voxelize,    :61-71)
 *   ks >= 0: the clipped cube of cells within +-ks of it (voxelize-ks, :47-59)
 * The result becomes the resident volume; voxels_out (res^3 bytes) may be NULL. */
int rm_voxelize_vertices(rm_ctx* ctx, const double* xyz, long long n_vertices, int res, int ks,
                         uint8_t* voxels_out);
/* voxelize-scatter (meshvoxel.clj:25-43): mesh-scale as above; every vertex writes byte 64 into the 3x3x3 cells
 * around a shifted copy of its cell ((x - dx + 0.4 res, y + 0.4 res, max(z - back, 0)) stored at index
 * y*res^2 + z*res + x -- y and z swapped, as the reference writes it) and, with probability 1/4, into up to five such
 * copies smeared along -x.  The reference draws dx / back / the copy count from the unseeded (rand); here draw k of
 * vertex v is the counter-based uniform u(seed, v, k) documented in csrc/rm_volgen.hip, so the volume is a function
 * of (vertices, res, seed) and can be checked.  The result becomes the resident volume; voxels_out may be NULL. */
int rm_voxelize_scatter(rm_ctx* ctx, const double* xyz, long long n_vertices, int res, unsigned long long seed,
                        uint8_t* voxels_out);
/* make-heatmap (meshvoxel.clj:73-87): a res x res ARGB image (row-major, as piksel's
 * get-pixels returns it) -> in slab y a column of ceil(h) voxels of byte 255 over
 * pixel (x, y), h = c > 0 ? (c > 224 ? 2 : max(2, c*amp)) : 0 with c = argb & 255.
 * Columns are cut at res voxels (the reference would run into the next slab). */
int rm_make_heatmap_volume(rm_ctx* ctx, const uint32_t* argb, int res, double amp, uint8_t* voxels_out);

/* One NDRange of the RenderImage kernel (renderer.cl:478-494; pipeline step
 * core.clj:84-89): write opts + table, run work-items 0..n-1, read the
 * accumulator back.  `pixels` is in/out (n float4). */
int rm_render_image(rm_ctx* ctx, const float* mc, const void* opts544, float* pixels, int n);
/* Same for work-items id0 <= id < id1 only (what a tile of the NDRange does). */
int rm_render_image_range(rm_ctx* ctx, const float* mc, const void* opts544, float* pixels, int n,
                          int id0, int id1);
/* As rm_render_image, additionally ADDS the kernel's event counts to *out. */
int rm_render_image_counted(rm_ctx* ctx, const float* mc, const void* opts544, float* pixels,
                            int n, rm_counters* out);

/* One NDRange of the TonemapImage kernel (renderer.cl:496-508; core.clj:91-97). */
int rm_tonemap_image(rm_ctx* ctx, const float* pixels, const void* opts544, uint32_t* argb, int n);

/* ops/execute-pipeline of the pipeline built by make-pipeline (core.clj:76-97,
 * 171): accumulator zeroed, `iter` RenderImage passes in order with
 * (opts_i, mc_i), TonemapImage with opts_0, read back.  pixels_out (n float4)
 * and argb_out (n uint32) may each be NULL.  (Buffers registered with
 * rm_pin_host_buffer move by DMA at PCIe speed; others go through the runtime's
 * pageable staging path.  At BASELINE config 2 a caller that reads back only the ARGB image,
 * as the reference's pipeline does (core.clj:91-97), pays the kernel + ~0.2 ms either way.)
 * aoIter > 7 (more than RM_WAVE_AO_PROBES = 8 AO probes per hit, the result slots of the exchange area through which a
 * wavefront shares its secondary rays; the reference's default is aoIter = 5): on cubic 256^3, 512^3 and 1024^3
 * volumes -- BASELINE's -- the frame kernel takes the probes in chunks of 8, the time follows the probe count
 * (config 2: 9 / 12 / 16 probes = 1.16 / 1.34 / 1.63 x the time of 6; round 5: 2-3 x per pass).  On every other grid
 * such passes still go out one launch per pass through the single-pass kernels (each lane traces its own probes and
 * shadow rays), about 2-3 x the time per pass.  Results are bit-identical either way. */
int rm_render_frame(rm_ctx* ctx, const void* opts_array, const float* mc_array, int iter, int n,
                    float* pixels_out, uint32_t* argb_out);

/* ---- QUALITY MODE: not a reference feature, not reference-equivalent ----
 * What the north star's prose describes and the reference does not implement: the
 * distance estimate comes from a float distance field sampled TRILINEARLY (cell centres,
 * world units, negative inside), the hit normal is its gradient (central differences,
 * step voxelSize), light visibility is a soft penumbra term min(k*clearance/distance),
 * k = 1/lightScatter.  Everything else -- the march (= sphere tracing with the estimate as
 * the step), AO, lighting, reflections, atmosphere, pass blending, tonemap, and the
 * 544-byte option record -- is the reference path.  Checked bit for bit against a CPU
 * restatement of the same algorithm (oracle/rm_restate.c sdf_*), which is its only pin.
 *
 * rm_set_sdf_volume: rx*ry*rz float32, x fastest, copied to HBM (independent of the byte
 * volume); 2..4096 cells per axis, fewer than 2^32 cells.  The call also builds what the kernel
 * samples -- one float4 per cell holding the four field values of the cell's xy-face, 4x the
 * field's size -- once per field.  rm_render_sdf_frame: as rm_render_frame; voxelRes of the
 * records must equal the field's size; isoVal is ignored. */
int rm_set_sdf_volume(rm_ctx* ctx, const float* sdf, int rx, int ry, int rz);
int rm_render_sdf_frame(rm_ctx* ctx, const void* opts544_array, const float* mc_array, int iter,
                        int n, float* pixels_out, uint32_t* argb_out);

/* ---- device-resident form of the same pipeline (inputs already in HBM) ----
 * The image is cut into 8x8-pixel tiles, numbered row-major; partition
 * (tile_first, tile_stride) owns tiles tile_first, tile_first+tile_stride, ...
 * -- (0,1) is the whole image, (rank, world) the multi-GPU split.  A
 * partition's accumulators are kept TILE-MAJOR: rm_tiles_per_part() tiles of
 * 64 float4 each, local tile j at d_tiles[j*64 .. j*64+63] (lane = (y&7)*8 +
 * (x&7)); this is the buffer ranks exchange.
 *
 * rm_frame_device: the partition's accumulators start from zero, then `iter` RenderImage
 * passes in order with (opts_i, mc_i) over the partition's tiles.  All records of a frame
 * must have the same resolution.x.
 * d_opts: iter*544 bytes, d_mc: iter tables, width: image width in pixels.
 * Asynchronous on the context's stream. */
int rm_tiles_per_part(int resx, int n, int parts);
int rm_frame_device(rm_ctx* ctx, const void* d_opts, const float* d_mc, int iter, int n,
                    int width, int tile_first, int tile_stride, float* d_tiles);
/* The same partition with its TonemapImage words next to the accumulators: d_argb_tiles receives, tile-major
 * like d_tiles, TonemapImage(d_opts[0]) of the partition's pixels after the last pass (renderer.cl:496-508 is
 * per pixel, so tonemapping before the exchange changes nothing).  A frame whose caller wants the ARGB image
 * only exchanges these words -- 4 bytes per pixel over the links instead of 16 -- and the root un-permutes
 * them with rm_resolve_device_argb.  d_argb_tiles: rm_tiles_per_part * 64 words. */
int rm_frame_device_argb(rm_ctx* ctx, const void* d_opts, const float* d_mc, int iter, int n,
                         int width, int tile_first, int tile_stride, float* d_tiles, uint32_t* d_argb_tiles);
/* Tile-major ARGB words of `parts` partitions, gathered as [parts][tiles_per_part][64] -> the row-major
 * ARGB image (the reference's q-buf, core.clj:91-97).  Asynchronous on the context's stream. */
int rm_resolve_device_argb(rm_ctx* ctx, const uint32_t* d_argb_tiles_all, int parts, int n, int width,
                           uint32_t* d_argb);
/* The unpartitioned frame in ONE kernel launch per group of 16 passes (one for a 16-pass frame; a run of 20-31 passes is one launch too): all passes, blended in order, the row-major
 * float4 image into d_pixels (nullable) and TonemapImage(d_opts[0]) into d_argb (nullable;
 * at least one of the two).  Same validation contract as rm_frame_device.
 * On a multi-device context (rm_create_multi) the frame is tiled over its devices; the records and
 * tables are copied from the root's memory to the other devices when (d_opts, d_mc, iter) differ
 * from the previous call's, after rm_check_device_opts, and after any rm_render_frame on the
 * context -- NOT when a caller rewrites the contents behind unchanged pointers: call
 * rm_check_device_opts again after changing records or scatter tables in place. */
int rm_frame_device_full(rm_ctx* ctx, const void* d_opts, const float* d_mc, int iter, int n,
                         int width, float* d_pixels, uint32_t* d_argb);
/* Un-permute `parts` partitions' accumulators (d_tiles_all = partition 0's
 * buffer, then partition 1's, ... each rm_tiles_per_part()*64 float4) into the
 * row-major float4 image d_pixels (nullable) and run TonemapImage with
 * d_opts[0] into d_argb (nullable).  Asynchronous on the context's stream. */
int rm_resolve_device(rm_ctx* ctx, const float* d_tiles_all, int parts, const void* d_opts,
                      int n, int width, float* d_pixels, uint32_t* d_argb);
/* The device-resident calls take the image width from the caller and do not
 * read the records back (no host<->device traffic per frame); `width` must be
 * TRenderOpts.resolution.x.  This synchronous helper fetches the `iter` records
 * once and applies the same validation the host-buffer entry points do
 * (resolution, voxelRes against the resident volume, numLights), and notes each
 * record's isoVal; rm_frame_device[_full] refuse d_opts that were not checked, or were
 * checked with another iter / n / width or against another resident volume.  It also
 * builds the structures derived from the volume for the first record's isoVal
 * (otherwise the first frame would).  Call it again after rewriting the records
 * in place, and after rm_set_volume* / rm_invalidate_volume. */
int rm_check_device_opts(rm_ctx* ctx, const void* d_opts, int iter, int n, int width);

/* Elapsed milliseconds of the RenderImage-pass kernels of the last
 * rm_frame_device / rm_render_frame call (first launch -> end of the last
 * one), measured with HIP events on the stream they ran on (synchronises).
 * launches = number of render kernel launches in that interval. */
int rm_last_frame_timing(rm_ctx* ctx, float* ms, int* launches);
/* The same for the last `max_frames` frames of the context (at most 32 are kept), oldest first:
 * ms[i] / launches[i] (launches may be NULL), *count = how many were written.  The events are
 * recorded around the launches whether anyone reads them or not, so a caller that times K
 * blocking frames with its own clock can read the kernels' device time of THOSE frames after
 * its loop (bench.py: ms_per_step >= roofline.kernel_ms by construction) -- no reference
 * counterpart (the reference does not time its kernels; core.clj:171 times the pipeline). */
int rm_frame_timing_history(rm_ctx* ctx, float* ms, int* launches, int max_frames, int* count);
/* Device time (HIP events) of the last build of the tables derived from the resident volume
 * (dist8 + oct8 + surf32; once per (volume, isoVal), inside the first call that needs them). */
/* Device time of the last frame by device of the context: share_ms[r] = the render kernels of device r's
 * tile partition (r < min(rm_num_devices, max_devices)); *frame_ms = from the start of the root's share to
 * the end of its resolve (multi-device: includes the wait for the slowest device's tiles and the peer
 * copies).  A device that took no part in the last frame (quality-mode frames and frames through a single device's
 * entry points run on the root alone) reports 0.  Measurement only; blocks until the frame is done. */
int rm_last_frame_breakdown(rm_ctx* ctx, float* share_ms, int max_devices, float* frame_ms);
int rm_last_table_build_ms(rm_ctx* ctx, float* ms);

/* Test hook: copy out the derived structures the kernels use for the resident
 * volume at hit threshold `iso` (see csrc/rm_accel.hip): dist_out = rx*ry*rz
 * bytes (0 = cell the march hits, else Chebyshev distance to the nearest hit
 * cell or grid edge, capped at 255), surf_out = rx*ry*rz uint32 (packed voxel
 * value + smooth/flat normal terms; meaningful where value > iso).  Either
 * pointer may be NULL. */
int rm_debug_get_accel(rm_ctx* ctx, int iso, uint8_t* dist_out, uint32_t* surf_out);
/* Same for the 8 directional tables (csrc/rm_accel.hip oct8): oct_out = 8 * rx*ry*rz
 * bytes, table o (bit 0/1/2 set = walking towards -x/-y/-z) holds per cell the edge of
 * the largest cube of empty in-grid cells with that cell as its corner, extending in
 * the walking direction (0 = hit cell, capped at 255). */
int rm_debug_get_octants(rm_ctx* ctx, int iso, uint8_t* oct_out);
/* The part of the image height (fractions, 0 = top row; *lo = *hi = 0: none) whose tile rows the frame kernel
 * dispatches FIRST: the rows in which the clip box [voxelBoundsMin, voxelBoundsMax] of the 544-byte record `opts`
 * covers at least half as much of the image's width as in the row where it covers most -- the box's edges through the
 * inverse of cameraRayLookat (renderer.cl:456-465).  A scheduling heuristic (longest jobs first; then the rows below
 * the band, the rows above it -- sky -- as the tail of a blocking frame), OPT-IN with RAYMARCH_ROW_ORDER=band: at
 * BASELINE's camera the box fills the view and there is no band; at farther cameras it measured -1.9 .. +3.5 %.
 * The default order is plain bottom to top.  Pixels never depend on it.  Host-side, no device needed. */
int rm_debug_volume_band(const void* opts, double* lo, double* hi);

/* Test hook: the dispatch order of ONE launch of the frame kernel (rm_kernels.hip logical_block, the function the kernel
 * compiles, evaluated on the host).  An image of n pixels in rows of resx, `passes` (1..64) passes in the launch, the
 * tiles tile_first, tile_first + tile_stride, ... of it (rm_frame_device's partition; 0, 1 = all).  xcd_rows / xcd_2d /
 * rows_desc / band_lo, band_hi: the switches RAYMARCH_XCD_ROWS, RAYMARCH_XCD_2D (-1 = chosen per launch), RAYMARCH_ROW_ORDER,
 * RAYMARCH_ROW_BAND set.  out[b] (b < cap) = tile << 8 | sub-block of the tile that hardware workgroup b renders, or -1
 * for a workgroup that leaves at once.  -> workgroups of the launch (the grid), or a negative error code.  Every
 * (tile, sub-block) of the partition appears exactly once: the order is a permutation (tests/test_host_abi.py).
 * Host-side, no device needed. */
long long rm_debug_block_order(int resx, int n, int passes, int tile_first, int tile_stride, int xcd_rows, int xcd_2d,
                               int rows_desc, double band_lo, double band_hi, long long* out, long long cap);

/* ---- host-side parameter layer (no device needed) ------------------------
 * The reference builds its inputs in Clojure; a non-Python host gets the same
 * helpers here.  NaN in a double field of rm_render_args means "not given"
 * (Clojure's `(or x default)`).  Results are byte-identical to the Python layer. */
typedef struct rm_render_args {
  int width, height;   /* :width :height */
  int vres[3];         /* :vres */
  int iter;            /* :iter  (frameBlend = 1/iter) */
  double t;            /* :t     (pass time; make-render-option-buffer uses i*0.333) */
  double eyepos[3];    /* :eyepos      default [2 0 2] */
  double targetpos[3]; /* :targetpos   default [0 -0.15 0] */
  double fov_deg;      /* :fov         default 90 */
  double dof;          /* :dof         default 0.001 */
  double gamma;        /* :gamma       default 1.5 */
  double ground_y;     /* :groundY     default 1.05 */
  double voxel_size;   /* :voxelSize   default 1/vres[0] */
  const char* mat;     /* :mat  "orange-stripes" | "metal" | "metal2" | "ao" (unknown/NULL -> "ao") */
} rm_render_args;
/* render-options (core.clj:28-74) + structgen encoding -> one 544-byte TRenderOpts */
int rm_render_options(const rm_render_args* args, void* out544);
/* compute-eyepos (core.clj:150-152) */
int rm_compute_eyepos(double theta_deg, double dist, double y, double out_xyz[3]);
/* generate-scatter-offsets (generators.clj:8-16) with a seed instead of nanoTime:
 * out = 0x4000 unit 4-vectors (RM_TABLE_FLOATS floats) */
int rm_make_scatter_table(uint64_t seed, float* out);
/* make-gyroid-volume (generators.clj:27-42) on the host */
int rm_make_gyroid_host(int rx, int ry, int rz, uint8_t* out);
/* save-volume / load-volume (io.clj:9-33) */
int rm_vox_save(const char* path, int rx, int ry, int rz, const uint8_t* voxels);
int rm_vox_info(const char* path, int* rx, int* ry, int* rz);
int rm_vox_load(const char* path, uint8_t* out, size_t capacity);

/* Device-vs-host checks of the float primitives the parity contract rests on.
 * op: 0 a/b, 1 sqrt(a), 2 exp(a), 3 exp2(a), 4 pow(a,b), 5 (int)a [x86],
 *     6 (uint)a [x86], 7 convert_int_sat(a), 8 a*b+c unfused (c = a).
 * a, b: n floats (b may be NULL for unary ops); out: n 32-bit words. */
int rm_selftest_prims(rm_ctx* ctx, int op, const float* a, const float* b, uint32_t* out, int n);
/* The exact-outcome shortcuts of the march next to the exact slab test (renderer.cl:153-161,
 * :214) they stand in for.  rays: n x 8 floats (origin xyz, direction xyz, march distance t,
 * ground term g); out[i] bits: 1 the filter says "this estimate certainly does not walk",
 * 2 "the position is certainly inside the clip box (the slab test returns exactly +0 < g)",
 * 4 the exact test says the estimate walks (0 <= t_in < g), 8 the exact slab test returned
 * exactly +0, 16 the position passes the inside-by-a-margin shortcut of the estimate.
 * A correct library never reports 1 with 4, nor 2 or 16 without 8. */
int rm_selftest_filter(rm_ctx* ctx, const void* opts544, const float* rays, int n, uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif
