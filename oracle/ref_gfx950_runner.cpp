// oracle/ref_gfx950_runner.cpp -- TEST INFRASTRUCTURE (not product code).
//
// Loads a code object that `make -C oracle ref_gfx950` produced from the UNMODIFIED reference
// kernel source (/root/reference/resources/renderer.cl, compiled where it lies with ROCm's
// OpenCL front end and linked against ROCm's own OpenCL built-in library: opencl.bc / ocml.bc /
// ockl.bc -- no stand-in for any built-in) and runs its two kernels on the GPU exactly as the
// reference host does (core.clj:76-97): RenderImage(voxels, mcSamples, opts, pixels, n) once per
// pass on a zero-filled accumulator (renderer.cl:478-494), then TonemapImage(pixels, opts_0,
// rgba, n) (renderer.cl:496-508).  1-D NDRange, global size rounded up to the work-group size
// (the kernels guard id < n).  The hidden OpenCL kernel arguments (global offset, group sizes)
// are filled by the HIP runtime from the code object's metadata.
//
// Nothing in raymarchcl_amd/ links or loads this file; only tests/ and tools/pin_gfx950.py do.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace {
thread_local std::string g_err;
int fail(const char* what, hipError_t e) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return -1;
}
#define CK(x)                                   \
    do {                                        \
        hipError_t e_ = (x);                    \
        if (e_ != hipSuccess) return fail(#x, e_); \
    } while (0)

struct Ref {
    hipModule_t mod = nullptr;
    hipFunction_t render = nullptr, tonemap = nullptr;
};
}  // namespace

extern "C" {

const char* refg_last_error() { return g_err.c_str(); }

int refg_load(const char* hsaco_path, void** out) {
    Ref* r = new Ref;
    hipError_t e = hipModuleLoad(&r->mod, hsaco_path);
    if (e != hipSuccess) { delete r; return fail("hipModuleLoad", e); }
    e = hipModuleGetFunction(&r->render, r->mod, "RenderImage");
    if (e != hipSuccess) { delete r; return fail("hipModuleGetFunction(RenderImage)", e); }
    e = hipModuleGetFunction(&r->tonemap, r->mod, "TonemapImage");
    if (e != hipSuccess) { delete r; return fail("hipModuleGetFunction(TonemapImage)", e); }
    *out = r;
    return 0;
}

void refg_unload(void* h) {
    Ref* r = static_cast<Ref*>(h);
    if (!r) return;
    if (r->mod) (void)hipModuleUnload(r->mod);
    delete r;
}

// The whole pipeline of core.clj:76-97 with host buffers.  pixels_io: n*4 floats, used as the
// initial accumulator when `keep_pixels` != 0 (else zero-filled, as the reference's p-buf);
// argb_out may be null.  local_size: work-group size (simplecl's choice is not in the tree).
// kernel_ms_out (nullable): summed device time of the RenderImage launches.
int refg_render_frame(void* h, const uint8_t* vox, size_t nvox, const float* mc_array, const void* opts_array,
                      int iters, float* pixels_io, int keep_pixels, uint32_t* argb_out, int n, int local_size,
                      float* kernel_ms_out) {
    Ref* r = static_cast<Ref*>(h);
    const size_t table_bytes = size_t(0x4000) * 16, opts_bytes = 544;
    void *d_vox = nullptr, *d_mc = nullptr, *d_opts = nullptr, *d_px = nullptr, *d_argb = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    auto body = [&]() -> int {
        CK(hipMalloc(&d_vox, nvox));
        CK(hipMalloc(&d_mc, table_bytes * iters));
        CK(hipMalloc(&d_opts, opts_bytes * iters));
        CK(hipMalloc(&d_px, size_t(n) * 16));
        CK(hipMalloc(&d_argb, size_t(n) * 4));
        CK(hipMemcpy(d_vox, vox, nvox, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_mc, mc_array, table_bytes * iters, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_opts, opts_array, opts_bytes * iters, hipMemcpyHostToDevice));
        if (keep_pixels) CK(hipMemcpy(d_px, pixels_io, size_t(n) * 16, hipMemcpyHostToDevice));
        else CK(hipMemset(d_px, 0, size_t(n) * 16));
        CK(hipMemset(d_argb, 0, size_t(n) * 4));
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const unsigned groups = unsigned((n + local_size - 1) / local_size);
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) {
            void* mc_i = static_cast<char*>(d_mc) + table_bytes * i;
            void* opts_i = static_cast<char*>(d_opts) + opts_bytes * i;
            int nn = n;
            void* params[] = {&d_vox, &mc_i, &opts_i, &d_px, &nn};
            CK(hipModuleLaunchKernel(r->render, groups, 1, 1, unsigned(local_size), 1, 1, 0, nullptr, params, nullptr));
        }
        CK(hipEventRecord(e1, nullptr));
        if (argb_out) {
            int nn = n;
            void* params[] = {&d_px, &d_opts, &d_argb, &nn};
            CK(hipModuleLaunchKernel(r->tonemap, groups, 1, 1, unsigned(local_size), 1, 1, 0, nullptr, params, nullptr));
        }
        CK(hipDeviceSynchronize());
        if (kernel_ms_out) CK(hipEventElapsedTime(kernel_ms_out, e0, e1));
        CK(hipMemcpy(pixels_io, d_px, size_t(n) * 16, hipMemcpyDeviceToHost));
        if (argb_out) CK(hipMemcpy(argb_out, d_argb, size_t(n) * 4, hipMemcpyDeviceToHost));
        return 0;
    };
    rc = body();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    for (void* p : {d_vox, d_mc, d_opts, d_px, d_argb})
        if (p) (void)hipFree(p);
    return rc;
}

}  // extern "C"
