// Internal helpers shared by the translation units of libdlesm_hip.so.
#ifndef DLESM_INTERNAL_H
#define DLESM_INTERNAL_H

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "dlesm_error.h"
#include "dlesm_hip.h"

namespace dlesm {

#define DLESM_HIP_TRY(expr)                                                                 \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return ::dlesm::fail(DLESM_EHIP, "%s failed: %s (%s:%d)", #expr,                \
                                 hipGetErrorString(_e), __FILE__, __LINE__);                \
    } while (0)

#define DLESM_REQUIRE(cond, ...)                                                            \
    do {                                                                                    \
        if (!(cond)) return ::dlesm::fail(DLESM_EINVAL, __VA_ARGS__);                       \
    } while (0)

// true once a device has been bound (dlesm_init, or lazily on first device call)
int ensure_device();
hipStream_t side_stream();     // created by ensure_device()
hipStream_t transfer_stream(); // for the non-blocking sync callbacks
int tuning(const char *key, int fallback);

// column-major 1-based -> linear offset (field_mod.f90:350)
__host__ __device__ inline size_t lin(int ld, int ji, int jj)
{
    return (size_t)(jj - 1) * (size_t)ld + (size_t)(ji - 1);
}

// shared by the frame/interior split of the distributed step (dlesm_halo.hip)
int launch_stencil5(const double *in, double *out, int ld, int ny, int xstart, int xstop,
                    int ystart, int ystop, hipStream_t s);
int launch_stencil5_frame(const double *in, double *out, int ld, int ny, int xstart, int xstop,
                          int ystart, int ystop, hipStream_t s);

// two fused Jacobi steps (dlesm_jacobi_x2.hip); 1-based inclusive output and intermediate boxes
int launch_stencil5_x2(const double *in, double *out, int ld, int ny, int xstart, int xstop, int ystart,
                       int ystop, int exstart, int exstop, int eystart, int eystop, hipStream_t s);

// block shape (waves per workgroup, padded tiles per row) of a linear tile sweep
void choose_block_shape(int *nxw_io, int *tpb_out);
int check_box(const char *who, int ld, int ny, int xstart, int xstop, int ystart, int ystop, int ring);
// shallow-water step, register-tiled linear sweep (dlesm_shallow.hip); 0-based inclusive box
void launch_shallow_tile(const dlesm_sw_params &q, int ld, int x0, int x1, int y0, int y1,
                         const double *u, const double *v, const double *p, const double *uold,
                         const double *vold, const double *pold, double *unew, double *vnew,
                         double *pnew, hipStream_t s);

} // namespace dlesm

struct dlesm_field {
    uint64_t magic;
    double *data;
    int ld, ny;
    bool owned;
};
static const uint64_t DLESM_FIELD_MAGIC = 0x444c45534d464c44ULL; // "DLESMFLD"

#endif
