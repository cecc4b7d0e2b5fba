/* libdlesm_lab.so -- measurement tooling that is loaded NEXT TO libdlesm_hip.so, never instead of it and never by it
 * (dl_esm_inf_amd/csrc/lab/).  No reference counterpart: the reference has no device code to hold a kernel against.
 * Shares no state with the product library: raw device pointers and a HIP stream in, 0 / -1 out. */
#ifndef DLESM_LAB_H
#define DLESM_LAB_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* text of the last failure of this thread */
const char *dlesm_lab_last_error(void);

/* The measured ceiling of a sweep with `nread` read and `nwrite` written arrays of n doubles each (n even; 16-byte aligned
 * bases): one 16-byte element per thread per array, workgroups front to back, nothing else.  Supported (nread + nwrite):
 * 1+1, 2+1, 3+1, 4+1, 6+3, 6+6 (dst[3..5] may be src[3..5]: the in-place pattern of the filtered step), 8+1.
 * nt bit 0: the second half of the read arrays loaded non-temporally; bit 1: non-temporal stores.
 * bench.py times it on the kernels' own arrays (`copy_ceiling`, `roofline.frac_of_copy_ceiling`). */
int dlesm_lab_stream_copy_f64(int nread, int nwrite, const double *const *src, double *const *dst, size_t n, int nt,
                              void *stream);

#ifdef __cplusplus
}
#endif
#endif
