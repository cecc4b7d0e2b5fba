// Error reporting shared by every translation unit of libdlesm_hip.so (no HIP dependency,
// so that the host-only index maps can also be built stand-alone, e.g. under sanitizers).
#ifndef DLESM_ERROR_H
#define DLESM_ERROR_H

#include "dlesm_hip.h"

namespace dlesm {

// records the message behind dlesm_last_error() (thread-local) and returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

} // namespace dlesm

#endif
