// sgmcmc_host.hpp -- host-side error reporting shared by the translation units of libsgmcmc_hip.so
// (defined in sgmcmc_kernels.hip). fail() formats the thread-local message returned by
// sgmcmc_last_error() and returns `code`.
#pragma once
#include <hip/hip_runtime.h>

namespace sgmcmc_host {
extern thread_local char g_err[512];
int fail(int code, const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);
}  // namespace sgmcmc_host
