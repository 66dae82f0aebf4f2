// Error plumbing, ABI version and device-attribute queries of the C ABI.
// Reference for the attribute ops: /root/reference/csrc/cuda_utils_kernels.cu:1-29.
#include <cstring>
#include "common.h"

#include <stdarg.h>
#include <stdio.h>

namespace nmv {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
void append_error(const char* fmt, ...) {
  const size_t n = strlen(g_err);
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err + n, sizeof(g_err) - n, fmt, ap);
  va_end(ap);
}
}  // namespace nmv

extern "C" const char* nmv_last_error(void) { return nmv::g_err; }
// 2: nmv_gptq_marlin_gemm_partial_splits takes num_groups; round-2 signature changes of the attention / fp8-marlin entry points
extern "C" int nmv_abi_version(void) { return 8; }

extern "C" int64_t nmv_get_device_attribute(int64_t attribute, int64_t device_id) {
  int device = (int)device_id, value = 0;
  if (device < 0) (void)hipGetDevice(&device);
  if (hipDeviceGetAttribute(&value, (hipDeviceAttribute_t)attribute, device) != hipSuccess) {
    nmv::set_error("get_device_attribute(%lld, %d) failed", (long long)attribute, device);
    return -1;
  }
  return value;
}

extern "C" int64_t nmv_get_max_shared_memory_per_block_device_attribute(int64_t device_id) {
  // the ROCm branch of the reference asks for hipDeviceAttributeMaxSharedMemoryPerBlock
  return nmv_get_device_attribute((int64_t)hipDeviceAttributeMaxSharedMemoryPerBlock, device_id);
}
