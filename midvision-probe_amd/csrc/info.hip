// Library introspection + error strings for the C ABI.
#include <string.h>

#include "mvp_common.h"

extern "C" int mvp_get_info(mvp_info_t* out) {
  if (!out) return MVP_EINVAL;
  memset(out, 0, sizeof(*out));
  out->abi_version = MVP_ABI_VERSION;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  out->device_count = n;
  if (n > 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) {
      strncpy(out->arch, prop.gcnArchName, sizeof(out->arch) - 1);
      out->cu_count = prop.multiProcessorCount;
      out->gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    }
  }
  return MVP_OK;
}

extern "C" const char* mvp_strerror(int code) {
  switch (code) {
    case MVP_OK: return "ok";
    case MVP_EINVAL: return "invalid argument (shape, alignment or NULL pointer)";
    case MVP_ELAUNCH: return "HIP kernel launch failed";
    case MVP_ENODEV: return "no gfx950 device";
    default: return "unknown error";
  }
}
