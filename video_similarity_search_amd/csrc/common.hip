// Library-wide entry points: version, error string, device check.
#include "common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void slic_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int slic_version(void) { return (0 << 16) | (1 << 8) | 0; }

extern "C" const char* slic_last_error(void) { return g_err; }

extern "C" int slic_device_check(void) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    slic_set_error("no HIP device visible");
    return SLIC_ENODEV;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    slic_set_error("device is %s, this library is built for gfx950 only", prop.gcnArchName);
    return SLIC_ENODEV;
  }
  return SLIC_OK;
}
