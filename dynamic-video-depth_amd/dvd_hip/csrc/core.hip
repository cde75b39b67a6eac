// Library-wide pieces of libdvd_hip.so: error channel, ABI version, device query.
#include "dvd_common.h"

namespace dvd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace dvd

extern "C" {

int dvd_abi_version(void) { return DVD_ABI_VERSION; }

const char* dvd_last_error(void) { return dvd::g_err; }

int dvd_device_cu_count(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return n;
}

}  // extern "C"
