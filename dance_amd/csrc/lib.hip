// Library-level entry points: version, error text, device probe.
#include "common.h"

namespace dh {

char* error_buffer() {
  static thread_local char buf[512] = "ok";
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace dh

extern "C" {

int dh_version(void) { return 100; /* 0.1.0 */ }

const char* dh_last_error_string(void) { return dh::error_buffer(); }

int dh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

}  // extern "C"
