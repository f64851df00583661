// Library-level plumbing for libdalm_hip.so: version + thread-local error string.
#include "common.hpp"

namespace dalm {
static thread_local std::string g_last_error = "";

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(int code, const char* fn, const char* what) {
  g_last_error = std::string(fn) + ": " + what;
  return code;
}

int check_launch(const char* fn) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  g_last_error = std::string(fn) + ": " + hipGetErrorString(e);
  return static_cast<int>(e);
}
}  // namespace dalm

extern "C" int dalm_version(void) { return 200; /* 0.2.0 */ }
extern "C" const char* dalm_last_error_string(void) { return dalm::g_last_error.c_str(); }
