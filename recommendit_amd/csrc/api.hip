// Library-wide C-ABI glue: version, arch, thread-local error string.
#include "common.h"
#include "recommendit_hip.h"

#include <atomic>
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void rihip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static std::atomic<uint64_t> g_generation{1};
void rihip_bump_generation(void) { g_generation.fetch_add(1, std::memory_order_relaxed); }
extern "C" uint64_t rihip_scratch_generation(void) { return g_generation.load(std::memory_order_relaxed); }

extern "C" int rihip_abi_version(void) { return RIHIP_ABI_VERSION; }
extern "C" const char* rihip_target_arch(void) { return "gfx950"; }
extern "C" const char* rihip_last_error(void) { return g_err; }

extern "C" int rihip_device_arch(char* buf, int buf_len) {
  RIHIP_REQUIRE(buf && buf_len > 0, RIHIP_ERR_ARG, "device_arch: bad buffer");
  hipDeviceProp_t p;
  RIHIP_CHECK_HIP(hipGetDeviceProperties(&p, 0));
  strncpy(buf, p.gcnArchName, (size_t)buf_len - 1);
  buf[buf_len - 1] = 0;
  return RIHIP_OK;
}
