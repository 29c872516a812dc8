// Error slot and version of libmser.so.
#include "common.h"
#include "../../include/mser.h"
#include <cstring>

namespace mser {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }
Prof g_prof;
}  // namespace mser

extern "C" {
int mser_version(void) { return MSER_VERSION; }
const char* mser_last_error(void) { return mser::get_error(); }

int mser_prof_enable(int32_t kernel_id, int32_t max_launches) {
  using mser::g_prof;
  for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
  delete[] g_prof.ev;
  g_prof = mser::Prof();
  if (kernel_id <= 0 || max_launches <= 0) return 0;
  g_prof.ev = new hipEvent_t[2 * (size_t)max_launches];
  for (int i = 0; i < 2 * max_launches; ++i) MSER_CHECK_HIP(hipEventCreate(&g_prof.ev[i]));
  g_prof.cap = max_launches;
  g_prof.kernel_id = kernel_id;
  return 0;
}

int mser_prof_collect(float* total_ms, int32_t* launches) {
  using mser::g_prof;
  MSER_REQUIRE(total_ms && launches, "mser_prof_collect: null pointer");
  float tot = 0.f;
  for (int i = 0; i < g_prof.used; ++i) {
    MSER_CHECK_HIP(hipEventSynchronize(g_prof.ev[2 * i + 1]));
    float ms = 0.f;
    MSER_CHECK_HIP(hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
    tot += ms;
  }
  *total_ms = tot;
  *launches = g_prof.used;
  g_prof.used = 0;
  return 0;
}
}
