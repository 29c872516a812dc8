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
}  // namespace mser

extern "C" {
int mser_version(void) { return MSER_VERSION; }
const char* mser_last_error(void) { return mser::get_error(); }
}
