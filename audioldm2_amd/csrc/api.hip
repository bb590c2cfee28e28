// api.hip — library-level entry points (version, thread-local error string).
#include "common.h"
#include <string.h>

namespace aldm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace aldm

extern "C" int aldm_version(void) { return 9; }
extern "C" const char* aldm_last_error(void) { return aldm::g_err; }
