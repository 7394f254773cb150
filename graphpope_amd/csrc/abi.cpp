// Error string and version behind the C ABI (include/graphpope_hip.h).
#include <cstring>

#include "common.h"

namespace pope {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

void clear_error() { g_error[0] = '\0'; }

}  // namespace pope

extern "C" const char *pope_last_error(void) { return pope::g_error; }

extern "C" const char *pope_version(void) { return "graphpope_hip 0.1 gfx950"; }
