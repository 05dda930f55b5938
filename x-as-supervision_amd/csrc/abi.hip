// Error reporting and ABI version for libxas_hip.so.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace xas {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace xas

extern "C" const char* xas_last_error(void) { return xas::g_err; }
extern "C" int xas_abi_version(void) { return 3; }   // 2: xas_conv_shape.mode, pre-split weights (round 3); 3: xas_conv_shape.x_amax,
                                                     // recorded maxima as slots of XAS_AMAX_SLOT_FLOATS floats (round 4)
