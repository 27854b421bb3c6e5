// Sanitizer build only: the two symbols encoder.cpp needs from api.hip (error text), so that the host C++ can be compiled
// and run under -fsanitize=address,undefined without the HIP runtime.
#include <stdarg.h>
#include <stdio.h>
namespace nngp {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace nngp
extern "C" const char* nngp_last_error(void) { return nngp::g_err; }
