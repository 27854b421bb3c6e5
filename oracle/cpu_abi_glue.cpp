// nngp::set_error for the host sources of the product that are compiled into the CPU build of the ABI as they are
// (nngp-src_amd/csrc/encoder.cpp): forwards to the error buffer of nngp_cpu_abi.c.  Test infrastructure (oracle/).
#include <stdarg.h>
#include <stdio.h>
extern "C" void nngp_cpu_set_error(const char* msg);
namespace nngp {
void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    nngp_cpu_set_error(buf);
}
}  // namespace nngp
