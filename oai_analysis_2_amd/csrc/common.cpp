#include "common.h"

#include <cstdlib>

namespace oai {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#ifdef OAI_DIAG
int diag_env(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
#endif

}  // namespace oai

extern "C" {

int oai_version(void) { return 100; }

const char* oai_last_error(void) { return oai::error_buffer(); }

int oai_device_info(char* name, int cap) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return -1;
    if (name && cap > 0) {
        strncpy(name, p.gcnArchName, cap - 1);
        name[cap - 1] = 0;
    }
    return p.multiProcessorCount;
}
}
