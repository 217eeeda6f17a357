// creg_common.h -- host-side helpers shared by the libcreg translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/creg.h"

namespace creg {

void set_error(const char* fmt, ...);

#define CREG_HIP(call)                                                                  \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess) {                                                        \
            creg::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),     \
                            __FILE__, __LINE__);                                        \
            return CREG_EHIP;                                                           \
        }                                                                               \
    } while (0)

#define CREG_REQUIRE(cond, ...)                                                         \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            creg::set_error(__VA_ARGS__);                                               \
            return CREG_EINVAL;                                                         \
        }                                                                               \
    } while (0)

#define CREG_LAUNCH_CHECK()                                                             \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            creg::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), \
                            __FILE__, __LINE__);                                        \
            return CREG_EHIP;                                                           \
        }                                                                               \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace creg
