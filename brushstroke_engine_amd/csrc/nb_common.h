// Shared helpers for the gfx950 NeuBE kernels (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include "../../include/neube_hip.h"

#define NB_ABI_VERSION 8

void nb_set_error(const char* fmt, ...);

#define NB_REQUIRE(cond, ...)                         \
    do {                                              \
        if (!(cond)) {                                \
            nb_set_error(__VA_ARGS__);                \
            return NB_EINVAL;                         \
        }                                             \
    } while (0)

#define NB_CHECK_LAUNCH(what)                                                      \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            nb_set_error("%s: launch failed: %s", what, hipGetErrorString(e_));    \
            return NB_ELAUNCH;                                                     \
        }                                                                          \
    } while (0)

static inline int nb_cdiv(int a, int b) { return (a + b - 1) / b; }
