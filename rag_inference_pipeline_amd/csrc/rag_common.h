// rag_common.h — helpers shared by the translation units of librag_amd.so.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/rag_amd.h"

// Sets the thread-local error string returned by rag_last_error() and returns `code`.
int ragc_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define RAGC_HIP_TRY(expr)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return ragc_fail(e_ == hipErrorOutOfMemory ? RAG_ERR_OOM : RAG_ERR_HIP, "%s: %s", #expr, \
                             hipGetErrorString(e_));                                                \
    } while (0)

struct RagcDeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit RagcDeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~RagcDeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
