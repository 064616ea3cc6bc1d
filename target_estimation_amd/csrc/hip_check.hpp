#pragma once
#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>

// A failed call also leaves its code in the runtime's per-thread "last error"; it is cleared here, so that the error is
// reported once, where it happened, and not again by the next unrelated hipGetLastError() check.
#define TE_HIP_CHECK(expr)                                                                       \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      (void)hipGetLastError();                                                                   \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " +    \
                               __FILE__ + ":" + std::to_string(__LINE__) + " (" #expr ")");     \
    }                                                                                            \
  } while (0)
