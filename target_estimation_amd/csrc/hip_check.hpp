#pragma once
#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>

#define TE_HIP_CHECK(expr)                                                                       \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess)                                                                        \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " +    \
                               __FILE__ + ":" + std::to_string(__LINE__) + " (" #expr ")");     \
  } while (0)
