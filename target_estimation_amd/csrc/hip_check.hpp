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

namespace te {

// hipFree synchronises the whole device: next to a resident ("live") kernel -- this batch's sibling's, another manager's -- it
// would block the host until that session ends.  While any session of the process is resident, device memory that is no longer
// needed goes on a list instead and is freed when the last session has left (live_session_ended).
void device_free(void* p);
// Resident sessions of the process: `share` = the fraction of the device's resident-wavefront capacity the session's grid takes
// (its wavefronts + relay over what the device holds of that kernel).  live_session_begin returns false -- and counts nothing --
// if the shares of the sessions already resident leave no room for it.
bool live_session_begin(double share);
void live_session_ended(double share);
double live_sessions_share();

}  // namespace te
