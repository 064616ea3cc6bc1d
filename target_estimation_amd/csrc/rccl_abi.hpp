// rccl_abi.hpp -- the part of RCCL's C ABI that pose_gather.cpp resolves at run time (dlsym), declared by hand so that the
// library has no build-time dependency on RCCL: the id struct, the one datatype code, and the signatures of the eight entry
// points as function-pointer members.  tests/host/rccl_abi_check.cpp includes the REAL <rccl/rccl.h> next to this header and
// static_asserts that every declaration here is the one there (RCCL 2.x: /opt/rocm/include/rccl/rccl.h), so a header change
// shows up as a failed CPU test instead of a silently wrong call.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace te {
namespace rccl_abi {

struct UniqueId { char internal[128]; };   // ncclUniqueId: NCCL_UNIQUE_ID_BYTES = 128, passed BY VALUE to ncclCommInitRank
constexpr int kNcclDouble = 8;            // ncclFloat64 / ncclDouble

struct Rccl {
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

}  // namespace rccl_abi
}  // namespace te
