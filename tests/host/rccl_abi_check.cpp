// rccl_abi_check.cpp -- compile-time check (no GPU, nothing runs) that the hand-declared RCCL ABI of
// target_estimation_amd/csrc/rccl_abi.hpp is the ABI of the installed <rccl/rccl.h>: the datatype code, the size of the
// unique id, and every entry point's signature up to the representation of the opaque handle / enum types (ncclComm_t is
// a pointer, ncclResult_t / ncclDataType_t are int-sized enums: checked below, which is what makes the casts in
// pose_gather.cpp sound).  Build: hipcc -fsyntax-only (tests/test_abi_exports.py).
#include <rccl/rccl.h>

#include <type_traits>

#include "../../target_estimation_amd/csrc/rccl_abi.hpp"

using namespace te::rccl_abi;

static_assert((int)ncclFloat64 == kNcclDouble && (int)ncclDouble == kNcclDouble, "ncclDouble changed");
static_assert(sizeof(ncclUniqueId) == sizeof(UniqueId) && NCCL_UNIQUE_ID_BYTES == 128, "ncclUniqueId changed size");
static_assert(alignof(ncclUniqueId) == alignof(UniqueId), "ncclUniqueId changed alignment (it is passed by value)");
static_assert(std::is_pointer<ncclComm_t>::value && sizeof(ncclComm_t) == sizeof(void*), "ncclComm_t is no longer a pointer");
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int), "RCCL enums are no longer int-sized");
static_assert((int)ncclSuccess == 0, "ncclSuccess is no longer 0");

// the real entry points must have exactly these parameter lists (with RCCL's own types in place of the erased ones)
static_assert(std::is_same<decltype(&ncclGetUniqueId), ncclResult_t (*)(ncclUniqueId*)>::value, "ncclGetUniqueId");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int)>::value, "ncclCommInitRank");
static_assert(std::is_same<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>::value, "ncclCommDestroy");
static_assert(std::is_same<decltype(&ncclGroupStart), ncclResult_t (*)()>::value, "ncclGroupStart");
static_assert(std::is_same<decltype(&ncclGroupEnd), ncclResult_t (*)()>::value, "ncclGroupEnd");
static_assert(std::is_same<decltype(&ncclSend), ncclResult_t (*)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>::value, "ncclSend");
static_assert(std::is_same<decltype(&ncclRecv), ncclResult_t (*)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>::value, "ncclRecv");
static_assert(std::is_same<decltype(&ncclGetErrorString), const char* (*)(ncclResult_t)>::value, "ncclGetErrorString");
// ... and the erased members line up with them position by position
static_assert(std::is_same<decltype(Rccl::Send), int (*)(const void*, size_t, int, int, void*, hipStream_t)>::value, "Rccl::Send");
static_assert(std::is_same<decltype(Rccl::Recv), int (*)(void*, size_t, int, int, void*, hipStream_t)>::value, "Rccl::Recv");
static_assert(std::is_same<decltype(Rccl::CommInitRank), int (*)(void**, int, UniqueId, int)>::value, "Rccl::CommInitRank");

int main() { return 0; }
