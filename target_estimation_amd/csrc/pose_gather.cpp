// pose_gather.cpp -- see pose_gather.hpp.
#include "pose_gather.hpp"

#include <dlfcn.h>

#include <chrono>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "hip_check.hpp"
#include "rccl_abi.hpp"
#include "target_manager.hpp"

namespace te {

namespace {
using rccl_abi::UniqueId;
using rccl_abi::Rccl;
constexpr int kNcclDouble = rccl_abi::kNcclDouble;
static_assert(sizeof(UniqueId) == PoseComm::kIdBytes, "the id the C ABI passes around is NCCL_UNIQUE_ID_BYTES long");

const Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // a copy the process already has (torch bundles one) must be the one used: two RCCL runtimes in a process do not mix
    void* h = RTLD_DEFAULT;
    if (!dlsym(RTLD_DEFAULT, "ncclCommInitRank")) {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) throw std::runtime_error(std::string("target_estimation_amd: RCCL is not available: ") + dlerror());
    }
    auto sym = [&](const char* name) {
      void* p = dlsym(h, name);
      if (!p) throw std::runtime_error(std::string("target_estimation_amd: RCCL symbol missing: ") + name);
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return r;
}

void check(int rc, const char* what) {
  if (rc != 0) throw std::runtime_error(std::string("target_estimation_amd: ") + what + ": " + rccl().GetErrorString(rc));
}
}  // namespace

void PoseComm::unique_id(char out[kIdBytes]) {
  UniqueId id;
  check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(out, id.internal, kIdBytes);
}

PoseComm::PoseComm(const char id[kIdBytes], int rank, int world) : rank_(rank), world_(world) {
  if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("target_estimation_amd: bad rank / world size");
  UniqueId uid;
  std::memcpy(uid.internal, id, kIdBytes);
  check(rccl().CommInitRank(&comm_, world, uid, rank), "ncclCommInitRank");
  TE_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  TE_HIP_CHECK(hipEventCreateWithFlags(&ready_, hipEventDisableTiming));
  TE_HIP_CHECK(hipEventCreate(&start_));
  TE_HIP_CHECK(hipEventCreate(&done_));
}

PoseComm::~PoseComm() {
  if (in_flight_) (void)hipEventSynchronize(done_);
  if (comm_) (void)rccl().CommDestroy(comm_);
  if (stream_) (void)hipStreamDestroy(stream_);
  if (ready_) (void)hipEventDestroy(ready_);
  if (start_) (void)hipEventDestroy(start_);
  if (done_) (void)hipEventDestroy(done_);
  device_free(send_);
}

void PoseComm::begin(TargetManager* m, int root, const long* counts, double* recv_dev) {
  if (root < 0 || root >= world_) throw std::invalid_argument("target_estimation_amd: gather: bad root");
  const long mine = counts[rank_];
  if (mine < 0) throw std::invalid_argument("target_estimation_amd: gather: negative row count");
  if (rank_ == root && !recv_dev && mine > 0) throw std::invalid_argument("target_estimation_amd: gather: the root needs a receive buffer");
  if (mine > send_cap_ && rank_ != root) {
    if (in_flight_) TE_HIP_CHECK(hipEventSynchronize(done_));
    device_free(send_);
    send_ = nullptr; send_cap_ = 0;
    TE_HIP_CHECK(hipMalloc((void**)&send_, sizeof(double) * 7 * mine));
    send_cap_ = mine;
  }
  long offset = 0;
  for (int r = 0; r < rank_; ++r) offset += counts[r];
  hipStream_t compute = nullptr;
  // One critical section of the manager: row count checked against counts[rank], stream read, waits and outputs kernels
  // enqueued -- a concurrent init / erase / setStream cannot fall between them (it either precedes the check, which then
  // throws before anything is queued, or follows the launches).
  m->posesForGather(mine, [&](long, hipStream_t st) -> double* {
    compute = st;
    // the previous gather reads the send buffer: the outputs kernels of this one wait for it ON THE DEVICE
    if (in_flight_) TE_HIP_CHECK(hipStreamWaitEvent(compute, done_, 0));
    // the root's own rows go straight into the receive buffer; every other rank fills its send buffer
    return (rank_ == root) ? recv_dev + offset * 7 : send_;
  });
  TE_HIP_CHECK(hipEventRecord(ready_, compute));
  TE_HIP_CHECK(hipStreamWaitEvent(stream_, ready_, 0));
  TE_HIP_CHECK(hipEventRecord(start_, stream_));
  if (world_ > 1) {
    check(rccl().GroupStart(), "ncclGroupStart");
    int rc = 0;
    const char* failed = nullptr;
    if (rank_ == root) {
      long off = 0;
      for (int r = 0; r < world_ && rc == 0; ++r) {
        if (r != root && counts[r] > 0) {
          rc = rccl().Recv(recv_dev + off * 7, (size_t)counts[r] * 7, kNcclDouble, r, comm_, stream_);
          if (rc != 0) failed = "ncclRecv";
        }
        off += counts[r];
      }
    } else if (mine > 0) {
      rc = rccl().Send(send_, (size_t)mine * 7, kNcclDouble, root, comm_, stream_);
      if (rc != 0) failed = "ncclSend";
    }
    const int rc_end = rccl().GroupEnd();   // always: a failed Send / Recv must not leave the group open for the next call
    if (rc != 0) check(rc, failed);
    check(rc_end, "ncclGroupEnd");
  }
  TE_HIP_CHECK(hipEventRecord(done_, stream_));
  in_flight_ = true;
}

void PoseComm::wait() {
  if (!in_flight_) return;
  TE_HIP_CHECK(hipEventSynchronize(done_));
  in_flight_ = false;
  timed_ = true;
}

bool PoseComm::wait_for(double timeout_s) {
  if (!in_flight_) return true;
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s);
  for (;;) {
    const hipError_t e = hipEventQuery(done_);
    if (e == hipSuccess) break;
    if (e != hipErrorNotReady) TE_HIP_CHECK(e);
    (void)hipGetLastError();   // hipErrorNotReady is not an error to keep
    if (std::chrono::steady_clock::now() >= t_end) return false;   // still in flight: the caller decides
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  in_flight_ = false;
  timed_ = true;
  return true;
}

float PoseComm::last_ms() {
  if (!timed_ || in_flight_) return 0.f;   // no finished gather yet: the events were never recorded
  float ms = 0.f;
  TE_HIP_CHECK(hipEventElapsedTime(&ms, start_, done_));
  return ms;
}

}  // namespace te
