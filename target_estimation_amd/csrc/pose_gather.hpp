// pose_gather.hpp -- gather of the estimated poses of all ranks over xGMI (RCCL), overlapped with the next ticks.
//
// Targets are independent, so the predict/update path has no collective.  The one exchange a multi-GPU
// deployment needs is what the reference's node does with the filtered poses every tick: publish them
// (src/target_manager_ros.cpp:78-87).  One process per GPU; rank r owns a contiguous shard of the ids.  The gather is
// a DIRECT one: every rank sends its pose rows to the root with one ncclSend, the root posts one ncclRecv per peer
// (xGMI is point-to-point: 7 independent links into the root; a ring would be per-link bound).  It runs on its own
// stream behind an event, so the step kernels of the following ticks are not held up:
//     compute stream : ... tick k | outputs kernel -> pose buffer | tick k+1 | tick k+2 ...
//     gather stream  :                  (event) ncclSend / ncclRecv ------------> (done event)
// RCCL is resolved at run time (the process's own copy if one is loaded -- e.g. torch's -- else librccl.so.1), so the
// library has no link-time dependency on it.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

namespace te {

class TargetManager;

class PoseComm {
 public:
  static constexpr int kIdBytes = 128;                 // NCCL_UNIQUE_ID_BYTES
  static void unique_id(char out[kIdBytes]);           // ncclGetUniqueId (rank 0; the caller broadcasts it)
  PoseComm(const char id[kIdBytes], int rank, int world);   // ncclCommInitRank on the current device
  ~PoseComm();
  PoseComm(const PoseComm&) = delete;
  PoseComm& operator=(const PoseComm&) = delete;
  int rank() const { return rank_; }
  int world() const { return world_; }

  // Gather the pose7 rows (doubles) of every target of `m` (batch order, slot order) to `root`.
  //   counts [world] : rows every rank contributes (counts[rank] must equal the manager's size)
  //   recv_dev       : root only, [sum(counts)][7] doubles; the rows of rank r start at row sum(counts[:r])
  // Returns as soon as the work is enqueued; wait() blocks the host until the gather has finished.  A second begin()
  // first waits (on the device) for the previous one: there is one send buffer.
  void begin(TargetManager* m, int root, const long* counts, double* recv_dev);
  void wait();
  // the same with a deadline: polls the done event (hipEventQuery) for at most timeout_s; false = still in flight
  // (nothing is cancelled: a later wait() / wait_for() / the destructor picks it up)
  bool wait_for(double timeout_s);
  // device time of the last gather, from its start event to its end event (after wait())
  float last_ms();

 private:
  void* comm_ = nullptr;
  int rank_, world_;
  hipStream_t stream_ = nullptr;
  hipEvent_t ready_ = nullptr, start_ = nullptr, done_ = nullptr;
  bool in_flight_ = false, timed_ = false;
  double* send_ = nullptr;
  long send_cap_ = 0;
};

}  // namespace te
