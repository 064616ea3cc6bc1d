// id_resolve.hpp -- target id -> (batch, slot) ON THE DEVICE, for the array-of-ids entry points.
//
// The reference looks every id up in a std::map under the manager mutex (src/target_manager.cpp:190-202, :243-250).
// The host mirror here is a hash table (id_table.hpp), but one host thread resolving 10^6 ids in caller order still
// costs 22 ms -- 20x the PCIe time of the measurements themselves.  So the by-id batch calls ship the ids as they are
// and resolve them on the GPU: an open-addressing table in HBM (linear probing, power-of-two size, load <= 1/2) that
// is rebuilt from the batches' slot -> id arrays whenever a target was created or erased since its last use.
//   resolve : loc[e] = batch << 26 | slot, or -1 for an unknown id; counts the known ids per batch and notices an id
//             that appears twice in one call (the reference's loop would step it twice, in order: such a call takes
//             the host path)
//   select  : idx[e] = slot if entry e belongs to batch b, else -1 (the indexed kernels skip negative slots)
#pragma once
#include <hip/hip_runtime.h>

namespace te {

constexpr unsigned kIdEmpty = 0xFFFFFFFFu;
constexpr int kIdSlotBits = 26;          // 6 bits of batch, 26 bits of slot (6.7e7 targets per batch)
constexpr int kIdMaxBatches = 8;         // per-batch counters of one resolve call

__device__ __forceinline__ unsigned id_hash(unsigned id, int log2cap) { return (id * 2654435761u) >> (32 - log2cap); }

// insert (ids[s], batch << 26 | s) for s < n; ids are unique across the manager
__attribute__((unused)) static __global__ void id_table_insert_kernel(unsigned* keys, unsigned* vals, int log2cap, const unsigned* ids, long n, unsigned batch) {
  const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const unsigned mask = (1u << log2cap) - 1u;
  const unsigned id = ids[s];
  const unsigned val = (batch << kIdSlotBits) | (unsigned)s;
  unsigned h = id_hash(id, log2cap);
  for (unsigned probe = 0; probe <= mask; ++probe) {
    if (atomicCAS(&vals[h], kIdEmpty, val) == kIdEmpty) {   // claimed: the key of a claimed cell is written by its owner only
      keys[h] = id;
      return;
    }
    h = (h + 1) & mask;
  }
}

struct ResolveCounters {
  int found[kIdMaxBatches];   // known ids per batch
  int duplicate;              // some id appeared more than once in the call
};

// epoch: a number that is different for every resolve call (seen[] keeps the epoch of the last call that touched a cell)
__attribute__((unused)) static __global__ void id_resolve_kernel(const unsigned* keys, const unsigned* vals, int* seen, int log2cap, const unsigned* ids, long n,
                                  int epoch, int* loc, ResolveCounters* counters) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int out = -1;
  if (e < n) {
    const unsigned mask = (1u << log2cap) - 1u;
    const unsigned id = ids[e];
    unsigned h = id_hash(id, log2cap);
    for (unsigned probe = 0; probe <= mask; ++probe) {
      const unsigned v = vals[h];
      if (v == kIdEmpty) break;
      if (keys[h] == id) {
        out = (int)v;
        if (atomicExch(&seen[h], epoch) == epoch) atomicOr(&counters->duplicate, 1);
        break;
      }
      h = (h + 1) & mask;
    }
    loc[e] = out;
  }
  // one atomic per wavefront and batch
  const int b_of = out < 0 ? -1 : (int)((unsigned)out >> kIdSlotBits);
#pragma unroll
  for (int b = 0; b < kIdMaxBatches; ++b) {
    const unsigned long long m = __ballot(b_of == b);
    if (m != 0 && (threadIdx.x & 63) == 0) atomicAdd(&counters->found[b], __popcll(m));
  }
}

__attribute__((unused)) static __global__ void id_select_kernel(const int* loc, long n, int batch, int* idx, unsigned char* found /* or null */) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int v = loc[e];
  idx[e] = (v >= 0 && (int)((unsigned)v >> kIdSlotBits) == batch) ? (int)((unsigned)v & ((1u << kIdSlotBits) - 1u)) : -1;
  if (found) found[e] = v >= 0 ? 1 : 0;
}

}  // namespace te
