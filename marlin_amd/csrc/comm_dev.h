// Device-side half of the transport (comm.hip): arrival flags raised by the producing kernel itself.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mrl {

// Passed by value to a producing kernel whose workgroups store straight into the peers' receive buffers (PEER_STORE).
// Every workgroup, after its last store, makes its stores visible system-wide and counts itself; the last one of the
// launch raises this rank's arrival flag (value `epoch`) in every peer's flag row of `channel`.
// counter == nullptr: no signalling (the host posts the exchange after the kernel instead).
struct SignalArgs {
  unsigned int *counter;       // device word, zero between launches
  unsigned long long *const *flag_tab;  // device table [nranks]: base of rank p's flag array (IPC mapped)
  unsigned long long epoch;
  unsigned int expected;       // workgroups of the launch
  int nranks, me;
  int row;                     // channel * kFlagRow
};

__device__ __forceinline__ void signal_tail(const SignalArgs &s) {
  if (!s.counter) return;
  // release at system scope: the workgroup's stores to peer memory are written back before the count is taken
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int old = __hip_atomic_fetch_add(s.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old == s.expected - 1u) {
      __hip_atomic_store(s.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      for (int p = 0; p < s.nranks; ++p)
        __hip_atomic_store(s.flag_tab[p] + s.row + s.me, s.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

}  // namespace mrl
