// Cost of a grid-wide barrier inside one persistent kernel on MI355X (256 CUs, 8 XCDs with private L2s), against the
// ~4.5 us a kernel boundary costs in a hipGraph: is a persistent per-token decode kernel worth building?
//   variant 0: relaxed device-scope atomics only (data would have to move with L2-bypassing loads / stores)
//   variant 1: + agent-scope release fence before arriving and acquire fence after leaving (L2 write-back / invalidate)
//   variant 2: relaxed atomics, two levels: one counter per XCD (workgroups b % 8), the last arrival of each XCD
//              arrives at a top-level counter, the last of those publishes the flag (counters 256 bytes apart)
// Every spin is bounded (the kernel cannot hang the box); a timed-out barrier is reported.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier_probe grid_barrier_probe.hip && ./grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int VARIANT>
__global__ __launch_bounds__(256) void barrier_loop(unsigned* counter, unsigned* flag, int rounds, unsigned* timeouts,
                                                    float* sink, const float* data) {
  const int nwg = gridDim.x;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    // a little work between barriers so that arrivals are not perfectly aligned
    acc += data[(blockIdx.x * 256 + threadIdx.x + r * 7) & 65535];
    __syncthreads();
    if (threadIdx.x == 0) {
      if (VARIANT == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const unsigned target = (unsigned)(r + 1);
      bool last;
      if (VARIANT == 2) {
        const int x = blockIdx.x & 7, per = (nwg + 7 - x) / 8;  // workgroups of this XCD
        const unsigned a1 = __hip_atomic_fetch_add(counter + 64 * (1 + x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = false;
        if (a1 == (unsigned)per * target - 1u) {
          const unsigned a2 = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          last = a2 == 8u * target - 1u;
        }
      } else {
        const unsigned arrived = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = arrived == (unsigned)nwg * target - 1u;
      }
      if (last) {
        __hip_atomic_store(flag, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          if (++spins > 2000000u) {
            atomicAdd(timeouts, 1u);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (VARIANT == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  if (acc == 123.456f) sink[0] = acc;
}

int main() {
  int dev = 0, ncu = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  unsigned *counter, *flag, *timeouts;
  float *sink, *data;
  hipMalloc(&counter, 4096); hipMalloc(&flag, 4); hipMalloc(&timeouts, 4); hipMalloc(&sink, 4); hipMalloc(&data, 65536 * 4);
  hipMemset(data, 0, 65536 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int rounds = 2000;
  for (int variant = 0; variant < 3; ++variant)
    for (int nwg : {ncu, ncu / 2, 64}) {
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(counter, 0, 4096); hipMemset(flag, 0, 4); hipMemset(timeouts, 0, 4);
        hipEventRecord(a, 0);
        if (variant == 0) hipLaunchKernelGGL(barrier_loop<0>, dim3(nwg), dim3(256), 0, 0, counter, flag, rounds, timeouts, sink, data);
        else if (variant == 1) hipLaunchKernelGGL(barrier_loop<1>, dim3(nwg), dim3(256), 0, 0, counter, flag, rounds, timeouts, sink, data);
        else hipLaunchKernelGGL(barrier_loop<2>, dim3(nwg), dim3(256), 0, 0, counter, flag, rounds, timeouts, sink, data);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        unsigned to = 0; hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost);
        if (rep == 2)
          printf("variant %d (%s) workgroups %3d: %.2f us per barrier (timeouts %u)\n", variant,
                 variant == 1 ? "release/acquire fences" : variant == 2 ? "relaxed, per-XCD counters" : "relaxed atomics only", nwg, ms * 1e3 / rounds, to);
      }
    }
  return 0;
}
