// GPU box probe (round 4): what a dependent kernel node of a hipGraph costs against a grid barrier inside one persistent kernel.
//   hipcc --offload-arch=gfx950 -O2 -o grid_barrier_probe grid_barrier_probe.hip && ./grid_barrier_probe
// (a) a graph of K dependent kernels of G workgroups x 256 threads, each touching a little global memory;
// (b) ONE kernel of G workgroups running the same K phases separated by a generation barrier (agent-scope release / acquire fences: the
//     phases of a conv chain read what other workgroups -- on other XCDs -- wrote in the phase before).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__device__ __forceinline__ void work(float *buf, int phase, int G) {
  // every workgroup reads a line another workgroup wrote in the previous phase and writes its own
  const int w = blockIdx.x, t = threadIdx.x;
  const float v = buf[((phase & 1) ^ 1) * G * 256 + ((w + 1) % G) * 256 + t];
  buf[(phase & 1) * G * 256 + w * 256 + t] = v + 1.0f;
}
__global__ __launch_bounds__(256) void phase_kernel(float *buf, int phase, int G) { work(buf, phase, G); }

__global__ __launch_bounds__(256) void mega_kernel(float *buf, int K, int G, unsigned *ctr, unsigned *gen, int *err) {
  unsigned my_gen = 0;
  if (threadIdx.x == 0) my_gen = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int p = 0; p < K; ++p) {
    work(buf, p, G);
    if (p + 1 == K) break;
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)G - 1) {
        __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        long spins = 0;
        while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_gen) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 2000000) { *err = 1; break; }
        }
      }
      ++my_gen;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}

int main() {
  const int K = 12, reps = 200;
  for (int G : {8, 16, 64}) {
    float *buf; unsigned *ctr; int *err;
    CK(hipMalloc(&buf, 2 * G * 256 * sizeof(float)));
    CK(hipMemset(buf, 0, 2 * G * 256 * sizeof(float)));
    CK(hipMalloc(&ctr, 8)); CK(hipMemset(ctr, 0, 8));
    CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    // (a) graph of K kernels
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < K; ++p) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(256), 0, st, buf, p, G);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
    const double ta = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    // (b) one kernel with K - 1 grid barriers (also from a graph, like the product would)
    hipGraph_t g2; hipGraphExec_t ge2;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(mega_kernel, dim3(G), dim3(256), 0, st, buf, K, G, ctr, ctr + 1, err);
    CK(hipStreamEndCapture(st, &g2));
    CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge2, st));
    CK(hipStreamSynchronize(st));
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) { CK(hipGraphLaunch(ge2, st)); CK(hipStreamSynchronize(st)); }
    const double tb = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    float v0 = 0; CK(hipMemcpy(&v0, buf + ((K - 1) & 1) * G * 256, 4, hipMemcpyDeviceToHost));
    std::printf("G = %3d workgroups, K = %d phases: graph of K kernels %.1f us per replay (%.2f per kernel) | one kernel with %d grid barriers %.1f us (%.2f per phase) | barrier timeouts %d, value %.0f\n",
                G, K, ta, ta / K, K - 1, tb, tb / K, herr, v0);
  }
  return 0;
}
