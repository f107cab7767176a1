// Raw producer -> consumer hand-off latency between workgroups of ONE launch on gfx950, the floor under every
// colour step of the single-launch triangular solves.  Workgroup b waits for the word of workgroup b - d (d = 1:
// a pure chain; d = 8: eight interleaved chains, one per XCD when workgroup b lands on XCD b % 8), then publishes
// its own.  mode 0: agent-scope (sc1) load + store; mode 1: workgroup-scope (sc0) poll, sc1 store;
// mode 2: as 0 plus the post-arrival work of the solve kernel (LDS write, barrier, 16 dependent LDS reads, shuffle).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value hop.hip -o hop && ./hop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr unsigned long long kS = 0x7FF8DEADBEEF0001ull;
template <int MODE>
__global__ __launch_bounds__(256) void chain(unsigned long long *w, int d, long long *t) {
  __shared__ double lds[2048];
  const int b = blockIdx.x;
  unsigned long long v = 1;
  if (b >= d) {
    const unsigned long long *p = w + (size_t)(b - d) * 16;   // one 128-byte line per workgroup
    v = kS;
    int spins = 0;
    while (v == kS && ++spins < (1 << 20)) {   // bounded: a wrong assumption about the dispatch order must not hang the GPU
      if (MODE == 1) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (v == kS) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  double acc = (double)v;
  if (MODE == 2) {
    lds[threadIdx.x] = acc;
    __syncthreads();
    double s = 0.0;
    for (int j = threadIdx.x & 3; j < 64; j += 4) s += lds[(j + threadIdx.x) & 255];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    acc = s * 1e-30 + acc;
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(w + (size_t)b * 16, (unsigned long long)acc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t) t[b] = (long long)__builtin_amdgcn_s_memrealtime();
  }
}
__global__ void fill(unsigned long long *w, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) w[i] = kS;
}
int main() {
  const int N = 1536;   // all resident at once (256 CUs x 8 workgroups of 256 threads = 2048)
  unsigned long long *w;
  long long *t;
  hipMalloc(&w, sizeof(*w) * 16 * N);
  hipMalloc(&t, sizeof(*t) * N);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode)
    for (int d : {1, 8, 64}) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(fill, dim3((16 * N + 255) / 256), dim3(256), 0, 0, w, (size_t)16 * N);
        hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(N), dim3(256), 0, 0, w, d, t);
        else if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(N), dim3(256), 0, 0, w, d, t);
        else hipLaunchKernelGGL(chain<2>, dim3(N), dim3(256), 0, 0, w, d, t);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      std::vector<long long> ht(N);
      std::vector<unsigned long long> hw(16 * N);
      hipMemcpy(ht.data(), t, sizeof(long long) * N, hipMemcpyDeviceToHost);
      hipMemcpy(hw.data(), w, sizeof(*w) * 16 * N, hipMemcpyDeviceToHost);
      const int hops = (N - 1) / d;
      const double span = (ht[N - 1] - ht[d]) * 0.01;   // 100 MHz counter
      printf("mode %d stride %2d: launch %.1f us, chain of %4d hops: %.3f us per hop (stamps: %.3f), last value %llu\n", mode, d,
             best * 1e3, hops, best * 1e3 / hops, span / ((N - 1 - d) / d), hw[(size_t)16 * (N - 1)]);
    }
  return 0;
}
