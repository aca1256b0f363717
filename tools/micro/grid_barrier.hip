// Micro-benchmark: a decode-chain-shaped sequence of 26 dependent "layers" (64 workgroups x 512 threads; each workgroup
// streams a 96 KB weight slice + its 8 activation rows, reduces, writes 8 x 16 outputs that every workgroup of the next layer
// reads) executed as
//   (a) 26 kernel launches per frame (stream order = the dependency), eager and as a hipGraph
//   (b) ONE persistent kernel per frame with a software grid barrier (device-scope atomic counter) between layers
// to decide whether a persistent chain kernel could beat the 8.3 us per dependent layer the decode measures today.
// Spins are bounded: a barrier that does not complete sets an error flag and every workgroup leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NWG = 64, NT = 512, LAYERS = 26;
constexpr int ROWS = 32, CH = 256, K = 768;            // activations (ROWS, CH); a layer reads 3 "taps" = K values per row
constexpr int WSLICE = K * 32;                          // floats of weights per workgroup per layer (96 KB)

__device__ __forceinline__ void layer_work(const float* __restrict__ w, const float* __restrict__ xin, float* __restrict__ xout,
                                           int wg, float* red) {
  const int tid = threadIdx.x;
  const int rg = wg & 3, cg = wg >> 2;                  // 4 row groups x 16 channel groups
  const float4* w4 = reinterpret_cast<const float4*>(w + (size_t)wg * WSLICE);
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < WSLICE / 4 / NT; ++i) {           // 12 float4 per thread
    const float4 v = w4[tid + i * NT];
    acc += v.x + v.y + v.z + v.w;
  }
  // 8 rows x 256 channels of the previous layer's output (all 16 channel groups' columns: the all-to-all dependency)
  const float* xr = xin + (size_t)(rg * 8) * CH;
  float xa = 0.f;
  for (int i = tid; i < 8 * CH; i += NT) xa += xr[i];
  acc = acc * 1e-6f + xa * 1e-3f;
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid < 128) {                                      // 8 rows x 16 channels out
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += red[i];
    xout[(size_t)(rg * 8 + (tid >> 4)) * CH + cg * 16 + (tid & 15)] = s * 1e-3f + (float)(tid & 15) * 1e-4f;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(NT) k_layer(const float* w, const float* xin, float* xout) {
  __shared__ float red[8];
  layer_work(w, xin, xout, blockIdx.x, red);
}

__global__ void __launch_bounds__(NT) k_persistent(const float* w, float* xa, float* xb, unsigned* counter, unsigned base, int* err) {
  __shared__ float red[8];
  const int wg = blockIdx.x;
  for (int l = 0; l < LAYERS; ++l) {
    layer_work(w + (size_t)l * NWG * WSLICE, (l & 1) ? xb : xa, (l & 1) ? xa : xb, wg, red);
    if (l + 1 < LAYERS) {
      // grid barrier: release our stores, count in, spin (bounded) until all NWG arrived, acquire
      __syncthreads();
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = base + (unsigned)(l + 1) * NWG;
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 200000 || ((spins & 1023) == 0 && *(volatile int*)err)) { *err = 1; break; }
        }
      }
      __syncthreads();
      __atomic_thread_fence(__ATOMIC_ACQUIRE);          // workgroup's other waves: see the other workgroups' stores
    }
  }
}

int main() {
  float *w, *xa, *xb; unsigned* counter; int* err;
  const size_t wbytes = (size_t)LAYERS * NWG * WSLICE * 4;       // 163 MB total: like the decode weights it does not fit any L2
  hipMalloc(&w, wbytes); hipMalloc(&xa, ROWS * CH * 4); hipMalloc(&xb, ROWS * CH * 4);
  hipMalloc(&counter, 4); hipMalloc(&err, 4);
  hipMemset(w, 0, wbytes); hipMemset(xa, 0, ROWS * CH * 4); hipMemset(xb, 0, ROWS * CH * 4); hipMemset(counter, 0, 4); hipMemset(err, 0, 4);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int frames = 200;
  float ms;
  // (a1) eager launches
  auto frame_launches = [&]() {
    for (int l = 0; l < LAYERS; ++l)
      hipLaunchKernelGGL(k_layer, dim3(NWG), dim3(NT), 0, s, w + (size_t)l * NWG * WSLICE, (l & 1) ? xb : xa, (l & 1) ? xa : xb);
  };
  for (int i = 0; i < 5; ++i) frame_launches();
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int f = 0; f < frames; ++f) frame_launches();
  hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("eager launches      : %.2f us per layer (%.1f us per 26-layer frame)\n", ms * 1e3f / frames / LAYERS, ms * 1e3f / frames);
  // (a2) one graph per frame
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  frame_launches();
  hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int f = 0; f < frames; ++f) hipGraphLaunch(ge, s);
  hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("hipGraph per frame  : %.2f us per layer (%.1f us per frame)\n", ms * 1e3f / frames / LAYERS, ms * 1e3f / frames);
  // (b) persistent kernel with grid barriers
  unsigned base = 0;
  auto frame_persistent = [&]() {
    hipLaunchKernelGGL(k_persistent, dim3(NWG), dim3(NT), 0, s, w, xa, xb, counter, base, err);
    base += (unsigned)(LAYERS - 1) * NWG;
  };
  for (int i = 0; i < 5; ++i) frame_persistent();
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int f = 0; f < frames; ++f) frame_persistent();
  hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
  printf("persistent + barrier: %.2f us per layer (%.1f us per frame)%s\n", ms * 1e3f / frames / LAYERS, ms * 1e3f / frames,
         herr ? "  [BARRIER TIMEOUT]" : "");
  return herr;
}
