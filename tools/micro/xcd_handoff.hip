// xcd_handoff.hip -- what does a dependent "layer" cost when the workgroups that hand data to each other all sit on ONE XCD?
//
// The decode chain is ~19 dependent layers per frame, each an all-to-all between 64 workgroups (every workgroup needs every column of the
// previous layer).  As launches a layer costs ~5.3 us (1.45 us launch boundary + a memory round trip + the work); as one persistent launch
// with hand-offs through device memory (sc1 stores, sc1 polls: round 2's hcgroup_kernel) about the same, because the eight XCDs' L2s are
// not coherent and every hand-off is a fabric round trip.  Inside one XCD the L2 IS the coherence point: plain stores stay in it, loads that
// bypass the CU's L1 (sc1) are served from it, and an atomic without sc1 executes in it.  This micro-benchmark measures that:
//   (1) census: which XCD (s_getreg HW_REG_XCC_ID) does block b land on?
//   (2) a team of T workgroups runs L dependent layers (each: publish 512 B, barrier, read everybody's 512 B, verify):
//       mode 0 = team on one XCD, L2-local protocol (plain stores, workgroup-scope atomic, sc1 polls / loads)
//       mode 1 = team spread over all XCDs, agent-scope protocol (sc1 stores / atomics / loads)
//       mode 2 = team on one XCD, no read-modify-write: every workgroup stores the layer number into ITS word of one line, wave 0 polls the
//                T words with one request (what xgroup_kernel / xcone_kernel use since: T atomics on one word serialise in the L2, and the
//                pollers' reads of that word queue in between)
// Every spin is bounded; a time-out raises an error word and every workgroup leaves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/xcd_handoff.hip -o tools/micro/kp_xcd_handoff
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xfu;
}

__global__ void census_kernel(unsigned* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

__global__ void __launch_bounds__(512) hog_kernel(long long* sink, int ticks) {          // holds its CU for `ticks` x 10 ns
  extern __shared__ float hog_lds[];
  const long long t0 = wall_clock64();
  long long t = t0;
  while (t - t0 < ticks) { t = wall_clock64(); }
  if (threadIdx.x == 0 && blockIdx.x == 0) { hog_lds[0] = 1.f; sink[0] = t; }
}

struct TeamParams {
  unsigned* census;      // [16] arrivals per XCD (mode 0) / [0] arrivals (mode 1)
  unsigned* bar;         // barrier counter (monotonic)
  float* buf;            // [2][T][128]
  long long* stamps;     // [T][2]
  int* err;              // [0] time-out, [1] bad words seen
  int T, L, mode, want_xcc;
};

__device__ __forceinline__ float4 ld_sc1(const float* p) {
  float4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__global__ void __launch_bounds__(512) team_kernel(const TeamParams p) {
  __shared__ int s_slot;
  __shared__ int s_ok;
  const int tid = threadIdx.x;
  const unsigned xcc = xcc_id();
  if (p.mode != 1 && (int)xcc != p.want_xcc) return;
  if (tid == 0) {
    unsigned* cnt = p.census + (p.mode != 1 ? xcc : 0);
    const unsigned s = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_slot = (s < (unsigned)p.T) ? (int)s : -1;
    int ok = 1;
    if (s < (unsigned)p.T) {                 // wait until the team is complete (bounded)
      int spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.T) {
        if (++spins > (1 << 20)) { ok = 0; atomicOr(p.err, 1); break; }
        __builtin_amdgcn_s_sleep(4);
      }
    }
    s_ok = ok;
  }
  __syncthreads();
  const int slot = s_slot;
  if (slot < 0 || !s_ok) return;
  long long t0 = 0;
  int bad = 0;
  for (int l = 0; l < p.L; ++l) {
    if (l == 8 && tid == 0) t0 = wall_clock64();
    float* mine = p.buf + ((size_t)(l & 1) * p.T + slot) * 128;
    const float val = (float)(l * 131 + slot);
    if (tid < 128) {
      if (p.mode != 1) mine[tid] = val + (float)tid;                                                     // plain store: stays in this XCD's L2
      else __hip_atomic_store(mine + tid, val + (float)tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 store: written through
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (p.mode == 2) {
      if (tid < 64) {
        if (tid == 0) __hip_atomic_store(p.bar + slot, (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       // plain store
        int spins = 0, ok = 1;
        for (;;) {
          const unsigned v = tid < p.T ? __hip_atomic_load(p.bar + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned)(l + 1);
          if (__builtin_amdgcn_ballot_w64(v < (unsigned)(l + 1)) == 0ull) break;
          if (++spins > (1 << 18)) { ok = 0; if (tid == 0) atomicOr(p.err, 1); break; }
        }
        if (tid == 0) s_ok = ok;
      }
    } else if (tid == 0) {
      const unsigned target = (unsigned)(l + 1) * (unsigned)p.T;
      if (p.mode == 0) __hip_atomic_fetch_add(p.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // no sc1: executes in the XCD's L2
      else __hip_atomic_fetch_add(p.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0, ok = 1;
      while (__hip_atomic_load(p.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {               // sc1 load: past the L1, served by the L2
        if (++spins > (1 << 18) || (((spins & 1023) == 0) && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { ok = 0; atomicOr(p.err, 1); break; }
      }
      s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) return;
    // read everybody's slice of this layer (T * 128 floats) past the L1 and check every word
    const float* all = p.buf + (size_t)(l & 1) * p.T * 128;
    for (int i = tid; i < p.T * 32; i += 512) {
      const float4 v = ld_sc1(all + (size_t)i * 4);
      const int s = i >> 5, c = (i & 31) * 4;
      const float e = (float)(l * 131 + s) + (float)c;
      bad += (v.x != e) + (v.y != e + 1.f) + (v.z != e + 2.f) + (v.w != e + 3.f);
    }
    __syncthreads();
  }
  if (tid == 0) { p.stamps[slot * 2] = t0; p.stamps[slot * 2 + 1] = wall_clock64(); }
  if (bad) atomicAdd(p.err + 1, bad);
}

int main() {
  // ---- (1) census
  const int NB = 1024;
  unsigned* d_c; CK(hipMalloc(&d_c, NB * 4));
  hipLaunchKernelGGL(census_kernel, dim3(NB), dim3(64), 0, 0, d_c);
  CK(hipDeviceSynchronize());
  std::vector<unsigned> c(NB);
  CK(hipMemcpy(c.data(), d_c, NB * 4, hipMemcpyDeviceToHost));
  int per[16] = {0}, rr = 0;
  for (int b = 0; b < NB; ++b) { per[c[b] & 15]++; rr += ((int)c[b] == b % 8); }
  printf("census of %d blocks x 64 threads: per XCD", NB);
  for (int x = 0; x < 8; ++x) printf(" %d", per[x]);
  printf("; block b on XCD b %% 8 for %d of %d blocks\n", rr, NB);
  // with 512-thread blocks (what the team kernel launches)
  hipLaunchKernelGGL(census_kernel, dim3(512), dim3(512), 0, 0, d_c);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(c.data(), d_c, 512 * 4, hipMemcpyDeviceToHost));
  int per2[16] = {0}; rr = 0;
  for (int b = 0; b < 512; ++b) { per2[c[b] & 15]++; rr += ((int)c[b] == b % 8); }
  printf("census of 512 blocks x 512 threads: per XCD");
  for (int x = 0; x < 8; ++x) printf(" %d", per2[x]);
  printf("; b %% 8 rule holds for %d of 512\n", rr);

  // ---- (1b) census under pressure: does block b of a launch still land on XCD (b + k) % 8 when another stream's workgroups hold most CUs and the
  //      launch's workgroups have to wait for free ones?  (The team kernels' placement rule; a violation is detected there, never trusted.)
  {
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    long long* d_t; CK(hipMalloc(&d_t, 8));
    for (int hogs : {96, 128, 160, 200, 240, 256}) {
      int viol = 0, launches = 0;
      for (int rep = 0; rep < 40; ++rep) {
        hipLaunchKernelGGL(hog_kernel, dim3(hogs), dim3(512), 60 * 1024, sa, d_t, 3000 + 500 * (rep % 5));      // 60 KB of LDS: one per CU next to a 512-thread census block
        hipLaunchKernelGGL(census_kernel, dim3(512), dim3(512), 0, sb, d_c);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(c.data(), d_c, 512 * 4, hipMemcpyDeviceToHost));
        const int k = (int)c[0];
        int v = 0;
        for (int b = 0; b < 512; ++b) v += ((int)c[b] != (b + k) % 8);
        viol += (v != 0); ++launches;
      }
      printf("census of 512 x 512-thread blocks next to %3d long-running workgroups on another stream: %d of %d launches broke the (b + k) %% 8 rule\n", hogs, viol, launches);
    }
    CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb)); CK(hipFree(d_t));
  }

  // ---- (2) team runs
  const int L = 208;
  for (int T : {16, 32, 64}) {
    for (int mode : {0, 2, 1}) {
      unsigned *d_census, *d_bar; float* d_buf; long long* d_st; int* d_err;
      CK(hipMalloc(&d_census, 64)); CK(hipMalloc(&d_bar, 256)); CK(hipMalloc(&d_buf, (size_t)2 * T * 128 * 4)); CK(hipMalloc(&d_st, (size_t)T * 16)); CK(hipMalloc(&d_err, 8));
      double best = 1e30; int errs[2] = {0, 0}; int done = 0;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(d_census, 0, 64)); CK(hipMemset(d_bar, 0, 256)); CK(hipMemset(d_buf, 0, (size_t)2 * T * 128 * 4)); CK(hipMemset(d_st, 0, (size_t)T * 16)); CK(hipMemset(d_err, 0, 8));
        TeamParams p{d_census, d_bar, d_buf, d_st, d_err, T, L, mode, 3};
        const int grid = (mode != 1) ? 8 * T + 64 : T;           // mode 0: every XCD gets >= T candidates, only XCD 3's first T stay
        hipLaunchKernelGGL(team_kernel, dim3(grid), dim3(512), 0, 0, p);
        CK(hipDeviceSynchronize());
        std::vector<long long> st(T * 2);
        CK(hipMemcpy(st.data(), d_st, (size_t)T * 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(errs, d_err, 8, hipMemcpyDeviceToHost));
        long long a = 0, b = 0; done = 0;
        for (int s = 0; s < T; ++s) if (st[s * 2 + 1]) { if (!done || st[s * 2] < a) a = st[s * 2]; if (st[s * 2 + 1] > b) b = st[s * 2 + 1]; ++done; }
        if (done == T && !errs[0]) { const double us = (b - a) / 100.0 / (L - 8); if (us < best) best = us; }
      }
      printf("team of %2d workgroups x 512 threads, %s: %.3f us per dependent layer (publish 512 B, barrier, read %d KB past L1, verify)   [%d finished, time-out %d, bad words %d]\n",
             T, mode == 0 ? "ONE XCD, L2 atomic counter       " : mode == 2 ? "ONE XCD, one flag word per group " : "all XCDs, agent-scope protocol   ", best, T / 2, done, errs[0], errs[1]);
      CK(hipFree(d_census)); CK(hipFree(d_bar)); CK(hipFree(d_buf)); CK(hipFree(d_st)); CK(hipFree(d_err));
    }
  }
  return 0;
}
