// hconv_lab.hip -- standalone measurement lab for the SSRN / TextEnc throughput kernel (dc_tts_amd/csrc/hconv_kernel.h).
// Not product code.  It times the production kernel next to ablated copies (what does the epilogue / the per-chunk barrier / the weight
// stream / the LDS fragment reads cost?) and next to candidate restructurings, on the layer shapes of networks.py:214-292 at the bench
// batch (B = 32 -> 26 880 rows at 4T), and checks every non-ablated candidate against the production kernel's output.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dc_tts_amd/csrc tools/micro/hconv_lab.hip -o tools/micro/kp_hconv_lab
//   gpurun -- 'tools/micro/kp_hconv_lab'
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <functional>
#include <algorithm>

#include "hconv_kernel.h"
#include "hconv16_kernel.h"
#include "hconv_lab_kernels.h"

using namespace dctts;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

namespace dctts {
hipError_t launch_hconv(const ConvShape&, const ConvParams&, hipStream_t, int) { return hipErrorInvalidConfiguration; }
hipError_t launch_hconv16(const ConvShape&, const ConvParams&, int, hipStream_t) { return hipErrorInvalidConfiguration; }
}

struct Shape { const char* name; int epi, nt, nw, cin, cin_p, ntaps, dil, cout, act; };

static float frand(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((s >> 40) & 0xffffff) / 8388608.0f - 1.0f; }

struct Bufs {
  float *in = nullptr, *out = nullptr, *ref = nullptr, *wp = nullptr, *bias = nullptr, *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr, *extra = nullptr;
  size_t out_floats = 0;
};

// Between two timed launches the pipeline runs ~12 ms of other layers that stream several hundred MB through the L2s and the 256 MB Infinity
// Cache, so a layer's weights come from HBM every time.  Repeating one kernel back to back keeps them cache-resident and flatters every variant
// (and changes their ranking): with `thrash` set, a 1 GB read-modify-write runs before every timed launch and each launch gets its own event pair.
static float* g_thrash = nullptr; static size_t g_thrash_n = 0;
__global__ void thrash_kernel(float4* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; v.x += 1.f; p[i] = v; }
}
template <typename F>
static float time_launch_thrash(F&& launch, int reps) {
  std::vector<hipEvent_t> e0(reps), e1(reps);
  for (int i = 0; i < reps; ++i) { CK(hipEventCreate(&e0[i])); CK(hipEventCreate(&e1[i])); }
  launch();
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL(thrash_kernel, dim3(2048), dim3(256), 0, 0, (float4*)g_thrash, g_thrash_n / 4);
    CK(hipEventRecord(e0[i], 0)); launch(); CK(hipEventRecord(e1[i], 0));
  }
  CK(hipDeviceSynchronize());
  std::vector<float> t(reps);
  for (int i = 0; i < reps; ++i) { CK(hipEventElapsedTime(&t[i], e0[i], e1[i])); CK(hipEventDestroy(e0[i])); CK(hipEventDestroy(e1[i])); }
  std::sort(t.begin(), t.end());
  return t[reps / 2] * 1000.f;            // median
}

template <typename F>
static float time_launch(F&& launch, int reps) {
  if (g_thrash) return time_launch_thrash(launch, reps);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1000.f / reps;
}

static double maxdiff(const Bufs& b, size_t n) {
  std::vector<float> a(n), r(n);
  CK(hipMemcpy(a.data(), b.out, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r.data(), b.ref, n * 4, hipMemcpyDeviceToHost));
  double m = 0;
  for (size_t i = 0; i < n; ++i) { const double d = std::fabs((double)a[i] - (double)r[i]); if (!(d <= m)) m = d; }
  return m;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 6;
  const int B = 32, R = 840, PADR = 64;
  const Shape shapes[] = {
      {"HC_11 (1024->2048, k=3)", EPI_HC, 8, 8, 1024, 1024, 3, 1, 1024, ACT_NONE},
      {"HC_8  (512->1024, k=3)", EPI_HC, 4, 8, 512, 512, 3, 1, 512, ACT_NONE},
      {"C_14  (1025->1025, k=1)", EPI_C, 3, 11, 1056, 1056, 1, 1, 1025, ACT_RELU},
      {"C_10  (512->1024, k=1)", EPI_C, 4, 8, 512, 512, 1, 1, 1024, ACT_NONE},
  };
  int only = argc > 2 ? atoi(argv[2]) : -1;
  const bool fixed = argc > 3 && atoi(argv[3]) != 0;
  const bool stamps = argc > 4 && atoi(argv[4]) != 0;
  if (argc > 5 && atoi(argv[5]) != 0) { g_thrash_n = (size_t)256 << 20; CK(hipMalloc(&g_thrash, g_thrash_n * 4)); CK(hipMemset(g_thrash, 0, g_thrash_n * 4)); }       // n1 / n3 record s_memtime at their phase boundaries        // K = 32 only: what an item costs besides its contraction
  for (int si = 0; si < 4; ++si) {
    if (only >= 0 && si != only) continue;
    const Shape& S = shapes[si];
    const int stride_in = (S.cin + 31) / 32 * 32, stride_out = (S.cout + 31) / 32 * 32;
    const long rows = PADR + R + PADR;
    Bufs b;
    const size_t nin = (size_t)B * rows * stride_in, nout = (size_t)B * rows * stride_out;
    b.out_floats = nout;
    const int tiles = S.nt * S.nw, KG = S.ntaps * (S.cin_p / 32) * 4;
    const size_t nw = (size_t)tiles * KG * 64 * 4;
    uint64_t seed = 1234 + si;
    std::vector<float> h(nin);
    for (auto& v : h) v = frand(seed);
    CK(hipMalloc(&b.in, nin * 4)); CK(hipMemcpy(b.in, h.data(), nin * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&b.out, nout * 4)); CK(hipMalloc(&b.ref, nout * 4));
    CK(hipMemset(b.out, 0, nout * 4)); CK(hipMemset(b.ref, 0, nout * 4));
    std::vector<float> w(nw);
    const float ws = 1.0f / std::sqrt((float)(S.ntaps * S.cin));
    for (auto& v : w) v = frand(seed) * ws;
    CK(hipMalloc(&b.wp, nw * 4)); CK(hipMemcpy(b.wp, w.data(), nw * 4, hipMemcpyHostToDevice));
    std::vector<float> v1(4096);
    auto up = [&](float** d, float base, float amp) { for (auto& v : v1) v = base + amp * frand(seed); CK(hipMalloc(d, 4096 * 4)); CK(hipMemcpy(*d, v1.data(), 4096 * 4, hipMemcpyHostToDevice)); };
    up(&b.bias, 0.f, 0.1f); up(&b.g1, 1.f, 0.1f); up(&b.b1, 0.f, 0.1f); up(&b.g2, 1.f, 0.1f); up(&b.b2, 0.f, 0.1f);

    ConvParams p; memset(&p, 0, sizeof(p));
    p.in = b.in; p.in_bstride = rows; p.in_row0 = PADR; p.in_stride = stride_in; p.cin = S.cin; p.cin_p = S.cin_p; p.ntaps = S.ntaps;
    if (S.ntaps == 3) { p.tap_off[0] = -S.dil; p.tap_off[1] = 0; p.tap_off[2] = S.dil; }
    if (fixed) { p.ntaps = 1; p.cin_p = 32; }
    p.R = R; p.wp = b.wp; p.bias = b.bias; p.g1 = b.g1; p.b1 = b.b1; p.g2 = b.g2; p.b2 = b.b2; p.cout = S.cout;
    p.out = b.out; p.out_bstride = rows; p.out_row0 = PADR; p.out_stride = stride_out; p.out_tmul = 1; p.out_tadd = 0; p.act = S.act;
    p.out_zero_to = stride_out;
    const double flop_row = 2.0 * S.ntaps * S.cin * (S.epi == EPI_HC ? 2 : 1) * S.cout;
    printf("== %s   tiles=%d KG=%d  %.2f MFLOP/row%s\n", S.name, tiles, KG, flop_row / 1e6, fixed ? "   [K = 32 only: fixed cost per item; TF figures meaningless]" : "");

    for (int items : {256, 768}) {
      p.M = items * 32;
      if (stamps && items != 256) continue;
      const dim3 grid(items);
      auto report = [&](const char* what, float us, double err) {
        const double tf = flop_row * p.M / us / 1e6;
        if (err >= 0) printf("  items=%3d %-58s %8.1f us  %6.1f TF  %.3f of peak   max|d|=%.2e\n", items, what, us, tf, tf / 157.3, err);
        else printf("  items=%3d %-58s %8.1f us  %6.1f TF  %.3f of peak\n", items, what, us, tf, tf / 157.3);
        fflush(stdout);
      };
      long long* d_st = nullptr;
      if (stamps) { CK(hipMalloc(&d_st, (size_t)items * 8 * sizeof(long long))); CK(hipMemset(d_st, 0, (size_t)items * 8 * sizeof(long long))); }
      auto show_stamps = [&](const char* what) {
        if (!stamps) return;
        std::vector<long long> h((size_t)items * 8);
        CK(hipMemcpy(h.data(), d_st, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        double seg[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0}; int n = 0;
        long long tmin = 0x7fffffffffffffffLL, tmax = 0, smax = 0;
        for (int i = 0; i < items; ++i) {
          const long long* t = &h[(size_t)i * 8];
          if (!t[5] || !t[0]) continue;
          const long long mid = t[3] ? t[3] : t[4];
          const double v[5] = {(double)(t[1] - t[0]), (double)(t[2] - t[1]), (double)(mid - t[2]), (double)(t[5] - mid), (double)(t[5] - t[0])};
          for (int k = 0; k < 5; ++k) { seg[k] += v[k]; if (v[k] > mx[k]) mx[k] = v[k]; }
          ++n;
          if (t[0] < tmin) tmin = t[0];
          if (t[0] > smax) smax = t[0];
          if (t[5] > tmax) tmax = t[5];
        }
        printf("      stamps %-4s (s_memtime ticks; mean / max over %d items): prologue %.0f / %.0f   K loop %.0f / %.0f   -> stats %.0f / %.0f   rest of epilogue %.0f / %.0f   total %.0f / %.0f\n"
               "                  first entry -> last entry %lld, first entry -> last exit %lld\n",
               what, n, seg[0] / n, mx[0], seg[1] / n, mx[1], seg[2] / n, mx[2], seg[3] / n, mx[3], seg[4] / n, mx[4], smax - tmin, tmax - tmin);
        CK(hipMemset(d_st, 0, (size_t)items * 8 * sizeof(long long)));
      };
      struct Variant { std::string label; std::function<void()> launch; bool check; bool ref; std::vector<float> t; };
      std::vector<Variant> vars;
#define RUN_REF(KERN, THREADS, LABEL) { ConvParams q = p; q.out = b.ref; vars.push_back({LABEL, [=] { hipLaunchKernelGGL(KERN, grid, dim3(THREADS), 0, 0, q); }, false, true, {}}); }
#define RUN(KERN, THREADS, LABEL, CHECK) { ConvParams q = p; vars.push_back({LABEL, [=] { hipLaunchKernelGGL(KERN, grid, dim3(THREADS), 0, 0, q); }, CHECK, false, {}}); }
      if (si == 0) {
        RUN_REF((hconv_kernel<EPI_HC, 8, 8>), 512, "production hconv_kernel<HC,8,8>")
        RUN((abl_kernel<EPI_HC, 8, 8, 1>), 512, "  ablate: no epilogue", false)
        RUN((abl_kernel<EPI_HC, 8, 8, 2>), 512, "  ablate: no per-chunk barrier / LDS store", false)
        RUN((abl_kernel<EPI_HC, 8, 8, 4>), 512, "  ablate: no weight loads in the loop", false)
        RUN((abl_kernel<EPI_HC, 8, 8, 8>), 512, "  ablate: no LDS fragment reads in the loop", false)
        RUN((abl_kernel<EPI_HC, 8, 8, 16>), 512, "  ablate: no activation loads in the loop", false)
        RUN((abl_kernel<EPI_HC, 8, 8, 31>), 512, "  ablate: all of the above (MFMAs only)", false)
        p.presum_out = (float*)d_st;
        RUN((n1_kernel<EPI_HC, 8, 8>), 512, "N1: 3 LDS buffers, mid-chunk barrier, A frag prefetch", true)
        RUN((n2_kernel<EPI_HC, 8, 8>), 512, "N2: N1 + batched epilogue", true)
        RUN((n3_kernel<EPI_HC, 8, 8>), 512, "N3: N1 + transposing epilogue (dwordx4 stores)", true)
        p.presum_out = nullptr;
        if (!fixed) RUN((n2_kernel<EPI_HC, 8, 8, 64>), 512, "N2, 64-channel chunks", true)
        RUN((hconv_kernel<EPI_HC, 8, 8>), 512, "production again (listed last)", true)
        RUN((abl_kernel<EPI_HC, 8, 8, 0>), 512, "round-2 loop (2 buffers, barrier at the chunk end)", true)
        RUN((n4_kernel<EPI_HC, 8, 8, 32, 1>), 512, "N4 (scalar bases) BD=1", true)
        RUN((n4_kernel<EPI_HC, 8, 8, 32, 2>), 512, "N4 (scalar bases) BD=2", true)
        RUN((n4_kernel<EPI_HC, 16, 4, 32, 1>), 256, "N4 4 waves x NT=16, BD=1", true)
        RUN((n4_kernel<EPI_HC, 16, 4, 32, 2>), 256, "N4 4 waves x NT=16, BD=2", true)
      } else if (si == 1) {
        RUN_REF((hconv_kernel<EPI_HC, 4, 8>), 512, "production hconv_kernel<HC,4,8>")
        RUN((abl_kernel<EPI_HC, 4, 8, 1>), 512, "  ablate: no epilogue", false)
        RUN((abl_kernel<EPI_HC, 4, 8, 2>), 512, "  ablate: no per-chunk barrier / LDS store", false)
        RUN((abl_kernel<EPI_HC, 4, 8, 4>), 512, "  ablate: no weight loads in the loop", false)
        RUN((abl_kernel<EPI_HC, 4, 8, 8>), 512, "  ablate: no LDS fragment reads in the loop", false)
        RUN((abl_kernel<EPI_HC, 4, 8, 31>), 512, "  ablate: all of the above (MFMAs only)", false)
        p.presum_out = (float*)d_st;
        RUN((n1_kernel<EPI_HC, 4, 8>), 512, "N1: 3 LDS buffers, mid-chunk barrier, A frag prefetch", true)
        RUN((n2_kernel<EPI_HC, 4, 8>), 512, "N2: N1 + batched epilogue", true)
        RUN((n3_kernel<EPI_HC, 4, 8>), 512, "N3: N1 + transposing epilogue (dwordx4 stores)", true)
        p.presum_out = nullptr;
        if (!fixed) RUN((n2_kernel<EPI_HC, 4, 8, 64>), 512, "N2, 64-channel chunks", true)
        RUN((n2_kernel<EPI_HC, 4, 8, 32, 2>), 512, "N2, weights 2 groups ahead (no pin)", true)
        if (!fixed) RUN((n2_kernel<EPI_HC, 4, 8, 64, 2>), 512, "N2, 64-channel chunks, weights 2 ahead", true)
        RUN((abl_kernel<EPI_HC, 4, 8, 0>), 512, "round-2 loop (2 buffers, barrier at the chunk end)", true)
        RUN((n4_kernel<EPI_HC, 4, 8, 32, 1>), 512, "N4 (scalar bases) BD=1", true)
        RUN((n4_kernel<EPI_HC, 4, 8, 32, 2>), 512, "N4 (scalar bases) BD=2", true)
        RUN((n4_kernel<EPI_HC, 4, 8, 32, 4>), 512, "N4 (scalar bases) BD=4", true)
        RUN((n4_kernel<EPI_HC, 8, 4, 32, 2>), 256, "N4 4 waves x NT=8, BD=2", true)
        RUN((n5_kernel<EPI_HC, 8, 4, 32, 1, 2>), 256, "N5 4 waves x NT=8, BD=1, 2 workgroups per CU", true)
        RUN((n5_kernel<EPI_HC, 4, 8, 32, 1, 4>), 512, "N5 8 waves x NT=4, BD=1, 128 registers: 2 workgroups per CU", true)
        RUN((n4_kernel<EPI_HC, 8, 4, 32, 4>), 256, "N4 4 waves x NT=8, BD=4", true)
      } else if (si == 2) {
        RUN_REF((hconv_kernel<EPI_C, 3, 11>), 704, "production hconv_kernel<C,3,11>")
        RUN((abl_kernel<EPI_C, 3, 11, 1>), 704, "  ablate: no epilogue", false)
        RUN((abl_kernel<EPI_C, 3, 11, 2>), 704, "  ablate: no per-chunk barrier / LDS store", false)
        RUN((abl_kernel<EPI_C, 3, 11, 31>), 704, "  ablate: all (MFMAs only)", false)
        RUN((n1_kernel<EPI_C, 3, 11>), 704, "N1: 3 LDS buffers, mid-chunk barrier, A frag prefetch", true)
        RUN((n2_kernel<EPI_C, 3, 11>), 704, "N2: N1 + batched epilogue", true)
        RUN((n3_kernel<EPI_C, 3, 11>), 704, "N3: N1 + transposing epilogue (dwordx4 stores)", true)
        RUN((n2_kernel<EPI_C, 3, 11, 32, 2>), 704, "N2, weights 2 groups ahead (no pin)", true)
        RUN((abl_kernel<EPI_C, 3, 11, 0>), 704, "round-2 loop (2 buffers, barrier at the chunk end)", true)
        RUN((n4_kernel<EPI_C, 3, 11, 32, 1>), 704, "N4 (scalar bases) BD=1", true)
        RUN((n4_kernel<EPI_C, 3, 11, 32, 2>), 704, "N4 (scalar bases) BD=2", true)
        RUN((n4_kernel<EPI_C, 3, 11, 32, 4>), 704, "N4 (scalar bases) BD=4", true)
      } else {
        RUN_REF((hconv_kernel<EPI_C, 4, 8>), 512, "production hconv_kernel<C,4,8>")
        RUN((abl_kernel<EPI_C, 4, 8, 1>), 512, "  ablate: no epilogue", false)
        RUN((abl_kernel<EPI_C, 4, 8, 31>), 512, "  ablate: all (MFMAs only)", false)
        RUN((n1_kernel<EPI_C, 4, 8>), 512, "N1: 3 LDS buffers, mid-chunk barrier, A frag prefetch", true)
        RUN((n2_kernel<EPI_C, 4, 8>), 512, "N2: N1 + batched epilogue", true)
        RUN((n3_kernel<EPI_C, 4, 8>), 512, "N3: N1 + transposing epilogue (dwordx4 stores)", true)
        if (!fixed) RUN((n2_kernel<EPI_C, 4, 8, 64>), 512, "N2, 64-channel chunks", true)
        RUN((n2_kernel<EPI_C, 4, 8, 32, 2>), 512, "N2, weights 2 groups ahead (no pin)", true)
        RUN((abl_kernel<EPI_C, 4, 8, 0>), 512, "round-2 loop (2 buffers, barrier at the chunk end)", true)
        RUN((n4_kernel<EPI_C, 4, 8, 32, 1>), 512, "N4 (scalar bases) BD=1", true)
        RUN((n4_kernel<EPI_C, 4, 8, 32, 2>), 512, "N4 (scalar bases) BD=2", true)
        RUN((n4_kernel<EPI_C, 4, 8, 32, 4>), 512, "N4 (scalar bases) BD=4", true)
        RUN((n5_kernel<EPI_C, 8, 4, 32, 1, 2>), 256, "N5 4 waves x NT=8, BD=1, 2 workgroups per CU", true)
        RUN((n5_kernel<EPI_C, 4, 8, 32, 1, 4>), 512, "N5 8 waves x NT=4, BD=1, 128 registers: 2 workgroups per CU", true)
      }
      // ---- run: every variant once for warm-up and the output check, then `reps` rounds over ALL variants in turn (each launch with its own
      // event pair, a cache-thrashing pass in front of it when asked for): no variant owns the cold clocks or the hot caches.
      for (auto& v : vars) {
        if (!v.ref) CK(hipMemset(b.out, 0, nout * 4));
        v.launch(); CK(hipGetLastError()); CK(hipDeviceSynchronize());
        if (v.check) { const double e = maxdiff(b, nout); v.label += e < 1e-4 ? "" : "  [MISMATCH]"; char t[64]; snprintf(t, 64, "   max|d|=%.2e", e); v.label += t; }
      }
      for (int r = 0; r < reps; ++r)
        for (auto& v : vars) {
          if (g_thrash) hipLaunchKernelGGL(thrash_kernel, dim3(2048), dim3(256), 0, 0, (float4*)g_thrash, g_thrash_n / 4);
          hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
          CK(hipEventRecord(e0, 0)); v.launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.t.push_back(ms * 1000.f);
          CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
      for (auto& v : vars) { std::sort(v.t.begin(), v.t.end()); report(v.label.c_str(), v.t[v.t.size() / 2], -1.0); }
      if (stamps) {
        for (auto& v : vars) if (v.label.rfind("N1", 0) == 0 || v.label.rfind("N3", 0) == 0) { v.launch(); CK(hipDeviceSynchronize()); show_stamps(v.label.substr(0, 2).c_str()); }
      }
    }
    CK(hipFree(b.in)); CK(hipFree(b.out)); CK(hipFree(b.ref)); CK(hipFree(b.wp));
    CK(hipFree(b.bias)); CK(hipFree(b.g1)); CK(hipFree(b.b1)); CK(hipFree(b.g2)); CK(hipFree(b.b2));
  }
  return 0;
}
