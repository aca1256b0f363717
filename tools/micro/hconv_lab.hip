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

template <typename F>
static float time_launch(F&& launch, int reps) {
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
    p.R = R; p.wp = b.wp; p.bias = b.bias; p.g1 = b.g1; p.b1 = b.b1; p.g2 = b.g2; p.b2 = b.b2; p.cout = S.cout;
    p.out = b.out; p.out_bstride = rows; p.out_row0 = PADR; p.out_stride = stride_out; p.out_tmul = 1; p.out_tadd = 0; p.act = S.act;
    p.out_zero_to = stride_out;
    const double flop_row = 2.0 * S.ntaps * S.cin * (S.epi == EPI_HC ? 2 : 1) * S.cout;
    printf("== %s   tiles=%d KG=%d  %.2f MFLOP/row\n", S.name, tiles, KG, flop_row / 1e6);

    for (int items : {256, 768}) {
      p.M = items * 32;
      const dim3 grid(items);
      auto report = [&](const char* what, float us, double err) {
        const double tf = flop_row * p.M / us / 1e6;
        if (err >= 0) printf("  items=%3d %-58s %8.1f us  %6.1f TF  %.3f of peak   max|d|=%.2e\n", items, what, us, tf, tf / 157.3, err);
        else printf("  items=%3d %-58s %8.1f us  %6.1f TF  %.3f of peak\n", items, what, us, tf, tf / 157.3);
        fflush(stdout);
      };
#define RUN_REF(KERN, THREADS, LABEL)                                                                       \
  { p.out = b.ref; float us = time_launch([&] { hipLaunchKernelGGL(KERN, grid, dim3(THREADS), 0, 0, p); }, reps); \
    CK(hipGetLastError()); report(LABEL, us, -1); p.out = b.out; }
#define RUN(KERN, THREADS, LABEL, CHECK)                                                                    \
  { CK(hipMemset(b.out, 0, nout * 4));                                                                       \
    float us = time_launch([&] { hipLaunchKernelGGL(KERN, grid, dim3(THREADS), 0, 0, p); }, reps);          \
    CK(hipGetLastError()); CK(hipDeviceSynchronize());                                                        \
    report(LABEL, us, CHECK ? maxdiff(b, nout) : -1.0); }
      if (si == 0) {
        RUN_REF((hconv_kernel<EPI_HC, 8, 8>), 512, "production hconv_kernel<HC,8,8>")
        RUN((abl_kernel<EPI_HC, 8, 8, 1>), 512, "  ablate: no epilogue", false)
        RUN((abl_kernel<EPI_HC, 8, 8, 2>), 512, "  ablate: no per-chunk barrier / LDS store", false)
        RUN((abl_kernel<EPI_HC, 8, 8, 4>), 512, "  ablate: no weight loads in the loop", false)
        RUN((abl_kernel<EPI_HC, 8, 8, 8>), 512, "  ablate: no LDS fragment reads in the loop", false)
        RUN((abl_kernel<EPI_HC, 8, 8, 16>), 512, "  ablate: no activation loads in the loop", false)
        RUN((abl_kernel<EPI_HC, 8, 8, 31>), 512, "  ablate: all of the above (MFMAs only)", false)
        RUN((n1_kernel<EPI_HC, 8, 8>), 512, "N1: 3 LDS buffers, mid-chunk barrier, A frag prefetch", true)
        RUN((hconv_kernel<EPI_HC, 4, 16>), 1024, "N3: 16 waves x NT=4", true)
        RUN((n1_kernel<EPI_HC, 4, 16>), 1024, "N1+N3", true)
      } else if (si == 1) {
        RUN_REF((hconv_kernel<EPI_HC, 4, 8>), 512, "production hconv_kernel<HC,4,8>")
        RUN((abl_kernel<EPI_HC, 4, 8, 1>), 512, "  ablate: no epilogue", false)
        RUN((abl_kernel<EPI_HC, 4, 8, 2>), 512, "  ablate: no per-chunk barrier / LDS store", false)
        RUN((abl_kernel<EPI_HC, 4, 8, 4>), 512, "  ablate: no weight loads in the loop", false)
        RUN((abl_kernel<EPI_HC, 4, 8, 8>), 512, "  ablate: no LDS fragment reads in the loop", false)
        RUN((abl_kernel<EPI_HC, 4, 8, 31>), 512, "  ablate: all of the above (MFMAs only)", false)
        RUN((n1_kernel<EPI_HC, 4, 8>), 512, "N1: 3 LDS buffers, mid-chunk barrier, A frag prefetch", true)
        RUN((hconv_kernel<EPI_HC, 2, 16>), 1024, "N3: 16 waves x NT=2", true)
        RUN((n1_kernel<EPI_HC, 2, 16>), 1024, "N1+N3", true)
      } else if (si == 2) {
        RUN_REF((hconv_kernel<EPI_C, 3, 11>), 704, "production hconv_kernel<C,3,11>")
        RUN((abl_kernel<EPI_C, 3, 11, 1>), 704, "  ablate: no epilogue", false)
        RUN((abl_kernel<EPI_C, 3, 11, 2>), 704, "  ablate: no per-chunk barrier / LDS store", false)
        RUN((abl_kernel<EPI_C, 3, 11, 31>), 704, "  ablate: all (MFMAs only)", false)
        RUN((n1_kernel<EPI_C, 3, 11>), 704, "N1: 3 LDS buffers, mid-chunk barrier, A frag prefetch", true)
      } else {
        RUN_REF((hconv_kernel<EPI_C, 4, 8>), 512, "production hconv_kernel<C,4,8>")
        RUN((abl_kernel<EPI_C, 4, 8, 1>), 512, "  ablate: no epilogue", false)
        RUN((abl_kernel<EPI_C, 4, 8, 31>), 512, "  ablate: all (MFMAs only)", false)
        RUN((n1_kernel<EPI_C, 4, 8>), 512, "N1: 3 LDS buffers, mid-chunk barrier, A frag prefetch", true)
        RUN((hconv_kernel<EPI_C, 2, 16>), 1024, "N3: 16 waves x NT=2", true)
      }
    }
    CK(hipFree(b.in)); CK(hipFree(b.out)); CK(hipFree(b.ref)); CK(hipFree(b.wp));
    CK(hipFree(b.bias)); CK(hipFree(b.g1)); CK(hipFree(b.b1)); CK(hipFree(b.g2)); CK(hipFree(b.b2));
  }
  return 0;
}
