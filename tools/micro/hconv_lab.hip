// hconv_lab.hip -- standalone timing lab for the SSRN / TextEnc throughput kernel.  Not product code, and since round 5 it holds NO kernel of its own: every variant
// is an instantiation of the PRODUCT template dc_tts_amd/csrc/hconv_kernel.h (weight ring depth BD, scalar tile bases SB, the extra-column form XC, the opt-in
// split-bf16 contraction BF) on the layer shapes of networks.py:214-292 at the bench batch (B = 32 -> 26 880 rows at 4T).  (Rounds 3-4 kept 1 267 lines of copied loop /
// epilogue variants and ablations beside it -- tools/micro/hconv_lab_kernels.h, deleted: git history; what they measured is in profiles/r03_hconv_lab.txt and
// DESIGN.md section 4.)  Protocol: every variant is timed IN TURN over `reps` rounds with a 1 GB cache-thrashing pass in front of every timed launch, median reported --
// repeating one kernel back to back keeps its weights cache-resident and re-ranks the variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dc_tts_amd/csrc tools/micro/hconv_lab.hip -o tools/micro/kp_hconv_lab && gpurun -- 'tools/micro/kp_hconv_lab 9'
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "hconv_kernel.h"

using namespace dctts;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

namespace dctts {      // (hconv_kernel.h declares the library's launchers; the lab launches instantiations itself)
hipError_t launch_hconv(const ConvShape&, const ConvParams&, hipStream_t, int) { return hipErrorInvalidConfiguration; }
}

static float frand(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((s >> 40) & 0xffffff) / 8388608.0f - 1.0f; }

__global__ void thrash_kernel(float4* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; v.x += 1.f; p[i] = v; }
}

struct Shape { const char* name; int epi, cin, cin_p, ntaps, dil, cout, act; };
struct Variant { std::string label; std::function<void()> launch; bool check; std::vector<float> t; };

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 9;
  const int only = argc > 2 ? atoi(argv[2]) : -1;
  const int B = 32, R = 840, PADR = 64;
  const Shape shapes[] = {
      {"HC_11 (1024->2048, k=3)", EPI_HC, 1024, 1024, 3, 1, 1024, ACT_NONE},
      {"HC_8  (512->1024, k=3)", EPI_HC, 512, 512, 3, 1, 512, ACT_NONE},
      {"C_14  (1025->1025, k=1)", EPI_C, 1056, 1056, 1, 1, 1025, ACT_RELU},
      {"C_10  (512->1024, k=1)", EPI_C, 512, 512, 1, 1, 1024, ACT_NONE},
  };
  float* thrash = nullptr; const size_t thrash_n = (size_t)256 << 20;
  CK(hipMalloc(&thrash, thrash_n * 4)); CK(hipMemset(thrash, 0, thrash_n * 4));
  for (int si = 0; si < 4; ++si) {
    if (only >= 0 && si != only) continue;
    const Shape& S = shapes[si];
    const int stride_in = (S.cin + 31) / 32 * 32, stride_out = (S.cout + 31) / 32 * 32;
    const long rows = PADR + R + PADR;
    const size_t nin = (size_t)B * rows * stride_in, nout = (size_t)B * rows * stride_out;
    const int tiles = (S.epi == EPI_HC ? 2 : 1) * ((S.cout + 31) / 32), K = S.ntaps * S.cin_p;
    const size_t nw = (size_t)tiles * K * 32;                 // floats of the fp32 packing = bytes/4 of the bf16 (hi, mid) packing: the same buffer serves both (timing only for BF)
    uint64_t seed = 1234 + si;
    std::vector<float> h(nin);
    for (auto& v : h) v = frand(seed);
    float *in, *out, *ref, *wp, *wx, *raw, *bias, *g1, *b1, *g2, *b2;
    CK(hipMalloc(&in, nin * 4)); CK(hipMemcpy(in, h.data(), nin * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, nout * 4)); CK(hipMalloc(&ref, nout * 4)); CK(hipMalloc(&raw, (size_t)B * R * 2 * S.cout * 4));
    std::vector<float> w(nw);
    const float ws = 1.0f / std::sqrt((float)(S.ntaps * S.cin));
    for (auto& v : w) v = frand(seed) * ws;
    CK(hipMalloc(&wp, nw * 4)); CK(hipMemcpy(wp, w.data(), nw * 4, hipMemcpyHostToDevice));
    std::vector<float> v1(4096);
    auto up = [&](float** d, float base, float amp) { for (auto& v : v1) v = base + amp * frand(seed); CK(hipMalloc(d, 4096 * 4)); CK(hipMemcpy(*d, v1.data(), 4096 * 4, hipMemcpyHostToDevice)); };
    up(&bias, 0.f, 0.1f); up(&g1, 1.f, 0.1f); up(&b1, 0.f, 0.1f); up(&g2, 1.f, 0.1f); up(&b2, 0.f, 0.1f); up(&wx, 0.f, ws);
    ConvParams p; memset(&p, 0, sizeof(p));
    p.in = in; p.in_bstride = rows; p.in_row0 = PADR; p.in_stride = stride_in; p.cin = S.cin; p.cin_p = S.cin_p; p.ntaps = S.ntaps;
    if (S.ntaps == 3) { p.tap_off[0] = -S.dil; p.tap_off[1] = 0; p.tap_off[2] = S.dil; }
    p.R = R; p.wp = wp; p.bias = bias; p.g1 = g1; p.b1 = b1; p.g2 = g2; p.b2 = b2; p.cout = S.cout;
    p.out = out; p.out_bstride = rows; p.out_row0 = PADR; p.out_stride = stride_out; p.out_tmul = 1; p.out_tadd = 0; p.act = S.act; p.out_zero_to = stride_out;
    const double flop_row = 2.0 * S.ntaps * S.cin * (S.epi == EPI_HC ? 2 : 1) * S.cout;
    printf("== %s   tiles=%d K=%d  %.2f MFLOP/row\n", S.name, tiles, K, flop_row / 1e6);
    for (int items : {256, 768}) {
      p.M = items * 32;
      std::vector<Variant> vars;
#define RUN(KERN, THREADS, GY, LABEL, CHECK, PATCH) { ConvParams q = p; PATCH; const dim3 grid_(items, GY); vars.push_back({LABEL, [=] { hipLaunchKernelGGL(KERN, grid_, dim3(THREADS), 0, 0, q); }, CHECK, {}}); }
      if (si == 0) {
        RUN((hconv_kernel<EPI_HC, 8, 8, 1, 1>), 512, 1, "round 5: hconv_kernel<HC,8,8,BD=1,SB=1>", false, q.out = ref)
        RUN((hconv_kernel<EPI_HC, 8, 8, 2, 1>), 512, 1, "BD=2", true, )
        RUN((hconv_kernel<EPI_HC, 8, 8, 1, 0>), 512, 1, "SB=0", true, )
        RUN((hconv_kernel<EPI_HC, 8, 8, 1, 1, 0, 0, 0, 1>), 512, 1, "production (round 6): SB=1 SG=1 (requests spread: one behind every three MFMAs)", true, )
        RUN((hconv_kernel<EPI_HC, 8, 8, 1, 0, 0, 0, 0, 1>), 512, 1, "SB=0 SG=1", true, )
        RUN((hconv_kernel<EPI_HC, 8, 8, 1, 0, 0, 0, 0, 2>), 512, 1, "SB=0 SG=2 (every four)", true, )
        RUN((hconv_kernel<EPI_HC, 8, 8, 1, 0, 0, 0, 0, 3>), 512, 1, "SB=0 SG=3 (every two, first half)", true, )
        RUN((hconv_kernel<EPI_HC, 8, 8, 2, 0, 0, 0, 0, 1>), 512, 1, "BD=2 SB=0 SG=1", true, )
        RUN((hconv_kernel<EPI_HC, 4, 8, 2, 1, 2, 0, 1>), 512, 2, "split-bf16: two column halves (pre-norm values to HBM; + the finishing pass, not timed)", false, q.raw_out = raw; q.raw_ld = 2 * q.cout)
      } else if (si == 1) {
        RUN((hconv_kernel<EPI_HC, 4, 8, 2, 0>), 512, 1, "production hconv_kernel<HC,4,8,BD=2,SB=0>", false, q.out = ref)
        RUN((hconv_kernel<EPI_HC, 4, 8, 1, 1>), 512, 1, "BD=1 SB=1", true, )
        RUN((hconv_kernel<EPI_HC, 8, 4, 1, 1, 0, 0, 0, 1>), 256, 1, "4 waves x 8 tiles, two workgroups per CU, SG=1", true, )
        RUN((hconv_kernel<EPI_HC, 8, 4, 1, 1, 0, 0, 0, 0>), 256, 1, "4 waves x 8 tiles, two workgroups per CU, SG=0", true, )
        RUN((hconv_kernel<EPI_HC, 8, 4, 1, 0, 0, 0, 0, 1>), 256, 1, "4 waves x 8 tiles, two workgroups per CU, SB=0 SG=1", true, )
        RUN((hconv_kernel<EPI_HC, 4, 8, 2, 1>), 512, 1, "BD=2 SB=1", true, )
        RUN((hconv_kernel<EPI_HC, 4, 8, 2, 1, 0, 0, 0, 1>), 512, 1, "BD=2 SB=1 SG=1", true, )
        RUN((hconv_kernel<EPI_HC, 4, 8, 2, 0, 0, 0, 0, 1>), 512, 1, "BD=2 SB=0 SG=1", true, )
        RUN((hconv_kernel<EPI_HC, 4, 8, 2, 1, 0, 0, 0, 2>), 512, 1, "BD=2 SB=1 SG=2", true, )
        RUN((hconv_kernel<EPI_HC, 4, 8, 2, 1, 0, 0, 0, 3>), 512, 1, "BD=2 SB=1 SG=3", true, )
        RUN((hconv_kernel<EPI_HC, 4, 8, 4, 1, 0, 0, 0, 1>), 512, 1, "BD=4 SB=1 SG=1", true, )
        RUN((hconv_kernel<EPI_HC, 4, 8, 2, 1, 0, 0, 1>), 512, 1, "split-bf16 BD=2", false, )
        RUN((hconv_kernel<EPI_HC, 4, 8, 1, 1, 0, 0, 1>), 512, 1, "split-bf16 BD=1", false, )
      } else if (si == 2) {
        RUN((hconv_kernel<EPI_C, 3, 11, 2, 1>), 704, 1, "rounds 1-4: hconv_kernel<C,3,11,BD=2,SB=1> (33 tiles on 11 waves)", false, q.out = ref)
        RUN((hconv_kernel<EPI_C, 4, 8, 1, 1, 0, 2>), 512, 1, "round 5: 8 waves x 4 tiles + column 1024 on the vector ALU, 128 registers (XC=2)", false, q.wx = wx)
        RUN((hconv_kernel<EPI_C, 4, 8, 2, 1, 0, 1>), 512, 1, "the same at ring depth 2, registers uncapped (one workgroup per CU)", false, q.wx = wx)
        RUN((hconv_kernel<EPI_C, 4, 8, 1, 1, 0, 2, 0, 1>), 512, 1, "XC=2 SG=1", false, q.wx = wx)
        RUN((hconv_kernel<EPI_C, 4, 8, 1, 1, 0, 2, 0, 2>), 512, 1, "XC=2 SG=2", false, q.wx = wx)
        RUN((hconv_kernel<EPI_C, 4, 8, 2, 1, 0, 1, 0, 1>), 512, 1, "production (round 6): XC=1 BD=2 SG=1", false, q.wx = wx)
        RUN((hconv_kernel<EPI_C, 4, 8, 2, 1, 0, 1, 1>), 512, 1, "split-bf16 XC BD=2", false, q.wx = wx)
      } else {
        RUN((hconv_kernel<EPI_C, 4, 8, 1, 1>), 512, 1, "production hconv_kernel<C,4,8,BD=1,SB=1>", false, q.out = ref)
        RUN((hconv_kernel<EPI_C, 4, 8, 2, 1>), 512, 1, "BD=2", true, )
        RUN((hconv_kernel<EPI_C, 8, 4, 1, 1, 0, 0, 0, 1>), 256, 1, "4 waves x 8 tiles, two workgroups per CU, SG=1", true, )
        RUN((hconv_kernel<EPI_C, 8, 4, 1, 1>), 256, 1, "4 waves x 8 tiles, two workgroups per CU, SG=0", true, )
        RUN((hconv_kernel<EPI_C, 4, 8, 2, 1, 0, 0, 0, 1>), 512, 1, "BD=2 SG=1", true, )
        RUN((hconv_kernel<EPI_C, 4, 8, 1, 1, 0, 0, 0, 2>), 512, 1, "BD=1 SG=2", true, )
        RUN((hconv_kernel<EPI_C, 4, 8, 2, 1, 0, 0, 1>), 512, 1, "split-bf16 BD=2", false, )
      }
      for (auto& v : vars) {
        CK(hipMemset(out, 0, nout * 4));
        v.launch(); CK(hipGetLastError()); CK(hipDeviceSynchronize());
        if (v.check) {
          std::vector<float> a(nout), r(nout);
          CK(hipMemcpy(a.data(), out, nout * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), ref, nout * 4, hipMemcpyDeviceToHost));
          double m = 0; for (size_t i = 0; i < nout; ++i) { const double d = std::fabs((double)a[i] - (double)r[i]); if (!(d <= m)) m = d; }
          char t[64]; snprintf(t, 64, "   max|d|=%.2e%s", m, m < 1e-4 ? "" : "  [MISMATCH]"); v.label += t;
        }
      }
      for (int r = 0; r < reps; ++r)
        for (auto& v : vars) {
          hipLaunchKernelGGL(thrash_kernel, dim3(2048), dim3(256), 0, 0, (float4*)thrash, thrash_n / 4);
          hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
          CK(hipEventRecord(e0, 0)); v.launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.t.push_back(ms * 1000.f);
          CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
      for (auto& v : vars) {
        std::sort(v.t.begin(), v.t.end());
        const float us = v.t[v.t.size() / 2];
        const double tf = flop_row * p.M / us / 1e6;
        printf("  items=%3d %-100s %8.1f us  %6.1f TF  %.3f of the fp32 MFMA peak\n", items, v.label.c_str(), us, tf, tf / 157.3);
      }
      fflush(stdout);
    }
    CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(ref)); CK(hipFree(wp)); CK(hipFree(wx)); CK(hipFree(raw));
    CK(hipFree(bias)); CK(hipFree(g1)); CK(hipFree(b1)); CK(hipFree(g2)); CK(hipFree(b2));
  }
  return 0;
}
