// fft_wave.h -- 1024-point complex FFT held by ONE 64-lane wavefront: 16 points per lane in registers, mixed-radix Stockham
// 16 x 16 x 4, two exchanges through a (padded) LDS buffer.  No workgroup-wide barrier is ever needed: a Griffin-Lim frame
// is one wave.  Every function is __host__ __device__ so tests/ can run the exact lane program on the CPU (fft_wave_test.cpp).
//
// Stockham pass of radix R, T = N/R lanes-worth of butterflies, p = product of the radices already done:
//   k = i & (p-1);  j = (i-k)*R + k;  u_q = in[i + q*T] * W^{q k N/(R p)};  out[j + r*p] = sum_q u_q W_R^{q r}
// Pass A (R=16, p=1):   lane i reads in[i + 64 q] (its registers), writes out[16 i + r]
// Pass B (R=16, p=16):  lane i reads in[i + 64 q], twiddle W_1024^{4 q k}, k = i & 15, writes out[256 (i>>4) + k + 16 r]
// Pass C (R=4,  p=256): butterfly i = lane + 64 c (c < 4) reads in[i + 256 q], twiddle W_1024^{q i}, result index i + 256 r
//                       -> the lane ends up holding Z[lane + 64 (c + 4 r)]: the same "lane + 64 q" layout pass A starts from.
#pragma once
#include <hip/hip_runtime.h>

namespace dctts {

#define FW_HD __host__ __device__ __forceinline__

constexpr int FW_N = 1024;
constexpr int FW_EX = FW_N + FW_N / 16 + 2;       // padded exchange buffer (float2 units), + room for one extra entry

FW_HD int fw_idx(int i) { return i + (i >> 4); }   // one pad slot per 16: lane-strided 16-float2 rows stop sharing banks

FW_HD float2 fw_mul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
FW_HD float2 fw_mulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)
template <bool INV> FW_HD float2 fw_tw(float2 a, float2 w) { return INV ? fw_mulc(a, w) : fw_mul(a, w); }      // w = e^{-i..}

// 4-point DFT in place: (a,b,c,d) <- (y0,y1,y2,y3), y_r = sum_q x_q (-/+ i)^{q r}
template <bool INV> FW_HD void fw_dft4(float2& a, float2& b, float2& c, float2& d) {
  const float2 s0 = make_float2(a.x + c.x, a.y + c.y), d0 = make_float2(a.x - c.x, a.y - c.y);
  const float2 s1 = make_float2(b.x + d.x, b.y + d.y), d1 = make_float2(b.x - d.x, b.y - d.y);
  const float2 r = INV ? make_float2(-d1.y, d1.x) : make_float2(d1.y, -d1.x);     // (b-d) * (+i | -i)
  a = make_float2(s0.x + s1.x, s0.y + s1.y);
  b = make_float2(d0.x + r.x, d0.y + r.y);
  c = make_float2(s0.x - s1.x, s0.y - s1.y);
  d = make_float2(d0.x - r.x, d0.y - r.y);
}

// v * W_16^m (forward) or its conjugate (inverse); m is a compile-time constant at every call site after unrolling
template <bool INV> FW_HD float2 fw_w16(float2 v, int m) {
  const float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
  float c, s;       // W_16^m = c - i s
  switch (m) {
    case 0: return v;
    case 1: c = C1; s = S1; break;
    case 2: c = H; s = H; break;
    case 3: c = S1; s = C1; break;
    case 4: c = 0.f; s = 1.f; break;
    case 6: c = -H; s = H; break;
    default: c = -C1; s = -S1; break;     // m == 9
  }
  if (INV) s = -s;
  return make_float2(v.x * c + v.y * s, v.y * c - v.x * s);
}

// 16-point DFT, natural order in and out:  n = n1 + 4 n2,  k = 4 k1 + k2
template <bool INV> FW_HD void fw_dft16(float2 v[16]) {
#pragma unroll
  for (int n1 = 0; n1 < 4; ++n1) fw_dft4<INV>(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);      // v[n1 + 4 k2] = A[n1][k2]
#pragma unroll
  for (int n1 = 1; n1 < 4; ++n1)
#pragma unroll
    for (int k2 = 1; k2 < 4; ++k2) v[n1 + 4 * k2] = fw_w16<INV>(v[n1 + 4 * k2], n1 * k2);
  float2 o[16];
#pragma unroll
  for (int k2 = 0; k2 < 4; ++k2) {
    float2 a = v[4 * k2], b = v[1 + 4 * k2], c = v[2 + 4 * k2], d = v[3 + 4 * k2];
    fw_dft4<INV>(a, b, c, d);                                                                  // over n1 -> k1
    o[k2] = a; o[4 + k2] = b; o[8 + k2] = c; o[12 + k2] = d;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = o[i];
}

// ---- the three passes, each split into "compute on registers" and "store / load through the exchange buffer" so a caller
// ---- can put its (wave-level) barrier between them.  v[q] <-> element lane + 64 q on entry of A and on exit of C.
// Exchange addresses are written as (per-lane base) + (compile-time constant) so they become DS immediate offsets instead
// of ~100 live address registers:  fw_idx(lane + 64 q) = fw_base(lane) + 68 q,  fw_idx(16 lane + r) = 17 lane + r,
// fw_idx(256 (lane>>4) + (lane&15) + 16 r) = 272 (lane>>4) + (lane&15) + 17 r.
FW_HD int fw_base(int lane) { return lane + (lane >> 4); }

template <bool INV> FW_HD void fw_passA_store(float2 v[16], float2* ex, int lane) {
  fw_dft16<INV>(v);
  float2* o = ex + 17 * lane;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = v[r];
}
template <bool INV> FW_HD void fw_passB_load(float2 v[16], const float2* ex, int lane, const float2* w1024) {
  const float2* in = ex + fw_base(lane);
  const unsigned k4 = 4u * (unsigned)(lane & 15);          // unsigned offsets from the uniform table base: one VGPR per address
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = in[68 * q];
#pragma unroll
  for (int q = 1; q < 16; ++q) v[q] = fw_tw<INV>(v[q], w1024[k4 * (unsigned)q]);      // W_1024^{4 q k}
  fw_dft16<INV>(v);
}
FW_HD void fw_passB_store(const float2 v[16], float2* ex, int lane) {
  float2* o = ex + 272 * (lane >> 4) + (lane & 15);
#pragma unroll
  for (int r = 0; r < 16; ++r) o[17 * r] = v[r];
}
template <bool INV> FW_HD void fw_passC_load(float2 v[16], const float2* ex, int lane, const float2* w1024) {
  const float2* in = ex + fw_base(lane);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const unsigned i = (unsigned)(lane + 64 * c);
    float2 u0 = in[68 * c], u1 = in[68 * c + 272], u2 = in[68 * c + 544], u3 = in[68 * c + 816];
    u1 = fw_tw<INV>(u1, w1024[i]); u2 = fw_tw<INV>(u2, w1024[2u * i]); u3 = fw_tw<INV>(u3, w1024[3u * i]);
    fw_dft4<INV>(u0, u1, u2, u3);
    v[c] = u0; v[c + 4] = u1; v[c + 8] = u2; v[c + 12] = u3;       // index i + 256 r = lane + 64 (c + 4 r)
  }
}

}  // namespace dctts
