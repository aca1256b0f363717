// Host-side run of the lane program in dc_tts_amd/csrc/fft_wave.h: 64 "lanes" executed one after the other, the exchange
// buffer a plain array.  Prints the max abs error of the forward and inverse 1024-point transform against a double DFT.
// Built and run by tests/test_host.py (no GPU needed: no HIP API call is made).
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>

#include "../dc_tts_amd/csrc/fft_wave.h"

using namespace dctts;

template <bool INV>
static void run(const std::vector<float2>& in, std::vector<float2>& out, const std::vector<float2>& w) {
  std::vector<float2> ex(FW_EX);
  float2 v[64][16];
  for (int l = 0; l < 64; ++l) for (int q = 0; q < 16; ++q) v[l][q] = in[l + 64 * q];
  for (int l = 0; l < 64; ++l) fw_passA_store<INV>(v[l], ex.data(), l);
  for (int l = 0; l < 64; ++l) fw_passB_load<INV>(v[l], ex.data(), l, w.data());
  for (int l = 0; l < 64; ++l) fw_passB_store(v[l], ex.data(), l);
  for (int l = 0; l < 64; ++l) fw_passC_load<INV>(v[l], ex.data(), l, w.data());
  out.resize(FW_N);
  for (int l = 0; l < 64; ++l) for (int q = 0; q < 16; ++q) out[l + 64 * q] = v[l][q];
}

int main() {
  std::vector<float2> w(FW_N), x(FW_N), y;
  for (int m = 0; m < FW_N; ++m) w[m] = make_float2((float)std::cos(2.0 * M_PI * m / FW_N), (float)-std::sin(2.0 * M_PI * m / FW_N));
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& e : x) e = make_float2(rnd(), rnd());
  double worst[2] = {0, 0};
  for (int inv = 0; inv < 2; ++inv) {
    if (inv) run<true>(x, y, w); else run<false>(x, y, w);
    for (int k = 0; k < FW_N; k += 7) {
      std::complex<double> acc = 0;
      for (int n = 0; n < FW_N; ++n) {
        const double a = (inv ? 2.0 : -2.0) * M_PI * (double)((long)k * n % FW_N) / FW_N;
        acc += std::complex<double>(x[n].x, x[n].y) * std::complex<double>(std::cos(a), std::sin(a));
      }
      worst[inv] = std::fmax(worst[inv], std::abs(acc - std::complex<double>(y[k].x, y[k].y)));
    }
  }
  std::printf("%.3e %.3e\n", worst[0], worst[1]);
  return 0;
}
