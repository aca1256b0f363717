// xmlp_kernel.h -- the seven k = 1 layers around the mel frame of a decode step as ONE launch in TEAM form (round 4).
//
// AudioDec C_8 .. C_11, the mel frame (sigmoid of C_11's layer-norm output, networks.py:192-210) and AudioEnc C_1 .. C_3 of the next frame
// (networks.py:82-105) are seven dependent (<= 256) x (<= 256) layers on the newest row of every utterance.  Round 2 ran them split by ROWS
// (mlp_rows_kernel: a workgroup owns two utterances and every column, so nothing is exchanged -- and every workgroup streams each layer's whole
// 256 KB weight matrix through ONE compute unit: ~3.9 us per layer, 27.7 us per frame, the longest launch of the chain).  Here they are split by
// COLUMNS exactly like xgroup_kernel.h: a team of 16 workgroups that the command processor placed on one XCD owns four utterances, workgroup
// `grp` owns output columns 16 grp .. 16 grp + 15 (its 16 KB slice of a layer's weights: one 16x16x4 fp32 MFMA tile, K split over the 8 waves),
// publishes its slice of the pre-norm row + the partial layer-norm statistics of those 16 columns with plain stores (they stay in the XCD's L2),
// passes the team's flag-word barrier and reads the other slices past its L1.  The next layer's weights, bias and layer-norm parameters are
// requested before the barrier.  Correctness does not rest on the placement: same barrier words (sequence number + XCD of the writer), same
// bounded spins and error word as xgroup_kernel.h.
//
// Input: the pre-norm rows [b][512] + partial statistics [b][16][4] AudioDec's last highway layer left (xgroup_kernel's pout / stats_out) and
// that layer's input rows (the highway residual).  Output: mel frame j -> ypad / logits; the LAST layer's pre-norm rows [b][256] + partial
// statistics [b][16][4] in the form xgroup_kernel's AudioEnc run reads as its P0 / stats0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xgroup_kernel.h"

namespace dctts {

struct XMlpLayer {
  const float* wp;                       // 16-column tiles [tile][nkg][lane][4] (pack_bw, MF = 16)
  const float* bias; const float* g; const float* be;
  int nkg, cout, act, pad_;              // k-groups of 16 input channels (16 for 256 channels, 6 for the 80 mel channels); ACT_NONE / ACT_RELU / ACT_SIGMOID (the mel layer)
};
struct XMlpParams {
  int B, nl, mel_layer, pad0;
  const float* P0; int p0_bs; int res_bs; const float* stats0;          // the highway producer: pre-norm rows (gate | info), partial statistics [b][16][4]
  const float* g1; const float* b1; const float* g2; const float* b2;   // its layer-norm parameters
  const float* res;                                                     // its input row of utterance b at res + b * res_bs (the highway residual)
  XMlpLayer lay[7];
  float* ymel; float* logits; int y_bs, l_bs;                           // mel frame of utterance b -> ymel + b * y_bs, its logits -> logits + b * l_bs
  float* pout; float* stats_out;                                        // the last layer's pre-norm rows [b][256] + statistics [b][16][4] (unused when the mel layer is the last)
  float* xch; float* sch; int xch_set, sch_set;                         // exchange: [2][B_pad][256] pre-norm rows, [2][B_pad][16][2] statistics; floats between the parity copies
  unsigned* bar; unsigned bar_base; int* err;
};

// grid: 128 blocks of 512 threads, whatever the batch (a team takes the utterance groups team, team + 8, ... in turn)
__global__ void __launch_bounds__(512) xmlp_kernel(const XMlpParams* __restrict__ pp) {
  __shared__ __attribute__((aligned(16))) float red[8 * 4 * 64];
  __shared__ int s_go;
  __shared__ XMlpLayer s_lay[7];
  __shared__ __attribute__((aligned(16))) float s_xs[8][4 * 32];
  typedef const __attribute__((address_space(4))) XMlpParams CP;
  CP& p = *(CP*)pp;
  asm volatile("; xmlp: parameters, one batch"
               :: "s"(p.B), "s"(p.nl), "s"(p.mel_layer), "s"(p.P0), "s"(p.p0_bs), "s"(p.res_bs), "s"(p.stats0), "s"(p.g1), "s"(p.b1), "s"(p.g2), "s"(p.b2), "s"(p.res),
                  "s"(p.lay[0].wp), "s"(p.lay[0].bias), "s"(p.lay[0].nkg), "s"(p.lay[0].cout), "s"(p.xch), "s"(p.sch), "s"(p.xch_set), "s"(p.sch_set),
                  "s"(p.bar), "s"(p.bar_base), "s"(p.err));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int team = (int)blockIdx.x & 7, grp = ((int)blockIdx.x >> 3) & 15;
  if (team * 4 >= p.B) return;
  const int arow = lane & 15, aq = lane >> 4, c4 = aq * 4;
  const int cr = lane >> 4, cc = lane & 15;              // compact layout: lane (cr, cc) owns row cr and the channels 16 w + cc, 128 + 16 w + cc
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  unsigned* const bar = p.bar + team * 32;
  const unsigned xcc = xg_xcc_id();
  const int nl = p.nl;
  if (tid < (int)(sizeof(XMlpLayer) * 7 / 4)) reinterpret_cast<uint32_t*>(s_lay)[tid] = reinterpret_cast<const uint32_t*>(pp->lay)[tid];
  if (tid == 0) s_go = __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;     // an earlier launch of this decode already failed: no more waiting
  float* const xs = s_xs[wave];
  const unsigned ch0 = (unsigned)(16 * wave + cc), ch1 = ch0 + 128u;

  for (int round = 0, m0 = team * 4; m0 < p.B; ++round, m0 += 32) {
    if (round > 0) __syncthreads();
    const unsigned rbase = p.bar_base + (unsigned)round * (unsigned)nl * 16u;
    const unsigned crow = (m0 + cr < p.B) ? (unsigned)(m0 + cr) : 0u;
    const bool crow_ok = m0 + cr < p.B;
    const int erow = aq * 4 + wave;                      // reduce phase (waves 0 .. 3): the row this lane finishes; real rows are 0 .. 3
    const int eb = m0 + erow;
    const bool wr = wave < 4 && erow < 4 && eb < p.B;

    // ---- layer 0's slice of the weights + the first rows (the producer is an earlier launch on this stream: plain loads)
    f32x4 vb[2];
    float cbias = 0.f;
    {
      const int nkg = p.lay[0].nkg;
      const float* wb = p.lay[0].wp + lane * 4;
#pragma unroll
      for (int e = 0; e < 2; ++e) { const int kg = wave + 8 * e; vb[e] = ldv(wb, (unsigned)(grp * nkg + (kg < nkg ? kg : nkg - 1)) * 256u); }
      cbias = p.lay[0].bias[grp * 16 + (lane & 15)];
    }
    float xc[2];
    {
      const float* pr = p.P0 + crow * (unsigned)p.p0_bs;
      const float hg0 = pr[ch0], hg1 = pr[ch1], hi0 = pr[256u + ch0], hi1 = pr[256u + ch1];
      const f32x4 stc = ldv(p.stats0, crow * 64u + (unsigned)cc * 4u);
      const float r0 = p.res[crow * (unsigned)p.res_bs + ch0], r1_ = p.res[crow * (unsigned)p.res_bs + ch1];
      const float a1 = p.g1[ch0], a2 = p.g1[ch1], c1 = p.b1[ch0], c2 = p.b1[ch1], d1_ = p.g2[ch0], d2_ = p.g2[ch1], e1 = p.b2[ch0], e2 = p.b2[ch1];
      const float m1 = row16_sum(stc[0]) * (1.0f / 16.0f), m2 = row16_sum(stc[2]) * (1.0f / 16.0f);
      const float dd1 = stc[0] - m1, dd2 = stc[2] - m2;
      const float r1 = rsqrt_fast(row16_sum(stc[1] + 16.0f * dd1 * dd1) * (1.0f / 256.0f) + 1e-12f);
      const float r2 = rsqrt_fast(row16_sum(stc[3] + 16.0f * dd2 * dd2) * (1.0f / 256.0f) + 1e-12f);
      { const float s_ = sigmoid_fast((hg0 - m1) * r1 * a1 + c1); xc[0] = s_ * ((hi0 - m2) * r2 * d1_ + e1) + (1.0f - s_) * r0; }
      { const float s_ = sigmoid_fast((hg1 - m1) * r1 * a2 + c2); xc[1] = s_ * ((hi1 - m2) * r2 * d2_ + e2) + (1.0f - s_) * r1_; }
    }
    if (round == 0) __syncthreads();                     // s_lay, s_go
    const bool team_ok = s_go != 0;
    xs[cr * 32 + cc] = xc[0]; xs[cr * 32 + 16 + cc] = xc[1];
    float4 x[2];
    x[0] = *reinterpret_cast<const float4*>(&xs[(arow & 3) * 32 + c4]);
    x[1] = *reinterpret_cast<const float4*>(&xs[(arow & 3) * 32 + 16 + c4]);

    for (int l = 0; l < nl; ++l) {
      const int nkg = __builtin_amdgcn_readfirstlane(s_lay[l].nkg), cout = __builtin_amdgcn_readfirstlane(s_lay[l].cout), act = __builtin_amdgcn_readfirstlane(s_lay[l].act);
      const bool last = (l + 1 == nl);
      const bool mine = grp * 16 < cout;                 // this workgroup owns real columns of the layer (80 columns: groups 0 .. 4)
      // ---- contraction: one 16-column tile, K split over the waves (k-groups w and w + 8)
      f32x4 acc = z4;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const f32x4 b = (mine && wave + 8 * e < nkg) ? vb[e] : z4;
        const float4 a = x[e];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[3], acc, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(wave * 4 + j) * 64 + lane] = acc[j];
      // ---- requests that do not depend on the other workgroups: the next layer's slice and bias, this layer's layer-norm parameters
      float lg[2], lb[2], nbias = 0.f;
      {
        const unsigned i0 = ch0 < (unsigned)cout ? ch0 : 0u, i1 = ch1 < (unsigned)cout ? ch1 : 0u;
        lg[0] = ldg1(s_lay[l].g, i0); lg[1] = ldg1(s_lay[l].g, i1); lb[0] = ldg1(s_lay[l].be, i0); lb[1] = ldg1(s_lay[l].be, i1);
      }
      if (!last) {
        const int nkg2 = __builtin_amdgcn_readfirstlane(s_lay[l + 1].nkg), cout2 = __builtin_amdgcn_readfirstlane(s_lay[l + 1].cout);
        const int tile2 = (grp * 16 < cout2) ? grp : 0;
        const float* wb = s_lay[l + 1].wp + lane * 4;
#pragma unroll
        for (int e = 0; e < 2; ++e) { const int kg = wave + 8 * e; vb[e] = ldg4(wb, (unsigned)(tile2 * nkg2 + (kg < nkg2 ? kg : nkg2 - 1)) * 256u); }
        nbias = ldg1(s_lay[l + 1].bias, (unsigned)(tile2 * 16 + (lane & 15)));
      }
      __syncthreads();
      float v_ = 0.f, mg = 0.f, m2g = 0.f;
      if (wave < 4) {
#pragma unroll
        for (int w = 0; w < 8; ++w) v_ += red[(w * 4 + wave) * 64 + lane];
        v_ += cbias;
        mg = row16_sum(v_) * (1.0f / 16.0f);
        const float dv = v_ - mg;
        m2g = row16_sum(dv * dv);
      }
      const int pcol = grp * 16 + (lane & 15);
      if (last && l != p.mel_layer) {
        // the AudioEnc run that follows on this stream reads these as its P0 / stats0 (xgroup_kernel.h, layer 0)
        if (wr && mine) {
          p.pout[(long)eb * 256 + pcol] = v_;
          if ((lane & 15) == 0) { float* so = p.stats_out + ((long)eb * 16 + grp) * 4; so[0] = mg; so[1] = m2g; }
        }
        break;
      }
      // ---- publish this workgroup's slice with PLAIN stores (they stay in this XCD's L2), arrive at the team's barrier
      const int par = l & 1;
      if (wr && mine) {
        p.xch[(long)par * p.xch_set + (long)eb * 256 + pcol] = v_;
        if ((lane & 15) == 0) { float* so = p.sch + (long)par * p.sch_set + ((long)eb * 16 + grp) * 2; so[0] = mg; so[1] = m2g; }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (wave == 0) {
        const unsigned target = rbase + (unsigned)(l + 1) * 16u, me = (target << 4) | xcc;
        if (lane == 0) __hip_atomic_store(bar + grp, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (team_ok) {
          int spins = 0;
          for (;;) {
            const unsigned v = lane < 16 ? __hip_atomic_load(bar + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : me;
            const bool there = (int)((v >> 4) - target) >= 0;
            if (__builtin_amdgcn_ballot_w64(there && (v & 15u) != xcc) != 0ull) { if (lane == 0) atomicOr(p.err, 2); break; }      // a split team
            if (__builtin_amdgcn_ballot_w64(!there) == 0ull) break;
            if (++spins > (1 << 16) || ((spins & 255) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { if (lane == 0) atomicOr(p.err, 1); break; }
          }
        }
      }
      __syncthreads();
      // ---- the team's rows of layer l, past the L1: this lane's two channels of row cr + column group cc's partial statistics
      const int ng = cout >> 4;
      float h0, h1, s0, s1;
      {
        const float* xr = p.xch + (long)par * p.xch_set + (long)crow * 256 + (ch0 < (unsigned)cout ? ch0 : 0u);      // (+128 floats: the second channel; inside the row either way)
        const float* sr = p.sch + (long)par * p.sch_set + ((long)crow * 16 + (cc < ng ? cc : 0)) * 2;
        asm volatile(
            "global_load_dword %0, %4, off sc1\n\t"
            "global_load_dword %1, %4, off offset:512 sc1\n\t"
            "global_load_dword %2, %5, off sc1\n\t"
            "global_load_dword %3, %5, off offset:4 sc1\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(h0), "=&v"(h1), "=&v"(s0), "=&v"(s1)
            : "v"(xr), "v"(sr)
            : "memory");
      }
      {
        const bool gok = cc < ng;
        const float inv_g = 1.0f / (float)ng;
        const float m1 = row16_sum(gok ? s0 : 0.f) * inv_g;
        const float d1 = s0 - m1;
        const float r1 = rsqrt_fast(row16_sum(gok ? s1 + 16.0f * d1 * d1 : 0.f) * (inv_g * (1.0f / 16.0f)) + 1e-12f);
        float y0 = (h0 - m1) * r1 * lg[0] + lb[0], y1 = (h1 - m1) * r1 * lg[1] + lb[1];
        if (l == p.mel_layer) {
          // the mel frame (networks.py:210): logits = the layer-norm output, Y = sigmoid(logits); column group 0's workgroup stores the team's rows
          const float q0 = sigmoid_fast(y0), q1 = sigmoid_fast(y1);
          if (grp == 0 && crow_ok) {
            if (ch0 < (unsigned)cout) { p.logits[(long)crow * p.l_bs + ch0] = y0; p.ymel[(long)crow * p.y_bs + ch0] = q0; }
            if (ch1 < (unsigned)cout) { p.logits[(long)crow * p.l_bs + ch1] = y1; p.ymel[(long)crow * p.y_bs + ch1] = q1; }
          }
          y0 = q0; y1 = q1;
        } else if (act == ACT_RELU) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
        xs[cr * 32 + cc] = ch0 < (unsigned)cout ? y0 : 0.f;
        xs[cr * 32 + 16 + cc] = ch1 < (unsigned)cout ? y1 : 0.f;
        x[0] = *reinterpret_cast<const float4*>(&xs[(arow & 3) * 32 + c4]);
        x[1] = *reinterpret_cast<const float4*>(&xs[(arow & 3) * 32 + 16 + c4]);
      }
      cbias = nbias;
      if (last) break;                                    // (the mel layer as the last one: the final frame of a decode)
    }
  }
}

}  // namespace dctts
