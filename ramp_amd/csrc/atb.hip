// atb_kernel: the attention backward on sample-owning waves (d(q, k, v) from q, k, v, d(o)), split from atk.hip in round 6 so that atk.hip
// (ato_kernel) can be compiled with -mllvm -amdgpu-mfma-vgpr-form while this kernel -- 368 registers at T = 48, every MFMA result in the
// accumulation half by the compiler's default -- keeps the code it was tuned with.  The product path reaches it only where abl_kernel is
// switched off (RAMP_ABL=0) and through ramp_op_atb / the stress cases.
#include "args_attention.h"
#include "tokmma.h"
#include "atkmma.h"

#include <algorithm>

namespace ramp {

// ---- attention backward on sample-owning waves ---------------------------------------------------------------------------------------------
// d(q, k, v) of softmax(q k^T / 8) v given d(o) (CrossAttention.forward differentiated, layers_attention_mini.py:101-127), replacing
// attn2_bwd_kernel on the levels whose token count divides 48 or 32.  A wave owns T = 48 (32) tokens = whole samples, one head at a time; no
// LDS, no barrier, two waves per SIMD.  Everything is 16 x 16 tiles on v_mfma_f32_16x16x32_f16 in the fp16x3 split with exact per-wave
// power-of-two operand scales.  "T-layout" registers (lane = token, registers = 4 consecutive features: one float4 load from the row-major rows)
// are MFMA operands whose free index is the token and whose contracted index is the feature: S^T = K Q^T and dP^T = V dO^T come straight from
// them.  The products that contract over tokens (dQ^T = K^T dS^T, dV^T = dO^T P, dK^T = Q^T dS) need operands with the FEATURE (or the other
// token index) on the lane: those are made in registers by an MFMA against a 0 / 1 selection matrix -- C[token][j] = sum_k X[token][k] [k == j]
// has the token in the accumulator's row (registers) and the feature on its column (lane); the two planes accumulate hi + lo exactly -- and
// split again; P and dS are turned the same way.  No transposed memory access, no LDS tile.
template <int NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NG == 3 ? 1 : 2, NG == 3 ? 1 : 2)))
void atb_kernel(AtbArgs a, int n_tiles) {
  constexpr int T = 16 * NG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int tile = (int)blockIdx.x;
  const int tok0 = tile * (4 * T) + wave * T;
  if (tok0 >= a.M) return;                                  // (no barrier anywhere: an empty wave may leave)
  const bool full = tok0 + T <= a.M;
  const int m_last = a.M - 1;

  unsigned kmask[NG];                                       // keys (16 kg + 4 g + i) in the sample of query 16 qg + c: bit 4 kg + i
#pragma unroll
  for (int qg = 0; qg < NG; ++qg) {
    unsigned m = 0;
#pragma unroll
    for (int kg = 0; kg < NG; ++kg)
#pragma unroll
      for (int i = 0; i < 4; ++i) m |= ((16 * kg + 4 * g + i) / a.L == (16 * qg + c) / a.L ? 1u : 0u) << (4 * kg + i);
    kmask[qg] = m;
  }
  // selection operands: as B operand of a k32 step whose A operand pairs two 16-wide blocks, sel[b] picks block b: B[k][j] = [k is element j of block b]
  u32x4 sel[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    unsigned w[4];
#pragma unroll
    for (int p2 = 0; p2 < 4; ++p2) {                        // dword p2 holds elements e = 2 p2, 2 p2 + 1 of the lane's 8
      unsigned v = 0;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int e = 2 * p2 + q;
        if ((e >> 2) == b && 4 * g + (e & 3) == c) v |= 0x3c00u << (16 * q);      // 1.0 in fp16
      }
      w[p2] = v;
    }
    sel[b] = u32x4{w[0], w[1], w[2], w[3]};
  }
  const u32x2 z2 = {0u, 0u};

  // row offsets (bytes) of the wave's tokens: token 16 t + c, + 16 g bytes (features 4 g ..)
  unsigned roff[NG], ooff[NG];
#pragma unroll
  for (int t = 0; t < NG; ++t) {
    const unsigned tk = (unsigned)min(tok0 + 16 * t + c, m_last);
    roff[t] = tk * 3072u + 16u * (unsigned)g;
    ooff[t] = tk * 1024u + 16u * (unsigned)g;
  }
  const char* qb = reinterpret_cast<const char*>(a.QKV);
  const char* ob = reinterpret_cast<const char*>(a.dO);
  char* gb = reinterpret_cast<char*>(a.dQKV);

  // a T-layout operand in two halves: the 12 row loads (raw), then -- a phase later, so that their latency hides behind the phase in between --
  // the exact power-of-two scale of its first use in the head and its planes [token group][k32 step = feature-block pair]
  auto issue = [&](const char* base, const unsigned (&off)[NG], int hoff, f32x4 (&raw)[4][NG]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NG; ++t)
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) raw[fb][t] = *reinterpret_cast<const f32x4*>(base + off[t] + hoff + 64 * fb);
  };
  auto finish = [&](const f32x4 (&raw)[4][NG], float& scale, u32x4 (&hi)[NG][2], u32x4 (&lo)[NG][2]) __attribute__((always_inline)) {
    if (scale == 0.f) {
      float mx = 0.f;
#pragma unroll
      for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int t = 0; t < NG; ++t) mx = amax4(raw[fb][t], mx);
      scale = pow2_scale(wave_max(mx), 13);
    }
#pragma unroll
    for (int t = 0; t < NG; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        u32x2 h0, l0, h1, l1;
        split4s(raw[2 * j][t], scale, h0, l0); split4s(raw[2 * j + 1][t], scale, h1, l1);
        hi[t][j] = cat2(h0, h1); lo[t][j] = cat2(l0, l1);
      }
  };
  // turn the T-layout planes of one k32 step (two feature blocks) of ALL token groups into the planes with the feature on the lane and
  // the tokens in the registers: out[f][pair] = tokens of groups (0, 1) | (2, -) of feature block 2 j + f, as a k32-step A operand
  auto turn = [&](const u32x4 (&hi)[NG][2], const u32x4 (&lo)[NG][2], int j, u32x4 (&oh)[2][2], u32x4 (&ol)[2][2]) __attribute__((always_inline)) {
    planes_fence();                                         // (asm-written low planes -> MFMA operand: atkmma.h; round 6, correct by construction)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      u32x2 th[NG], tl[NG];
#pragma unroll
      for (int t = 0; t < NG; ++t) {
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        x = mm32(hi[t][j], sel[f], x);
        x = mm32(lo[t][j], sel[f], x);
        unsigned h0, h1, l0, l1;
        split4m(x, h0, h1, l0, l1);
        th[t] = u32x2{h0, h1}; tl[t] = u32x2{l0, l1};
      }
      oh[f][0] = cat2(th[0], th[1]); ol[f][0] = cat2(tl[0], tl[1]);
      oh[f][1] = NG == 3 ? cat2(th[NG - 1], z2) : u32x4{0u, 0u, 0u, 0u};
      ol[f][1] = NG == 3 ? cat2(tl[NG - 1], z2) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  // out^T[d][token] (feature blocks 2 j, 2 j + 1) = sum over tokens' (A = turned operand, feature on the lane) x (B = bh / bl: planes with
  // the contracted token in the registers, [pair][free token group]); scaled and stored as rows of dqkv
  auto contract_store = [&](const u32x4 (&ah)[2][2], const u32x4 (&al)[2][2], const u32x4 (&bh)[2][NG], const u32x4 (&bl)[2][NG], int j, float oscale,
                            int col0) __attribute__((always_inline)) {
    planes_fence();
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int t = 0; t < NG; ++t) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pr = 0; pr < (NG == 3 ? 2 : 1); ++pr) {
          o = mm32(ah[f][pr], bl[pr][t], o);
          o = mm32(al[f][pr], bh[pr][t], o);
          o = mm32(ah[f][pr], bh[pr][t], o);
        }
        if (full || tok0 + 16 * t + c < a.M) *reinterpret_cast<f32x4*>(gb + roff[t] + col0 + 64 * (2 * j + f)) = o * oscale;
      }
  };

  f32x4 rawA[4][NG], rawB[4][NG];                           // rows in flight: requested one phase before they are split
  issue(qb, roff, 0, rawA); issue(qb, roff, 1024, rawB);    // head 0: q, k
#pragma unroll 1
  for (int h = 0; h < 4; ++h) {
    const int hoff = 256 * h;
    float sq = 0.f, sk = 0.f, sv = 0.f, sdo = 0.f, sds = 0.f;
    f32x4 pt[NG][NG], ds[NG][NG];                           // P^T, then dS^T: [key group][query group], keys in the registers, query on the lane
    // ---- S^T = K Q^T -> P^T (softmax over keys, masked to the query's sample)
    {
      u32x4 qh[NG][2], ql[NG][2], kh[NG][2], kl[NG][2];
      finish(rawA, sq, qh, ql);
      finish(rawB, sk, kh, kl);
      issue(qb, roff, hoff + 2048, rawA); issue(ob, ooff, hoff, rawB);      // v, d(o): in flight during the scores and the softmax
      const float ssc = 0.125f * 1.4426950408889634f / (sq * sk);
      planes_fence();
#pragma unroll
      for (int qg = 0; qg < NG; ++qg) {
        f32x4 st[NG];
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) {
          f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            x = mm32(kh[kg][j], ql[qg][j], x);
            x = mm32(kl[kg][j], qh[qg][j], x);
            x = mm32(kh[kg][j], qh[qg][j], x);
          }
          st[kg] = x;
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int kg = 0; kg < NG; ++kg)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = (kmask[qg] >> (4 * kg + i)) & 1u ? st[kg][i] * ssc : -3.0e38f;
            st[kg][i] = v;
            mx = fmaxf(mx, v);
          }
        mx = gmax(mx);
        float sum = 0.f;
#pragma unroll
        for (int kg = 0; kg < NG; ++kg)
#pragma unroll
          for (int i = 0; i < 4; ++i) { const float e = __builtin_amdgcn_exp2f(st[kg][i] - mx); st[kg][i] = e; sum += e; }
        sum = gsum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) pt[kg][qg] = st[kg] * inv;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- dP^T = V dO^T ; delta_q = sum_k P dP ; dS^T = P^T (dP^T - delta)
    u32x4 doh[NG][2], dol[NG][2];
    {
      u32x4 vh[NG][2], vl[NG][2];
      finish(rawA, sv, vh, vl);
      finish(rawB, sdo, doh, dol);
      issue(qb, roff, hoff + 1024, rawA);                   // k again (L1 / L2; held since the first phase it would cost 96 registers): in flight during dP
      const float dsc = 1.f / (sv * sdo);
      float dmax = 0.f;
      planes_fence();
#pragma unroll
      for (int qg = 0; qg < NG; ++qg) {
        f32x4 dp[NG];
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) {
          f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            x = mm32(vh[kg][j], dol[qg][j], x);
            x = mm32(vl[kg][j], doh[qg][j], x);
            x = mm32(vh[kg][j], doh[qg][j], x);
          }
          dp[kg] = x * dsc;
        }
        float delta = 0.f;
#pragma unroll
        for (int kg = 0; kg < NG; ++kg)
#pragma unroll
          for (int i = 0; i < 4; ++i) delta += pt[kg][qg][i] * dp[kg][i];
        delta = gsum(delta);
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) {
#pragma unroll
          for (int i = 0; i < 4; ++i) ds[kg][qg][i] = pt[kg][qg][i] * (dp[kg][i] - delta);
          dmax = amax4(ds[kg][qg], dmax);
        }
      }
      sds = pow2_scale(wave_max(dmax), 13);
    }
    __builtin_amdgcn_sched_barrier(0);
    // planes of P^T (x 2^13) and dS^T (x sds) as token-contracting B operands: [key-group pair][query group]
    u32x4 pbh[2][NG], pbl[2][NG], sbh[2][NG], sbl[2][NG];
    {
#pragma unroll
      for (int qg = 0; qg < NG; ++qg) {
        u32x2 ph[NG], pl[NG], sh[NG], sl[NG];
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) { split4s(pt[kg][qg], 8192.f, ph[kg], pl[kg]); split4s(ds[kg][qg], sds, sh[kg], sl[kg]); }
        pbh[0][qg] = cat2(ph[0], ph[1]); pbl[0][qg] = cat2(pl[0], pl[1]);
        sbh[0][qg] = cat2(sh[0], sh[1]); sbl[0][qg] = cat2(sl[0], sl[1]);
        pbh[1][qg] = NG == 3 ? cat2(ph[NG - 1], z2) : u32x4{0u, 0u, 0u, 0u}; pbl[1][qg] = NG == 3 ? cat2(pl[NG - 1], z2) : u32x4{0u, 0u, 0u, 0u};
        sbh[1][qg] = NG == 3 ? cat2(sh[NG - 1], z2) : u32x4{0u, 0u, 0u, 0u}; sbl[1][qg] = NG == 3 ? cat2(sl[NG - 1], z2) : u32x4{0u, 0u, 0u, 0u};
      }
    }
    // ---- dQ^T = K^T dS^T / 8: A = k turned (feature on the lane, keys in the registers), B = dS^T (keys in the registers, query on the lane)
    {
      u32x4 kh[NG][2], kl[NG][2];
      finish(rawA, sk, kh, kl);
      issue(qb, roff, hoff, rawA);                          // q again: in flight during dQ and dV
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        u32x4 th[2][2], tl[2][2];
        turn(kh, kl, j, th, tl);
        contract_store(th, tl, sbh, sbl, j, 0.125f / (sk * sds), hoff);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // P and dS with the QUERY in the registers and the key on the lane: turned tile by tile (the selection MFMA on a pair of key groups)
    u32x4 pqh[2][NG], pql[2][NG], sqh[2][NG], sql[2][NG];   // [query-group pair][key group]
    {
      u32x2 tph[NG][NG], tpl[NG][NG], tsh[NG][NG], tsl[NG][NG];      // [query group][key group]
      planes_fence();
#pragma unroll
      for (int qg = 0; qg < NG; ++qg)
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) {
          const int pr = kg >> 1, b = kg & 1;
          f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = {0.f, 0.f, 0.f, 0.f};
          x = mm32(pbh[pr][qg], sel[b], x); x = mm32(pbl[pr][qg], sel[b], x);
          y = mm32(sbh[pr][qg], sel[b], y); y = mm32(sbl[pr][qg], sel[b], y);
          unsigned h0, h1, l0, l1;
          split4m(x, h0, h1, l0, l1); tph[qg][kg] = u32x2{h0, h1}; tpl[qg][kg] = u32x2{l0, l1};
          split4m(y, h0, h1, l0, l1); tsh[qg][kg] = u32x2{h0, h1}; tsl[qg][kg] = u32x2{l0, l1};
        }
#pragma unroll
      for (int kg = 0; kg < NG; ++kg) {
        pqh[0][kg] = cat2(tph[0][kg], tph[1][kg]); pql[0][kg] = cat2(tpl[0][kg], tpl[1][kg]);
        sqh[0][kg] = cat2(tsh[0][kg], tsh[1][kg]); sql[0][kg] = cat2(tsl[0][kg], tsl[1][kg]);
        pqh[1][kg] = NG == 3 ? cat2(tph[NG - 1][kg], z2) : u32x4{0u, 0u, 0u, 0u}; pql[1][kg] = NG == 3 ? cat2(tpl[NG - 1][kg], z2) : u32x4{0u, 0u, 0u, 0u};
        sqh[1][kg] = NG == 3 ? cat2(tsh[NG - 1][kg], z2) : u32x4{0u, 0u, 0u, 0u}; sql[1][kg] = NG == 3 ? cat2(tsl[NG - 1][kg], z2) : u32x4{0u, 0u, 0u, 0u};
      }
    }
    // ---- dV^T = dO^T P: A = d(o) turned (feature on the lane, queries in the registers), B = P (queries in the registers, key on the lane)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      u32x4 th[2][2], tl[2][2];
      turn(doh, dol, j, th, tl);
      contract_store(th, tl, pqh, pql, j, 1.f / (sdo * 8192.f), hoff + 2048);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- dK^T = Q^T dS / 8
    {
      u32x4 qh[NG][2], ql[NG][2];
      finish(rawA, sq, qh, ql);
      {                                                     // the next head's q, k: in flight during dK (the last head re-requests its own: unused)
        const int hn = (h < 3 ? h + 1 : h) * 256;
        issue(qb, roff, hn, rawA); issue(qb, roff, hn + 1024, rawB);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        u32x4 th[2][2], tl[2][2];
        turn(qh, ql, j, th, tl);
        contract_store(th, tl, sqh, sql, j, 0.125f / (sq * sds), hoff + 1024);
      }
    }
  }
}

int launch_atb(const AtbArgs& a, hipStream_t s) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  int ng = 0;
  RAMP_REQUIRE(ato_applicable(a.M, a.L, &ng), "atb: tokens per sample must divide 48 or 32 (and M be whole samples)");
  RAMP_REQUIRE(a.QKV && a.dO && a.dQKV && al16(a.QKV) && al16(a.dO) && al16(a.dQKV), "atb: operands must be non-null and 16-byte aligned");
  RAMP_REQUIRE((long)a.M * 3072 < (1l << 32), "atb: 32-bit row offsets bound M to 1398100 tokens");
  RAMP_REQUIRE(!ranges_overlap(a.dQKV, (size_t)a.M * 3072, a.QKV, (size_t)a.M * 3072) && !ranges_overlap(a.dQKV, (size_t)a.M * 3072, a.dO, (size_t)a.M * 1024),
               "atb: the output must not overlap the operands");
  const int T = 16 * ng, n_tiles = (a.M + 4 * T - 1) / (4 * T);
  if (ng == 3) hipLaunchKernelGGL((atb_kernel<3>), dim3(n_tiles), dim3(256), 0, s, a, n_tiles);
  else hipLaunchKernelGGL((atb_kernel<2>), dim3(n_tiles), dim3(256), 0, s, a, n_tiles);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace ramp
