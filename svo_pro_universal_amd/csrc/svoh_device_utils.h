// svoh_device_utils.h -- device helpers shared by the Gauss-Newton kernels (sparse_align.hip, pose.hip):
// the wave-level reduce-scatter of normal-equation accumulators and the register-resident pivoted LDL^T.
#pragma once

#include <hip/hip_runtime.h>
#include <cfloat>

#include "svoh_internal.h"

namespace svoh {

// ---- wave-level reduce-scatter of the normal-equation accumulators ----
// A butterfly in which every stage halves the number of live values: the lane pair
// (l, l ^ off) splits the index range, each side keeps one half and adds the partner's
// copy of it.  NACC -> ceil/2 -> ... -> 1 takes 31 exchanges for the 29 accumulators of
// the SE3 case (47 for 45) instead of 6 x NACC with an all-reduce per value, and the
// two widest stages (off 1 and 2) run on the DPP network instead of ds_bpermute.
// Afterwards lane l holds the wave total of accumulator `idx` (if `valid`).
template <int OFF>
__device__ __forceinline__ double xor_exchange(double v)
{
  if constexpr (OFF == 1 || OFF == 2) {
    constexpr int ctrl = OFF == 1 ? 0xB1 : 0x4E;  // quad_perm [1,0,3,2] / [2,3,0,1]
    const unsigned long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, ctrl, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), ctrl, 0xF, 0xF, false);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
  } else if constexpr (OFF == 4 || OFF == 8) {
    // inside a row of 16 lanes: both rotations over the DPP network, each lane keeps the one that comes from lane ^ OFF
    // (row_ror:n hands lane l the value of lane (l - n) mod 16)
    const unsigned long long b = __double_as_longlong(v);
    const int lo = (int)(unsigned)b, hi = (int)(unsigned)(b >> 32);
    const int lo_dn = __builtin_amdgcn_update_dpp(0, lo, 0x120 + OFF, 0xF, 0xF, false), hi_dn = __builtin_amdgcn_update_dpp(0, hi, 0x120 + OFF, 0xF, 0xF, false);
    const int lo_up = __builtin_amdgcn_update_dpp(0, lo, 0x120 + 16 - OFF, 0xF, 0xF, false), hi_up = __builtin_amdgcn_update_dpp(0, hi, 0x120 + 16 - OFF, 0xF, 0xF, false);
    const bool upper = (__lane_id() & OFF) != 0;   // this lane's partner is the lower one: lane - OFF
    const int rlo = upper ? lo_dn : lo_up, rhi = upper ? hi_dn : hi_up;
    return __longlong_as_double(((unsigned long long)(unsigned)rhi << 32) | (unsigned)rlo);
  } else {
    // across rows: gfx950's v_permlane16_swap (odd rows of the first operand <-> even rows of the second) and
    // v_permlane32_swap (upper half <-> lower half) instead of two trips through the LDS crossbar (ds_bpermute)
    static_assert(OFF == 16 || OFF == 32, "wave64 butterfly");
    const unsigned long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    const bool upper = (__lane_id() & OFF) != 0;
    unsigned rlo, rhi;
    if constexpr (OFF == 16) {
      const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
      const auto c = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
      rlo = upper ? a[0] : a[1]; rhi = upper ? c[0] : c[1];
    } else {
      const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
      const auto c = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
      rlo = upper ? a[0] : a[1]; rhi = upper ? c[0] : c[1];
    }
    return __longlong_as_double(((unsigned long long)rhi << 32) | rlo);
  }
}

template <int N, int OFF>
__device__ __forceinline__ void reduce_scatter_stage(double* v, int lane, int& base, int& cnt)
{
  constexpr int HALF = (N + 1) / 2;
  const bool up = (lane & OFF) != 0;
#pragma unroll
  for (int i = 0; i < HALF; ++i) {
    const double a = v[i];
    const double b = (i + HALF < N) ? v[i + HALF] : 0.0;
    const double recv = xor_exchange<OFF>(up ? a : b);
    v[i] = (up ? b : a) + recv;
  }
  if (up) { base += HALF; cnt -= HALF; }
  else cnt = cnt < HALF ? cnt : HALF;
}

template <int NACC>
__device__ __forceinline__ void wave_reduce_scatter(double (&v)[NACC], int lane, int& idx, bool& valid)
{
  constexpr int N1 = (NACC + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2, N5 = (N4 + 1) / 2;
  int base = 0, cnt = NACC;
  reduce_scatter_stage<NACC, 1>(v, lane, base, cnt);
  reduce_scatter_stage<N1, 2>(v, lane, base, cnt);
  reduce_scatter_stage<N2, 4>(v, lane, base, cnt);
  reduce_scatter_stage<N3, 8>(v, lane, base, cnt);
  reduce_scatter_stage<N4, 16>(v, lane, base, cnt);
  reduce_scatter_stage<N5, 32>(v, lane, base, cnt);
  idx = base;
  valid = cnt >= 1;
}

// 8x8 LDL^T with diagonal pivoting, Eigen-3.4 semantics (see svoh_math.h), with
// the matrix held in registers: every index is a compile-time constant after
// unrolling, the run-time pivot position only steers predicated swaps.  Used by
// the one lane that runs the Gauss-Newton bookkeeping, where the LDS-resident
// variant paid an LDS round trip per matrix access.
__device__ __forceinline__ void swap_d(double& a, double& b) { const double t = a; a = b; b = t; }

// packed lower triangle: element (r, c), r >= c, at r*(r+1)/2 + c
#ifndef SVOH_L
#define SVOH_L(r, c) m[(r) * ((r) + 1) / 2 + (c)]
#endif
// Eigen 3.4 LDLT<Lower> (unblocked, diagonal pivoting) in registers, split into the factorisation and the solve
// so that a caller whose matrix does not change between solves (inverse-compositional Gauss-Newton: the Hessian
// of a level is constant while the visible set is) can keep the factor.  m: packed lower triangle, overwritten
// with L (unit diagonal implied) and D; tr: transpositions.  Returns false when the whole diagonal is zero:
// Eigen then stops with identity transpositions and the D^-1 step zeroes every component of the solution.
template <int N>
__device__ __forceinline__ bool ldlt_factor_regs(double (&m)[N * (N + 1) / 2], int (&tr)[N])
{
#pragma unroll
  for (int k = 0; k < N; ++k) tr[k] = k;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    int big = k;
    double bigv = fabs(SVOH_L(k, k));
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const double v = fabs(SVOH_L(i, i));
      const bool gt = v > bigv;
      bigv = gt ? v : bigv;
      big = gt ? i : big;
    }
    tr[k] = big;
#pragma unroll
    for (int bb = k + 1; bb < N; ++bb) {
      if (big == bb) {
#pragma unroll
        for (int c = 0; c < k; ++c) swap_d(SVOH_L(k, c), SVOH_L(bb, c));
#pragma unroll
        for (int r = bb + 1; r < N; ++r) swap_d(SVOH_L(r, k), SVOH_L(r, bb));
        swap_d(SVOH_L(k, k), SVOH_L(bb, bb));
#pragma unroll
        for (int i = k + 1; i < bb; ++i) swap_d(SVOH_L(i, k), SVOH_L(bb, i));
      }
    }
    if (k > 0) {
      double tmp[N];
#pragma unroll
      for (int c = 0; c < k; ++c) tmp[c] = SVOH_L(c, c) * SVOH_L(k, c);
      double accd = 0.0;
#pragma unroll
      for (int c = 0; c < k; ++c) accd += SVOH_L(k, c) * tmp[c];
      SVOH_L(k, k) -= accd;
#pragma unroll
      for (int r = k + 1; r < N; ++r) {
        double sacc = 0.0;
#pragma unroll
        for (int c = 0; c < k; ++c) sacc += SVOH_L(r, c) * tmp[c];
        SVOH_L(r, k) -= sacc;
      }
    }
    const double akk = SVOH_L(k, k);
    const bool pivot_ok = fabs(akk) > 0.0;
    if (k == 0 && !pivot_ok) return false;
    if (pivot_ok) {
#pragma unroll
      for (int r = k + 1; r < N; ++r) SVOH_L(r, k) /= akk;
    }
  }
  return true;
}

// x <- A^-1 x with the factor of ldlt_factor_regs (nonzero = its return value); false when the solution is NaN.
template <int N>
__device__ __forceinline__ bool ldlt_apply_regs(const double (&m)[N * (N + 1) / 2], const int (&tr)[N], bool nonzero, double (&x)[N])
{
  if (!nonzero) {
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] = 0.0;
    return true;
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int bb = k + 1; bb < N; ++bb)
      if (tr[k] == bb) swap_d(x[k], x[bb]);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double sacc = x[i];
#pragma unroll
    for (int c = 0; c < i; ++c) sacc -= SVOH_L(i, c) * x[c];
    x[i] = sacc;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double d = SVOH_L(i, i);
    x[i] = (fabs(d) > DBL_MIN) ? x[i] / d : 0.0;
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    double sacc = x[i];
#pragma unroll
    for (int c = i + 1; c < N; ++c) sacc -= SVOH_L(c, i) * x[c];
    x[i] = sacc;
  }
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
#pragma unroll
    for (int bb = k + 1; bb < N; ++bb)
      if (tr[k] == bb) swap_d(x[k], x[bb]);
  }
  return !(x[0] != x[0]);
}


// ---- the same solve by the lanes of ONE wave (round 4) -------------------------------------------------------------
// ldlt_apply_regs is a chain of ~400 instructions on one lane: two permutations by predicated swaps, 2 x 15 dependent
// multiply-adds, N divisions.  By the lanes i < N of a wave, lane i owning component i: the permutation is an index
// (perm[], written once by the lane that factorised), the N divisions are ONE division, a step of the forward
// substitution is one multiply-add for all lanes behind it.  Every component goes through exactly the operations of the
// one-lane code in its order (x_i -= L(i,c) x_c for ascending c, forward and backward): the same bits.
__device__ __forceinline__ double wave_bcast_f64(double v, int src_lane)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// perm[i]: the input component that the transpositions of the factorisation move to position i
template <int N>
__device__ __forceinline__ void ldlt_perm_from_transpositions(const int (&tr)[N], int* perm)
{
  int idx[N];
#pragma unroll
  for (int k = 0; k < N; ++k) idx[k] = k;
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int bb = k + 1; bb < N; ++bb)
      if (tr[k] == bb) { const int t = idx[k]; idx[k] = idx[bb]; idx[bb] = t; }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) perm[k] = idx[k];
}

// fact: packed lower triangle as ldlt_factor_regs leaves it (LDS or global); rhs_at_perm: component perm[lane] of the
// right-hand side (lanes < N).  Returns this lane's solved component, which belongs at index perm[lane] of the solution.
// Called by all 64 lanes of a wave (lanes >= N take part in the broadcasts only).
template <int N>
__device__ __forceinline__ double ldlt_apply_wave(const double* fact, bool nonzero, double rhs_at_perm, int lane)
{
  if (!nonzero) return 0.0;
  const int i = lane < N ? lane : N - 1;
  double M[N];     // M[c] = L(i, c) for c < i, L(c, i) for c > i (the lane's row and column of the factor)
#pragma unroll
  for (int c = 0; c < N; ++c) {
    const int r2 = c < i ? i : c, c2 = c < i ? c : i;
    M[c] = fact[r2 * (r2 + 1) / 2 + c2];
  }
  const double dd = fact[i * (i + 1) / 2 + i];
  double x = rhs_at_perm;
#pragma unroll
  for (int c = 0; c < N - 1; ++c) {
    const double xc = wave_bcast_f64(x, c);
    const double t = __builtin_fma(-M[c], xc, x);
    x = lane > c ? t : x;
  }
  x = (fabs(dd) > DBL_MIN) ? x / dd : 0.0;
#pragma unroll
  for (int r = N - 2; r >= 0; --r) {
#pragma unroll
    for (int c = r + 1; c < N; ++c) {
      const double xc = wave_bcast_f64(x, c);
      const double t = __builtin_fma(-M[c], xc, x);
      x = lane == r ? t : x;
    }
  }
  return x;
}

template <int N>
__device__ __forceinline__ bool ldlt_solve_regs(double (&m)[N * (N + 1) / 2], double (&x)[N])
{
  int tr[N];
  const bool nonzero = ldlt_factor_regs<N>(m, tr);
  return ldlt_apply_regs<N>(m, tr, nonzero, x);
}


}  // namespace svoh
