// advantage.hip -- V-trace / GAE advantage ("puff advantage") for gfx950, behind the C-ABI (mg_compute_advantage).
//
// Replaces torch.ops.pufferlib.compute_puff_advantage
//   (packages/pufferlib-core/src/pufferlib/extensions/cuda/pufferlib.cu:7-86, CPU twin extensions/pufferlib.cpp:28-85),
// which the reference cannot use on ROCm: metta/rl/advantage.py:94-97 falls back to a Python loop over the horizon
// (metta/rl/mps.py:5-41, about five kernel launches per time step).
//
// Layout: five f32 arrays [num_rows][horizon], row-major.  The recurrence runs backwards along a row and is kept
// strictly sequential per row, operation by operation as the reference's CPU function does it (no fused
// multiply-add: this file is compiled with -ffp-contract=off), so results are bit-identical to it.
//
// The reference kernel gives each thread a row, so the 64 lanes of a wave touch 64 different cache lines per load.
// Here a wave owns ADV_ROWS rows and walks the horizon in tiles of ADV_TT steps: the tile of each input is read with
// coalesced 16-byte loads (whole 256-byte row segments) issued one tile ahead, transposed through LDS (row stride
// ADV_TT + 1: bank conflict free both lane-per-step and lane-per-row); all lanes compute the terms that do not depend
// on the running value, one lane per row runs the recurrence, and the advantages leave coalesced.
// Bound: HBM (20 B per element: four reads, one write); with few rows it is latency bound (one tile = one round trip).
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/mettagrid_hip.h"

#define ADV_ROWS 8
#define ADV_TT 64
#define ADV_LD (ADV_TT + 1)

extern int mg_fail_msg(const std::string& m);  // mettagrid_hip.hip: sets mg_last_error(), returns 1

// Staging registers of one tile: 4 inputs x ADV_ROWS x ADV_TT floats over 64 lanes.
template <bool VEC4>
struct AdvStage {
  static constexpr int PER_ROW = ADV_TT / 4, ROWS_PER_IT = 64 / PER_ROW, ITS = ADV_ROWS / ROWS_PER_IT;
  float4 v[VEC4 ? ITS : ADV_ROWS / 4], r[VEC4 ? ITS : ADV_ROWS / 4], d[VEC4 ? ITS : ADV_ROWS / 4],
      w[VEC4 ? ITS : ADV_ROWS / 4];
};

template <bool VEC4>
__device__ __forceinline__ void adv_load_tile(AdvStage<VEC4>& x, const float* __restrict__ values,
                                              const float* __restrict__ rewards, const float* __restrict__ dones,
                                              const float* __restrict__ importance, long row0, int num_rows,
                                              int horizon, int lo, int n, int lane) {
  if (VEC4) {
    constexpr int PER_ROW = AdvStage<VEC4>::PER_ROW, ROWS_PER_IT = AdvStage<VEC4>::ROWS_PER_IT;
    const int j = (lane % PER_ROW) * 4, ri = lane / PER_ROW;
#pragma unroll
    for (int it = 0; it < AdvStage<VEC4>::ITS; it++) {
      const int i = it * ROWS_PER_IT + ri;
      x.v[it] = x.r[it] = x.d[it] = x.w[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + i < num_rows && j < n) {  // horizon % 4 == 0: a float4 lies inside the row or outside
        const size_t g = (size_t)(row0 + i) * horizon + lo + j;
        x.v[it] = *(const float4*)(values + g);
        x.r[it] = *(const float4*)(rewards + g);
        x.d[it] = *(const float4*)(dones + g);
        x.w[it] = *(const float4*)(importance + g);
      }
    }
  } else {  // one 64-step row segment per instruction, element by element
    static_assert(ADV_TT == 64, "scalar staging: one 64-step row segment per instruction");
    float* xv = (float*)x.v; float* xr = (float*)x.r; float* xd = (float*)x.d; float* xw = (float*)x.w;
#pragma unroll
    for (int i = 0; i < ADV_ROWS; i++) {
      xv[i] = xr[i] = xd[i] = xw[i] = 0.f;
      if (row0 + i < num_rows && lane < n) {
        const size_t g = (size_t)(row0 + i) * horizon + lo + lane;
        xv[i] = values[g]; xr[i] = rewards[g]; xd[i] = dones[g]; xw[i] = importance[g];
      }
    }
  }
}

template <bool VEC4>
__device__ __forceinline__ void adv_tile_to_lds(const AdvStage<VEC4>& x, float* tv, float* tr, float* td, float* tw,
                                                int lane) {
  if (VEC4) {
    constexpr int PER_ROW = AdvStage<VEC4>::PER_ROW, ROWS_PER_IT = AdvStage<VEC4>::ROWS_PER_IT;
    const int j = (lane % PER_ROW) * 4, ri = lane / PER_ROW;
#pragma unroll
    for (int it = 0; it < AdvStage<VEC4>::ITS; it++) {
      const int o = (it * ROWS_PER_IT + ri) * ADV_LD + j;
      tv[o] = x.v[it].x; tv[o + 1] = x.v[it].y; tv[o + 2] = x.v[it].z; tv[o + 3] = x.v[it].w;
      tr[o] = x.r[it].x; tr[o + 1] = x.r[it].y; tr[o + 2] = x.r[it].z; tr[o + 3] = x.r[it].w;
      td[o] = x.d[it].x; td[o + 1] = x.d[it].y; td[o + 2] = x.d[it].z; td[o + 3] = x.d[it].w;
      tw[o] = x.w[it].x; tw[o + 1] = x.w[it].y; tw[o + 2] = x.w[it].z; tw[o + 3] = x.w[it].w;
    }
  } else {
    const float* xv = (const float*)x.v; const float* xr = (const float*)x.r;
    const float* xd = (const float*)x.d; const float* xw = (const float*)x.w;
#pragma unroll
    for (int i = 0; i < ADV_ROWS; i++) {
      const int o = i * ADV_LD + lane;
      tv[o] = xv[i]; tr[o] = xr[i]; td[o] = xd[i]; tw[o] = xw[i];
    }
  }
}

// Per tile, newest first:  stage (registers -> LDS; the next tile's loads are issued right away and stay in flight)
//   -> all 64 lanes compute what does not depend on the running value: nn[t+1], delta[t], k[t] = gamma*lambda*c[t]
//   -> one lane per row runs  last = delta + k * last * nn  (the reference's operation order, pufferlib.cpp:37-38)
//   -> advantages leave coalesced.
template <bool VEC4>
__global__ __launch_bounds__(64) void mg_advantage_kernel(const float* __restrict__ values,
                                                          const float* __restrict__ rewards,
                                                          const float* __restrict__ dones,
                                                          const float* __restrict__ importance,
                                                          float* __restrict__ advantages, float gamma, float lambda,
                                                          float rho_clip, float c_clip, int num_rows, int horizon) {
  __shared__ float tv[ADV_ROWS * ADV_LD], tr[ADV_ROWS * ADV_LD], td[ADV_ROWS * ADV_LD], tw[ADV_ROWS * ADV_LD];
  __shared__ float car[ADV_ROWS * 3];  // step lo's value / reward / done of the tile above = step t+1 of this tile's top
  const int lane = threadIdx.x;
  const long row0 = (long)blockIdx.x * ADV_ROWS;
  const int ntiles = (horizon + ADV_TT - 1) / ADV_TT;
  const float gl = gamma * lambda;
  float last = 0.f;  // lanes < ADV_ROWS: the row's running advantage
  AdvStage<VEC4> x;
  {
    const int lo = (ntiles - 1) * ADV_TT;
    adv_load_tile<VEC4>(x, values, rewards, dones, importance, row0, num_rows, horizon, lo, min(ADV_TT, horizon - lo), lane);
  }
  for (int k = ntiles - 1; k >= 0; k--) {
    const int lo = k * ADV_TT;
    const int n = min(ADV_TT, horizon - lo);  // steps in this tile
    adv_tile_to_lds<VEC4>(x, tv, tr, td, tw, lane);
    if (k > 0) adv_load_tile<VEC4>(x, values, rewards, dones, importance, row0, num_rows, horizon, lo - ADV_TT, ADV_TT, lane);
    __syncthreads();
    // ---- everything but the recurrence, lane = step, rows in turn (conflict-free LDS rows) ----------------------
    float e_delta[ADV_ROWS], e_k[ADV_ROWS], e_nn[ADV_ROWS], c_v = 0.f, c_r = 0.f, c_d = 0.f;
    {
      const int j = lane;
      const bool has_out = j < n && lo + j < horizon - 1;
#pragma unroll
      for (int i = 0; i < ADV_ROWS; i++) {
        const int o = i * ADV_LD + j;
        e_delta[i] = e_k[i] = e_nn[i] = 0.f;  // the row's last step: no output, the running value stays 0
        const float v = tv[o];
        if (has_out) {
          const bool top = j == n - 1;  // step t+1 lies in the tile above
          const float nv = top ? car[i * 3 + 0] : tv[o + 1];
          const float nr = top ? car[i * 3 + 1] : tr[o + 1];
          const float nd = top ? car[i * 3 + 2] : td[o + 1];
          const float w = tw[o];
          const float nextnonterminal = (float)(1.0 - (double)nd);
          const float rho_t = fminf(w, rho_clip);
          const float c_t = fminf(w, c_clip);
          e_delta[i] = rho_t * (nr + gamma * nv * nextnonterminal - v);
          e_k[i] = gl * c_t;
          e_nn[i] = nextnonterminal;
        }
      }
      // this tile's first step is step t+1 of the tile below
      if (lane < ADV_ROWS) { c_v = tv[lane * ADV_LD]; c_r = tr[lane * ADV_LD]; c_d = td[lane * ADV_LD]; }
    }
    __syncthreads();
    {
      const int j = lane;
#pragma unroll
      for (int i = 0; i < ADV_ROWS; i++) {
        const int o = i * ADV_LD + j;
        tv[o] = e_delta[i]; tw[o] = e_k[i]; td[o] = e_nn[i];
      }
      if (lane < ADV_ROWS) { car[lane * 3 + 0] = c_v; car[lane * 3 + 1] = c_r; car[lane * 3 + 2] = c_d; }
    }
    __syncthreads();
    // ---- the recurrence: one lane per row, eight steps' operands fetched at a time ------------------------------
    if (lane < ADV_ROWS) {
      const int base = lane * ADV_LD;
      for (int j0 = ((n + 7) & ~7) - 8; j0 >= 0; j0 -= 8) {
        float dl[8], kk[8], nq[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { dl[u] = tv[base + j0 + u]; kk[u] = tw[base + j0 + u]; nq[u] = td[base + j0 + u]; }
#pragma unroll
        for (int u = 7; u >= 0; u--)
          if (j0 + u < n) {
            last = dl[u] + kk[u] * last * nq[u];
            tr[base + j0 + u] = last;
          }
      }
    }
    __syncthreads();
    // ---- advantages out, coalesced; advantages[horizon - 1] is never written (as in the reference) ---------------
    if (VEC4) {
      constexpr int PER_ROW = ADV_TT / 4, ROWS_PER_IT = 64 / PER_ROW, ITS = ADV_ROWS / ROWS_PER_IT;
      const int j = (lane % PER_ROW) * 4, ri = lane / PER_ROW;
#pragma unroll
      for (int it = 0; it < ITS; it++) {
        const int i = it * ROWS_PER_IT + ri;
        if (row0 + i < num_rows && j < n) {
          const int o = i * ADV_LD + j;
          const size_t g = (size_t)(row0 + i) * horizon + lo + j;
          if (lo + j + 3 < horizon - 1) {
            *(float4*)(advantages + g) = make_float4(tr[o], tr[o + 1], tr[o + 2], tr[o + 3]);
          } else {  // the row's last float4: three elements
            advantages[g] = tr[o]; advantages[g + 1] = tr[o + 1]; advantages[g + 2] = tr[o + 2];
          }
        }
      }
    } else {
      const int j = lane;
#pragma unroll 4
      for (int i = 0; i < ADV_ROWS; i++)
        if (row0 + i < num_rows && j < n && lo + j < horizon - 1)
          advantages[(size_t)(row0 + i) * horizon + lo + j] = tr[i * ADV_LD + j];
    }
    __syncthreads();
  }
}

extern "C" int mg_compute_advantage(const float* values, const float* rewards, const float* dones,
                                    const float* importance, float* advantages, float gamma, float lambda,
                                    float rho_clip, float c_clip, int num_rows, int horizon, void* stream) {
  if (!values || !rewards || !dones || !importance || !advantages)
    return mg_fail_msg("mg_compute_advantage: NULL tensor");
  if (num_rows < 0 || horizon < 0) return mg_fail_msg("mg_compute_advantage: negative shape");
  if (num_rows == 0 || horizon < 2) return 0;  // nothing to write (the reference's loop body never runs)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return mg_fail_msg("mg_compute_advantage: no HIP device available (this library has no CPU fallback)");
  const uintptr_t al = (uintptr_t)values | (uintptr_t)rewards | (uintptr_t)dones | (uintptr_t)importance |
                       (uintptr_t)advantages;
  const bool vec4 = (horizon % 4) == 0 && (al & 15) == 0;
  const unsigned blocks = (unsigned)((num_rows + ADV_ROWS - 1) / ADV_ROWS);
  hipStream_t s = (hipStream_t)stream;
  if (vec4)
    hipLaunchKernelGGL(mg_advantage_kernel<true>, dim3(blocks), dim3(64), 0, s, values, rewards, dones, importance,
                       advantages, gamma, lambda, rho_clip, c_clip, num_rows, horizon);
  else
    hipLaunchKernelGGL(mg_advantage_kernel<false>, dim3(blocks), dim3(64), 0, s, values, rewards, dones, importance,
                       advantages, gamma, lambda, rho_clip, c_clip, num_rows, horizon);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return mg_fail_msg(std::string("mg_compute_advantage: ") + hipGetErrorString(err));
  return 0;
}
