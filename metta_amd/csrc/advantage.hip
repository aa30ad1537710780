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
// coalesced 16-byte loads (whole 256-byte row segments), transposed through LDS (row stride ADV_TT + 1: bank
// conflict free for the lane-per-row phase), scanned by one lane per row, and the advantages leave the same way.
// Bound: HBM (20 B per element: four reads, one write); with few rows it is latency bound (one tile = one round trip).
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/mettagrid_hip.h"

#define ADV_ROWS 16
#define ADV_TT 64
#define ADV_LD (ADV_TT + 1)

extern int mg_fail_msg(const std::string& m);  // mettagrid_hip.hip: sets mg_last_error(), returns -1

template <bool VEC4>
__global__ __launch_bounds__(64) void mg_advantage_kernel(const float* __restrict__ values,
                                                          const float* __restrict__ rewards,
                                                          const float* __restrict__ dones,
                                                          const float* __restrict__ importance,
                                                          float* __restrict__ advantages, float gamma, float lambda,
                                                          float rho_clip, float c_clip, int num_rows, int horizon) {
  __shared__ float tv[ADV_ROWS * ADV_LD], tr[ADV_ROWS * ADV_LD], td[ADV_ROWS * ADV_LD], tw[ADV_ROWS * ADV_LD];
  const int lane = threadIdx.x;
  const long row0 = (long)blockIdx.x * ADV_ROWS;
  const int ntiles = (horizon + ADV_TT - 1) / ADV_TT;
  // per-row carry (lanes < ADV_ROWS): the running advantage and step t + 1's value / reward / done
  float last = 0.f, nv = 0.f, nr = 0.f, nd = 0.f;
  const bool my_row = lane < ADV_ROWS && row0 + lane < num_rows;
  for (int k = ntiles - 1; k >= 0; k--) {
    const int lo = k * ADV_TT;
    const int n = min(ADV_TT, horizon - lo);  // steps in this tile
    // ---- stage the tile: coalesced loads, transposed into LDS -------------------------------------------------
    if (VEC4) {
      constexpr int PER_ROW = ADV_TT / 4, ROWS_PER_IT = 64 / PER_ROW, ITS = ADV_ROWS / ROWS_PER_IT;
      const int j = (lane % PER_ROW) * 4, ri = lane / PER_ROW;
      float4 xv[ITS], xr[ITS], xd[ITS], xw[ITS];
#pragma unroll
      for (int it = 0; it < ITS; it++) {
        const int i = it * ROWS_PER_IT + ri;
        xv[it] = xr[it] = xd[it] = xw[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + i < num_rows && j < n) {  // horizon % 4 == 0: a float4 is inside the row or outside
          const size_t g = (size_t)(row0 + i) * horizon + lo + j;
          xv[it] = *(const float4*)(values + g);
          xr[it] = *(const float4*)(rewards + g);
          xd[it] = *(const float4*)(dones + g);
          xw[it] = *(const float4*)(importance + g);
        }
      }
#pragma unroll
      for (int it = 0; it < ITS; it++) {
        const int o = (it * ROWS_PER_IT + ri) * ADV_LD + j;
        tv[o] = xv[it].x; tv[o + 1] = xv[it].y; tv[o + 2] = xv[it].z; tv[o + 3] = xv[it].w;
        tr[o] = xr[it].x; tr[o + 1] = xr[it].y; tr[o + 2] = xr[it].z; tr[o + 3] = xr[it].w;
        td[o] = xd[it].x; td[o + 1] = xd[it].y; td[o + 2] = xd[it].z; td[o + 3] = xd[it].w;
        tw[o] = xw[it].x; tw[o + 1] = xw[it].y; tw[o + 2] = xw[it].z; tw[o + 3] = xw[it].w;
      }
    } else {
      static_assert(ADV_TT == 64, "scalar staging: one 64-step row segment per instruction");
      const int j = lane;
#pragma unroll 4
      for (int i = 0; i < ADV_ROWS; i++) {
        float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
        if (row0 + i < num_rows && j < n) {
          const size_t g = (size_t)(row0 + i) * horizon + lo + j;
          a = values[g]; b = rewards[g]; c = dones[g]; d = importance[g];
        }
        const int o = i * ADV_LD + j;
        tv[o] = a; tr[o] = b; td[o] = c; tw[o] = d;
      }
    }
    __syncthreads();
    // ---- one lane per row: the reference's loop, t = hi - 1 ... lo (pufferlib.cpp:31-40) -------------------------
    if (my_row) {
      const int base = lane * ADV_LD;
      for (int jj = n - 1; jj >= 0; jj--) {
        const float v = tv[base + jj], r = tr[base + jj], d = td[base + jj];
        if (lo + jj < horizon - 1) {
          const float w = tw[base + jj];
          const float nextnonterminal = (float)(1.0 - (double)nd);
          const float rho_t = fminf(w, rho_clip);
          const float c_t = fminf(w, c_clip);
          const float delta = rho_t * (nr + gamma * nv * nextnonterminal - v);
          last = delta + gamma * lambda * c_t * last * nextnonterminal;
          tw[base + jj] = last;  // the advantage takes the importance weight's slot
        }
        nv = v; nr = r; nd = d;
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
            *(float4*)(advantages + g) = make_float4(tw[o], tw[o + 1], tw[o + 2], tw[o + 3]);
          } else {  // the row's last float4: three elements
            advantages[g] = tw[o]; advantages[g + 1] = tw[o + 1]; advantages[g + 2] = tw[o + 2];
          }
        }
      }
    } else {
      const int j = lane;
#pragma unroll 4
      for (int i = 0; i < ADV_ROWS; i++)
        if (row0 + i < num_rows && j < n && lo + j < horizon - 1)
          advantages[(size_t)(row0 + i) * horizon + lo + j] = tw[i * ADV_LD + j];
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
