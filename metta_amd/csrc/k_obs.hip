// k_obs.hip -- see mg_kernels.h / DESIGN.md section 4
#include "mg_kernels.h"

// ------------------------------------------------------------------------------------------------------------
// kernel 2: observations + rewards + episode flags.  One workgroup (4 waves) per environment; each wave encodes
// whole agents' token rows in LDS and streams them to HBM (bindings/mettagrid_c.cpp:652-677).
// ------------------------------------------------------------------------------------------------------------
// BIG: grid tile and agent records are read in HBM (through L2); the token lists still live in LDS.
template <bool DCL, bool BIG>
// BIG environments (hundreds of agents, one workgroup per CU at best) get 1024 threads: sixteen waves encode rows.
__global__ __launch_bounds__(BIG ? MG_OBS_BIG_THREADS : MG_THREADS, BIG ? 1 : MG_OBS_WAVES) void mg_obs_kernel(KArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, nt = blockDim.x, e = blockIdx.x + a.env_base;
  Lds l;
  mg_carve_obs<true>(smem, &l, BIG ? 0 : a.HW, a.A, a.NAmax, a.ROWB, nt >> 6, DCL ? a.dc_obs_words : 0, a.NOBJ, a.HEAP,
               a.dc[DC_PATTERN_N], BIG ? 0 : a.A, a.NCmax);
  if (BIG) {
    l.grid = a.big_grid + (size_t)e * a.BHW;
    l.ag = a.agents + (size_t)e * a.A;
    l.ch = a.chests + (size_t)e * a.NCmax;
  }
  const bool slots = a.map_cfg != nullptr;  // per-slot configurations: the stream is known only after the env words
  if (DCL && !slots)
    for (int i = tid; i < (a.dc_obs_words >> 2); i += nt) ((uint4*)l.dc)[i] = ((const uint4*)a.dc)[i];  // padded to x4
  MG_STAMP(0);
  DEnv env = a.env[e];
  mg_use_map(a, env.map_id);
  if (DCL) {
    if (slots)
      for (int i = tid; i < (a.dc_obs_words >> 2); i += nt) ((uint4*)l.dc)[i] = ((const uint4*)a.gdc)[i];
    a.dc = l.dc;
  }
  const bool grid_kept = BIG && a.grid_valid && a.grid_valid[e];  // (the action kernel left the grid as it is now)
  if (!grid_kept) mg_load_grid(a, l, (int)env.map_id, tid, nt);
  {
    if (!BIG) {
      const uint4* src = (const uint4*)(a.agents + (size_t)e * a.A);
      uint4* dst = (uint4*)l.ag;
      for (int i = tid; i < a.A * (int)(sizeof(DAgent) / 16); i += nt) dst[i] = src[i];
      const uint4* s3 = (const uint4*)(a.chests + (size_t)e * a.NCmax);
      uint4* d3 = (uint4*)l.ch;
      for (int i = tid; i < a.NCmax * (int)(sizeof(DAgent) / 16); i += nt) d3[i] = s3[i];
    }
    const uint4* s2 = (const uint4*)(a.asms + (size_t)e * a.NAmax);
    uint4* d2 = (uint4*)l.as;
    for (int i = tid; i < a.NAmax; i += nt) d2[i] = s2[i];
    for (int i = tid; i < a.A; i += nt) l.exec[i] = a.exec[(size_t)e * a.A + i];
    const int4* ai_src = a.map_asm + (size_t)env.map_id * a.NAmax;
    for (int i = tid; i < a.NAmax; i += nt) l.ainfo[i] = ai_src[i];
    if (tid < 3) l.tokacc[tid] = a.gstats[(size_t)e * a.SG + tid];
    if (BIG)
      for (int i = tid; i < a.A; i += nt) l.pos[i] = *(const uint32_t*)&l.ag[i];
  }
  __syncthreads();
  MG_STAMP(1);
  if (!grid_kept) mg_overlay_agents(a, l, tid, nt);
  __syncthreads();
  MG_STAMP(2);
  if (!(a.diag & 32)) mg_build_tokens(a, l, (int)env.map_id, env.step, tid, nt);
  __syncthreads();
  MG_STAMP(3);
  if (!(a.diag & 16)) mg_encode_all<uint16_t, true, false, BIG>(a, l, e, env.step, l.exec, true, tid, nt);
  __syncthreads();
  MG_STAMP(4);
  if (a.diag & 64) return;
  // The closing phases read their pointers from the kernel-argument segment again (an opaque pointer, so that the
  // loads are not merged with the ones at kernel entry): otherwise some thirty SGPRs stay live across the encode
  // loop for their sake and are spilled there.
  KArgs b = a;
#if defined(__HIP_DEVICE_COMPILE__)
  {
    const __attribute__((address_space(4))) KArgs* ka =
        (const __attribute__((address_space(4))) KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    b = *ka;
  }
#endif
  b.dc = a.dc; b.gdc = a.gdc; b.H = a.H; b.W = a.W; b.max_steps = a.max_steps;
  if (tid == 0) {
    mg_token_stats(b, l, 1);
    float* gs_ = b.gstats + (size_t)e * b.SG;
    gs_[0] = l.tokacc[0];
    gs_[1] = l.tokacc[1];
    gs_[2] = l.tokacc[2];
  }
  __syncthreads();
  mg_rewards_and_flags(b, l, e, env, tid, nt);
}


void mg_launch_obs(const KArgs& k, unsigned n, size_t lds, bool dcl, bool big, hipStream_t st) {
  if (big) hipLaunchKernelGGL((mg_obs_kernel<false, true>), dim3(n), dim3(MG_OBS_BIG_THREADS), lds, st, k);
  else if (dcl) hipLaunchKernelGGL((mg_obs_kernel<true, false>), dim3(n), dim3(MG_THREADS), lds, st, k);
  else hipLaunchKernelGGL((mg_obs_kernel<false, false>), dim3(n), dim3(MG_THREADS), lds, st, k);
}
hipError_t mg_raise_lds_cap_obs(int cap) {
  hipError_t e = hipFuncSetAttribute((const void*)mg_obs_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)mg_obs_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)mg_obs_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
}
