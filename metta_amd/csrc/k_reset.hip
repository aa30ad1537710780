// k_reset.hip -- see mg_kernels.h / DESIGN.md section 4
#define MG_WITH_CLIPPER
#include "mg_kernels.h"

// Reset kernel: `MettaGrid` construction + `_init_buffers` for the listed environments.
template <bool BIG>
__global__ __launch_bounds__(MG_THREADS) void mg_reset_kernel(KArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
#ifdef MG_ISA_LITE  // ISA inspection builds (tools/isa.sh): keep only the kernel under study
  return;
#endif
  if (a.reset_count && (int)blockIdx.x >= *a.reset_count) return;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int e = a.reset_envs ? a.reset_envs[blockIdx.x] : (int)blockIdx.x;
  Lds l;
  mg_carve<true>(smem, &l, BIG ? 0 : a.HW, a.A, a.NAmax, a.SA, a.SG, a.NWIN, a.ROWB, nt >> 6, 0, a.NOBJ, a.HEAP,
           a.dc[DC_PATTERN_N], BIG ? 0 : a.A, a.NCmax);
  if (BIG) {
    l.grid = a.big_grid + (size_t)e * a.BHW;
    l.ag = a.agents + (size_t)e * a.A;
    l.ch = a.chests + (size_t)e * a.NCmax;
  }
  DEnv env = a.env[e];
  env.seed = a.reset_seeds[blockIdx.x];
  env.map_id = (uint32_t)a.reset_maps[blockIdx.x];
  env.step = 0;
  env.mt_idx = 624;
  env.done = 0;
  env.error = 0;
  Ctx cx;
  mg_use_map(a, env.map_id);
  mg_make_ctx(a, l, cx, e, (int)env.map_id, 0);
  mg_load_grid(a, l, (int)env.map_id, tid, nt);
  mg_reset_env(a, l, cx, e, (int)env.map_id, env.seed, tid, nt);
  mg_overlay_agents(a, l, tid, nt);
  {
    const int4* ai_src = a.map_asm + (size_t)env.map_id * a.NAmax;
    for (int i = tid; i < a.NAmax; i += nt) l.ainfo[i] = ai_src[i];
  }
  __syncthreads();
  if (a.build_templates && a.t_asm_tok) mg_build_asm_templates<uint16_t>(a, l, e, (int)env.map_id, tid, nt);  // (e = the map slot)
  mg_build_tokens(a, l, (int)env.map_id, 0, tid, nt);
  __syncthreads();
  mg_encode_all(a, l, e, 0, nullptr, true, tid, nt);
  __syncthreads();
  int passes = a.dc[DC_INIT_PASSES];
  if (a.clip_w && a.dc[DC_CLIPPER]) {
    // The reference builds its Clipper after the constructor's own observation pass (mettagrid_c.cpp:139-154): the
    // stations that start clipped show no protocol tokens in that pass, and the one their clipper drew in the passes
    // after it (set_buffers).
    if (tid == 0) mg_token_stats(a, l, 1);
    __syncthreads();
    mg_clipper_init(a, a.dc, l.as, e, (int)env.map_id, a.H, a.W, tid, nt);
    passes -= 1;
    if (passes > 0) {
      mg_build_tokens(a, l, (int)env.map_id, 0, tid, nt);
      __syncthreads();
      mg_encode_all(a, l, e, 0, nullptr, true, tid, nt);
      __syncthreads();
    }
  }
  if (tid == 0) {
    mg_token_stats(a, l, passes);
    float* gs_ = a.gstats + (size_t)e * a.SG;
    gs_[0] = l.tokacc[0];
    gs_[1] = l.tokacc[1];
    gs_[2] = l.tokacc[2];
    atomicOr(&a.gtouched[(size_t)e * a.GW], 7u);
  }
  __syncthreads();
  mg_flush_stats(a, l, cx, e, tid, nt);
  if (!BIG) {
    uint4* dst = (uint4*)(a.agents + (size_t)e * a.A);
    const uint4* src = (const uint4*)l.ag;
    for (int i = tid; i < a.A * (int)(sizeof(DAgent) / 16); i += nt) dst[i] = src[i];
    uint4* d3 = (uint4*)(a.chests + (size_t)e * a.NCmax);
    const uint4* s3 = (const uint4*)l.ch;
    for (int i = tid; i < a.NCmax * (int)(sizeof(DAgent) / 16); i += nt) d3[i] = s3[i];
  }
  uint4* d2 = (uint4*)(a.asms + (size_t)e * a.NAmax);
  const uint4* s2 = (const uint4*)l.as;
  for (int i = tid; i < a.NAmax; i += nt) d2[i] = s2[i];
  if (tid == 0) {
    a.env[e] = env;
    if (BIG && a.grid_valid) a.grid_valid[e] = 1;
  }
}


void mg_launch_reset(const KArgs& k, unsigned n, size_t lds, bool big, hipStream_t st) {
  if (big) hipLaunchKernelGGL(mg_reset_kernel<true>, dim3(n), dim3(MG_THREADS), lds, st, k);
  else hipLaunchKernelGGL(mg_reset_kernel<false>, dim3(n), dim3(MG_THREADS), lds, st, k);
}
hipError_t mg_raise_lds_cap_reset(int cap) {
  hipError_t e = hipFuncSetAttribute((const void*)mg_reset_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)mg_reset_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
}
