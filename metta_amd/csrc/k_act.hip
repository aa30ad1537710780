// k_act.hip -- see mg_kernels.h / DESIGN.md section 4
#include "mg_kernels.h"

// ------------------------------------------------------------------------------------------------------------
// kernel 1: actions.  One wave per environment; thread 0 resolves the actions in the reference's shuffled
// sequential order on LDS state (bindings/mettagrid_c.cpp:572-650).
// ------------------------------------------------------------------------------------------------------------
// BIG: the map does not fit LDS next to the records (256 x 256, 1024 agents): the env's grid
// lives in an HBM scratch tile (served by L2), everything else as usual.
template <bool DCL, bool BIG>
__global__ __launch_bounds__(MG_ACT_THREADS) void mg_act_kernel(KArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
#ifdef MG_ISA_LITE  // ISA inspection builds (tools/isa.sh): keep only the kernel under study
  return;
#endif
  const int tid = threadIdx.x, nt = blockDim.x, e = blockIdx.x + a.env_base;
  Lds l;
  mg_carve<true>(smem, &l, BIG ? 0 : a.HW, a.A, a.NAmax, a.SA, a.SG, a.NWIN, a.ROWB, 0, DCL ? a.dc_words : 0, 0, 0, 0, -1, a.NCmax);
  if (BIG) l.grid = a.big_grid + (size_t)e * a.HW;
  MG_STAMP(8);
  if (DCL) {
    for (int i = tid; i < (a.dc_words >> 2); i += nt) ((uint4*)l.dc)[i] = ((const uint4*)a.dc)[i];  // padded to x4
    a.dc = l.dc;
    __syncthreads();
  }
  DEnv env = a.env[e];
  const int* dc = a.dc;
  Ctx cx;

  const bool do_reset = a.auto_reset && env.done;  // block-uniform
  if (do_reset) mg_restart_env(a, e, env, tid, nt);
  mg_use_map(a, env.map_id);
  mg_load_grid(a, l, (int)env.map_id, tid, nt);
  const uint4* src = (const uint4*)(a.agents + (size_t)e * a.A);
  uint4* dst = (uint4*)l.ag;
  for (int i = tid; i < a.A * (int)(sizeof(DAgent) / 16); i += nt) dst[i] = src[i];
  const uint4* s2 = (const uint4*)(a.asms + (size_t)e * a.NAmax);
  uint4* d2 = (uint4*)l.as;
  for (int i = tid; i < a.NAmax; i += nt) d2[i] = s2[i];
  {
    const uint4* s3 = (const uint4*)(a.chests + (size_t)e * a.NCmax);
    uint4* d3 = (uint4*)l.ch;
    for (int i = tid; i < a.NCmax * (int)(sizeof(DAgent) / 16); i += nt) d3[i] = s3[i];
    if (tid == 0) *l.gsetm = 0;
  }
  for (int i = tid; i < a.SG; i += nt) l.gd[i] = 0;
  for (int i = tid; i < a.A; i += nt) {
    l.dirty_amt[i] = 0;
    l.dirty_misc[i] = 0;
  }
  __syncthreads();
  mg_overlay_agents(a, l, tid, nt);
  mg_stage_rng(a, l, e, env.mt_idx, tid, nt);
  env.step += 1;
  mg_make_ctx(a, l, cx, e, (int)env.map_id, env.step);
  __syncthreads();

  if (tid == 0) {
    RngWin rng{l.win, a.NWIN, 0, 0};
    mg_shuffle(l.order, a.A, rng);
    const int n_actions = dc[DC_N_ACTIONS], oa = dc[DC_OFF_ACTIONS];
    const int passes = dc[DC_MAX_PRIORITY] + 1;
    const int* acts = a.actions + (size_t)e * a.A;
    for (int i = 0; i < a.A; i++) l.exec[i] = 0;
    for (int k = 0; k < a.A; k++) {
      int ai = l.order[k];
      int act = acts[ai];
      bool ok = false;
      if (act < 0 || act >= n_actions) {
        // _handle_invalid_action runs once per priority level (the check precedes the priority filter)
        mg_stat_add(cx, ai, cx.AS.invalid(), passes);
        int variant = act == -1 ? 0 : (act >= n_actions && act < n_actions + 7) ? 1 + (act - n_actions) : 8;
        mg_stat_add(cx, ai, cx.AS.invalid_variant(variant), passes);
      } else {
        ok = mg_handle_action(cx, ai, dc[oa + 2 * act], dc[oa + 2 * act + 1]);
        if (ok) l.exec[ai] = act;
      }
      a.success[(size_t)e * a.A + ai] = ok ? 1 : 0;
    }
    // inventory regeneration (bindings/mettagrid_c.cpp:625-640)
    int ri = dc[DC_REGEN_INTERVAL];
    if (ri > 0 && env.step % (uint32_t)ri == 0) {
      for (int ai = 0; ai < a.A; ai++) {
        const int* gr = group_rec(dc, l.ag[ai].group);
        int orr = gr[GR_OFF_REGEN], nr = dc[orr];
        if (nr == 0) continue;
        int lst = -1;
        for (int q = 0; q < nr; q++)
          if (dc[orr + 1 + 2 * q] == l.ag[ai].vibe) lst = dc[orr + 2 + 2 * q];
        if (lst < 0)
          for (int q = 0; q < nr; q++)
            if (dc[orr + 1 + 2 * q] == 0) lst = dc[orr + 2 + 2 * q];
        if (lst >= 0) {
          int n = L_n(dc, lst);
          for (int q = 0; q < n; q++) inv_update(cx, ai, KV_k(dc, lst, q), KV_v(dc, lst, q));
        }
      }
    }
    if (dc[DC_HAS_DAMAGE])  // bindings/mettagrid_c.cpp:642-650: after regeneration, agent order
      for (int ai = 0; ai < a.A; ai++) mg_check_damage(cx, ai, rng);
    l.flags[0] = rng.used;
    l.flags[1] = rng.overflow | (cx.overflow ? 8 : 0);
    l.flags[2] = cx.asm_dirty;
  }
  __syncthreads();
  for (int i = tid; i < a.A; i += nt) a.exec[(size_t)e * a.A + i] = l.exec[i];
  mg_flush_stats(a, l, cx, e, tid, nt);
  {
    uint4* dst = (uint4*)(a.agents + (size_t)e * a.A);
    const uint4* src = (const uint4*)l.ag;
    for (int i = tid; i < a.A * (int)(sizeof(DAgent) / 16); i += nt) dst[i] = src[i];
    if (l.flags[2] || do_reset) {
      uint4* d2 = (uint4*)(a.asms + (size_t)e * a.NAmax);
      const uint4* s2 = (const uint4*)l.as;
      for (int i = tid; i < a.NAmax; i += nt) d2[i] = s2[i];
    }
    {
      uint4* d3 = (uint4*)(a.chests + (size_t)e * a.NCmax);
      const uint4* s3 = (const uint4*)l.ch;
      for (int i = tid; i < a.NCmax * (int)(sizeof(DAgent) / 16); i += nt) d3[i] = s3[i];
    }
    uint32_t new_idx = env.mt_idx + (uint32_t)l.flags[0];
    if (new_idx > 624) {  // == 624: the block is exhausted but untwisted; the next step twists it
      uint32_t* mt = a.mt + (size_t)e * 624;
      for (int i = tid; i < 624; i += nt) mt[i] = l.mtnext[i];
      new_idx -= 624;
    }
    if (tid == 0) {
      env.mt_idx = new_idx;
      if (l.flags[1] & 1) env.error |= 1u;
      if (l.flags[1] & 8) env.error |= 8u;
      a.env[e] = env;
      if (env.error) atomicOr(&a.status[0], env.error);
    }
  }
}


void mg_launch_act(const KArgs& k, unsigned n, size_t lds, bool dcl, bool big, hipStream_t st) {
  if (big) hipLaunchKernelGGL((mg_act_kernel<false, true>), dim3(n), dim3(MG_ACT_THREADS), lds, st, k);
  else if (dcl) hipLaunchKernelGGL((mg_act_kernel<true, false>), dim3(n), dim3(MG_ACT_THREADS), lds, st, k);
  else hipLaunchKernelGGL((mg_act_kernel<false, false>), dim3(n), dim3(MG_ACT_THREADS), lds, st, k);
}
hipError_t mg_raise_lds_cap_act(int cap) {
  hipError_t e = hipFuncSetAttribute((const void*)mg_act_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)mg_act_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)mg_act_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
}
