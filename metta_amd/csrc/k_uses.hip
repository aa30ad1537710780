// k_uses.hip -- see mg_kernels.h / DESIGN.md section 4
#define MG_NO_CLIPPER  // (configurations with a Clipper take the general kernels)
#include "mg_kernels.h"

// ------------------------------------------------------------------------------------------------------------
// kernel 1b (fast path): assembler uses handed over by mg_act_fast_kernel, in action order, on LDS-staged
// agent/assembler records.  One wave per environment; environments without uses leave at once.  The LDS
// footprint is kept small (no map tile, no RNG window) so that every environment with work is resident at once:
// the kernel's duration is the latency of one environment's chain, not a sum of rounds.
// ------------------------------------------------------------------------------------------------------------
template <bool DCL>
__global__ __launch_bounds__(MG_USES_THREADS) void mg_uses_kernel(KArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
#ifdef MG_ISA_LITE  // ISA inspection builds (tools/isa.sh): keep only the kernel under study
  return;
#endif
  const int tid = threadIdx.x, nt = MG_USES_THREADS, e = blockIdx.x + a.env_base;
  // the mode word and the env words depend on nothing but e: one round trip for both
  const uint32_t mode = a.mode[e];
  const uint32_t step = a.env[e].step;  // already advanced by the fast kernel
  const uint32_t map_id = a.env[e].map_id;
  if ((mode & 0xFF) != 1) return;
  const int count = (int)(mode >> 8);
  Lds l;
  mg_carve_uses<true>(smem, &l, a.A, a.NAmax, a.SG, DCL ? a.dc_words : 0);
  mg_use_map(a, map_id);  // (per-slot configurations: a.gdc becomes the slot's stream)
  MG_STAMP(8);
  // every staging load is independent of the others: issue them together, one barrier
  {
    const uint4* __restrict__ s1 = (const uint4*)(a.agents + (size_t)e * a.A);
    uint4* d1 = (uint4*)l.ag;
    for (int i = tid; i < a.A * (int)(sizeof(DAgent) / 16); i += nt) d1[i] = s1[i];
    const uint4* __restrict__ s2 = (const uint4*)(a.asms + (size_t)e * a.NAmax);
    uint4* d2 = (uint4*)l.as;
    for (int i = tid; i < a.NAmax; i += nt) d2[i] = s2[i];
    const uint4* __restrict__ s3 = (const uint4*)(a.pend + (size_t)e * a.A * 32);
    uint4* d3 = (uint4*)l.rows;
    for (int i = tid; i < count * 2; i += nt) d3[i] = s3[i];
    const int4* __restrict__ s4 = a.map_asm + (size_t)map_id * a.NAmax;  // read by every use: keep it off the chain
    for (int i = tid; i < a.NAmax; i += nt) l.ainfo[i] = s4[i];
    if (DCL) {
      const uint4* __restrict__ s0 = (const uint4*)a.gdc;
      uint4* d0 = (uint4*)l.dc;
      for (int i = tid; i < (a.dc_words >> 2); i += nt) d0[i] = s0[i];  // padded to x4
    }
    for (int i = tid; i < a.SG; i += nt) l.gd[i] = 0;
    for (int i = tid; i < a.A; i += nt) {
      l.dirty_amt[i] = 0;
      l.dirty_misc[i] = 0;
    }
  }
  if (DCL) a.dc = l.dc;
  Ctx cx;
  mg_make_ctx(a, l, cx, e, (int)map_id, step);
  cx.grid = nullptr;
  cx.asm_info = l.ainfo;
  __syncthreads();
  MG_STAMP(9);
  // Uses that share neither an agent nor the assembler commute, so they run on separate lanes; conflicting ones are
  // ordered into successive rounds (round of use q = 1 + the latest round among earlier conflicting uses).
  const uint8_t* recs = l.rows;
  uint8_t* rounds = (uint8_t*)l.order;  // [A] bytes
  if (tid == 0) {
    int max_round = 0;
    for (int q = 0; q < count; q++) {
      const uint8_t* rq = recs + q * 32;
      unsigned long long mq = 0;
      for (int i = 0; i < rq[1]; i++) mq |= 1ull << rq[8 + i];
      int aq = rq[2] | (rq[3] << 8), rd = 0;
      for (int p2 = 0; p2 < q; p2++) {
        const uint8_t* rp = recs + p2 * 32;
        unsigned long long mp = 0;
        for (int i = 0; i < rp[1]; i++) mp |= 1ull << rp[8 + i];
        int ap = rp[2] | (rp[3] << 8);
        if ((mp & mq) != 0 || ap == aq) rd = max(rd, rounds[p2] + 1);
      }
      rounds[q] = (uint8_t)rd;
      max_round = max(max_round, rd);
    }
    l.flags[0] = max_round;
    l.flags[2] = 0;
  }
  __syncthreads();
  const int max_round = l.flags[0];
  for (int rd = 0; rd <= max_round; rd++) {
    for (int q0 = 0; q0 < count; q0 += 8) {  // 8 scratch slots
      int q = q0 + tid;
      if (tid < 8 && q < count && rounds[q] == rd) {
        const uint8_t* rec = recs + q * 32;
        int actor = rec[0], nn = rec[1], asm_idx = rec[2] | (rec[3] << 8);
        int act = (int)(rec[4] | (rec[5] << 8) | (rec[6] << 16) | ((uint32_t)rec[7] << 24));
        cx.tmp = l.tmp + tid * 96;
        int* near = cx.tmp;
        uint8_t* vibes = (uint8_t*)(cx.tmp + 8);
        for (int i = 0; i < 8; i++) {
          near[i] = rec[8 + i];
          vibes[i] = rec[16 + i];
        }
        bool ok = mg_assembler_use_core(cx, actor, asm_idx, near, vibes, nn);
        if (ok) {
          l.ag[actor].succ[MG_ACT_MOVE] += 1;
          a.exec[(size_t)e * a.A + actor] = act;
          a.success[(size_t)e * a.A + actor] = 1;
        } else {
          l.ag[actor].fail[MG_ACT_MOVE] += 1;
        }
        if (cx.asm_dirty) l.flags[2] = 1;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  MG_STAMP(10);
  mg_flush_stats(a, l, cx, e, tid, nt);
  uint4* wd = (uint4*)(a.agents + (size_t)e * a.A);
  const uint4* ws = (const uint4*)l.ag;
  for (int i = tid; i < a.A * (int)(sizeof(DAgent) / 16); i += nt) wd[i] = ws[i];
  if (l.flags[2]) {
    uint4* wd2 = (uint4*)(a.asms + (size_t)e * a.NAmax);
    const uint4* ws2 = (const uint4*)l.as;
    for (int i = tid; i < a.NAmax; i += nt) wd2[i] = ws2[i];
  }
  MG_STAMP(11);
}


void mg_launch_uses(const KArgs& k, unsigned n, size_t lds, bool dcl, hipStream_t st) {
  if (dcl) hipLaunchKernelGGL(mg_uses_kernel<true>, dim3(n), dim3(MG_USES_THREADS), lds, st, k);
  else hipLaunchKernelGGL(mg_uses_kernel<false>, dim3(n), dim3(MG_USES_THREADS), lds, st, k);
}
hipError_t mg_raise_lds_cap_uses(int cap) {
  hipError_t e = hipFuncSetAttribute((const void*)mg_uses_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)mg_uses_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
}
