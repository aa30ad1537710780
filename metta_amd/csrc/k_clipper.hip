// k_clipper.hip -- the Clipper's turn in a step (`_clipper->maybe_clip_new_assembler()`, mettagrid_c.cpp:647-650): after
// the actions, regeneration and damage of the action kernel, before the observations.  One small workgroup per
// environment on the station records in HBM; launched only for configurations with a clipper (general kernel path).
#define MG_WITH_CLIPPER
#include "mg_kernels.h"

#define MG_CLIP_THREADS 64

__global__ __launch_bounds__(MG_CLIP_THREADS) void mg_clipper_kernel(KArgs a) {
  const int tid = threadIdx.x, nt = MG_CLIP_THREADS, e = blockIdx.x + a.env_base;
  const DEnv env = a.env[e];  // (the action kernel has advanced env.step to the current step)
  const int map_id = (int)env.map_id;
  mg_use_map(a, env.map_id);
  const int* dc = a.gdc;
  if (!dc[DC_CLIPPER]) return;
  const int nasm = a.map_nasm[map_id];
  const int4* info = a.map_asm + (size_t)map_id * a.NAmax;
  DAsm* as = a.asms + (size_t)e * a.NAmax;
  uint32_t* w = a.clip_w + (size_t)e * a.NAmax;
  const uint2 st = a.clip_state[e];
  const uint32_t ls = st.y, cutoff = (uint32_t)dc[DC_CLIP_CUTOFF];
  // on_unclip_assembler (clipper.hpp:196-205) for the stations a use has unclipped during this step (assembler.hpp:
  // 519-522,583-592): take their weight off their neighbours.  Commutative: atomics, any order.
  for (int i = tid; i < nasm; i += nt) {
    if (!(w[i] >> 31) || as[i].clipped != 0u) continue;
    if (dc[DC_CLIP_PERIOD] > 0)
      for (int j = 0; j < nasm; j++) {
        if (j == i || !mg_clip_member(dc, info, j)) continue;
        const uint32_t wt = mg_clip_weight(info[i], info[j], ls, cutoff);
        if (wt) atomicSub(&w[j], wt);
      }
    atomicAnd(&w[i], 0x7FFFFFFFu);
  }
  __threadfence_block();
  __syncthreads();
  if (tid != 0) return;
  // maybe_clip_new_assembler (clipper.hpp:231-238)
  ClipRng rng{a.clip_mt + (size_t)e * 624, st.x};
  bool hit;
  const uint32_t period = (uint32_t)dc[DC_CLIP_PERIOD];
  if (period > 0) {
    hit = rng.bounded(period) == 0u;  // uniform_int_distribution<uint32_t>(1, period)(rng) == 1
  } else {
    // (1, 0): libstdc++ forms the range in 64 bits, 2^64 - 1, and composes two draws (uniform_int_dist.h:283-300):
    // 2^32 x1 + x2, truncated to 32 bits, + 1  ==  1  iff  x2 == 0
    rng.next();
    hit = rng.next() == 0u;
  }
  if (hit) {  // pick_assembler_to_clip (clipper.hpp:207-229)
    uint32_t total = 0, n_un = 0;
    for (int i = 0; i < nasm; i++)
      if (mg_clip_member(dc, info, i) && as[i].clipped == 0u) {
        n_un++;
        total += w[i] & 0x7FFFFFFFu;
      }
    int target = -1;
    if (n_un > 0) {
      if (total == 0) {
        uint32_t k = rng.bounded(n_un);
        for (int i = 0; i < nasm && target < 0; i++)
          if (mg_clip_member(dc, info, i) && as[i].clipped == 0u && k-- == 0u) target = i;
      } else {
        uint32_t rw = 1u + rng.bounded(total);  // uniform_int_distribution<>(1, total)
        for (int i = 0; i < nasm && target < 0; i++) {
          if (!mg_clip_member(dc, info, i) || as[i].clipped != 0u) continue;
          const uint32_t wi = w[i] & 0x7FFFFFFFu;
          if (wi == 0u) continue;
          if (rw <= wi) target = i;
          else rw -= wi;
        }
      }
    }
    if (target >= 0) mg_clip_assembler(dc, as, info, nasm, w, ls, rng, target, env.step);
  }
  a.clip_state[e] = make_uint2(rng.idx, ls);
}

void mg_launch_clipper(const KArgs& k, unsigned n, hipStream_t st) {
  hipLaunchKernelGGL(mg_clipper_kernel, dim3(n), dim3(MG_CLIP_THREADS), 0, st, k);
}
