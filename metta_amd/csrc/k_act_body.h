// k_act_body.h -- general action kernel (attack / transfer / swaps / chests / regeneration / damage, any number of agents).
//
// Reference: the shuffled sequential loop of MettaGrid::_step (bindings/mettagrid_c.cpp:588-650): agents act one
// after the other in std::shuffle order and interact through the grid, through each other's inventories (attack,
// loot, transfer, swap) and through assemblers and chests.  Here one workgroup (up to 1024 threads, thread = agent)
// owns an environment and resolves the same sequence in parallel rounds:
//
//   * order      std::shuffle(iota(A)) from the same MT19937 draws: the Lemire draws and both swap partners of every
//                draw are computed lane-parallel, then every thread traces ONE final position back through the swap
//                list (a rejected draw -- about once in 1e6 steps -- takes the serial form);
//   * footprint  the cells an action can read or write: the agent's cell, the target cell of a move, and for a move
//                into an assembler its whole ring plus the chest search square.  Everything an action touches hangs
//                off those cells (the agents standing on them, the assembler / chest in them);
//   * rounds     every pending agent claims its footprint cells with an atomic max of (round, -rank) on a hashed LDS
//                table; an agent whose claims all survived has no earlier pending agent on any of its cells, so its
//                action commutes with everything still pending and is executed now.  Hash collisions only add
//                conflicts (extra rounds), never remove one.  The lowest pending rank always wins, so at most A rounds;
//                sparse maps need two or three.
//   * after      inventory regeneration (per agent), damage (two passes: thresholds, then the discrete_distribution
//                draws at RNG offsets given by a prefix sum over the agents that need one), stat flush, write-back.
//
// BIG (map tile + records do not fit LDS: 256 x 256 x 1024 agents): grid tile and agent records are used in place in
// HBM (L2); LDS holds the claim table, order / rank, the RNG window and the per-use scratch pool.
#define MG_WITH_CLIPPER  // (mg_restart_env rebuilds the environment's clipper)
#include "mg_kernels.h"

#ifndef MG_ACT_MAX_THREADS
#define MG_ACT_MAX_THREADS 1024
#endif
#define MG_ACT_TMP_SLOTS 32   // assembler uses executed concurrently in one round (scratch slots)
#define MG_ACT_TMP_INTS 128   // 96 + 32 input holders (mg_assembler_use)

struct ActLds {
  uint32_t* claim;  // [CL] (round << 16) | (0xFFFF - rank) of the best claimant of the cell that owns the slot
  uint32_t* ckey;   // [CL] cell + 1 of the slot's owner (open addressing, linear probing), 0 = free
  int cl_shift;     // 32 - log2(CL)
  uint16_t* rank;   // [A] position of agent i in the shuffled order
  uint8_t* pendf;   // [A] 1 while the agent's action is pending
  uint32_t* swaps;  // [D] swap partners of draw t: a | b << 16
  int* tmp_pool;    // [MG_ACT_TMP_SLOTS][MG_ACT_TMP_INTS]
  int* wsum;        // [64] scan scratch
};

__host__ __device__ inline int mg_act_claims(int A) {
  int cl = 256;
  while (cl < 8 * A && cl < 8192) cl <<= 1;
  return cl;
}

template <bool FILL>
__host__ __device__ inline size_t mg_carve_act(unsigned char* base, Lds* l, ActLds* x, int HW, int a_rec, int A, int NA, int NC,
                                               int SG, int NWIN, int dcw) {
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = mg_align16(o + bytes);
    return at;
  };
  const size_t na = (size_t)(NA > 0 ? NA : 1);
  size_t dco = take((size_t)dcw * 4);
  size_t g = take((size_t)HW * 2), ag = take((size_t)(a_rec + (a_rec ? NC : 0)) * sizeof(DAgent)), as = take(na * sizeof(DAsm));
  size_t gd = take((size_t)SG * 4), da = take((size_t)A * 4), dm = take((size_t)A * 4);
  size_t ord = take((size_t)A * 2), rk = take((size_t)A * 2), ex = take((size_t)A * 4), pf = take((size_t)A);
  size_t win = take((size_t)NWIN * 4), mtn = take(624 * 4), fl = take(64), gs = take(MGD_MAX_R * 8);
  size_t sw = take((size_t)(A / 2 + 2) * 4), ws = take(64 * 4);
  const int CL = mg_act_claims(A);
  size_t cl = take((size_t)CL * 4), tp = take((size_t)MG_ACT_TMP_SLOTS * MG_ACT_TMP_INTS * 4);
  size_t ck = take((size_t)CL * 4);
  if (FILL) {
    l->dc = (int*)(base + dco);
    l->grid = (uint16_t*)(base + g); l->ag = (DAgent*)(base + ag); l->ch = l->ag + a_rec; l->as = (DAsm*)(base + as);
    l->gd = (int*)(base + gd); l->dirty_amt = (uint32_t*)(base + da); l->dirty_misc = (uint32_t*)(base + dm);
    l->order = (uint16_t*)(base + ord); l->exec = (int*)(base + ex);
    l->win = (uint32_t*)(base + win); l->mtnext = (uint32_t*)(base + mtn); l->flags = (int*)(base + fl);
    l->gset64 = (unsigned long long*)(base + gs);
    l->gsetv = nullptr; l->gsetm = nullptr; l->tmp = nullptr;
    x->rank = (uint16_t*)(base + rk); x->pendf = base + pf; x->swaps = (uint32_t*)(base + sw); x->wsum = (int*)(base + ws);
    x->claim = (uint32_t*)(base + cl); x->tmp_pool = (int*)(base + tp); x->ckey = (uint32_t*)(base + ck);
    int lg = 0;
    while ((1 << lg) < CL) lg++;
    x->cl_shift = 32 - lg;
  }
  return o;
}

// The claim slot of a cell.  Slots used to be shared by all cells hashing to them: at 1024 agents a fifth of the claims
// met a stranger's higher claim, the agent waited a round for nothing, and a step took four or five rounds of three
// barriers each where the true conflicts need two.  Now a cell owns its slot (the key array; keys stay for the whole
// step, the round number in the value makes old claims lose).  After MG_CLAIM_PROBES occupied slots the cell shares its
// home slot with that slot's owner -- every claimant of the cell sees the same occupied slots and lands there too, so
// real conflicts are never missed, only false ones added.
#define MG_CLAIM_PROBES 16
__device__ __forceinline__ uint32_t mg_claim_slot(const ActLds& x, const int cell) {
  const uint32_t key = (uint32_t)cell + 1u, mask = (1u << (32 - x.cl_shift)) - 1u;
  const uint32_t home = ((uint32_t)cell * 2654435761u) >> x.cl_shift;
  uint32_t h = home;
  for (int i = 0; i < MG_CLAIM_PROBES; i++, h = (h + 1u) & mask) {
    uint32_t k = x.ckey[h];
    if (k == 0u) k = atomicCAS(&x.ckey[h], 0u, key);
    if (k == 0u || k == key) return h;
  }
  return home;
}

// The cells agent `ai`'s action can read or write, as the state stands (every cell is in the map).  Returns whether the
// action may run an assembler use (which needs a scratch slot).
template <class F>
__device__ __forceinline__ bool mg_footprint(const Ctx& cx, const int ai, const int act, const int n_actions, const int oa,
                                             F&& f) {
  const DAgent* me = &cx.ag[ai];
  const int r = me->r, c = me->c;
  f(r * cx.W + c);
  if (act < 0 || act >= n_actions) return false;
  const int kind = cx.dc[oa + 2 * act], arg = cx.dc[oa + 2 * act + 1];
  if (kind != MG_ACT_MOVE || me->frozen != 0) return false;  // (only agents that claim my cell can change `frozen`)
  const int dx = (arg == 2 || arg == 4 || arg == 6) ? -1 : ((arg == 3 || arg == 5 || arg == 7) ? 1 : 0);
  const int dy = (arg == 0 || arg == 4 || arg == 5) ? -1 : ((arg == 1 || arg == 6 || arg == 7) ? 1 : 0);
  const int tr = r + dy, tc = c + dx;
  if (tr < 0 || tc < 0 || tr >= cx.H || tc >= cx.W) return false;
  f(tr * cx.W + tc);
  const uint16_t target = cx.grid[tr * cx.GW + tc];
  if (cx.cc.kind(target) != MG_CELL_ASSEMBLER) return false;  // static objects never move: safe to read at any time
  // Assembler::onUse reads and writes the agents on the ring and the chests inside the search square (assembler.hpp:
  // 48-122)
  int d = 1;
  if (cx.NC > 0) {
    const int dist = asm_rec(cx.dc, cx.asm_info[cx.cc.asm_idx(target)].z)[AR_CHEST_DIST];
    if (dist > d) d = dist;
  }
  for (int rr = tr - d; rr <= tr + d; rr++)
    for (int c2 = tc - d; c2 <= tc + d; c2++)
      if (rr >= 0 && c2 >= 0 && rr < cx.H && c2 < cx.W) f(rr * cx.W + c2);
  return true;
}

// Exclusive prefix sum of one 0/1 flag per thread over the workgroup, in thread order.  `*total` = sum over all threads.
__device__ __forceinline__ int mg_block_scan_flag(const bool flag, int* wsum, const int tid, const int nt, int* total) {
  const unsigned long long b = __ballot(flag);
  const int lane = tid & 63, wave = tid >> 6, nwaves = (nt + 63) >> 6;
  const int within = __popcll(b & ((1ull << lane) - 1ull));
  __syncthreads();  // (wsum may still be read from the previous call)
  if (lane == 0) wsum[wave] = __popcll(b);
  __syncthreads();
  int before = 0, all = 0;
  for (int w = 0; w < nwaves; w++) {
    const int v = wsum[w];
    if (w < wave) before += v;
    all += v;
  }
  *total = all;
  return before + within;
}

template <bool DCL, bool BIG>
__global__ __launch_bounds__(MG_ACT_MAX_THREADS) void mg_act_kernel(KArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, nt = blockDim.x, e = blockIdx.x + a.env_base;
  Lds l;
  ActLds x;
  mg_carve_act<true>(smem, &l, &x, BIG ? 0 : a.HW, BIG ? 0 : a.A, a.A, a.NAmax, a.NCmax, a.SG, a.NWIN, DCL ? a.dc_words : 0);
  if (BIG) {
    l.grid = a.big_grid + (size_t)e * a.BHW;
    l.ag = a.agents + (size_t)e * a.A;
    l.ch = a.chests + (size_t)e * a.NCmax;
  }
  const bool slots = a.map_cfg != nullptr;  // per-slot configurations: the stream is known only after the env words
  if (DCL && !slots)
    for (int i = tid; i < (a.dc_words >> 2); i += nt) ((uint4*)l.dc)[i] = ((const uint4*)a.dc)[i];  // padded to x4
  DEnv env = a.env[e];
  const int A = a.A;
  MG_STAMP(7);

  const bool do_reset = a.auto_reset && env.done;  // block-uniform
  if (do_reset) mg_restart_env(a, e, env, tid, nt);
  mg_use_map(a, env.map_id);
  if (DCL) {
    if (slots)
      for (int i = tid; i < (a.dc_words >> 2); i += nt) ((uint4*)l.dc)[i] = ((const uint4*)a.gdc)[i];
    a.dc = l.dc;
  }
  const int* dc = a.dc;
  // BIG: the environment's grid in HBM is carried from step to step (moves and swaps update it in place); only a new
  // episode copies its map and overlays the agents
  const bool grid_kept = BIG && a.grid_valid && a.grid_valid[e] && !do_reset;  // block-uniform
  if (!grid_kept) mg_load_grid(a, l, (int)env.map_id, tid, nt);
  if (!BIG) {
    const uint4* src = (const uint4*)(a.agents + (size_t)e * A);
    uint4* dst = (uint4*)l.ag;
    for (int i = tid; i < A * (int)(sizeof(DAgent) / 16); i += nt) dst[i] = src[i];
    const uint4* s3 = (const uint4*)(a.chests + (size_t)e * a.NCmax);
    uint4* d3 = (uint4*)l.ch;
    for (int i = tid; i < a.NCmax * (int)(sizeof(DAgent) / 16); i += nt) d3[i] = s3[i];
  }
  {
    const uint4* s2 = (const uint4*)(a.asms + (size_t)e * a.NAmax);
    uint4* d2 = (uint4*)l.as;
    for (int i = tid; i < a.NAmax; i += nt) d2[i] = s2[i];
  }
  for (int i = tid; i < a.SG; i += nt) l.gd[i] = 0;
  for (int i = tid; i < A; i += nt) {
    l.dirty_amt[i] = 0;
    l.dirty_misc[i] = 0;
    l.exec[i] = 0;
    x.pendf[i] = 1;
  }
  if (tid < MGD_MAX_R) l.gset64[tid] = 0ull;
  for (int i = tid; i < mg_act_claims(A); i += nt) {
    x.claim[i] = 0u;
    x.ckey[i] = 0u;
  }
  if (tid < 16) l.flags[tid] = 0;
  __syncthreads();
  MG_STAMP(8);
  if (!grid_kept) mg_overlay_agents(a, l, tid, nt);
  if (BIG && a.grid_valid && tid == 0) a.grid_valid[e] = 1;
  mg_stage_rng(a, l, e, env.mt_idx, tid, nt);
  env.step += 1;
  Ctx cx;
  mg_make_ctx(a, l, cx, e, (int)env.map_id, env.step);
  cx.gset64 = l.gset64;
  __syncthreads();
  MG_STAMP(9);
  if (do_reset && a.clip_w && dc[DC_CLIPPER] && dc[DC_PROTO_OBS] && dc[DC_INIT_PASSES] > 1 && mg_act_claims(A) >= 2 * A + 4 &&
      !(a.diag & 1048576))  // (1048576: the template's figures, to show that the recount is what the test sees)
    mg_recount_initial_tokens(a, l, e, (int)env.map_id, (int*)x.claim, tid, nt);  // (the claim table is idle until the rounds)

  // ---- std::shuffle(iota(A), rng) (stl_algo.h:3726-3792) ---------------------------------------
  const int even = (A & 1) == 0 ? 1 : 0;
  const int D = A <= 1 ? 0 : even + (A - (even ? 2 : 1)) / 2;  // draws: U(2) first when A is even, then pairs
  {
    // One draw per swap step, each from the RNG output at its own index -- unless an earlier draw was rejected (Lemire,
    // uniform_int_dist.h:258-269) and took a second output: every later draw then reads one output further on.  At 1024
    // agents 4 % of the environment-steps hold a rejection, so nearly every launch of 256 environments had a few
    // workgroups on the serial redraw below while the others waited.  Instead: draw, count the rejections before each
    // step (block scan), draw again where that count changed; settled after (rejections + 1) passes.  A draw rejected
    // twice in a row, a window that ends, or more steps than threads fall back to the serial form.
    bool reject = false;
    int n_rej = 0;
    if (D <= nt && !(a.diag & 8)) {
      int shift = 0;
      for (int it = 0;; it++) {
        bool rej = false, hard = false;
        if (tid < D) {
          const int t = tid;
          uint32_t range = 2, s1 = 0;
          if (!(even && t == 0)) {
            const uint32_t i = (uint32_t)((even ? 2 : 1) + 2 * (t - even));
            range = (i + 1) * (i + 2);
            s1 = i + 2;
          }
          const uint32_t thr = (0u - range) % range;
          const int k = t + shift;
          unsigned long long product = 0;
          if (k + 1 >= a.NWIN) {
            hard = true;
          } else {
            product = (unsigned long long)l.win[k] * range;
            if ((uint32_t)product < range && (uint32_t)product < thr) {
              rej = true;
              product = (unsigned long long)l.win[k + 1] * range;
              if ((uint32_t)product < range && (uint32_t)product < thr) hard = true;
            }
          }
          const uint32_t val = (uint32_t)(product >> 32);
          x.swaps[t] = s1 ? ((val / s1) | ((val % s1) << 16)) : val;
        }
        int total = 0;
        const int before = mg_block_scan_flag(rej, x.wsum, tid, nt, &total);
        const int state = hard ? 2 : ((tid < D && before != shift) ? 1 : 0);
        shift = before;
        n_rej = total;
        const int any = __syncthreads_or(state);  // bit 0: someone draws again; bit 1: the serial form
        if ((any & 2) || (any && it >= 8)) {
          reject = true;
          break;
        }
        if (!any) break;
      }
    } else {
      for (int t = tid; t < D; t += nt) {
        uint32_t range = 2, s1 = 0;
        if (!(even && t == 0)) {
          const uint32_t i = (uint32_t)((even ? 2 : 1) + 2 * (t - even));
          range = (i + 1) * (i + 2);
          s1 = i + 2;
        }
        const unsigned long long product = (unsigned long long)(t < a.NWIN ? l.win[t] : 0u) * range;
        const uint32_t low = (uint32_t)product, val = (uint32_t)(product >> 32);
        if (low < range && low < (0u - range) % range) reject = true;  // Lemire rejection: the serial form redraws
        x.swaps[t] = s1 ? ((val / s1) | ((val % s1) << 16)) : val;
      }
    }
    const int any_reject = __syncthreads_or((reject || (a.diag & 8)) ? 1 : 0);
    if (any_reject) {
      if (tid == 0) {
        RngWin rng{l.win, a.NWIN, 0, 0};
        mg_shuffle(l.order, A, rng);
        l.flags[0] = rng.used;
        l.flags[1] = rng.overflow;
      }
      __syncthreads();
      for (int p = tid; p < A; p += nt) x.rank[l.order[p]] = (uint16_t)p;
    } else if (A >= 128 && (mg_act_claims(A) * 2) / A >= 2) {
      // The element that ends at position p is p walked back through all D swap steps -- D/2 dependent steps per
      // agent, A^2/4 in total: 38 % of this kernel at 1024 agents.  In blocks instead: B threads each run S = D/B
      // steps of the forward shuffle on an iota array of their own (M_b = what block b alone does to the positions),
      // then every agent walks back through the B arrays: p -> M_{B-1}[p] -> ... -> M_0[.].  The arrays borrow the
      // claim table (unused until the action rounds, cleared again below).
      uint16_t* M = (uint16_t*)x.claim;
      int B = (mg_act_claims(A) * 2) / A;  // u16 entries the table holds / A
      if (B > 32) B = 32;
      if (B > (nt >> 6)) B = nt >> 6;  // one block per wave: the serial parts run side by side on different SIMDs
      const int S = (D + B - 1) / B;
      for (int b = 0; b < B; b++)
        for (int i = tid; i < A; i += nt) M[b * A + i] = (uint16_t)i;
      __syncthreads();
      if ((tid & 63) == 0 && (tid >> 6) < B) {
        const int b = tid >> 6;
        uint16_t* m = M + b * A;
        const int t1 = (b + 1) * S < D ? (b + 1) * S : D;
        for (int t = b * S; t < t1; t++) {
          const uint32_t sw = x.swaps[t];
          if (even && t == 0) {
            const int j = (int)sw;  // swap(v[1], v[j])
            const uint16_t u = m[1], w = m[j];
            m[1] = w;
            m[j] = u;
          } else {
            const int i = (even ? 2 : 1) + 2 * (t - even), pa = (int)(sw & 0xFFFF), pb = (int)(sw >> 16);
            uint16_t u = m[i], w = m[pa];  // swap(v[i], v[pa]) ...
            m[i] = w;
            m[pa] = u;
            u = m[i + 1];  // ... then swap(v[i+1], v[pb])
            w = m[pb];
            m[i + 1] = w;
            m[pb] = u;
          }
        }
      }
      __syncthreads();
      for (int p = tid; p < A; p += nt) {
        int q = p;
        for (int b = B - 1; b >= 0; b--) q = M[b * A + q];
        l.order[p] = (uint16_t)q;
        x.rank[q] = (uint16_t)p;
      }
      if (tid == 0) {
        l.flags[0] = D + n_rej;
        l.flags[1] = 0;
      }
      __syncthreads();
      for (int i = tid; i < (B * A + 1) / 2; i += nt) x.claim[i] = 0u;
    } else {
      // the element that ends at position p: walk p back through the swaps (the array starts as iota)
      for (int p = tid; p < A; p += nt) {
        int q = p;
        for (int t = D - 1; t >= 0; t--) {
          const uint32_t sw = x.swaps[t];
          if (even && t == 0) {
            const int j = (int)sw;  // swap(v[1], v[j])
            q = q == 1 ? j : (q == j ? 1 : q);
          } else {
            const int i = (even ? 2 : 1) + 2 * (t - even), pa = (int)(sw & 0xFFFF), pb = (int)(sw >> 16);
            q = q == i + 1 ? pb : (q == pb ? i + 1 : q);  // undo swap(v[i+1], v[pb]) ...
            q = q == i ? pa : (q == pa ? i : q);          // ... then swap(v[i], v[pa])
          }
        }
        l.order[p] = (uint16_t)q;
        x.rank[q] = (uint16_t)p;
      }
      if (tid == 0) {
        l.flags[0] = D + n_rej;
        l.flags[1] = 0;
      }
    }
  }
  __syncthreads();
  MG_STAMP(10);

  // ---- the action loop, in parallel rounds (mettagrid_c.cpp:599-622) ---------------------------------------------------
  const int n_actions = dc[DC_N_ACTIONS], oa = dc[DC_OFF_ACTIONS];
  const int passes = dc[DC_MAX_PRIORITY] + 1;
  const int* acts = a.actions + (size_t)e * A;
  int my_overflow = 0, my_asm_dirty = 0;
  for (uint32_t round = 1;; round++) {
    for (int ai = tid; ai < A; ai += nt) {
      if (!x.pendf[ai]) continue;
      const uint32_t val = (round << 16) | (0xFFFFu - (uint32_t)x.rank[ai]);
      mg_footprint(cx, ai, acts[ai], n_actions, oa,
                   [&](int cell) { atomicMax(&x.claim[mg_claim_slot(x, cell)], val); });
    }
    if (tid == 0) l.flags[4] = 0;  // scratch slots handed out this round
    __syncthreads();
    uint32_t ready = 0, needs_tmp = 0;  // bit k: my k-th agent (A <= 8191, nt >= 256 for A > 256: at most 32 per thread)
    {
      int k = 0;
      for (int ai = tid; ai < A; ai += nt, k++) {
        if (!x.pendf[ai]) continue;
        const uint32_t val = (round << 16) | (0xFFFFu - (uint32_t)x.rank[ai]);
        bool ok = true;
        const bool nt_ = mg_footprint(cx, ai, acts[ai], n_actions, oa, [&](int cell) {
          ok = ok && x.claim[mg_claim_slot(x, cell)] == val;
        });
        if (ok) ready |= 1u << k;
        if (nt_) needs_tmp |= 1u << k;
      }
    }
    __syncthreads();
    int still = 0;
    {
      int k = 0;
      for (int ai = tid; ai < A; ai += nt, k++) {
        if (!x.pendf[ai]) continue;
        if (!((ready >> k) & 1u)) {
          still = 1;
          continue;
        }
        if ((needs_tmp >> k) & 1u) {
          const int slot = atomicAdd(&l.flags[4], 1);
          if (slot >= MG_ACT_TMP_SLOTS) {  // no scratch slot left this round: stays pending (and ready) for the next
            still = 1;
            continue;
          }
          cx.tmp = x.tmp_pool + slot * MG_ACT_TMP_INTS;
        }
        cx.seq = ((uint32_t)x.rank[ai] + 1u) << 8;
        const int act = acts[ai];
        bool ok = false;
        if (act < 0 || act >= n_actions) {
          // _handle_invalid_action runs once per priority level (the check precedes the priority filter)
          // (only the agent's own turn touches these slots)
          mg_count_invalid(cx.astats + (size_t)ai * cx.SA, cx.AS, act, n_actions, passes);
        } else {
          ok = mg_handle_action(cx, ai, dc[oa + 2 * act], dc[oa + 2 * act + 1]);
          if (ok) l.exec[ai] = act;
        }
        a.success[(size_t)e * A + ai] = ok ? 1 : 0;
        x.pendf[ai] = 0;
      }
    }
    if (round == 1) MG_STAMP(15);
    if (__syncthreads_or(still) == 0) {
      if (a.dbg && tid == 0) a.dbg[((size_t)blockIdx.x + a.env_base) * 16 + 14] = round;
      break;
    }
  }
  MG_STAMP(11);
  my_overflow |= cx.overflow;
  my_asm_dirty |= cx.asm_dirty;

  // ---- inventory regeneration (mettagrid_c.cpp:625-640): per agent ------------------------------------------------------
  {
    const int ri = dc[DC_REGEN_INTERVAL];
    if (ri > 0 && env.step % (uint32_t)ri == 0) {
      for (int ai = tid; ai < A; ai += nt) {
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
  }
  // ---- damage (mettagrid_c.cpp:642-650; agent.cpp:134-177): agent order decides only which RNG outputs an agent gets ----
  int rng_used = l.flags[0];
  if (dc[DC_HAS_DAMAGE]) {
    __syncthreads();
    int base = rng_used;
    for (int a0 = 0; a0 < A; a0 += nt) {  // chunks of nt consecutive agents keep the agent order
      const int ai = a0 + tid;
      int n_avail = -1;
      if (ai < A) n_avail = mg_damage_pay(cx, ai);
      int total = 0;
      const int before = mg_block_scan_flag(n_avail >= 2, x.wsum, tid, nt, &total);
      if (ai < A && n_avail >= 0) {
        RngWin rng{l.win, a.NWIN, base + 2 * before, 0};
        mg_damage_apply(cx, ai, n_avail, rng);
        if (rng.overflow) atomicOr(&l.flags[1], 1);
      }
      base += 2 * total;
    }
    rng_used = base;
  }
  if (my_overflow) atomicOr(&l.flags[1], 8);
  if (my_asm_dirty) atomicOr(&l.flags[2], 1);
  __syncthreads();
  MG_STAMP(12);

  for (int i = tid; i < A; i += nt) a.exec[(size_t)e * A + i] = l.exec[i];
  mg_flush_stats(a, l, cx, e, tid, nt);
  {
    if (!BIG) {
      uint4* dst = (uint4*)(a.agents + (size_t)e * A);
      const uint4* src = (const uint4*)l.ag;
      for (int i = tid; i < A * (int)(sizeof(DAgent) / 16); i += nt) dst[i] = src[i];
      uint4* d3 = (uint4*)(a.chests + (size_t)e * a.NCmax);
      const uint4* s3 = (const uint4*)l.ch;
      for (int i = tid; i < a.NCmax * (int)(sizeof(DAgent) / 16); i += nt) d3[i] = s3[i];
    }
    if (l.flags[2] || do_reset) {
      uint4* d2 = (uint4*)(a.asms + (size_t)e * a.NAmax);
      const uint4* s2 = (const uint4*)l.as;
      for (int i = tid; i < a.NAmax; i += nt) d2[i] = s2[i];
    }
    uint32_t new_idx = env.mt_idx + (uint32_t)rng_used;
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
  MG_STAMP(13);
}

// ---- one kernel instantiation per translation unit (k_act_dcl.hip / k_act_nodcl.hip / k_act_big.hip define
// MG_ACT_VARIANT and include this file): the three of them compile side by side instead of one after the other ----
static inline int mg_act_threads_i(const KArgs& k) {
  int t = (k.A + 63) & ~63;
  return t < 256 ? 256 : (t > MG_ACT_MAX_THREADS ? MG_ACT_MAX_THREADS : t);
}
#if MG_ACT_VARIANT == 0
#define MG_ACT_KERNEL mg_act_kernel<true, false>
#define MG_ACT_NAME(x) x##_dcl
#elif MG_ACT_VARIANT == 1
#define MG_ACT_KERNEL mg_act_kernel<false, false>
#define MG_ACT_NAME(x) x##_nodcl
#else
#define MG_ACT_KERNEL mg_act_kernel<false, true>
#define MG_ACT_NAME(x) x##_big
#endif
void MG_ACT_NAME(mg_launch_act)(const KArgs& k, unsigned n, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((MG_ACT_KERNEL), dim3(n), dim3((unsigned)mg_act_threads_i(k)), lds, st, k);
}
hipError_t MG_ACT_NAME(mg_raise_lds_cap_act)(int cap) {
  return hipFuncSetAttribute((const void*)MG_ACT_KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
}
#if MG_ACT_VARIANT == 0
void mg_launch_act_nodcl(const KArgs& k, unsigned n, size_t lds, hipStream_t st);
void mg_launch_act_big(const KArgs& k, unsigned n, size_t lds, hipStream_t st);
hipError_t mg_raise_lds_cap_act_nodcl(int cap);
hipError_t mg_raise_lds_cap_act_big(int cap);
int mg_act_threads(const KArgs& k) { return mg_act_threads_i(k); }
size_t mg_act_lds_bytes(const KArgs& k, bool dcl, bool big) {
  return mg_carve_act<false>(nullptr, nullptr, nullptr, big ? 0 : k.HW, big ? 0 : k.A, k.A, k.NAmax, k.NCmax, k.SG, k.NWIN,
                             dcl ? k.dc_words : 0);
}
void mg_launch_act(const KArgs& k, unsigned n, size_t lds, bool dcl, bool big, hipStream_t st) {
  if (big) mg_launch_act_big(k, n, lds, st);
  else if (dcl) mg_launch_act_dcl(k, n, lds, st);
  else mg_launch_act_nodcl(k, n, lds, st);
}
hipError_t mg_raise_lds_cap_act(int cap) {
  hipError_t e = mg_raise_lds_cap_act_dcl(cap);
  if (e != hipSuccess) return e;
  e = mg_raise_lds_cap_act_nodcl(cap);
  if (e != hipSuccess) return e;
  return mg_raise_lds_cap_act_big(cap);
}
#endif
