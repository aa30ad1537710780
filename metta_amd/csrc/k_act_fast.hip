// k_act_fast.hip -- see mg_kernels.h / DESIGN.md section 4
#define MG_NO_CLIPPER  // (configurations with a Clipper take the general kernels)
#include "mg_kernels.h"

// ------------------------------------------------------------------------------------------------------------
// kernel 1 (fast path): actions, wave-parallel.  One wave per environment, lane i = agent i.
//
// Eligible configurations (host-checked): no vibe-triggered attack/transfer, no required/consumed resources on
// noop/move/change_vibe, no inventory regeneration, A <= 64.  Then within a step positions change only through
// moves into empty cells, and the reference's shuffled sequential loop (bindings/mettagrid_c.cpp:588-622) is
// reproduced exactly by
//   1. the std::shuffle ranks, from the same MT19937 draws (lanes hold draws; swaps applied to ranks),
//   2. a fixed-point over "did agent j's move succeed", whose dependencies only point to lower ranks
//      (unique solution; converges in a few ballots' worth of iterations),
//   3. assembler uses (moves into an assembler), collected in rank order with the neighbour ring as it stood at
//      that moment, and executed by mg_uses_kernel on LDS-staged inventories.
// Episode restarts (template copies) and the rare rejected RNG draw (serial redraw) are handled in place.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MG_FAST_THREADS) void mg_act_fast_kernel(KArgs a) {
  __shared__ uint32_t mtnext[624];
  const int lane = threadIdx.x, e = blockIdx.x + a.env_base;
  const int A = a.A;
  const bool active = lane < A;
  const size_t g = (size_t)e * A + (active ? lane : 0);
  DAgent* rec = a.agents + g;
  // Everything that depends on nothing but e is requested at once (env words, agent fields, action): one round
  // trip instead of two.  A restart rewrites the records, so that rare case reads them again.
  DEnv env = a.env[e];
  uint4 f0 = make_uint4(0, 0, 0, 0);
  uint32_t max_swm = 0, succ0 = 0, succ1 = 0, succ2 = 0, fail0 = 0, fail1 = 0, fail2 = 0;
  int act = 0;
  auto load_agent = [&]() {
    if (active) {
      f0 = *(const uint4*)rec;
      max_swm = rec->max_swm;
      succ0 = rec->succ[0]; succ1 = rec->succ[1]; succ2 = rec->succ[2];
      fail0 = rec->fail[0]; fail1 = rec->fail[1]; fail2 = rec->fail[2];
    }
  };
  load_agent();
  if (active) act = a.actions[g];
  if (a.auto_reset && env.done && !(a.diag & 2048)) {  // wave-uniform (2048: timing ablation, wrong results)
    mg_restart_env(a, e, env, lane, 64);
    load_agent();
  }
  mg_use_map(a, env.map_id);
  const int* dc = a.dc;
  const int H = a.H, W = a.W;
  int r = (int)(f0.x & 0xFFFF), c = (int)(f0.x >> 16);
  int frozen = (int)(int16_t)(f0.z & 0xFFFF);
  uint32_t vibe = (f0.z >> 16) & 0xFF, group = f0.z >> 24, swm = f0.w;
  // Eligible configurations have no attacks, so nothing can freeze an agent; flag it loudly if it ever happens.
  if (__ballot(active && frozen != 0) != 0) env.error |= 4u;

  // ---- MT19937 window -------------------------------------------------------------------
  const uint32_t* mt = a.mt + (size_t)e * 624;
  const int NW_ = a.NWIN < 64 ? a.NWIN : 64;
  const bool need_twist = (int)env.mt_idx + NW_ > 624 && !(a.diag & 1024);  // 1024: timing ablation, wrong results
  if (need_twist) {
    for (int i = lane; i < 227; i += 64) mtnext[i] = mt_mix(mt[i], mt[i + 1], mt[i + 397]);
    __syncthreads();
    for (int i = 227 + lane; i < 454; i += 64) mtnext[i] = mt_mix(mt[i], mt[i + 1], mtnext[i - 227]);
    __syncthreads();
    for (int i = 454 + lane; i < 623; i += 64) mtnext[i] = mt_mix(mt[i], mt[i + 1], mtnext[i - 227]);
    __syncthreads();
    if (lane == 0) mtnext[623] = mt_mix(mt[623], mtnext[0], mtnext[396]);
    __syncthreads();
  }
  uint32_t w = 0;
  if (lane < NW_) {
    int j = (int)env.mt_idx + lane;
    w = mt_temper(j < 624 ? mt[j] : mtnext[j - 624]);
  }
  // draw t of std::shuffle(iota(A)): t = 0 -> U(2) when A is even; then positions i = i0 + 2(t - t0), range (i+1)(i+2)
  const int even = (A & 1) == 0 ? 1 : 0;
  const int D = A <= 1 ? 0 : even + (A - (even ? 2 : 1)) / 2;
  uint32_t val = 0;
  bool reject = false;
  if (lane < D) {
    uint32_t range;
    if (even && lane == 0) {
      range = 2;
    } else {
      uint32_t i = (uint32_t)((even ? 2 : 1) + 2 * (lane - even));
      range = (i + 1) * (i + 2);
    }
    unsigned long long product = (unsigned long long)w * range;
    uint32_t low = (uint32_t)product;
    reject = low < range && low < (0u - range) % range;  // Lemire rejection: extremely rare, needs a serial redraw
    val = (uint32_t)(product >> 32);
  }
  uint32_t used = (uint32_t)D;  // MT19937 outputs consumed by the shuffle
  if (__ballot(reject) != 0 || (a.diag & 8)) {
    // A rejected draw shifts every later draw by one output: redo the draws one after the other (wave-uniform scalars;
    // about one step in a million takes this branch, MG_DIAG bit 8 forces it for the tests).
    int idx = 0;
    bool overflow = false;
    for (int t = 0; t < D; t++) {
      uint32_t range = 2;
      if (!(even && t == 0)) {
        const uint32_t i = (uint32_t)((even ? 2 : 1) + 2 * (t - even));
        range = (i + 1) * (i + 2);
      }
      const uint32_t threshold = (0u - range) % range;
      uint32_t hi = 0;
      for (;;) {
        if (idx >= NW_) {
          overflow = true;
          break;
        }
        const uint32_t wv = (uint32_t)__builtin_amdgcn_readlane((int)w, idx);
        idx++;
        const unsigned long long product = (unsigned long long)wv * range;
        const uint32_t low = (uint32_t)product;
        hi = (uint32_t)(product >> 32);
        if (!(low < range && low < threshold)) break;
      }
      if (lane == t) val = hi;
    }
    used = (uint32_t)idx;
    if (overflow) env.error |= 1u;
  }
  // ranks: agent x sits at position rank_x of the shuffled order; a swap of positions (p, q) exchanges the ranks.
  // The two swap partners of draw t (x / (s+1), x % (s+1)) are derived once, lane-parallel.
  int pa_l = 0, pb_l = 0;
  if (lane < D && !(even && lane == 0)) {
    const uint32_t s1 = (uint32_t)((even ? 2 : 1) + 2 * (lane - even)) + 2u;  // s + 1
    pa_l = (int)(val / s1);
    pb_l = (int)(val - (uint32_t)pa_l * s1);
  }
  int rank = lane;
  for (int t = 0; t < D; t++) {
    if (even && t == 0) {
      int j = __builtin_amdgcn_readlane((int)val, 0);  // swap(v[1], v[j])
      rank = rank == 1 ? j : (rank == j ? 1 : rank);
    } else {
      int i = (even ? 2 : 1) + 2 * (t - even);
      int pa = __builtin_amdgcn_readlane(pa_l, t), pb = __builtin_amdgcn_readlane(pb_l, t);
      rank = rank == i ? pa : (rank == pa ? i : rank);
      rank = rank == i + 1 ? pb : (rank == pb ? i + 1 : rank);
    }
  }

  // ---- decode actions ------------------------------------------------------------------------------------
  const int n_actions = dc[DC_N_ACTIONS], oa = dc[DC_OFF_ACTIONS];
  const bool valid = active && act >= 0 && act < n_actions;
  int kind = -1, arg = 0;
  if (valid) {
    kind = dc[oa + 2 * act];
    arg = dc[oa + 2 * act + 1];
  }
  const bool is_move = valid && kind == MG_ACT_MOVE;
  const bool is_vibe = valid && kind == MG_ACT_CHANGE_VIBE;
  int tr = r, tc = c;
  uint16_t scell = 0;
  bool inb = false;
  if (is_move) {
    // deltas of orientation.hpp:26-48: N S W E NW NE SW SE
    const int dx = (arg == 2 || arg == 4 || arg == 6) ? -1 : ((arg == 3 || arg == 5 || arg == 7) ? 1 : 0);
    const int dy = (arg == 0 || arg == 4 || arg == 5) ? -1 : ((arg == 1 || arg == 6 || arg == 7) ? 1 : 0);
    tr = r + dy;
    tc = c + dx;
    inb = tr >= 0 && tc >= 0 && tr < H && tc < W;
    if (inb) scell = a.map_cells[(size_t)env.map_id * a.HW + tr * W + tc];
  }
  const CellCodec cc{a.NW, A, a.NAmax};
  const int skind = cc.kind(scell);
  const bool mover = is_move && inb && skind == MG_CELL_EMPTY;
  const bool user = is_move && inb && skind == MG_CELL_ASSEMBLER;
  const uint32_t P = ((uint32_t)r << 16) | (uint32_t)c;
  const uint32_t Tm = mover ? (((uint32_t)tr << 16) | (uint32_t)tc) : 0xFFFFFFFFu;

  // ---- fixed point over move success ----------------------------------------------------------------------
  // Per lane, once: who stands on my target, who else wants it, who acts before me.  Each iteration is then a
  // handful of mask operations on the ballot of the current guess.
  unsigned long long occ = 0, tgt = 0, lower = 0;
  for (int j = 0; j < A; j++) {
    const uint32_t Pj = (uint32_t)__builtin_amdgcn_readlane((int)P, j);
    const uint32_t Tj = (uint32_t)__builtin_amdgcn_readlane((int)Tm, j);
    const int rj = __builtin_amdgcn_readlane(rank, j);
    const unsigned long long bit = 1ull << j;
    if (Pj == Tm) occ |= bit;
    if (Tj == Tm && j != lane) tgt |= bit;
    if (rj < rank) lower |= bit;
  }
  bool m = mover;
  for (int it = 0; it <= A; it++) {
    const unsigned long long before = lower & __ballot(m);  // moved (away / in) before my turn
    const bool nm = mover && (occ & ~before) == 0 && (tgt & before) == 0;
    const bool changed = nm != m;
    m = nm;
    if (__ballot(changed) == 0) break;
  }
  const unsigned long long moved_mask = __ballot(m);

  // ---- assembler uses, in rank order, with the ring as it stood at that moment ------------------------------
  unsigned long long umask = __ballot(user);
  int n_pending = 0;
  uint8_t* pend = a.pend + (size_t)e * A * 32;
  while (umask) {
    // next user = lowest rank among the remaining ones
    int best = -1, best_rank = 1 << 30;
    unsigned long long scan = umask;
    while (scan) {
      int j = __ffsll((long long)scan) - 1;
      scan &= scan - 1;
      int rj = __builtin_amdgcn_readlane(rank, j);
      if (rj < best_rank) {
        best_rank = rj;
        best = j;
      }
    }
    umask &= ~(1ull << best);
    const int k = best_rank;
    const int ar = __builtin_amdgcn_readlane(tr, best), ac = __builtin_amdgcn_readlane(tc, best);
    const int asm_cell = __builtin_amdgcn_readlane((int)scell, best);
    const int uact = __builtin_amdgcn_readlane(act, best);
    // my position and vibe at time k
    const bool before = rank < k;
    const int pr = (before && m) ? tr : r, pc = (before && m) ? tc : c;
    const uint32_t vk = (before && is_vibe) ? (uint32_t)arg : vibe;
    const int dr = pr - ar, dcl = pc - ac;
    const bool adj = active && dr >= -1 && dr <= 1 && dcl >= -1 && dcl <= 1 && !(dr == 0 && dcl == 0);
    // ring index among the in-map neighbour cells (NW, N, NE, W, E, SW, S, SE)
    uint32_t vmask8 = 0;
    for (int q = 0; q < 8; q++) {
      int o9 = q < 4 ? q : q + 1;
      int rr = ar + o9 / 3 - 1, c2 = ac + o9 % 3 - 1;
      if (rr >= 0 && c2 >= 0 && rr < H && c2 < W) vmask8 |= 1u << q;
    }
    const int np = __popc(vmask8);
    int o9m = (dr + 1) * 3 + (dcl + 1);
    int om = o9m < 4 ? o9m : o9m - 1;
    int posidx = adj ? __popc(vmask8 & ((1u << om) - 1u)) : 0;
    const int start = __builtin_amdgcn_readlane(posidx, best);  // the actor is adjacent by construction
    int key = adj ? (posidx - start + np) % np : -1;
    // emit the record: neighbours in key order
    uint8_t* rec8 = pend + (size_t)n_pending * 32;
    int nn = 0;
    for (int q = 0; q < np; q++) {
      unsigned long long sel = __ballot(adj && key == q);
      if (sel) {
        int j = __ffsll((long long)sel) - 1;
        uint32_t vj = (uint32_t)__builtin_amdgcn_readlane((int)vk, j);
        if (lane == 0) {
          rec8[8 + nn] = (uint8_t)j;
          rec8[16 + nn] = (uint8_t)vj;
        }
        nn++;
      }
    }
    if (lane == 0) {
      rec8[0] = (uint8_t)best;
      rec8[1] = (uint8_t)nn;
      int ai_ = cc.asm_idx((uint16_t)asm_cell);
      rec8[2] = (uint8_t)(ai_ & 0xFF);
      rec8[3] = (uint8_t)(ai_ >> 8);
      rec8[4] = (uint8_t)(uact & 0xFF);
      rec8[5] = (uint8_t)((uact >> 8) & 0xFF);
      rec8[6] = (uint8_t)((uact >> 16) & 0xFF);
      rec8[7] = (uint8_t)((uact >> 24) & 0xFF);
    }
    n_pending++;
  }
  (void)moved_mask;
  (void)group;

  // ---- per-agent bookkeeping (action_handler.hpp:118-150) ----------------------------------------------------
  if (active) {
    bool ok = false;
    if (valid) {
      if (kind == MG_ACT_NOOP) ok = true;
      else if (is_vibe) { vibe = (uint32_t)arg; ok = true; }
      else if (mover) ok = m;
      // users: decided by the general kernel; static failures (wall / outside): false
      if (m) { r = tr; c = tc; swm = 0; }
      else { swm += 1; if (swm > max_swm) max_swm = swm; }
      if (!user) {
        if (ok) { if (kind == 0) succ0++; else if (kind == 1) succ1++; else succ2++; }
        else { if (kind == 0) fail0++; else if (kind == 1) fail1++; else fail2++; }
      }
      // prev_location = location
      f0.x = ((uint32_t)c << 16) | (uint32_t)r;
      f0.y = f0.x;
      f0.z = (f0.z & 0xFF00FFFFu) | (vibe << 16);
      f0.w = swm;
      *(uint4*)rec = f0;
      rec->max_swm = max_swm;
      rec->succ[0] = succ0; rec->succ[1] = succ1; rec->succ[2] = succ2;
      rec->fail[0] = fail0; rec->fail[1] = fail1; rec->fail[2] = fail2;
    } else {
      // _handle_invalid_action, once per priority level (bindings/mettagrid_c.cpp:565-570,599-608)
      mg_count_invalid(a.astats + g * a.SA, MgAgentSlots{a.R, a.NG}, act, n_actions, dc[DC_MAX_PRIORITY] + 1);
    }
    a.exec[g] = ok ? act : 0;
    a.success[g] = ok ? 1 : 0;
  }
  // ---- env write-back ------------------------------------------------------------------------------------------
  uint32_t new_idx = env.mt_idx + used;
  if (new_idx > 624) {
    uint32_t* mtw = a.mt + (size_t)e * 624;
    for (int i = lane; i < 624; i += 64) mtw[i] = mtnext[i];
    new_idx -= 624;
  }
  if (lane == 0) {
    env.step += 1;
    env.mt_idx = new_idx;
    a.env[e] = env;
    a.mode[e] = n_pending > 0 ? (1u | ((uint32_t)n_pending << 8)) : 0u;
    if (env.error) atomicOr(&a.status[0], env.error);  // sticky: the host reads one word for all environments
  }
}


void mg_launch_act_fast(const KArgs& k, unsigned n, hipStream_t st) {
  hipLaunchKernelGGL(mg_act_fast_kernel, dim3(n), dim3(MG_FAST_THREADS), 0, st, k);
}
