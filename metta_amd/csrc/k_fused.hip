// k_fused.hip -- the whole step of one environment in ONE workgroup of two waves (attack-free configurations, A <= 64:
// arena, navigation).  Replaces the three launches mg_act_fast_kernel -> mg_uses_kernel -> mg_obs_kernel:
//
//   P0  both waves   env words, episode restart (rare), stage the config prefix, the agent records and the MT19937
//                    window into LDS -- every load is issued before the first wait
//   P1  one wave     the reference's shuffled sequential action loop, lane = agent (mettagrid_c.cpp:588-622): shuffle
//                    ranks from the same MT19937 draws, move resolution as a fixed point over ballots, assembler uses
//                    collected in rank order with the neighbour ring as it stood at that turn
//   P2  one wave     the assembler uses (assembler.hpp:455-526) on the LDS records; independent uses on separate lanes
//   P3  both waves   per-object token lists, observation rows (mettagrid_c.cpp:397-555), tokens_* stats, stat rewards,
//                    episode flags (:652-677); the records go back to HBM once
//
// A step of one environment is a chain of dependent phases, most of them latency bound; what hides the latency is the
// number of environments resident on a CU.  The kernel is therefore built around its LDS footprint: about 10 KB per
// environment (arena), so that 16 workgroups x 2 waves fill a CU (32 waves, 160 KB) and 4096 environments are resident
// on the chip at once.  To get there nothing that other workgroups can share is copied into LDS: the static map layer
// is read from the byte pool in HBM/L2 (window lookups, move targets) and the agents are scattered over the window
// through a 128-byte per-wave buffer; assembler records and the map's assembler table are used in place in HBM; only
// the prefix of the config stream that the use / token code walks with dependent reads is staged.
// Against the three-kernel path the agent records are read and written once instead of three times, and the hand-over
// buffers (mode / pending uses / executed action) never leave LDS.
#define MG_NO_CLIPPER  // (configurations with a Clipper take the general kernels)
#include "mg_kernels.h"
// Wave priority in the two serial phases (s_setprio): the workgroups that end a launch are those with assembler uses,
// and in both phases one wave works while its partner waits at the barrier.  One launch over 4096 environments
// 0.0605 -> 0.0570 ms; the four-stream form is indifferent (levels 1..3 tried, no difference between them).
#ifndef MG_PRIO_USES
#define MG_PRIO_USES 2
#endif
#ifndef MG_PRIO_ACT
#define MG_PRIO_ACT 1
#endif

// Built twice: as is, and through k_fused_nomod.hip with MG_NO_MODIFIERS (configurations without inventory-limit
// modifiers, e.g. arena and navigation: the use code then needs far fewer registers).
// and a third time (k_fused_nomod_o7.hip) for 7 instead of 8 waves per SIMD: 72 registers, no vector spills, 14
// workgroups per CU -- the better build whenever the launch is not exactly one round of 16 workgroups per CU (small
// groups stepped on several streams, or more environments than one round holds).
#if defined(MG_NO_MODIFIERS) && defined(MG_FUSED_W18)
#define mg_step_fused_kernel mg_step_fused_nomod_w18_kernel
#define mg_launch_fused mg_launch_fused_nomod_w18
#define mg_fused_lds_bytes mg_fused_lds_bytes_nomod_w18
#define mg_raise_lds_cap_fused mg_raise_lds_cap_fused_nomod_w18
#define MG_FUSED_OCC 8
#define MG_FUSED_THREADS 64
#elif defined(MG_NO_MODIFIERS) && defined(MG_FUSED_W1)
// (k_fused_nomod_w1.hip: ONE wave per environment, for environments of a few agents -- navigation has 4 -- where a
// second wave only repeats the scalar work and idles through the single-wave phases)
#define mg_step_fused_kernel mg_step_fused_nomod_w1_kernel
#define mg_launch_fused mg_launch_fused_nomod_w1
#define mg_fused_lds_bytes mg_fused_lds_bytes_nomod_w1
#define mg_raise_lds_cap_fused mg_raise_lds_cap_fused_nomod_w1
#define MG_FUSED_OCC 7
#define MG_FUSED_THREADS 64
#elif defined(MG_NO_MODIFIERS) && defined(MG_FUSED_O7)
#define mg_step_fused_kernel mg_step_fused_nomod_o7_kernel
#define mg_launch_fused mg_launch_fused_nomod_o7
#define mg_fused_lds_bytes mg_fused_lds_bytes_nomod_o7
#define mg_raise_lds_cap_fused mg_raise_lds_cap_fused_nomod_o7
#define MG_FUSED_OCC 7
#elif defined(MG_FUSED_O6)  // (k_fused_o6.hip: the build with modifiers needs 80 registers to stop spilling)
#define mg_step_fused_kernel mg_step_fused_o6_kernel
#define mg_launch_fused mg_launch_fused_o6
#define mg_fused_lds_bytes mg_fused_lds_bytes_o6
#define mg_raise_lds_cap_fused mg_raise_lds_cap_fused_o6
#define MG_FUSED_OCC 6
#elif defined(MG_NO_MODIFIERS)
#define mg_step_fused_kernel mg_step_fused_nomod_kernel
#define mg_launch_fused mg_launch_fused_nomod
#define mg_fused_lds_bytes mg_fused_lds_bytes_nomod
#define mg_raise_lds_cap_fused mg_raise_lds_cap_fused_nomod
#endif
#ifdef MG_FUSED_OCC_OVERRIDE  // (experiments: tools/fvariant.sh)
#undef MG_FUSED_OCC
#define MG_FUSED_OCC MG_FUSED_OCC_OVERRIDE
#endif
#ifndef MG_FUSED_THREADS
#define MG_FUSED_THREADS 128
#endif
// The Lds pointers of the fused kernel from the layout in the kernel arguments (`smem` = the workgroup's dynamic LDS).
__device__ __forceinline__ void mg_fused_lds(Lds& l, const KArgs& a, unsigned char* smem) {
  const FusedLds& f = a.fl;
  l.dc = (int*)(smem + f.dc);
  l.grid = nullptr; l.ag = (DAgent*)(smem + f.ag); l.ch = l.ag + a.A;
  l.exec = (int*)(smem + f.ex); l.tok_att = (int*)(smem + f.ta); l.tokacc = (float*)(smem + f.tk);
  l.rows = (uint8_t*)(smem + f.rows); l.pend = smem + f.rows;
  l.ocnt = (uint32_t*)(smem + f.u) + 1; l.heap = (uint16_t*)(smem + f.heap); l.pat = nullptr;
  l.tmp = (int*)(smem + f.u); l.mtnext = (uint32_t*)(smem + f.u);
  l.wscr = smem + f.ws; l.ipat = smem + f.ip; l.WS = f.WS;  // (ipat: re-pointed at the staged config by mg_fused_phase<true>)
  l.gd = (int*)(smem + f.gd); l.dirty_amt = (uint32_t*)(smem + f.da); l.dirty_misc = (uint32_t*)(smem + f.dm);
  l.order = (uint16_t*)(smem + f.ord); l.flags = (int*)(smem + f.fl); l.succ = smem + f.su;
  l.gsetv = nullptr; l.gsetm = nullptr; l.win = nullptr;
}

// ---- P1: the action phase of mg_act_fast_kernel on LDS-staged state (wave 0, lane = agent) ---------------------------
// `act` = this lane's action, `w` = tempered MT19937 output mt_idx + lane.  Returns the MT outputs consumed; sets
// *n_pending_out (wave-uniform) and env.error bits.
__device__ __forceinline__ uint32_t mg_fused_act(const KArgs& a, const Lds& l, DEnv& env, const int lane, const int act,
                                                 const uint32_t w, const int NW_, int* n_pending_out) {
  const int A = a.A;
  const int* dc = a.dc;
  const bool active = lane < A;
  const int H = a.H, W = a.W;
  DAgent* rec = l.ag + (active ? lane : 0);
  uint4 f0 = make_uint4(0, 0, 0, 0);
  uint32_t max_swm = 0;
  if (active) {
    f0 = *(const uint4*)rec;
    max_swm = rec->max_swm;
  }
  int r = (int)(f0.x & 0xFFFF), c = (int)(f0.x >> 16);
  const int frozen = (int)(int16_t)(f0.z & 0xFFFF);
  uint32_t vibe = (f0.z >> 16) & 0xFF, swm = f0.w;
  // Eligible configurations have no attacks, so nothing can freeze an agent; flag it loudly if it ever happens.
  if (__ballot(active && frozen != 0) != 0) env.error |= 4u;

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
  if (valid) {  // (the action table lies behind the LDS-staged prefix of the stream)
    const int2 ka = *(const int2*)(a.gdc + oa + 2 * act);
    kind = ka.x;
    arg = ka.y;
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
    if (inb) scell = (uint16_t)l.gmap[tr * (W + a.PADC) + tc];  // the static layer of the map (HBM pool, L2; padded rows)
  }
  const CellCodec cc{a.NW, A, a.NAmax};
  const int skind = cc.kind(scell);
  const bool mover = is_move && inb && skind == MG_CELL_EMPTY;
  const bool bump = is_move && inb && skind == MG_CELL_ASSEMBLER;
  // A use of an assembler that is cooling down (without partial usage) or used up fails whatever happens earlier in
  // this step (nothing in a step shortens a cooldown or lowers a use count; assembler.hpp:455-470) and has no effect
  // but the failed move: decided here, lane-parallel, instead of queueing it for the serial use phase.  The record
  // loads are issued now and consumed after the move fixed point.
  DAsm bump_as = {0, 0, 0, 0};
  int bump_cfg = 0;
  if (bump) {
    const int ai_ = cc.asm_idx(scell);
    bump_as = l.as[ai_];
    bump_cfg = l.ainfo[ai_].z;
  }
  const uint32_t P = ((uint32_t)r << 16) | (uint32_t)c;
  const uint32_t Tm = mover ? (((uint32_t)tr << 16) | (uint32_t)tc) : 0xFFFFFFFFu;

  // ---- fixed point over move success ----------------------------------------------------------------------
  // (agent masks: 32 bits wide when the agents fit -- half the vector instructions of the 64-bit form)
  bool m = mover;
#if defined(MG_FUSED_O7) || defined(MG_FUSED_O6) || defined(MG_FUSED_W1)
  auto fixed_point = [&](auto zero) {
    typedef decltype(zero) mask_t;
    mask_t occ = 0, tgt = 0, lower = 0;
    for (int j = 0; j < A; j++) {
      const uint32_t Pj = (uint32_t)__builtin_amdgcn_readlane((int)P, j);
      const uint32_t Tj = (uint32_t)__builtin_amdgcn_readlane((int)Tm, j);
      const int rj = __builtin_amdgcn_readlane(rank, j);
      const mask_t bit = (mask_t)1 << j;
      if (Pj == Tm) occ |= bit;
      if (Tj == Tm && j != lane) tgt |= bit;
      if (rj < rank) lower |= bit;
    }
    for (int it = 0; it <= A; it++) {
      const mask_t before = lower & (mask_t)__ballot(m);  // moved (away / in) before my turn
      const bool nm = mover && (occ & ~before) == 0 && (tgt & before) == 0;
      const bool changed = nm != m;
      m = nm;
      if (__ballot(changed) == 0) break;
    }
  };
  if (A <= 32) fixed_point((uint32_t)0);
  else fixed_point((unsigned long long)0);
#else  // (the 64-register builds keep the one 64-bit form: a second copy of the loop costs them on one stream, 0.064 -> 0.066 ms)
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
  for (int it = 0; it <= A; it++) {
    const unsigned long long before = lower & __ballot(m);  // moved (away / in) before my turn
    const bool nm = mover && (occ & ~before) == 0 && (tgt & before) == 0;
    const bool changed = nm != m;
    m = nm;
    if (__ballot(changed) == 0) break;
  }
#endif

  // ---- assembler uses, in rank order, with the ring as it stood at that moment ------------------------------
  bool user = bump;
  if (bump) {
    const int* arec = asm_rec(dc, bump_cfg);
    const uint32_t max_uses = (uint32_t)arec[AR_MAX_USES];
    const bool used_up = max_uses > 0 && bump_as.uses_count >= max_uses;
    const bool cooling = bump_as.cooldown_end > env.step && arec[AR_ALLOW_PARTIAL] == 0;
    user = !(used_up || cooling);
  }
  unsigned long long umask = __ballot(user);
  // A use that is alone in its step meets the inventories exactly as they were staged (nothing else on this path
  // changes them), so whether the surrounding agents can pay the protocol's inputs (can_afford_protocol,
  // assembler.hpp:124-137) is decided here, across the lanes; an unaffordable use has no effect but the failed move
  // and never reaches the serial phase.  (Two thirds of the queued uses failed there under random actions.)
#ifdef MG_NO_AFFORD_FILTER
  const bool lone_use = false;
#else
  const bool lone_use = __popcll(umask) == 1;
#endif
  int n_pending = 0;
  uint8_t* pend = l.pend;
  while (umask) {
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
    const bool before = rank < k;
    const int pr = (before && m) ? tr : r, pc = (before && m) ? tc : c;
    const uint32_t vk = (before && is_vibe) ? (uint32_t)arg : vibe;
    const int dr = pr - ar, dcl = pc - ac;
    const bool adj = active && dr >= -1 && dr <= 1 && dcl >= -1 && dcl <= 1 && !(dr == 0 && dcl == 0);
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
    if (lone_use) {
      const int cfg_b = __builtin_amdgcn_readlane(bump_cfg, best);
      const uint32_t clipped_b = (uint32_t)__builtin_amdgcn_readlane((int)bump_as.clipped, best);
      const uint32_t cd_end_b = (uint32_t)__builtin_amdgcn_readlane((int)bump_as.cooldown_end, best);
      const int po = asm_rec(dc, cfg_b)[AR_SIMPLE_PO];
      if (po >= 0 && clipped_b == 0 && cd_end_b <= env.step) {  // (partial usage while cooling: left to the serial phase)
        const int oin = dc[po + PR_OFF_IN], n_in = L_n(dc, oin);
        bool afford = true;
        for (int i = 0; i < n_in; i++) {
          int x = adj ? (int)rec->amt[KV_k(dc, oin, i)] : 0;
          x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // wave sum: row shifts, then row broadcasts
          x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
          x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
          x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
          x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
          x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
          const int total = __builtin_amdgcn_readlane(x, 63) & 0xFFFF;  // (the reference adds in uint16)
          if (total < (KV_v(dc, oin, i) & 0xFFFF)) afford = false;
        }
        if (!afford) {
          if (lane == best) user = false;  // counted as a failed move by the bookkeeping below
          continue;
        }
      }
    }
    const int start = __builtin_amdgcn_readlane(posidx, best);  // the actor is adjacent by construction
    int key = adj ? (posidx - start + np) % np : -1;
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
      *(int*)(rec8 + 4) = uact;
    }
    n_pending++;
  }

  // ---- per-agent bookkeeping (action_handler.hpp:118-150) ----------------------------------------------------
  if (active) {
    bool ok = false;
    if (valid) {
      if (kind == MG_ACT_NOOP) ok = true;
      else if (is_vibe) { vibe = (uint32_t)arg; ok = true; }
      else if (mover) ok = m;
      if (m) { r = tr; c = tc; swm = 0; }
      else { swm += 1; if (swm > max_swm) max_swm = swm; }
      if (!user) {  // users: success / failure is decided by the use itself (P2)
        if (ok) rec->succ[kind] += 1;
        else rec->fail[kind] += 1;
      }
      f0.x = ((uint32_t)c << 16) | (uint32_t)r;
      f0.y = f0.x;  // prev_location = location
      f0.z = (f0.z & 0xFF00FFFFu) | (vibe << 16);
      f0.w = swm;
      *(uint4*)rec = f0;
      rec->max_swm = max_swm;
    } else {
      // _handle_invalid_action, once per priority level (bindings/mettagrid_c.cpp:565-570,599-608).  (No use of this
      // launch touches these slots: plain updates.)
      float* row = a.astats + ((size_t)(blockIdx.x + a.env_base) * A + lane) * a.SA;
      mg_count_invalid(row, MgAgentSlots{a.R, a.NG}, act, n_actions, dc[DC_MAX_PRIORITY] + 1);
    }
    l.exec[lane] = ok ? act : 0;
    l.succ[lane] = ok ? 1 : 0;
  }
  *n_pending_out = n_pending;
  return used;
}

// ---- P2: assembler uses in action order.  Uses that share neither an agent nor the assembler commute: they run in the
// same round, on separate lanes, the even slots on the acting wave (`wsel` 0) and the odd ones on the other (`wsel` 1), so
// that two divergent uses do not serialise inside one wave.  All threads of the workgroup call this. ----------------
#ifndef MG_FUSED_USES_ATTR
#define MG_FUSED_USES_ATTR __forceinline__
#endif
__device__ MG_FUSED_USES_ATTR void mg_fused_uses(const KArgs& a, const Lds& l, Ctx& cx, const int count, const int lane,
                                                 const int wsel) {
  const uint8_t* recs = l.pend;
  uint8_t* rounds = (uint8_t*)l.order;  // [A] bytes
  // round of use q = 1 + the latest round among earlier uses that share an agent or the assembler with it
  if (wsel != 0) {
    // (the acting wave orders the uses)
  } else if (count == 1) {
    if (lane == 0) {
      rounds[0] = 0;
      l.flags[0] = 0;
      l.flags[2] = 0;
    }
  } else if (lane == 0) {
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
  if (a.dbg && lane == 0 && wsel == 0) {
    a.dbg[((size_t)blockIdx.x + a.env_base) * 16 + 8] = (unsigned long long)count;
    a.dbg[((size_t)blockIdx.x + a.env_base) * 16 + 9] = (unsigned long long)max_round;
    a.dbg[((size_t)blockIdx.x + a.env_base) * 16 + 10] = 0;
  }
  for (int rd = 0; rd <= max_round; rd++) {
    for (int q0 = 0; q0 < count; q0 += MG_FUSED_SLOTS) {
      const int q = q0 + lane;
      if (lane < MG_FUSED_SLOTS && (MG_FUSED_THREADS == 64 || (lane & 1) == wsel) && q < count && rounds[q] == rd) {
        const uint8_t* rec = recs + q * 32;
        const int actor = rec[0], nn = rec[1], asm_idx = rec[2] | (rec[3] << 8);
        const int act = *(const int*)(rec + 4);
        cx.tmp = l.tmp + lane * 96;
        int* near = cx.tmp;
        uint8_t* vibes = (uint8_t*)(cx.tmp + 8);
        for (int i = 0; i < 8; i++) {
          near[i] = rec[8 + i];
          vibes[i] = rec[16 + i];
        }
        const bool ok = mg_assembler_use_core(cx, actor, asm_idx, near, vibes, nn);
        if (ok) {
          if (a.dbg) atomicAdd(&a.dbg[((size_t)blockIdx.x + a.env_base) * 16 + 10], 1ull);
          l.ag[actor].succ[MG_ACT_MOVE] += 1;
          l.exec[actor] = act;
          l.succ[actor] = 1;
        } else {
          l.ag[actor].fail[MG_ACT_MOVE] += 1;
        }
        if (cx.asm_dirty) l.flags[2] = 1;
      }
      __syncthreads();
    }
  }
}

// tokens_* game stats (mettagrid_c.cpp:552-554): the reference adds three f32 values per agent, in agent order.  While
// every partial sum stays below 2^24 those adds are exact, so the sums may be formed as integers by one wave (DPP
// reduction) and added once; beyond, thread 0 repeats the reference's sequential f32 adds.  First wave only.
__device__ __forceinline__ void mg_fused_token_stats(const KArgs& a, const Lds& l, const int lane) {
  int sw = 0, sd = 0, sf = 0;
  for (int i = lane; i < a.A; i += 64) {
    const int att = l.tok_att[i];
    const int wr = att < a.T ? att : a.T;
    sw += wr;
    sd += att - wr;
    sf += a.T - wr;
  }
  // wave sums on the vector unit's own cross-lane paths (DPP row shifts + row broadcasts; the total lands in lane 63) --
  // a __shfl_xor butterfly is six dependent trips through the LDS crossbar per value
  auto wave_sum = [](int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(x, 63);
  };
  sw = wave_sum(sw);
  sd = wave_sum(sd);
  sf = wave_sum(sf);
  const float w0 = l.tokacc[0], d0 = l.tokacc[1], f0 = l.tokacc[2];
  const float lim = 16777216.0f;
  const bool exact = w0 + (float)sw < lim && d0 + (float)sd < lim && f0 + (float)sf < lim && sw < (1 << 23) &&
                     sd < (1 << 23) && sf < (1 << 23);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    if (exact) {
      l.tokacc[0] = w0 + (float)sw;
      l.tokacc[1] = d0 + (float)sd;
      l.tokacc[2] = f0 + (float)sf;
    } else {
      mg_token_stats(a, l, 1);
    }
  }
}

// Fresh copies of the kernel arguments for one phase, read through an opaque kernarg pointer: the values of the
// previous phase cannot be kept alive across the phase boundary (they would be spilled there: the phases together need
// several times the scalar registers any one of them does, and a spilled scalar is a v_writelane / v_readlane pair on
// the VECTOR unit -- at the start of round 3 more than half of the stage phase's vector instructions).  The compiler
// loads only the fields a phase touches.
__device__ __forceinline__ void mg_fused_kargs(KArgs& a) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __attribute__((address_space(4))) KArgs* ka =
      (const __attribute__((address_space(4))) KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  a = *ka;
#endif
}
template <bool DCL>
__device__ __forceinline__ void mg_fused_phase(KArgs& a, Lds& l, unsigned char* smem, int H, int W, uint32_t map_id,
                                               int max_steps, int cfg_off) {
  mg_fused_kargs(a);
  mg_fused_lds(l, a, smem);
  const int e = blockIdx.x + a.env_base;
  l.gmap = a.map_cells8 + (size_t)map_id * a.HW8;
  l.as = a.asms + (size_t)e * a.NAmax;                             // used in place (HBM)
  l.ainfo = (int4*)(a.map_asm + (size_t)map_id * a.NAmax);         // read only
  a.gdc += cfg_off;  // per-slot configurations (0 without)
  a.dc = DCL ? l.dc : a.gdc;
  if (DCL) l.ipat = (uint8_t*)(l.dc + l.dc[DC_OFF_IPAT]);
  a.H = H;
  a.W = W;
  a.max_steps = max_steps;
}

// Thread ids of one phase, derived afresh behind an opaque copy: shared across the phases, the allocator keeps values
// derived from them (the zero-extended lane index of every 64-bit address) alive through the whole kernel and spills
// them in the tight builds (8 waves per SIMD: 17 spilled vector registers, reloaded in every phase).
#define MG_PHASE_IDS                         \
  int tid = (int)threadIdx.x;                \
  asm volatile("" : "+v"(tid));              \
  const int lane = tid & 63, wave = tid >> 6; \
  (void)lane; (void)wave;

template <bool DCL>
#ifndef MG_FUSED_OCC
#define MG_FUSED_OCC 8  // waves per SIMD the register budget allows (64 VGPRs): 16 workgroups of 2 waves per CU
#endif
__global__ __launch_bounds__(MG_FUSED_THREADS, MG_FUSED_OCC) void mg_step_fused_kernel(KArgs a0) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int nt = MG_FUSED_THREADS;
  DEnv env;
  int H, W, act = 0, NW_, actor = 0, max_steps, cfg_off;
  uint32_t mt_raw = 0;
  uint4 v1 = make_uint4(0, 0, 0, 0), v2 = v1;
  float tok0 = 0.f;  // tokens_* accumulators of the running episode (threads 0..2), requested with the first loads
  bool do_reset, slots;

  // ---- P0: stage.  In blocks that each read their own few kernel arguments (see mg_fused_kargs) --------------------
  {  // P0a: the loads that depend on nothing but the environment's index, issued before the env words are waited for
    KArgs a;
    MG_PHASE_IDS
    mg_fused_kargs(a);
    const int e = blockIdx.x + a.env_base;
    MG_STAMP(0);
    const uint4* __restrict__ s1 = (const uint4*)(a.agents + (size_t)e * a.A);
    const int n1 = a.A * (int)(sizeof(DAgent) / 16);
    if (n1 <= 2 * nt) {  // the usual shapes: at most two record chunks per thread (a restart loads them again)
      if (tid < n1) v1 = s1[tid];
      if (tid + nt < n1) v2 = s1[tid + nt];
    }
    slots = a.map_cfg != nullptr;  // per-slot configurations: the stream is known only after the env words
    if (DCL && !slots) {
      const uint4* __restrict__ s0 = (const uint4*)a.gdc;
      uint4* d0 = (uint4*)(smem + a.fl.dc);
      for (int i = tid; i < (a.dc_obs_words >> 2); i += nt) d0[i] = s0[i];  // padded to x4
    }
    if (lane < a.A) act = a.actions[(size_t)e * a.A + lane];  // (every wave: cheaper than choosing the acting wave first)
    if (tid < 3) tok0 = a.gstats[(size_t)e * a.SG + tid];
    env = a.env[e];
    {  // the env words are the same in every lane: pin them to scalar registers.  Left to itself the allocator may
       // carry them through the phases as vector registers and spill them there -- 16 bytes per LANE per spill.
      env.step = (uint32_t)__builtin_amdgcn_readfirstlane((int)env.step);
      env.mt_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)env.mt_idx);
      env.seed = (uint32_t)__builtin_amdgcn_readfirstlane((int)env.seed);
      env.map_id = (uint32_t)__builtin_amdgcn_readfirstlane((int)env.map_id);
      env.episodes = (uint32_t)__builtin_amdgcn_readfirstlane((int)env.episodes);
      env.early_reset = (uint32_t)__builtin_amdgcn_readfirstlane((int)env.early_reset);
      env.done = (uint32_t)__builtin_amdgcn_readfirstlane((int)env.done);
      env.error = (uint32_t)__builtin_amdgcn_readfirstlane((int)env.error);
    }
    do_reset = a.auto_reset && env.done;  // block-uniform
    MG_STAMP(11);
  }
  if (do_reset) {  // P0b: episode restart (rare); rewrites the records
    KArgs a;
    MG_PHASE_IDS
    mg_fused_kargs(a);
    mg_restart_env(a, blockIdx.x + a.env_base, env, tid, nt);
  }
  bool need_twist;
  {  // P0c: the map this environment plays, the window of MT19937 outputs this step may consume
    KArgs a;
    MG_PHASE_IDS
    mg_fused_kargs(a);
    const int e = blockIdx.x + a.env_base;
    const int* gdc0 = a.gdc;
    mg_use_map(a, env.map_id);
    cfg_off = (int)(a.gdc - gdc0);
    if (DCL && slots) {
      const uint4* __restrict__ s0 = (const uint4*)a.gdc;
      uint4* d0 = (uint4*)(smem + a.fl.dc);
      for (int i = tid; i < (a.dc_obs_words >> 2); i += nt) d0[i] = s0[i];
    }
    H = a.H;
    W = a.W;
    max_steps = a.max_steps;
    NW_ = a.NWIN < 64 ? a.NWIN : 64;
    const uint32_t* mt = a.mt + (size_t)e * 624;
    need_twist = (int)env.mt_idx + NW_ > 624;  // block-uniform
    const int j = (int)env.mt_idx + lane;
    if (lane < NW_ && j < 624) mt_raw = mt[j];
    const int n1 = a.A * (int)(sizeof(DAgent) / 16);
    uint4* dag = (uint4*)(smem + a.fl.ag);
    if (n1 <= 2 * nt && !do_reset) {
      if (tid < n1) dag[tid] = v1;
      if (tid + nt < n1) dag[tid + nt] = v2;
    } else {
      const uint4* __restrict__ s1 = (const uint4*)(a.agents + (size_t)e * a.A);
      for (int i = tid; i < n1; i += nt) dag[i] = s1[i];
    }
    MG_STAMP(12);
  }
  {  // P0d: per-step LDS state
    KArgs a;
    MG_PHASE_IDS
    mg_fused_kargs(a);
    const int e = blockIdx.x + a.env_base;
    const int NP = a.NWINCELLS;
    int* gd = (int*)(smem + a.fl.gd);
    uint32_t* da = (uint32_t*)(smem + a.fl.da);
    uint32_t* dm = (uint32_t*)(smem + a.fl.dm);
    for (int i = tid; i < a.SG; i += nt) gd[i] = 0;
    for (int i = tid; i < a.A; i += nt) {
      da[i] = 0;
      dm[i] = 0;
    }
    uint32_t* wscr = (uint32_t*)(smem + a.fl.ws);
    for (int i = tid; i < ((nt >> 6) * 2 * a.fl.WS) >> 2; i += nt) wscr[i] = 0;
    if (!DCL) {  // window cell -> position in the pattern order (with the config prefix in LDS: read from there, DC_OFF_IPAT)
      const int* gdc = a.gdc + cfg_off;
      uint8_t* ipat = smem + a.fl.ip;
      const int hr = gdc[DC_OBS_H] >> 1, wr = gdc[DC_OBS_W] >> 1, ow = gdc[DC_OBS_W], op = gdc[DC_OFF_PATTERN];
      for (int p = tid; p < NP; p += nt) {
        const uint32_t packed = (uint32_t)gdc[op + p];
        const int dr = (int)(packed & 0xFF) - 8, dcc = (int)((packed >> 8) & 0xFF) - 8;
        ipat[(dr + hr) * (ow | 1) + dcc + wr] = (uint8_t)p;
      }
    }
    float* tokacc = (float*)(smem + a.fl.tk);
    int* flags = (int*)(smem + a.fl.fl);
    if (tid < 3) tokacc[tid] = do_reset ? a.gstats[(size_t)e * a.SG + tid] : tok0;  // (a restart rewrote them)
    if (tid == 0) {
      flags[3] = 0;
      flags[2] = 0;
    }
    if (need_twist) {  // next MT19937 block (rare: once per 624 outputs of an environment)
      const uint32_t* mt = a.mt + (size_t)e * 624;
      uint32_t* mtnext = (uint32_t*)(smem + a.fl.u);
      for (int i = tid; i < 227; i += nt) mtnext[i] = mt_mix(mt[i], mt[i + 1], mt[i + 397]);
      __syncthreads();
      for (int i = 227 + tid; i < 454; i += nt) mtnext[i] = mt_mix(mt[i], mt[i + 1], mtnext[i - 227]);
      __syncthreads();
      for (int i = 454 + tid; i < 623; i += nt) mtnext[i] = mt_mix(mt[i], mt[i + 1], mtnext[i - 227]);
      __syncthreads();
      if (tid == 0) mtnext[623] = mt_mix(mt[623], mtnext[0], mtnext[396]);
    }
    env.step += 1;
    actor = (MG_FUSED_THREADS == 64 || (a.diag & 4096)) ? 0 : (int)(blockIdx.x & 1u);  // spread the single-wave phases over the SIMDs
    __syncthreads();
    MG_STAMP(1);
  }

  // ---- P1: actions (one wave) ------------------------------------------------------------------------------------
  if ((int)(threadIdx.x >> 6) == actor) {  // (65536: timing / counter ablation, results wrong on purpose)
    KArgs a;
    MG_PHASE_IDS
    Lds l;
    mg_fused_phase<DCL>(a, l, smem, H, W, env.map_id, max_steps, cfg_off);
    const int e = blockIdx.x + a.env_base;
    if (!(a.diag & 65536)) {  // (65536: timing / counter ablation, results wrong on purpose)
#ifdef MG_PRIO_ACT
    __builtin_amdgcn_s_setprio(MG_PRIO_ACT);
#endif
    uint32_t w = 0;
    {
      const int j = (int)env.mt_idx + lane;
      if (lane < NW_) w = mt_temper(j < 624 ? mt_raw : l.mtnext[j - 624]);
    }
    int n_pending = 0;
    const uint32_t used = mg_fused_act(a, l, env, lane, act, w, NW_, &n_pending);
    uint32_t new_idx = env.mt_idx + used;
    if (new_idx > 624) {  // == 624: the block is exhausted but untwisted; the next step twists it
      uint32_t* mtw = a.mt + (size_t)e * 624;
      for (int i = lane; i < 624; i += 64) mtw[i] = l.mtnext[i];
      new_idx -= 624;
    }
    env.mt_idx = new_idx;
    if (lane == 0) {
      a.env[e] = env;  // (done / early_reset are updated by the closing phase)
      l.flags[3] = n_pending;
      if (env.error) atomicOr(&a.status[0], env.error);
    }
#ifdef MG_PRIO_ACT
    __builtin_amdgcn_s_setprio(0);
#endif
    }
  }
  __syncthreads();

  // ---- P2: assembler uses (one wave) -----------------------------------------------------------------------------------
  {
    KArgs a;
    MG_PHASE_IDS
    Lds l;
    mg_fused_phase<DCL>(a, l, smem, H, W, env.map_id, max_steps, cfg_off);
    const int e = blockIdx.x + a.env_base;
    MG_STAMP(2);
    const int n_uses = l.flags[3];
    if (n_uses > 0) {  // block-uniform
#ifdef MG_PRIO_USES
      __builtin_amdgcn_s_setprio(MG_PRIO_USES);
#endif
      Ctx cx;
      mg_make_ctx(a, l, cx, e, (int)env.map_id, env.step);
      cx.grid = nullptr;
      mg_fused_uses(a, l, cx, n_uses, lane, wave == actor ? 0 : 1);  // (ends with a workgroup barrier)
      mg_flush_stats(a, l, cx, e, tid, nt);
#ifdef MG_PRIO_USES
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    MG_STAMP(3);
  }

  // ---- P3: observations, rewards, flags -----------------------------------------------------------------------------
  {
    KArgs a;
    MG_PHASE_IDS
    Lds l;
    mg_fused_phase<DCL>(a, l, smem, H, W, env.map_id, max_steps, cfg_off);
    __syncthreads();  // (the use scratch becomes the token lists)
    if (!(a.diag & 32)) mg_build_tokens_fused(a, l, (int)env.map_id, env.step, tid, nt);
    __syncthreads();
    MG_STAMP(4);
  }
  {
    KArgs a;
    MG_PHASE_IDS
    Lds l;
    mg_fused_phase<DCL>(a, l, smem, H, W, env.map_id, max_steps, cfg_off);
    const int e = blockIdx.x + a.env_base;
    if (!(a.diag & 16)) mg_encode_all<uint8_t, false, true>(a, l, e, env.step, l.exec, true, tid, nt);
    __syncthreads();
    MG_STAMP(5);
  }
  {
    KArgs b;
    MG_PHASE_IDS
    Lds l;
    mg_fused_phase<DCL>(b, l, smem, H, W, env.map_id, max_steps, cfg_off);
    const int e = blockIdx.x + b.env_base;
    if (wave == 0) {
      mg_fused_token_stats(b, l, lane);
      if (lane == 0) {
        float* gs_ = b.gstats + (size_t)e * b.SG;
        gs_[0] = l.tokacc[0];
        gs_[1] = l.tokacc[1];
        gs_[2] = l.tokacc[2];
      }
    }
    __syncthreads();
    if (!(b.diag & 64)) mg_rewards_and_flags<true>(b, l, e, env, tid, nt);
    __syncthreads();
    uint4* wd = (uint4*)(b.agents + (size_t)e * b.A);
    const uint4* ws = (const uint4*)l.ag;
    for (int i = tid; i < b.A * (int)(sizeof(DAgent) / 16); i += nt) wd[i] = ws[i];
    if (b.dbg && tid == 0) b.dbg[(size_t)e * 16 + 6] = clock64();
  }
}

void mg_launch_fused(const KArgs& k, unsigned n, size_t lds, bool dcl, hipStream_t st) {
  if (dcl) hipLaunchKernelGGL(mg_step_fused_kernel<true>, dim3(n), dim3(MG_FUSED_THREADS), lds, st, k);
  else hipLaunchKernelGGL(mg_step_fused_kernel<false>, dim3(n), dim3(MG_FUSED_THREADS), lds, st, k);
}
size_t mg_fused_lds_bytes(const KArgs& k, bool dcl) {
  return mg_carve_fused(nullptr, k.A, k.SG, k.ROWB, MG_FUSED_THREADS / 64, dcl ? k.dc_obs_words : 0, k.NOBJ, k.HEAP,
                        k.NWINCELLS);
}
hipError_t mg_raise_lds_cap_fused(int cap) {
  hipError_t e = hipFuncSetAttribute((const void*)mg_step_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)mg_step_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
}
