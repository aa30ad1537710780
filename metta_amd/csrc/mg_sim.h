// mg_sim.h -- device-side simulation logic of the MettaGrid step path (included by mettagrid_hip.hip).
//
// One workgroup owns one environment.  All mutable per-env state (agent records, assembler records, the
// occupancy grid, per-step stat deltas) lives in LDS for the duration of a step; thread 0 resolves the actions
// in the reference's shuffled sequential order (exact by construction), then all waves encode observations.
//
// Every function cites the reference code whose result it must reproduce bit for bit
// (paths relative to packages/mettagrid/cpp/).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "mg_device_cfg.h"

// Grid cell encoding (uint16): 0 = empty, else 1 + object index in the env's unified object numbering
//   [0, NW) wall types | [NW, NW + A) agents | [NW + A, NW + A + NA) assembler instances of the map | chests.
// Walls carry no per-instance state, so a wall cell names its *type*; the static layer of a map (walls,
// assemblers) is shared by every environment that plays it, agents are overlaid per environment.
#define MG_MAX_OBJECTS 65534
struct CellCodec {
  int NW, A, NA;  // wall types, agents, assembler slots per map; chests follow the assemblers
  __host__ __device__ __forceinline__ int kind(uint16_t v) const {
    int o = (int)v - 1;
    return v == 0 ? MG_CELL_EMPTY
                  : (o < NW ? MG_CELL_WALL
                            : (o < NW + A ? MG_CELL_AGENT : (o < NW + A + NA ? MG_CELL_ASSEMBLER : MG_CELL_CHEST)));
  }
  __host__ __device__ __forceinline__ bool is_agent(uint16_t v) const { return (int)v > NW && (int)v <= NW + A; }
  __host__ __device__ __forceinline__ bool is_chest(uint16_t v) const { return (int)v > NW + A + NA; }
  __host__ __device__ __forceinline__ int agent_idx(uint16_t v) const { return (int)v - 1 - NW; }
  __host__ __device__ __forceinline__ int asm_idx(uint16_t v) const { return (int)v - 1 - NW - A; }
  __host__ __device__ __forceinline__ int chest_idx(uint16_t v) const { return (int)v - 1 - NW - A - NA; }
  __host__ __device__ __forceinline__ int wall_idx(uint16_t v) const { return (int)v - 1; }
  __host__ __device__ __forceinline__ uint16_t agent(int i) const { return (uint16_t)(1 + NW + i); }
  __host__ __device__ __forceinline__ uint16_t assembler(int i) const { return (uint16_t)(1 + NW + A + i); }
  __host__ __device__ __forceinline__ uint16_t chest(int i) const { return (uint16_t)(1 + NW + A + NA + i); }
  __host__ __device__ __forceinline__ uint16_t wall(int t) const { return (uint16_t)(1 + t); }
};

struct __align__(16) DAgent {  // 112 bytes; one per agent, [E][A] in HBM, staged in LDS during a step
  uint16_t r, c, prev_r, prev_c;
  int16_t frozen;
  uint8_t vibe, group;
  uint32_t swm;      // steps_without_motion (agent.hpp:39)
  float cur_reward;  // current_stat_reward (agent.hpp:36)
  uint32_t max_swm;  // mirror of stat status.max_steps_without_motion
  uint16_t amt[MGD_MAX_R];
  uint8_t order[MGD_MAX_R];  // unordered_map iteration order of the inventory
  uint8_t n_items, nb, next_resize, diversity;
  uint32_t presence;  // diversity-tracked resources currently held (bit per resource)
  // Hot per-step counters live in the record (it is streamed every step anyway) instead of the stat array:
  // action.<noop|move|change_vibe>.success / .failed; action.failed is their sum; max_swm above.
  // mg_episode_stats merges them back into the reference's key set.
  uint32_t succ[3], fail[3];
  float aligned;  // the collective's "aligned.agent" statistic: its members on this map (constant over an episode)
  uint32_t pad2;
};
static_assert(sizeof(DAgent) == (MGD_MAX_R == 16 ? 112 : 160), "DAgent layout");

struct __align__(16) DAsm {  // assembler.hpp:245-259
  uint32_t cooldown_end, cooldown_duration, uses_count, clipped;
};

struct __align__(16) DEnv {
  uint32_t step, mt_idx, seed, map_id;
  uint32_t episodes, early_reset, done, error;
};

struct Ctx {
  const int* __restrict__ dc;
  uint16_t* grid;
  DAgent* ag;
  DAsm* as;
  const int4* __restrict__ asm_info;  // per map: (r, c, cfg, obj_id)
  float* astats;                      // this env's [A][SA] stat slots in HBM (rare events add to them directly)
  uint32_t* atouched;                 // this env's [A][TW] explicit touched bits
  int TW;
  int* tmp;                           // [96] LDS scratch ints of the acting thread (lists of the assembler path)
  int* gd;                            // [SG] game stat deltas of this step
  uint32_t* dirty_amt;                // [A] bit r: <r>.amount was set this step
  uint32_t* dirty_misc;               // [A] bit0 diversity, bit1 max_steps_without_motion
  // Chests (chest.hpp): inventory holders like agents.  Holder index h < A is agent h, h >= A is chest h - A; a chest
  // record is a DAgent whose `group` names the pseudo group carrying its inventory configuration.
  DAgent* ch;
  const int4* __restrict__ chest_info;  // per map: (r, c, cfg, obj_id)
  int NC;
  int* gsetv;                         // [R] pending StatsTracker::set of chest.<r>.amount (last writer wins) ...
  uint32_t* gsetm;                    // ... and which of them are pending
  // Parallel action rounds (k_act_body.h): "last writer" means last in the reference's action order, not in time.  Each set
  // carries a sequence key (the acting agent's rank, then a per-action counter); the largest key wins (atomic max of
  // key << 16 | amount).  Null = the single-writer form above.
  unsigned long long* gset64;
  uint32_t seq;
  int H, W, A, NA, R, SA, SG, NG;
  int GW;                             // row stride of `grid` (W, or W + padding for the grids kept in HBM)
  uint32_t step;
  int asm_dirty;
  int overflow;                       // a fixed-size work list was too small (more than 24 chests around an assembler)
  MgAgentSlots AS;
  MgGameSlots GS;
  CellCodec cc;
};

// StatsTracker::add for the rarely touched per-agent keys (gained/lost, attack, transfer, swap, frozen ticks,
// invalid index): fire-and-forget f32 atomic on the HBM slot.  Counts are small integers, so the sum is exact and
// order independent.  The per-step hot counters live in the agent record instead (see DAgent).
__device__ __forceinline__ void mg_stat_add(Ctx& cx, int ai, int slot, int delta) {
  atomicAdd(&cx.astats[ai * cx.SA + slot], (float)delta);
  atomicOr(&cx.atouched[ai * cx.TW + (slot >> 5)], 1u << (slot & 31));
}

// _handle_invalid_action (bindings/mettagrid_c.cpp:565-570): "action.invalid_index" and "action.invalid_index.<id>",
// `passes` times (once per priority level, :599-608).  `row` = the agent's stat row in HBM; one thread per agent, so
// plain read-modify-writes (a key exists once its value is non-zero; see MG_INVALID_TABLE for ids without a slot).
__device__ inline void mg_count_invalid(float* row, const MgAgentSlots AS, int act, int n_actions, int passes) {
  row[AS.invalid()] += (float)passes;
  if (act == -1 || (act >= n_actions && act < n_actions + 7)) {
    row[AS.invalid_variant(act == -1 ? 0 : 1 + (act - n_actions))] += (float)passes;
    return;
  }
  for (int k = 0; k < MG_INVALID_TABLE; k++) {
    const float cnt = row[AS.invalid_table_count(k)];
    if (cnt == 0.f) {
      row[AS.invalid_table_id(k)] = __int_as_float(act);
      row[AS.invalid_table_count(k)] = (float)passes;
      return;
    }
    if (__float_as_int(row[AS.invalid_table_id(k)]) == act) {
      row[AS.invalid_table_count(k)] = cnt + (float)passes;
      return;
    }
  }
  row[AS.invalid_variant(8)] += (float)passes;
}

__device__ __forceinline__ DAgent* mg_holder(const Ctx& cx, int h) { return h < cx.A ? &cx.ag[h] : &cx.ch[h - cx.A]; }

// ---- small list helpers over the dcfg stream ----------------------------------------------------------------
__device__ __forceinline__ int L_n(const int* dc, int off) { return dc[off]; }
__device__ __forceinline__ int L_at(const int* dc, int off, int i) { return dc[off + 1 + i]; }
__device__ __forceinline__ int KV_k(const int* dc, int off, int i) { return dc[off + 1 + 2 * i]; }
__device__ __forceinline__ int KV_v(const int* dc, int off, int i) { return dc[off + 2 + 2 * i]; }
__device__ __forceinline__ const int* group_rec(const int* dc, int g) { return dc + dc[dc[DC_OFF_GROUPS] + g]; }
__device__ __forceinline__ const int* asm_rec(const int* dc, int a) { return dc + dc[dc[DC_OFF_ASMS] + a]; }
__device__ __forceinline__ const int* wall_rec(const int* dc, int w) { return dc + dc[dc[DC_OFF_WALLS] + w]; }

// ---- inventory: unordered_map<uint8,uint16> order emulation (hashtable.h:1888-1912) -------
__device__ inline int mg_next_bkt(DAgent* a, int n) {
  int r;
  if (n < 14) {
    if (n == 0) return 1;
    r = (int)((0xDDBBBB77553222ull >> (4 * n)) & 0xF);  // __fast_bkt[] = 2,2,2,3,5,5,7,7,11,11,11,11,13,13
  } else {
    r = n <= 17 ? 17 : n <= 19 ? 19 : n <= 23 ? 23 : n <= 29 ? 29 : n <= 31 ? 31 : n <= 37 ? 37 : n <= 41 ? 41
        : n <= 43 ? 43 : n <= 47 ? 47 : n <= 53 ? 53 : 59;
  }
  a->next_resize = (uint8_t)r;
  return r;
}
__device__ inline void mg_insert_node(DAgent* a, int k) {
  int nb = a->nb, b = k % nb, n = a->n_items, pos = 0;
  bool found = false;
  for (int i = 0; i < n; i++)
    if (a->order[i] % nb == b) {
      pos = i;
      found = true;
      break;
    }
  if (!found) pos = 0;
  for (int i = n; i > pos; i--) a->order[i] = a->order[i - 1];
  a->order[pos] = (uint8_t)k;
  a->n_items = (uint8_t)(n + 1);
}
__device__ inline void mg_rehash(DAgent* a, int m) {
  // the order list, packed eight entries per register (no private array -> no scratch)
  unsigned long long o0 = 0, o1 = 0;
#if MGD_MAX_R > 16
  unsigned long long o2 = 0, o3 = 0;
#endif
  int n = a->n_items;
  for (int i = 0; i < n; i++) {
    unsigned long long v = (unsigned long long)a->order[i] << (8 * (i & 7));
    if (i < 8) o0 |= v;
    else if (i < 16) o1 |= v;
#if MGD_MAX_R > 16
    else if (i < 24) o2 |= v;
    else o3 |= v;
#endif
  }
  a->n_items = 0;
  a->nb = (uint8_t)m;
  for (int i = 0; i < n; i++) {
    unsigned long long w = i < 8 ? o0 : o1;
#if MGD_MAX_R > 16
    if (i >= 16) w = i < 24 ? o2 : o3;
#endif
    mg_insert_node(a, (int)((w >> (8 * (i & 7))) & 0xFF));
  }
}
__device__ inline void mg_map_insert(DAgent* a, int k) {  // operator[] on an absent key
  int n_elt = a->n_items;
  if (n_elt + 1 > a->next_resize) {
    int min_bkts = max(n_elt + 1, a->next_resize ? 0 : 11);
    if (min_bkts >= a->nb)
      mg_rehash(a, mg_next_bkt(a, max(min_bkts + 1, a->nb * 2)));
    else
      a->next_resize = a->nb;
  }
  mg_insert_node(a, k);
}
__device__ inline void mg_map_erase(DAgent* a, int k) {
  int n = a->n_items, pos = -1;
  for (int i = 0; i < n; i++)
    if (a->order[i] == k) {
      pos = i;
      break;
    }
  if (pos < 0) return;
  for (int i = pos; i < n - 1; i++) a->order[i] = a->order[i + 1];
  a->n_items = (uint8_t)(n - 1);
}

// SharedInventoryLimit::effective_limit (inventory.hpp:21-33)
__device__ inline int mg_effective_limit(const int* dc, const int* gr, const DAgent* a, int li) {
  int base = dc[gr[GR_OFF_LIMITS] + 2 * li] & 0xFFFF;
  int eff = base;
#ifndef MG_NO_MODIFIERS
  if (gr[GR_HAS_MODIFIERS])
#else
  if (false)
#endif
  {
    int mo = dc[gr[GR_OFF_LIMITS] + 2 * li + 1];
    int n = L_n(dc, mo);
    for (int i = 0; i < n; i++) eff += (int)a->amt[KV_k(dc, mo, i)] * (KV_v(dc, mo, i) & 0xFFFF);
  }
  eff = eff < 0 ? 0 : (eff > 65535 ? 65535 : eff);
  return eff;
}
// SharedInventoryLimit::amount == sum of the amounts of the resources mapped to the limit (mod 2^16)
__device__ inline int mg_limit_sum(const int* dc, const int* gr, const DAgent* a, int li, int R) {
  int s = 0;
  for (uint32_t m = (uint32_t)dc[gr[GR_OFF_LIMIT_MASK] + li]; m; m &= m - 1) s += a->amt[__ffs(m) - 1];
  return s & 0xFFFF;
}
// Inventory::free_space (inventory.cpp:96-108)
__device__ inline int mg_free_space(const Ctx& cx, const DAgent* a, int item) {
  const int* gr = group_rec(cx.dc, a->group);
  int li = cx.dc[gr[GR_OFF_LIMIT_OF] + item];
  if (li < 0) return (65535 - a->amt[item]) & 0xFFFF;
  int used = mg_limit_sum(cx.dc, gr, a, li, cx.R), eff = mg_effective_limit(cx.dc, gr, a, li);
  return eff > used ? eff - used : 0;
}

// Agent::on_inventory_change (agent.cpp:74-101)
__device__ inline void mg_on_change(Ctx& cx, int ai, int item, int delta) {
  if (ai >= cx.A) {  // Chest::on_inventory_change (chest.hpp:56-63): game stats
    if (delta > 0)
      atomicAdd(&cx.gd[cx.GS.chest_deposited(item)], delta);
    else if (delta < 0)
      atomicAdd(&cx.gd[cx.GS.chest_withdrawn(item)], -delta);
    if (cx.gset64) {
      const unsigned long long key = cx.seq++;
      atomicMax(&cx.gset64[item], (key << 16) | (unsigned long long)cx.ch[ai - cx.A].amt[item]);
    } else {
      cx.gsetv[item] = (int)cx.ch[ai - cx.A].amt[item];
      *cx.gsetm |= 1u << item;
    }
    return;
  }
  DAgent* a = &cx.ag[ai];
  if (delta > 0)
    mg_stat_add(cx, ai, cx.AS.gained(item), delta);
  else
    mg_stat_add(cx, ai, cx.AS.lost(item), -delta);
  cx.dirty_amt[ai] |= (1u << item);
  const int* gr = group_rec(cx.dc, a->group);
  if ((uint32_t)gr[GR_DIVERSITY_MASK] & (1u << item)) {
    bool had = (a->presence >> item) & 1, has = a->amt[item] > 0;
    if (had != has) {
      a->presence = a->presence ^ (1u << item);
      a->diversity = (uint8_t)(a->diversity + (has ? 1 : -1));
      cx.dirty_misc[ai] |= 1u;
    }
  }
}

template <int D>
__device__ int mg_inv_update(Ctx& cx, int ai, int item, int attempted_delta, bool ignore_limits);

// Inventory::enforce_all_limits (inventory.cpp:126-158); nesting bounded at depth 3.
// A translation unit built with MG_NO_MODIFIERS serves configurations without limit modifiers only (host-checked): the
// four nested copies of the update code below cost every caller their registers, and those configurations never get here.
template <int D>
__device__ void mg_enforce_limits(Ctx& cx, int ai) {
#ifdef MG_NO_MODIFIERS
  return;
#endif
  DAgent* a = mg_holder(cx, ai);
  const int* dc = cx.dc;
  const int* gr = group_rec(dc, a->group);
  int nl = gr[GR_N_LIMITS], oo = gr[GR_OFF_LIMITS_ORDER], no = L_n(dc, oo);
  for (int li = 0; li < nl; li++) {
    int excess = mg_limit_sum(dc, gr, a, li, cx.R) - mg_effective_limit(dc, gr, a, li);
    if (excess <= 0) continue;
    for (int i = 0; i < no; i++) {
      if (KV_v(dc, oo, i) != li) continue;
      int item = KV_k(dc, oo, i);
      int to_drop = min((int)a->amt[item], excess);
      if (to_drop > 0) {
        if constexpr (D < 3) mg_inv_update<D + 1>(cx, ai, item, -to_drop, false);
        excess = mg_limit_sum(dc, gr, a, li, cx.R) - mg_effective_limit(dc, gr, a, li);
      }
      if (excess <= 0) break;
    }
  }
}

// Inventory::update (inventory.cpp:37-85)
template <int D>
__device__ int mg_inv_update(Ctx& cx, int ai, int item, int attempted_delta, bool ignore_limits) {
  DAgent* a = mg_holder(cx, ai);
  const int* dc = cx.dc;
  const int* gr = group_rec(dc, a->group);
  int initial = a->amt[item];
  if (initial == 0) mg_map_insert(a, item);
  int new_amount = initial + attempted_delta;
  int mx = 65535;
  int li = dc[gr[GR_OFF_LIMIT_OF] + item];
  if (!ignore_limits && li >= 0) {
    int eff = mg_effective_limit(dc, gr, a, li);
    int used_by_others = mg_limit_sum(dc, gr, a, li, cx.R) - initial;
    if (used_by_others < 0) used_by_others = 0;
    int max_int = eff - used_by_others;
    if (max_int < 0) max_int = 0;
    mx = max_int & 0xFFFF;
  }
  int clamped = (new_amount < 0) ? 0 : ((mx < new_amount) ? mx : new_amount);
  clamped &= 0xFFFF;
  a->amt[item] = (uint16_t)clamped;
  if (clamped == 0) mg_map_erase(a, item);
  int cd = clamped - initial;
  if (cd != 0) mg_on_change(cx, ai, item, cd);
#ifndef MG_NO_MODIFIERS
  if (cd < 0 && dc[gr[GR_OFF_MODIFIER] + item]) mg_enforce_limits<D>(cx, ai);
#endif
  return cd;
}
__device__ inline int inv_update(Ctx& cx, int ai, int item, int delta, bool ignore = false) {
  return mg_inv_update<0>(cx, ai, item, delta, ignore);
}

// HasInventory::shared_update (has_inventory.cpp:7-73) over the inventories (holder indices) list[0..n), n <= 32.
// The reference's "inventories to consider" vectors are sub-sequences of the list in list order, so they are kept as
// bitmasks over list positions.
__device__ inline int mg_shared_update(Ctx& cx, const int* list, int n, int item, int delta) {
  if (n == 0) return 0;
  // one inventory: both branches of the algorithm below end in a single update by `delta` (clamped by the limits)
  if (n == 1) return inv_update(cx, list[0], item, delta);
  uint32_t consider, next = (1u << n) - 1u;
  int delta_remaining = delta, num_remaining = n;
  int per = delta_remaining / num_remaining;
  do {
    consider = next;
    next = 0;
    for (uint32_t m = consider; m; m &= m - 1) {
      int pos = __ffs(m) - 1, ai = list[pos];
      bool now;
      if (delta_remaining > 0)
        now = mg_free_space(cx, mg_holder(cx, ai), item) <= per;
      else
        now = (int)mg_holder(cx, ai)->amt[item] <= -per;
      if (now) {
        delta_remaining -= inv_update(cx, ai, item, per);
        num_remaining--;
        if (num_remaining > 0) per = delta_remaining / num_remaining;
      } else {
        next |= 1u << pos;
      }
    }
  } while (consider != next);
  if (num_remaining == 0) return delta - delta_remaining;
  // reverse order, so that earlier inventories get the remainder
  int i = __popc(consider) - 1;
  for (uint32_t m = consider; m; i--) {
    int pos = 31 - __clz(m);
    m &= ~(1u << pos);
    int d = delta_remaining / (i + 1);
    delta_remaining -= inv_update(cx, list[pos], item, d);
  }
  return delta - delta_remaining;
}

// HasInventory::transfer_resources (has_inventory.cpp:76-108) between two holders.
__device__ inline int mg_transfer_resources(Ctx& cx, int src, int dst, int item, int delta, bool destroy_untransferred) {
  if (delta <= 0) return 0;
  int max_give = min((int)mg_holder(cx, src)->amt[item], delta);
  int max_recv = mg_free_space(cx, mg_holder(cx, dst), item);
  int amount = min(max_give, max_recv);
  int loss = destroy_untransferred ? max_give : amount;
  inv_update(cx, src, item, -loss);
  inv_update(cx, dst, item, amount);
  return amount;
}

// Chest::onUse (chest.hpp:31-54,103-122): the actor's vibe selects (item, delta) pairs; positive = deposit.
__device__ inline bool mg_chest_use(Ctx& cx, int ai, int chest_idx) {
  const int* dc = cx.dc;
  const int* rec = dc + dc[dc[DC_OFF_CHESTS] + cx.chest_info[chest_idx].z];
  const int ox = rec[CR_OFF_XFER], nx = dc[ox];
  if (nx == 0) return false;
  int deltas = -1;
  const int vibe = cx.ag[ai].vibe;
  for (int i = 0; i < nx; i++)
    if (dc[ox + 1 + 2 * i] == vibe) deltas = dc[ox + 2 + 2 * i];
  if (deltas < 0) return false;
  bool any = false;
  const int n = L_n(dc, deltas), h = cx.A + chest_idx;
  for (int i = 0; i < n; i++) {
    const int item = KV_k(dc, deltas, i), v = KV_v(dc, deltas, i);
    if (v > 0) {
      int t = mg_transfer_resources(cx, ai, h, item, v, true);
      if (t > 0) {
        any = true;
        mg_stat_add(cx, ai, cx.AS.chest_deposited(item), t);
      }
    } else if (v < 0) {
      int t = mg_transfer_resources(cx, h, ai, item, -v, true);
      if (t > 0) {
        any = true;
        mg_stat_add(cx, ai, cx.AS.chest_withdrawn(item), t);
      }
    }
  }
  return any;
}

// ---- assembler ---------------------------------------------------------------------------------------------
// Protocol lookup (assembler.hpp:378-408) from the vibes of the surrounding agents (any order; zeros ignored) and
// their number (byte q of `packed_vibes` = vibe of the q-th surrounding agent).  Returns the record offset or -1.
// `clipped` = DAsm::clipped: 0 = no; 1 = clipped with no way out (a station that starts clipped without a Clipper);
// 2 + k = clipped by the Clipper, which installed its k-th unclipping protocol (clipper.hpp:186-193).
__device__ inline int mg_select_protocol(const int* dc, int cfg, unsigned long long packed_vibes, int n_agents,
                                         uint32_t clipped) {
#ifdef MG_NO_CLIPPER  // (kernels that never run a configuration with a Clipper: k_fused, k_act_fast, k_uses)
  if (clipped) return -1;
#else
  if (clipped == 1u) return -1;
#endif
  // key = the non-zero vibes sorted ascending, 8 bits each (assembler.hpp:326-331): selection sort on the packed bytes
  unsigned long long key = 0;
  uint32_t used = 0;
  for (int t = 0; t < 8; t++) {
    int best = -1, bv = 256;
    for (int q = 0; q < n_agents && q < 8; q++) {
      int v = (int)((packed_vibes >> (8 * q)) & 0xFF);
      if (!((used >> q) & 1u) && v != 0 && v < bv) {
        bv = v;
        best = q;
      }
    }
    if (best < 0) break;
    used |= 1u << best;
    key = (key << 8) | (unsigned long long)bv;
  }
#ifndef MG_NO_CLIPPER
  if (clipped >= 2u) {
    // unclip_protocols = build_protocol_map({that protocol}): found under the local vibe, else under 0
    const int* t = dc + dc[DC_OFF_UNCLIP] + 3 * (int)(clipped - 2u);
    const unsigned long long kk = ((unsigned long long)(uint32_t)t[2] << 32) | (uint32_t)t[1];
    return ((kk == key || kk == 0ull) && dc[t[0] + PR_MIN_AGENTS] <= n_agents) ? t[0] : -1;
  }
#endif
  const int* rec = asm_rec(dc, cfg);
  int nk = rec[AR_N_KEYS], ok = rec[AR_OFF_KEYS], pt = rec[AR_OFF_PTABLE];
  for (int attempt = 0; attempt < 2; attempt++) {
    unsigned long long k = attempt == 0 ? key : 0ull;
    for (int q = 0; q < nk; q++) {
      unsigned long long kk = ((unsigned long long)(uint32_t)dc[ok + 3 * q + 1] << 32) | (uint32_t)dc[ok + 3 * q];
      if (kk != k) continue;
      int lo = dc[ok + 3 * q + 2], n = L_n(dc, lo);
      for (int i = 0; i < n; i++) {
        int po = dc[pt + L_at(dc, lo, i)];
        if (dc[po + PR_MIN_AGENTS] <= n_agents) return po;
      }
    }
  }
  return -1;
}

// Assembler::get_current_protocol on the grid as it stands (assembler.hpp:353-408).
template <class CT>
__device__ inline int mg_current_protocol(const int* dc, const CT* grid, const DAgent* ag, CellCodec codec, int H,
                                          int W, int GW, int ar, int ac, int cfg, uint32_t clipped) {
  unsigned long long vibes = 0;
  int n_agents = 0;
  if (!grid) {  // no occupancy tile (fused kernel): the surrounding agents by position (any order: the key is sorted)
    for (int j = 0; j < codec.A; j++) {
      const int dr = (int)ag[j].r - ar, dcl = (int)ag[j].c - ac;
      if (dr >= -1 && dr <= 1 && dcl >= -1 && dcl <= 1 && !(dr == 0 && dcl == 0)) {
        if (n_agents < 8) vibes |= (unsigned long long)ag[j].vibe << (8 * n_agents);
        n_agents++;
      }
    }
    return mg_select_protocol(dc, cfg, vibes, n_agents, clipped);
  }
  for (int i = -1; i <= 1; i++)
    for (int j = -1; j <= 1; j++) {
      if (i == 0 && j == 0) continue;
      int rr = ar + i, cc = ac + j;
      if (rr < 0 || cc < 0 || rr >= H || cc >= W) continue;
      uint16_t cv = grid[rr * GW + cc];
      if (codec.is_agent(cv)) {
        vibes |= (unsigned long long)ag[codec.agent_idx(cv)].vibe << (8 * n_agents);
        n_agents++;
      }
    }
  return mg_select_protocol(dc, cfg, vibes, n_agents, clipped);
}

__device__ inline int mg_kv_lookup(const int* dc, int off, int k) {
  int n = L_n(dc, off);
  for (int i = 0; i < n; i++)
    if (KV_k(dc, off, i) == k) return KV_v(dc, off, i);
  return 0;
}

// Assembler::onUse (assembler.hpp:455-526) given the surrounding agents `near[0..nn)` in the reference's order
// (NW..SE ring rotated to start at the actor, assembler.hpp:48-92) and their vibes at the time of use.
// `inputs[0..n_inputs)` = the holders the inputs are drawn from: the surrounding agents, then the chests within
// `chest_search_distance` (assembler.hpp:100-122); null = the surrounding agents only.
__device__ inline bool mg_assembler_use_core(Ctx& cx, int ai_actor, int asm_idx, const int* near, const uint8_t* near_vibes,
                                             int nn, const int* inputs = nullptr, int n_inputs = 0) {
  if (!inputs) {
    inputs = near;
    n_inputs = nn;
  }
  const int* dc = cx.dc;
  // (the record may live in HBM -- fused kernel -- so it is read once into registers and written back on success)
  DAsm* const as_mem = &cx.as[asm_idx];
  DAsm as_val = *as_mem;
  DAsm* as = &as_val;
  int4 info = cx.asm_info[asm_idx];
  const int* rec = asm_rec(dc, info.z);
  uint32_t max_uses = (uint32_t)rec[AR_MAX_USES];
  if (max_uses > 0 && as->uses_count >= max_uses) return false;
  uint32_t remaining = as->cooldown_end <= cx.step ? 0 : as->cooldown_end - cx.step;
  bool partial = rec[AR_ALLOW_PARTIAL] != 0;
  if (remaining > 0 && !partial) return false;
  unsigned long long pv = 0;
  for (int q = 0; q < nn && q < 8; q++) pv |= (unsigned long long)near_vibes[q] << (8 * q);
  // a lone protocol under key 0 that needs no agents is the current protocol whoever stands around (AR_SIMPLE_PO)
  int po = (rec[AR_SIMPLE_PO] >= 0 && as->clipped == 0) ? rec[AR_SIMPLE_PO]
                                                        : mg_select_protocol(dc, info.z, pv, nn, as->clipped);
  if (po < 0) return false;
  // protocol_to_use as (item, amount) lists
  int* in_k = cx.tmp + 16; int* in_v = cx.tmp + 32; int* out_k = cx.tmp + 48; int* out_v = cx.tmp + 64;
  int n_in, n_out;
  int oin = dc[po + PR_OFF_IN], oout = dc[po + PR_OFF_OUT];
  bool orig_pos_out = false;
  {
    int n = L_n(dc, oout);
    for (int i = 0; i < n; i++)
      if (KV_v(dc, oout, i) > 0) orig_pos_out = true;
  }
  if (remaining > 0 && partial) {
    uint32_t duration = as->cooldown_duration, elapsed = duration - remaining;
    int ois = dc[po + PR_OFF_IN_S], oos = dc[po + PR_OFF_OUT_S];
    n_in = L_n(dc, ois);
    for (int i = 0; i < n_in; i++) {
      int k = L_at(dc, ois, i);
      in_k[i] = k;
      in_v[i] = (int)((((uint32_t)mg_kv_lookup(dc, oin, k) * elapsed + duration - 1) / duration) & 0xFFFF);
    }
    n_out = L_n(dc, oos);
    bool pos_out = false;
    for (int i = 0; i < n_out; i++) {
      int k = L_at(dc, oos, i);
      out_k[i] = k;
      out_v[i] = (int)((((uint32_t)mg_kv_lookup(dc, oout, k) * elapsed) / duration) & 0xFFFF);
      if (out_v[i] > 0) pos_out = true;
    }
    if (!pos_out && orig_pos_out && !as->clipped) return false;
  } else {
    n_in = L_n(dc, oin);
    for (int i = 0; i < n_in; i++) {
      in_k[i] = KV_k(dc, oin, i);
      in_v[i] = KV_v(dc, oin, i);
    }
    n_out = L_n(dc, oout);
    for (int i = 0; i < n_out; i++) {
      out_k[i] = KV_k(dc, oout, i);
      out_v[i] = KV_v(dc, oout, i);
    }
  }
  // can_afford_protocol: uint16 totals (assembler.hpp:124-137)
  for (int i = 0; i < n_in; i++) {
    int total = 0;
    for (int q = 0; q < n_inputs; q++) total = (total + mg_holder(cx, inputs[q])->amt[in_k[i]]) & 0xFFFF;
    if (total < (in_v[i] & 0xFFFF)) return false;
  }
  // get_output_inventories (assembler.hpp:198-223)
  int* outs = cx.tmp + 80;
  int no = 0;
  int ov = dc[po + PR_OFF_VIBES], nvib = L_n(dc, ov);
  if (nvib <= 1) {
    outs[no++] = ai_actor;
  } else {
    uint32_t open_slots = (1u << (nvib < 8 ? nvib : 8)) - 1u;  // required vibes not yet matched (one bit per entry)
    for (int q = 0; q < nn; q++) {
      uint8_t v = near_vibes[q];
      if (v == 0) continue;
      int slot = -1;
      for (int i = 0; i < nvib && i < 8; i++)
        if (L_at(dc, ov, i) == v && ((open_slots >> i) & 1u)) {
          slot = i;
          break;
        }
      if (slot < 0) continue;
      open_slots &= ~(1u << slot);
      outs[no++] = near[q];
      if (no >= nvib) break;
    }
    if (no == 0) outs[no++] = ai_actor;
  }
  // can_receive_output (assembler.hpp:143-171)
  {
    bool pos_out = false;
    for (int i = 0; i < n_out; i++)
      if (out_v[i] > 0) pos_out = true;
    bool ok = true;
    if (pos_out) {
      ok = false;
      for (int i = 0; i < n_out && !ok; i++)
        if (out_v[i] > 0) {
          int total_free = 0;
          for (int q = 0; q < no; q++) total_free = (total_free + mg_free_space(cx, &cx.ag[outs[q]], out_k[i])) & 0xFFFF;
          if (total_free >= 1) ok = true;
        }
    }
    if (!ok && !as->clipped) return false;
  }
  for (int i = 0; i < n_in; i++) mg_shared_update(cx, inputs, n_inputs, in_k[i], -in_v[i]);
  for (int i = 0; i < n_out; i++) {
    int distributed = mg_shared_update(cx, outs, no, out_k[i], out_v[i]);
    if (distributed > 0) atomicAdd(&cx.gd[cx.GS.assembler_created(out_k[i])], distributed);  // LDS; uses may run on several lanes
  }
  as->cooldown_duration = (uint32_t)dc[po + PR_COOLDOWN];
  as->cooldown_end = cx.step + as->cooldown_duration;
  if (as->clipped)
    as->clipped = 0;
  else
    as->uses_count++;
  *as_mem = as_val;
  cx.asm_dirty = 1;
  return true;
}


// Assembler::onUse on the grid as it stands (general path): gather the surrounding agents, then the core.
__device__ inline bool mg_assembler_use(Ctx& cx, int ai_actor, int asm_idx) {
  int4 info = cx.asm_info[asm_idx];
  int ar = info.x, ac = info.y;
  const DAgent* actor = &cx.ag[ai_actor];
  // in-map ring positions NW,N,NE,W,E,SW,S,SE (assembler.hpp:30-45) as a bitmask; `start` = the actor's index
  uint32_t vmask = 0;
  int np = 0, start = -1;
  for (int q = 0; q < 8; q++) {
    int o9 = q < 4 ? q : q + 1;
    int rr = ar + o9 / 3 - 1, cc = ac + o9 % 3 - 1;
    if (rr < 0 || cc < 0 || rr >= cx.H || cc >= cx.W) continue;
    if (rr == actor->r && cc == actor->c && start < 0) start = np;
    vmask |= 1u << q;
    np++;
  }
  if (start < 0) return false;  // reference throws; cannot happen for a move into the assembler
  int* near = cx.tmp;
  uint8_t* vibes = (uint8_t*)(cx.tmp + 8);
  int nn = 0;
  for (int t = 0; t < np; t++) {
    int want = (start + t) % np;  // index among the in-map positions
    int q = 0;
    for (int seen = 0; q < 8; q++)
      if ((vmask >> q) & 1u) {
        if (seen == want) break;
        seen++;
      }
    int o9 = q < 4 ? q : q + 1;
    uint16_t cv = cx.grid[(ar + o9 / 3 - 1) * cx.GW + (ac + o9 % 3 - 1)];
    if (cx.cc.is_agent(cv)) {
      near[nn] = cx.cc.agent_idx(cv);
      vibes[nn] = cx.ag[near[nn]].vibe;
      nn++;
    }
  }
  const int dist = cx.NC > 0 ? asm_rec(cx.dc, info.z)[AR_CHEST_DIST] : 0;
  if (dist <= 0) return mg_assembler_use_core(cx, ai_actor, asm_idx, near, vibes, nn);
  // chests within the search distance, row-major (assembler.hpp:100-122); the list lives behind the use scratch
  int* inputs = cx.tmp + 96;
  int n_inputs = 0;
  for (int q = 0; q < nn; q++) inputs[n_inputs++] = near[q];
  for (int dr = -dist; dr <= dist; dr++)
    for (int dcl = -dist; dcl <= dist; dcl++) {
      if (dr == 0 && dcl == 0) continue;
      int rr = ar + dr, c2 = ac + dcl;
      if (rr < 0 || c2 < 0 || rr >= cx.H || c2 >= cx.W) continue;
      uint16_t cv = cx.grid[rr * cx.GW + c2];
      if (cx.cc.is_chest(cv) && n_inputs < 32) inputs[n_inputs++] = cx.A + cx.cc.chest_idx(cv);
      else if (cx.cc.is_chest(cv)) cx.overflow = 1;
    }
  return mg_assembler_use_core(cx, ai_actor, asm_idx, near, vibes, nn, inputs, n_inputs);
}

// ---- attack / transfer (attack.hpp:93-253, transfer.hpp:72-150) ----------------------------------------------
__device__ inline bool mg_try_attack(Ctx& cx, int ai, uint16_t target_cell) {
  const int* dc = cx.dc;
  if (!cx.cc.is_agent(target_cell)) return false;
  int ti = cx.cc.agent_idx(target_cell);
  DAgent* actor = &cx.ag[ai];
  DAgent* target = &cx.ag[ti];
  if (target->frozen > 0) return false;
  int oc = dc[DC_OFF_ATK_CONSUMED], nc = L_n(dc, oc);
  for (int i = 0; i < nc; i++)
    if ((int)actor->amt[KV_k(dc, oc, i)] < (KV_v(dc, oc, i) & 0xFFFF)) return false;
  int ag = group_rec(dc, actor->group)[GR_NAME_IDX], tg = group_rec(dc, target->group)[GR_NAME_IDX];
  bool blocked = false;
  if (dc[DC_HAS_DEFENSE]) {
    int weapon = 0, armor = 0;
    int ow = dc[DC_OFF_ATK_WEAPON], nw = L_n(dc, ow);
    for (int i = 0; i < nw; i++) weapon += (int)actor->amt[KV_k(dc, ow, i)] * (KV_v(dc, ow, i) & 0xFFFF);
    int oa = dc[DC_OFF_ATK_ARMOR], na = L_n(dc, oa);
    for (int i = 0; i < na; i++) {
      int item = KV_k(dc, oa, i);
      int amount = target->amt[item];
      bool vibing = target->vibe != 0 && (int)target->vibe < dc[DC_N_VIBES] && item < cx.R &&
                    dc[dc[DC_OFF_VIBE_RESOURCE] + target->vibe] == item;
      if (vibing) amount += dc[dc[DC_OFF_ATK_VIBE_BONUS] + target->vibe];
      armor += amount * (KV_v(dc, oa, i) & 0xFFFF);
    }
    int bonus = max(weapon - armor, 0);
    int od = dc[DC_OFF_ATK_DEFENSE], nd = L_n(dc, od);
    bool can = true;
    for (int i = 0; i < nd; i++) {
      int required = (KV_v(dc, od, i) & 0xFFFF) + bonus;
      if ((int)target->amt[KV_k(dc, od, i)] < (required & 0xFFFF)) {
        can = false;
        break;
      }
    }
    if (can) {
      // _consume_defense_resources recomputes the powers before consuming; nothing changed in between.
      for (int i = 0; i < nd; i++) inv_update(cx, ti, KV_k(dc, od, i), -((KV_v(dc, od, i) & 0xFFFF) + bonus));
      mg_stat_add(cx, ai, cx.AS.attack_blocked_by(tg), 1);
      blocked = true;
    }
  }
  if (!blocked) {
    if (dc[DC_ATK_FREEZE] > 0) target->frozen = (int16_t)dc[DC_ATK_FREEZE];
    int o1 = dc[DC_OFF_ATK_ACTOR], n1 = L_n(dc, o1);
    for (int i = 0; i < n1; i++) inv_update(cx, ai, KV_k(dc, o1, i), KV_v(dc, o1, i));
    int o2 = dc[DC_OFF_ATK_TARGET], n2 = L_n(dc, o2);
    for (int i = 0; i < n2; i++) inv_update(cx, ti, KV_k(dc, o2, i), KV_v(dc, o2, i));
    int ol = dc[DC_OFF_ATK_LOOT], nl = L_n(dc, ol);
    for (int i = 0; i < nl; i++) {
      int item = L_at(dc, ol, i);
      int amount = target->amt[item];
      if (amount > 0) {
        int stolen = inv_update(cx, ai, item, amount);
        inv_update(cx, ti, item, -stolen);
      }
    }
    if (ag == tg) {
      mg_stat_add(cx, ai, cx.AS.attack_friendly_fire(), 1);
    } else {
      mg_stat_add(cx, ai, cx.AS.attack_hit(tg), 1);
      mg_stat_add(cx, ti, cx.AS.attack_hit_by(ag), 1);
    }
  }
  for (int i = 0; i < nc; i++)
    if (KV_v(dc, oc, i) > 0) inv_update(cx, ai, KV_k(dc, oc, i), -(KV_v(dc, oc, i) & 0xFFFF));
  return true;
}

__device__ inline bool mg_try_transfer(Ctx& cx, int ai, uint16_t target_cell) {
  const int* dc = cx.dc;
  DAgent* actor = &cx.ag[ai];
  int orq = dc[DC_OFF_XFER_REQUIRED], nrq = L_n(dc, orq);
  for (int i = 0; i < nrq; i++)
    if ((int)actor->amt[KV_k(dc, orq, i)] < (KV_v(dc, orq, i) & 0xFFFF)) return false;
  if (!cx.cc.is_agent(target_cell)) return false;
  int ti = cx.cc.agent_idx(target_cell);
  DAgent* target = &cx.ag[ti];
  if (target->frozen > 0) return false;
  int oe = dc[DC_OFF_XFER_EFFECTS], ne = dc[oe];
  int oa = -1, ot = -1;
  for (int i = 0; i < ne; i++)
    if (dc[oe + 1 + 3 * i] == actor->vibe) {
      oa = dc[oe + 2 + 3 * i];
      ot = dc[oe + 3 + 3 * i];
    }
  if (oa < 0) return false;
  int tg = group_rec(dc, target->group)[GR_NAME_IDX];
  int na = L_n(dc, oa), nt = L_n(dc, ot);
  for (int i = 0; i < na; i++) {
    int d = KV_v(dc, oa, i);
    if (d < 0 && (int)actor->amt[KV_k(dc, oa, i)] < -d) return false;
  }
  for (int i = 0; i < nt; i++) {
    int d = KV_v(dc, ot, i);
    if (d < 0 && (int)target->amt[KV_k(dc, ot, i)] < -d) return false;
  }
  for (int i = 0; i < na; i++) {
    int d = KV_v(dc, oa, i);
    if (d > 0 && d > mg_free_space(cx, actor, KV_k(dc, oa, i))) return false;
  }
  for (int i = 0; i < nt; i++) {
    int d = KV_v(dc, ot, i);
    if (d > 0 && d > mg_free_space(cx, target, KV_k(dc, ot, i))) return false;
  }
  for (int i = 0; i < na; i++) {
    int d = KV_v(dc, oa, i), item = KV_k(dc, oa, i);
    if (d != 0) {
      int actual = inv_update(cx, ai, item, d);
      if (actual != 0) mg_stat_add(cx, ai, cx.AS.transfer_self(item), actual);
    }
  }
  for (int i = 0; i < nt; i++) {
    int d = KV_v(dc, ot, i), item = KV_k(dc, ot, i);
    if (d != 0) {
      int actual = inv_update(cx, ti, item, d);
      if (actual != 0) mg_stat_add(cx, ai, cx.AS.transfer_to(tg, item), actual);
    }
  }
  mg_stat_add(cx, ai, cx.AS.transfer_count(), 1);
  return true;
}

// Move::_handle_action (move.hpp:74-141)
__device__ inline bool mg_do_move(Ctx& cx, int ai, int orientation) {
  const int DX[8] = {0, 0, -1, 1, -1, 1, -1, 1};
  const int DY[8] = {-1, 1, 0, 0, -1, -1, 1, 1};
  DAgent* actor = &cx.ag[ai];
  int tr = (int)actor->r + DY[orientation], tc = (int)actor->c + DX[orientation];
  if (tr < 0 || tc < 0 || tr >= cx.H || tc >= cx.W) return false;
  uint16_t target = cx.grid[tr * cx.GW + tc];
  int vh = cx.dc[cx.dc[DC_OFF_VIBE_HANDLER] + actor->vibe];
  if (vh == MG_VH_ATTACK) {
    if (mg_try_attack(cx, ai, target)) return true;
  } else if (vh == MG_VH_TRANSFER) {
    if (mg_try_transfer(cx, ai, target)) return true;
  }
  int kind = cx.cc.kind(target);
  if (kind == MG_CELL_EMPTY) {
    cx.grid[tr * cx.GW + tc] = cx.cc.agent(ai);
    cx.grid[(int)actor->r * cx.GW + actor->c] = 0;
    actor->r = (uint16_t)tr;
    actor->c = (uint16_t)tc;
    return true;
  }
  if (kind == MG_CELL_AGENT) {
    int ti = cx.cc.agent_idx(target);
    DAgent* ta = &cx.ag[ti];
    if (ta->frozen > 0) {
      cx.grid[(int)actor->r * cx.GW + actor->c] = cx.cc.agent(ti);
      cx.grid[tr * cx.GW + tc] = cx.cc.agent(ai);
      ta->r = actor->r;
      ta->c = actor->c;
      actor->r = (uint16_t)tr;
      actor->c = (uint16_t)tc;
      mg_stat_add(cx, ai, cx.AS.swap(), 1);
      return true;
    }
    return false;  // Agent::onUse (agent.cpp:179-185)
  }
  if (kind == MG_CELL_ASSEMBLER) return mg_assembler_use(cx, ai, cx.cc.asm_idx(target));
  if (kind == MG_CELL_CHEST) return mg_chest_use(cx, ai, cx.cc.chest_idx(target));
  return false;  // walls are not Usable
}

// ActionHandler::handle_action (action_handler.hpp:99-153)
__device__ inline bool mg_handle_action(Ctx& cx, int ai, int kind, int arg) {
  DAgent* actor = &cx.ag[ai];
  const int* dc = cx.dc;
  if (actor->frozen != 0) {
    mg_stat_add(cx, ai, cx.AS.frozen_ticks(), 1);
    mg_stat_add(cx, ai, cx.AS.frozen_ticks_group(), 1);
    if (actor->frozen > 0) actor->frozen -= 1;
    return false;
  }
  int orq = dc[DC_OFF_HANDLERS + 2 * kind], nrq = L_n(dc, orq);
  bool has = true;
  for (int i = 0; i < nrq; i++)
    if ((int)actor->amt[KV_k(dc, orq, i)] < (KV_v(dc, orq, i) & 0xFFFF)) {
      has = false;
      break;
    }
  bool success = false;
  if (has) {
    if (kind == MG_ACT_NOOP)
      success = true;
    else if (kind == MG_ACT_MOVE)
      success = mg_do_move(cx, ai, arg);
    else {
      actor->vibe = (uint8_t)arg;
      success = true;
    }
  }
  if (actor->r == actor->prev_r && actor->c == actor->prev_c) {
    actor->swm += 1;
    if (actor->swm > actor->max_swm) actor->max_swm = actor->swm;  // stat status.max_steps_without_motion
  } else {
    actor->swm = 0;
  }
  actor->prev_r = actor->r;
  actor->prev_c = actor->c;
  if (success) {
    actor->succ[kind] += 1;
    int oc = dc[DC_OFF_HANDLERS + 2 * kind + 1], nc = L_n(dc, oc);
    for (int i = 0; i < nc; i++)
      if (KV_v(dc, oc, i) > 0) inv_update(cx, ai, KV_k(dc, oc, i), -(KV_v(dc, oc, i) & 0xFFFF));
  } else {
    actor->fail[kind] += 1;  // action.failed == sum of the per-handler failures
  }
  return success;
}

// ---- RNG: std::mt19937 tempering + libstdc++ Lemire bounded draw ---------------------------
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t c) {  // new = c ^ twist(a, b)
  uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
struct RngWin {  // window of pre-tempered outputs staged in LDS
  const uint32_t* win;
  int n, used;
  int overflow;
  __device__ uint32_t next() {
    if (used >= n) {
      overflow = 1;
      return 0;
    }
    return win[used++];
  }
  __device__ double canonical() {  // std::generate_canonical<double, 53> over a 32-bit engine: two outputs
    double a = (double)next();
    double b = (double)next();
    double r = (a + b * 4294967296.0) / 18446744073709551616.0;
    if (r >= 1.0) r = 0.99999999999999988897769753748;  // nextafter(1.0, 0.0)
    return r;
  }
  __device__ uint32_t bounded(uint32_t range) {  // uniform_int_dist.h:258-269
    unsigned long long product = (unsigned long long)next() * range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
      uint32_t threshold = (0u - range) % range;
      while (low < threshold && !overflow) {
        product = (unsigned long long)next() * range;
        low = (uint32_t)product;
      }
    }
    return (uint32_t)(product >> 32);
  }
};
// std::shuffle of iota(n) (stl_algo.h:3726-3792)
__device__ inline void mg_shuffle(uint16_t* v, int n, RngWin& g) {
  for (int i = 0; i < n; i++) v[i] = (uint16_t)i;
  if (n == 0) return;
  // (0xFFFFFFFF / n) >= n holds for every n <= 65535
  int i = 1;
  if ((n & 1) == 0) {
    uint32_t j = g.bounded(2);
    uint16_t t = v[1];
    v[1] = v[j];
    v[j] = t;
    i = 2;
  }
  while (i != n) {
    uint32_t s = (uint32_t)i + 1;
    uint32_t x = g.bounded(s * (s + 1));
    uint32_t a = x / (s + 1), b = x % (s + 1);
    uint16_t t = v[i];
    v[i] = v[a];
    v[a] = t;
    t = v[i + 1];
    v[i + 1] = v[b];
    v[b] = t;
    i += 2;
  }
}

// ---- observation tokens --------------------------------------------------------------------------------------
// Number of tokens of inventory item with `amount` (observation_encoder.hpp:165-180)
// Agent::check_and_apply_damage (agent.cpp:134-177): when every threshold resource is held, pay the thresholds and
// lose one unit of a resource drawn by std::discrete_distribution over the amounts above the protected minimum
// (random.tcc:2656-2713: cumulative probabilities in double, last forced to 1, lower_bound of a canonical draw).
// First half: when every threshold resource is held, pay the thresholds; returns the number of resources that can
// lose a unit (>= 2 means the second half draws from the RNG), or -1 when the damage does not trigger.
__device__ inline int mg_damage_pay(Ctx& cx, int ai) {
  const int* dc = cx.dc;
  DAgent* a = &cx.ag[ai];
  const int* gr = group_rec(dc, a->group);
  const int ot = gr[GR_OFF_DMG_THRESHOLD], orr = gr[GR_OFF_DMG_RESOURCES];
  const int nt = L_n(dc, ot), nr = L_n(dc, orr);
  if (nt == 0 || nr == 0) return -1;
  for (int i = 0; i < nt; i++)
    if ((int)a->amt[KV_k(dc, ot, i)] < (KV_v(dc, ot, i) & 0xFFFF)) return -1;
  for (int i = 0; i < nt; i++) inv_update(cx, ai, KV_k(dc, ot, i), -KV_v(dc, ot, i));
  int n_avail = 0;
  for (int i = 0; i < nr; i++)
    if ((int)a->amt[KV_k(dc, orr, i)] - KV_v(dc, orr, i) > 0) n_avail++;
  return n_avail;
}
// Second half (only for agents whose first half returned >= 0).
__device__ inline void mg_damage_apply(Ctx& cx, int ai, int n_avail, RngWin& rng) {
  const int* dc = cx.dc;
  DAgent* a = &cx.ag[ai];
  const int* gr = group_rec(dc, a->group);
  const int orr = gr[GR_OFF_DMG_RESOURCES];
  const int nr = L_n(dc, orr);
  if (n_avail > 0) {
    double sum = 0.0;
    for (int i = 0; i < nr; i++) {
      int removable = (int)a->amt[KV_k(dc, orr, i)] - KV_v(dc, orr, i);
      if (removable > 0) sum += (double)removable;
    }
    int sel = 0;
    if (n_avail >= 2) {
      const double p = rng.canonical();
      double acc = 0.0;
      int seen = 0;
      sel = n_avail - 1;  // the last cumulative probability is forced to 1
      for (int i = 0; i < nr; i++) {
        int removable = (int)a->amt[KV_k(dc, orr, i)] - KV_v(dc, orr, i);
        if (removable <= 0) continue;
        acc += (double)removable / sum;
        if (seen < n_avail - 1 && !(acc < p)) {  // lower_bound: first cumulative value >= p
          sel = seen;
          break;
        }
        seen++;
      }
    }
    int item = -1, seen = 0;
    for (int i = 0; i < nr; i++) {
      int removable = (int)a->amt[KV_k(dc, orr, i)] - KV_v(dc, orr, i);
      if (removable <= 0) continue;
      if (seen == sel) {
        item = KV_k(dc, orr, i);
        break;
      }
      seen++;
    }
    inv_update(cx, ai, item, -1);
    mg_stat_add(cx, ai, cx.AS.damage_items_lost(), 1);
    mg_stat_add(cx, ai, cx.AS.damaged(item), 1);
  }
  mg_stat_add(cx, ai, cx.AS.damage_triggered(), 1);
}
__device__ inline void mg_check_damage(Ctx& cx, int ai, RngWin& rng) {
  const int n_avail = mg_damage_pay(cx, ai);
  if (n_avail >= 0) mg_damage_apply(cx, ai, n_avail, rng);
}

__device__ __forceinline__ int mg_inv_ntok(int amount, int base, int n_inv_tokens) {
  int n = 1;
  unsigned rem = (unsigned)amount / (unsigned)base;
  for (int p = 1; p < n_inv_tokens && rem > 0; p++) {
    n++;
    rem /= (unsigned)base;
  }
  return n;
}

// Token sinks: (feature, value) pairs of one object.
struct PairOut {  // into an LDS token list of uint16 (feature | value << 8)
  uint16_t* dst;
  int n;
  __device__ __forceinline__ void put(int f, int v) { dst[n++] = (uint16_t)((f & 0xFF) | ((v & 0xFF) << 8)); }
};
struct CountOut {
  int n;
  __device__ __forceinline__ void put(int, int) { n++; }
};

// The tokens an object shows to any observer (wall.hpp:26-38, agent.cpp:187-213, assembler.hpp:528-578).  They do not
// depend on who looks, so the observation kernel builds each object's list once per step.
template <class OUT>
__device__ inline void mg_inventory_tokens(const int* dc, const DAgent* a, OUT& out) {  // observation_encoder.hpp:165-180
  const int base = dc[DC_TOKEN_BASE], nit = dc[DC_N_INV_TOKENS], ofi = dc[DC_OFF_FID_INV];
  const int ni = a->n_items;
  if (base == 256) {  // the default base: digits are bytes (an integer division costs some forty instructions)
    for (int i = 0; i < ni; i++) {
      const int item = a->order[i];
      const unsigned amount = a->amt[item];
      out.put(dc[ofi + item * nit], (int)(amount & 0xFFu));
      if (nit > 1 && (amount >> 8) != 0) out.put(dc[ofi + item * nit + 1], (int)(amount >> 8));  // amounts are uint16
    }
    return;
  }
  for (int i = 0; i < ni; i++) {
    int item = a->order[i], amount = a->amt[item];
    out.put(dc[ofi + item * nit], (unsigned)amount % (unsigned)base);
    unsigned rem = (unsigned)amount / (unsigned)base;
    for (int p = 1; p < nit && rem > 0; p++) {
      out.put(dc[ofi + item * nit + p], rem % (unsigned)base);
      rem /= (unsigned)base;
    }
  }
}

template <class OUT, class CT>
__device__ inline void mg_object_tokens(const int* dc, const CT* grid, const DAgent* ag, const DAsm* as,
                                        const int4* asm_info, CellCodec cc, int H, int W, int GW, uint32_t step, uint16_t cv,
                                        OUT& out, const DAgent* chests = nullptr, const int4* chest_info = nullptr) {
  int kind = cc.kind(cv);
  const int f_tag = dc[DC_FID0 + MG_F_TAG], f_vibe = dc[DC_FID0 + MG_F_VIBE];
  if (kind == MG_CELL_WALL) {
    const int* rec = wall_rec(dc, cc.wall_idx(cv));
    int ot = rec[WR_OFF_TAGS], nt = L_n(dc, ot);
    for (int i = 0; i < nt; i++) out.put(f_tag, L_at(dc, ot, i));
    if (rec[WR_VIBE] != 0) out.put(f_vibe, rec[WR_VIBE]);
  } else if (kind == MG_CELL_AGENT) {
    const DAgent* a = &ag[cc.agent_idx(cv)];
    const int* gr = group_rec(dc, a->group);
    out.put(dc[DC_FID0 + MG_F_GROUP], gr[GR_GROUP_ID]);
    out.put(dc[DC_FID0 + MG_F_FROZEN], a->frozen != 0 ? 1 : 0);
    if (a->vibe != 0) out.put(f_vibe, a->vibe);
    mg_inventory_tokens(dc, a, out);
    int ot = gr[GR_OFF_TAGS], nt = L_n(dc, ot);
    for (int i = 0; i < nt; i++) out.put(f_tag, L_at(dc, ot, i));
  } else if (kind == MG_CELL_ASSEMBLER) {
    int idx = cc.asm_idx(cv);
    const DAsm* s = &as[idx];
    int4 info = asm_info[idx];
    const int* rec = asm_rec(dc, info.z);
    uint32_t remaining = s->cooldown_end <= step ? 0 : s->cooldown_end - step;
    if (remaining > 255u) remaining = 255u;
    if (remaining > 0) out.put(dc[DC_FID0 + MG_F_COOLDOWN], (int)remaining);
    if (s->clipped) out.put(dc[DC_FID0 + MG_F_CLIPPED], 1);
    uint32_t max_uses = (uint32_t)rec[AR_MAX_USES];
    if (max_uses > 0) {
      uint32_t ru = s->uses_count < max_uses ? max_uses - s->uses_count : 0;
      if (ru > 255u) ru = 255u;
      out.put(dc[DC_FID0 + MG_F_REMAINING_USES], (int)ru);
    }
    if (dc[DC_PROTO_OBS]) {
      int po = rec[AR_SIMPLE_PO];
#ifdef MG_NO_CLIPPER
      if (s->clipped) po = -1;
      else if (po < 0) po = mg_current_protocol(dc, grid, ag, cc, H, W, GW, info.x, info.y, info.z, 0u);
#else
      if (s->clipped == 1u) po = -1;
      else if (po < 0 || s->clipped) po = mg_current_protocol(dc, grid, ag, cc, H, W, GW, info.x, info.y, info.z, s->clipped);
#endif
      if (po >= 0) {
        int oi = dc[po + PR_OFF_IN], ni = L_n(dc, oi);
        for (int i = 0; i < ni; i++)
          if (KV_v(dc, oi, i) > 0) out.put(dc[dc[DC_OFF_FID_IN] + KV_k(dc, oi, i)], KV_v(dc, oi, i));
        int oo = dc[po + PR_OFF_OUT], no = L_n(dc, oo);
        for (int i = 0; i < no; i++)
          if (KV_v(dc, oo, i) > 0) out.put(dc[dc[DC_OFF_FID_OUT] + KV_k(dc, oo, i)], KV_v(dc, oo, i));
      }
    }
    int ot = rec[AR_OFF_TAGS], nt = L_n(dc, ot);
    for (int i = 0; i < nt; i++) out.put(f_tag, L_at(dc, ot, i));
    if (rec[AR_VIBE] != 0) out.put(f_vibe, rec[AR_VIBE]);
  } else if (kind == MG_CELL_CHEST) {  // chest.hpp:124-146
    int idx = cc.chest_idx(cv);
    const int* rec = dc + dc[dc[DC_OFF_CHESTS] + chest_info[idx].z];
    if (rec[CR_VIBE] != 0) out.put(f_vibe, rec[CR_VIBE]);
    mg_inventory_tokens(dc, &chests[idx], out);
    int ot = rec[CR_OFF_TAGS], nt = L_n(dc, ot);
    for (int i = 0; i < nt; i++) out.put(f_tag, L_at(dc, ot, i));
  }
}
