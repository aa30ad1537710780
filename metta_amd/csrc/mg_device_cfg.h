// mg_device_cfg.h -- device-side configuration stream ("dcfg"): a compact int32 array with a fixed header of
// scalars and offsets, built on the host from the parsed MgConfig (mg_blob.h) and read by the kernels.
//
// The reference keeps this information in heap objects reached through pointers (GameConfig, AgentConfig,
// AssemblerConfig, Protocol ...).  On the device it is one read-only array: wave-uniform reads become scalar loads,
// everything stays L2-resident, and no pointer chasing is left on the step path.
#pragma once
#include <cstdint>
#include <vector>

#include "mg_blob.h"

enum MgDc {
  DC_R = 0, DC_A, DC_OBS_W, DC_OBS_H, DC_T, DC_TOKEN_BASE, DC_MAX_STEPS, DC_TRUNCATES,
  DC_G_COMPLETION, DC_G_LAST_ACTION, DC_G_LAST_REWARD, DC_G_COMPASS, DC_G_GOAL, DC_PROTO_OBS,
  DC_REGEN_INTERVAL, DC_INIT_PASSES, DC_N_VIBES, DC_NG, DC_N_INV_TOKENS, DC_N_ACTIONS, DC_MAX_PRIORITY,
  DC_ATK_FREEZE, DC_SA, DC_SG, DC_PATTERN_N, DC_N_GROUPS, DC_N_WALLS, DC_N_ASMS, DC_N_CELLS, DC_HAS_DEFENSE, DC_HAS_DAMAGE, DC_N_CHESTS,
  DC_OBS_WORDS,  // the stream's prefix that is staged in LDS for the observation / use code (pattern, action, attack and transfer tables come after it)
  DC_FID0,
  DC_OFF_FID_INV = DC_FID0 + MG_F_COUNT, DC_OFF_FID_IN, DC_OFF_FID_OUT, DC_OFF_ACTIONS,
  DC_OFF_HANDLERS,  // 6 offsets: required/consumed kv-lists of noop, move, change_vibe
  DC_OFF_VIBE_HANDLER = DC_OFF_HANDLERS + 6, DC_OFF_ATK_CONSUMED, DC_OFF_ATK_DEFENSE, DC_OFF_ATK_ARMOR,
  DC_OFF_ATK_WEAPON, DC_OFF_ATK_ACTOR, DC_OFF_ATK_TARGET, DC_OFF_ATK_LOOT, DC_OFF_ATK_VIBE_BONUS,
  DC_OFF_VIBE_RESOURCE, DC_OFF_XFER_REQUIRED, DC_OFF_XFER_EFFECTS, DC_OFF_GROUPS, DC_OFF_WALLS, DC_OFF_ASMS,
  DC_OFF_PATTERN, DC_OFF_CELLS, DC_OFF_CHESTS,
  // Clipper (systems/clipper.hpp): present flag, configured length scale (0 = from the map's sparsity), cutoff in units
  // of the length scale, clip period, and the unclipping protocols as (protocol record, key lo, key hi) triples
  DC_CLIPPER, DC_CLIP_LENGTH, DC_CLIP_CUTOFF, DC_CLIP_PERIOD, DC_N_UNCLIP, DC_OFF_UNCLIP,
  DC_OFF_IPAT,  // inverse observation pattern, one byte per window cell (row-major): the cell's position in the pattern order
  DC_HEADER_WORDS
};
// group record
enum MgGr {
  GR_GROUP_ID = 0, GR_NAME_IDX, GR_FREEZE_DURATION, GR_INITIAL_VIBE, GR_OFF_TAGS, GR_N_LIMITS, GR_OFF_LIMITS,
  GR_OFF_LIMIT_OF, GR_OFF_MODIFIER, GR_OFF_LIMITS_ORDER, GR_OFF_INITIAL, GR_OFF_STAT_REWARDS, GR_OFF_GOAL,
  GR_OFF_REGEN, GR_DIVERSITY_MASK, GR_HAS_MODIFIERS, GR_OFF_DMG_THRESHOLD, GR_OFF_DMG_RESOURCES,
  GR_OFF_LIMIT_MASK,  // per limit: bit r set = resource r counts against it (the inverse of GR_OFF_LIMIT_OF)
  GR_COLLECTIVE,  // the collective the group's agents end up in (last "collective:<name>" tag), or -1
  GR_WORDS
};
enum MgWr { WR_TYPE_ID = 0, WR_VIBE, WR_OFF_TAGS, WR_WORDS };
// chest record: CR_GROUP = index of the pseudo group record (after the agent groups) that carries the chest's inventory
// configuration in the GR_* layout; CR_OFF_XFER -> n, then n x (vibe, offset of the (item, delta) kv-list)
enum MgCr { CR_TYPE_ID = 0, CR_VIBE, CR_OFF_TAGS, CR_GROUP, CR_OFF_XFER, CR_WORDS };
enum MgAr {
  AR_TYPE_ID = 0, AR_VIBE, AR_ALLOW_PARTIAL, AR_MAX_USES, AR_CLIP_IMMUNE, AR_START_CLIPPED, AR_CHEST_DIST, AR_OFF_TAGS,
  AR_N_PROTOCOLS, AR_OFF_PTABLE, AR_N_KEYS, AR_OFF_KEYS, AR_SIMPLE_PO, AR_WORDS
};
enum MgPr { PR_MIN_AGENTS = 0, PR_COOLDOWN, PR_OFF_VIBES, PR_OFF_IN, PR_OFF_OUT, PR_OFF_IN_S, PR_OFF_OUT_S, PR_WORDS };

#define MG_SR_WORDS 7  // stat-reward entry: agent slot, game slot, weight, has_max, max, collective constant, aligned flag
// Device inventories hold up to MGD_MAX_R resources (uint16 amounts + an order list of that many entries, in the agent
// record).  The library is built twice: libmettagrid_hip.so with 16 (112-byte records: the arena / navigation / CvC
// resource sets) and libmettagrid_hip_r32.so with 32 (160-byte records) for wider resource lists; 32-bit masks over
// the resources (dirty amounts, diversity presence, pending chest amounts) set the ceiling.
#ifndef MGD_MAX_R
#define MGD_MAX_R 16
#endif
static_assert(MGD_MAX_R == 16 || MGD_MAX_R == 32, "MGD_MAX_R: 16 or 32");

struct MgDcBuilder {
  std::vector<int32_t> w;
  int put_list(const std::vector<int32_t>& v) {
    int off = (int)w.size();
    w.push_back((int32_t)v.size());
    for (int32_t x : v) w.push_back(x);
    return off;
  }
  int put_kv(const MgKVList& v) {
    int off = (int)w.size();
    w.push_back((int32_t)v.size());
    for (const auto& kv : v) {
      w.push_back(kv.k);
      w.push_back(kv.v);
    }
    return off;
  }
  int reserve(int n) {
    int off = (int)w.size();
    w.resize(w.size() + (size_t)n, 0);
    return off;
  }
};

inline int32_t mg_f32_bits(float f) {
  int32_t v;
  std::memcpy(&v, &f, 4);
  return v;
}

// Observation offsets in reference order (packed_coordinate.hpp:74-143), packed (dr+8) | (dc+8) << 8.
inline std::vector<int32_t> mg_pattern(int h, int w) {
  std::vector<int32_t> out;
  int emitted = 0, max_emitted = h * w;
  int row_min = -(h / 2), row_max = h / 2, col_min = -(w / 2), col_max = w / 2;
  for (int d = 0; emitted < max_emitted && d <= h + w + 2; d++)
    for (int dr = -d; dr <= d && emitted < max_emitted; dr++) {
      int dc = d - (dr < 0 ? -dr : dr);
      for (int step = 0; step < (dc == 0 ? 1 : 2) && emitted < max_emitted; step++) {
        int rr = dr, cc = (step == 0 ? -dc : dc);
        if (rr >= row_min && rr <= row_max && cc >= col_min && cc <= col_max) {
          out.push_back((rr + 8) | ((cc + 8) << 8));
          emitted++;
        }
      }
    }
  return out;
}

inline std::vector<int32_t> mg_build_dcfg(const MgConfig& c) {
  MgDcBuilder b;
  b.reserve(DC_HEADER_WORDS);
  auto H = [&](int i) -> int32_t& { return b.w[(size_t)i]; };
  H(DC_R) = c.R; H(DC_A) = c.A; H(DC_OBS_W) = c.obs_w; H(DC_OBS_H) = c.obs_h; H(DC_T) = c.T;
  H(DC_TOKEN_BASE) = c.token_base; H(DC_MAX_STEPS) = c.max_steps; H(DC_TRUNCATES) = c.episode_truncates;
  H(DC_G_COMPLETION) = c.g_completion; H(DC_G_LAST_ACTION) = c.g_last_action; H(DC_G_LAST_REWARD) = c.g_last_reward;
  H(DC_G_COMPASS) = c.g_compass; H(DC_G_GOAL) = c.g_goal; H(DC_PROTO_OBS) = c.protocol_details_obs;
  H(DC_REGEN_INTERVAL) = c.regen_interval; H(DC_INIT_PASSES) = c.init_obs_passes < 1 ? 1 : c.init_obs_passes;
  H(DC_N_VIBES) = c.n_vibes; H(DC_NG) = c.n_group_names; H(DC_N_INV_TOKENS) = c.n_inv_tokens;
  H(DC_N_ACTIONS) = (int)c.actions.size(); H(DC_MAX_PRIORITY) = c.max_priority; H(DC_ATK_FREEZE) = c.atk_freeze;
  H(DC_SA) = c.n_agent_slots; H(DC_SG) = c.n_game_slots; H(DC_N_GROUPS) = (int)c.groups.size();
  H(DC_N_WALLS) = (int)c.walls.size(); H(DC_N_ASMS) = (int)c.assemblers.size(); H(DC_N_CELLS) = (int)c.cells.size();
  H(DC_HAS_DEFENSE) = c.atk_defense.empty() ? 0 : 1;
  H(DC_HAS_DAMAGE) = 0;
  for (const auto& g : c.groups)
    if (!g.damage_threshold.empty() && !g.damage_resources.empty()) H(DC_HAS_DAMAGE) = 1;
  for (int i = 0; i < MG_F_COUNT; i++) H(DC_FID0 + i) = c.fid[i];
  { int o = b.put_list(c.fid_inv); H(DC_OFF_FID_INV) = o + 1; }
  { int o = b.put_list(c.fid_in); H(DC_OFF_FID_IN) = o + 1; }
  { int o = b.put_list(c.fid_out); H(DC_OFF_FID_OUT) = o + 1; }
  // groups (+ one pseudo group per chest type: the same record layout carries the chest's inventory configuration,
  // so every inventory routine works on agents and chests alike)
  {
    std::vector<int32_t> table;
    const MgGroup empty_group{};
    auto put_group = [&](const MgGroup& g, const MgInventoryCfg& inv) {
      int32_t rec[GR_WORDS] = {0};
      rec[GR_GROUP_ID] = g.group_id; rec[GR_NAME_IDX] = g.group_name_idx; rec[GR_FREEZE_DURATION] = g.freeze_duration;
      rec[GR_INITIAL_VIBE] = g.initial_vibe; rec[GR_OFF_TAGS] = b.put_list(g.tags);
      rec[GR_N_LIMITS] = (int32_t)inv.limits.size();
      std::vector<int32_t> lim;
      for (const auto& l : inv.limits) {
        lim.push_back(l.base);
        lim.push_back(b.put_kv(l.modifiers));
      }
      { int o = b.reserve((int)lim.size()); for (size_t i = 0; i < lim.size(); i++) b.w[(size_t)o + i] = lim[i]; rec[GR_OFF_LIMITS] = o; }
      std::vector<int32_t> limit_of((size_t)std::max(c.R, 1), -1), modifier((size_t)std::max(c.R, 1), 0);
      bool has_mod = false;
      for (const auto& kv : inv.limits_map_order) limit_of[(size_t)kv.k] = kv.v;
      for (const auto& kv : inv.limits_map_order)
        for (const auto& m : inv.limits[(size_t)kv.v].modifiers) { modifier[(size_t)m.k] = 1; has_mod = true; }
      { int o = b.reserve((int)limit_of.size()); for (size_t i = 0; i < limit_of.size(); i++) b.w[(size_t)o + i] = limit_of[i]; rec[GR_OFF_LIMIT_OF] = o; }
      {
        std::vector<int32_t> lmask(std::max<size_t>(inv.limits.size(), 1), 0);
        for (size_t r = 0; r < limit_of.size(); r++)
          if (limit_of[r] >= 0 && (size_t)limit_of[r] < lmask.size()) lmask[(size_t)limit_of[r]] |= (int32_t)(1u << r);
        int o = b.reserve((int)lmask.size());
        for (size_t i = 0; i < lmask.size(); i++) b.w[(size_t)o + i] = lmask[i];
        rec[GR_OFF_LIMIT_MASK] = o;
      }
      { int o = b.reserve((int)modifier.size()); for (size_t i = 0; i < modifier.size(); i++) b.w[(size_t)o + i] = modifier[i]; rec[GR_OFF_MODIFIER] = o; }
      rec[GR_HAS_MODIFIERS] = has_mod ? 1 : 0;
      rec[GR_OFF_LIMITS_ORDER] = b.put_kv(inv.limits_map_order);
      rec[GR_OFF_INITIAL] = b.put_kv(inv.initial);
      {
        // Collective statistics never change during an episode (nothing on the step path touches a collective's
        // inventory or membership): what `getCollective()->stats.get(name)` adds (agent.cpp:115-117) is a constant
        // of the entry -- "collective.<r>.deposited" = the initial amount (collective.hpp:36-41,101-108; uint16
        // inventory) -- or, for "aligned.agent", the member count, which the reset kernel leaves in the agent record.
        const int coll = g.collectives.empty() ? -1 : g.collectives.back();
        rec[GR_COLLECTIVE] = coll;
        int o = b.reserve(1 + MG_SR_WORDS * (int)g.stat_rewards.size());
        b.w[(size_t)o] = (int32_t)g.stat_rewards.size();
        for (size_t i = 0; i < g.stat_rewards.size(); i++) {
          const auto& s = g.stat_rewards[i];
          int32_t* p = &b.w[(size_t)o + 1 + MG_SR_WORDS * i];
          p[0] = s.agent_slot; p[1] = s.game_slot; p[2] = mg_f32_bits(s.weight); p[3] = s.has_max; p[4] = mg_f32_bits(s.max);
          float cv = 0.f;
          if (coll >= 0 && s.coll_kind == 2)
            for (const auto& kv : c.collective_initial[(size_t)coll])
              if (kv.k == s.coll_res && kv.v > 0) cv = (float)(kv.v > 65535 ? 65535 : kv.v);
          p[5] = mg_f32_bits(cv);
          p[6] = (coll >= 0 && s.coll_kind == 1) ? 1 : 0;
        }
        rec[GR_OFF_STAT_REWARDS] = o;
      }
      rec[GR_OFF_GOAL] = b.put_list(g.goal_values);
      {
        std::vector<int32_t> rr;
        for (const auto& v : g.regen) { rr.push_back(v.first); rr.push_back(b.put_kv(v.second)); }
        int o = b.reserve(1 + (int)rr.size());
        b.w[(size_t)o] = (int32_t)g.regen.size();
        for (size_t i = 0; i < rr.size(); i++) b.w[(size_t)o + 1 + i] = rr[i];
        rec[GR_OFF_REGEN] = o;
      }
      uint32_t dm = 0;
      for (int32_t r : g.diversity) dm |= (1u << r);
      rec[GR_DIVERSITY_MASK] = (int32_t)dm;
      rec[GR_OFF_DMG_THRESHOLD] = b.put_kv(g.damage_threshold);
      rec[GR_OFF_DMG_RESOURCES] = b.put_kv(g.damage_resources);
      int o = b.reserve(GR_WORDS);
      for (int i = 0; i < GR_WORDS; i++) b.w[(size_t)o + (size_t)i] = rec[i];
      table.push_back(o);
    };
    for (const auto& g : c.groups) put_group(g, g.inv);
    for (const auto& ch : c.chests) put_group(empty_group, ch.inv);
    int o = b.reserve((int)table.size());
    for (size_t i = 0; i < table.size(); i++) b.w[(size_t)o + i] = table[i];
    H(DC_OFF_GROUPS) = o;
  }
  {  // chests (chest.hpp:20-147)
    std::vector<int32_t> table;
    for (size_t ci = 0; ci < c.chests.size(); ci++) {
      const auto& ch = c.chests[ci];
      int ot = b.put_list(ch.tags);
      std::vector<int32_t> xf;
      for (const auto& v : ch.vibe_transfers) { xf.push_back(v.first); xf.push_back(b.put_kv(v.second)); }
      int ox = b.reserve(1 + (int)xf.size());
      b.w[(size_t)ox] = (int32_t)ch.vibe_transfers.size();
      for (size_t i = 0; i < xf.size(); i++) b.w[(size_t)ox + 1 + i] = xf[i];
      int o = b.reserve(CR_WORDS);
      b.w[(size_t)o + CR_TYPE_ID] = ch.type_id; b.w[(size_t)o + CR_VIBE] = ch.vibe; b.w[(size_t)o + CR_OFF_TAGS] = ot;
      b.w[(size_t)o + CR_GROUP] = (int32_t)(c.groups.size() + ci); b.w[(size_t)o + CR_OFF_XFER] = ox;
      table.push_back(o);
    }
    int o = b.reserve((int)table.size());
    for (size_t i = 0; i < table.size(); i++) b.w[(size_t)o + i] = table[i];
    H(DC_OFF_CHESTS) = o;
    H(DC_N_CHESTS) = (int32_t)c.chests.size();
  }
  {
    std::vector<int32_t> table;
    for (const auto& wl : c.walls) {
      int ot = b.put_list(wl.tags);
      int o = b.reserve(WR_WORDS);
      b.w[(size_t)o + WR_TYPE_ID] = wl.type_id; b.w[(size_t)o + WR_VIBE] = wl.vibe; b.w[(size_t)o + WR_OFF_TAGS] = ot;
      table.push_back(o);
    }
    int o = b.reserve((int)table.size());
    for (size_t i = 0; i < table.size(); i++) b.w[(size_t)o + i] = table[i];
    H(DC_OFF_WALLS) = o;
  }
  {
    std::vector<int32_t> table;
    for (const auto& a : c.assemblers) {
      int32_t rec[AR_WORDS] = {0};
      rec[AR_TYPE_ID] = a.type_id; rec[AR_VIBE] = a.vibe; rec[AR_ALLOW_PARTIAL] = a.allow_partial_usage;
      rec[AR_MAX_USES] = a.max_uses; rec[AR_CLIP_IMMUNE] = a.clip_immune; rec[AR_START_CLIPPED] = a.start_clipped;
      rec[AR_CHEST_DIST] = a.chest_search_distance; rec[AR_OFF_TAGS] = b.put_list(a.tags);
      rec[AR_N_PROTOCOLS] = (int32_t)a.protocols.size();
      std::vector<int32_t> pt;
      for (const auto& p : a.protocols) {
        int32_t pr[PR_WORDS];
        pr[PR_MIN_AGENTS] = p.min_agents; pr[PR_COOLDOWN] = p.cooldown; pr[PR_OFF_VIBES] = b.put_list(p.vibes);
        pr[PR_OFF_IN] = b.put_kv(p.inputs); pr[PR_OFF_OUT] = b.put_kv(p.outputs);
        pr[PR_OFF_IN_S] = b.put_list(p.inputs_scaled); pr[PR_OFF_OUT_S] = b.put_list(p.outputs_scaled);
        int o = b.reserve(PR_WORDS);
        for (int i = 0; i < PR_WORDS; i++) b.w[(size_t)o + (size_t)i] = pr[i];
        pt.push_back(o);
      }
      { int o = b.reserve((int)pt.size()); for (size_t i = 0; i < pt.size(); i++) b.w[(size_t)o + i] = pt[i]; rec[AR_OFF_PTABLE] = o; }
      rec[AR_N_KEYS] = (int32_t)a.keys.size();
      std::vector<int32_t> kt;
      for (const auto& k : a.keys) {
        kt.push_back((int32_t)(uint32_t)(k.key & 0xFFFFFFFFull));
        kt.push_back((int32_t)(uint32_t)(k.key >> 32));
        kt.push_back(b.put_list(k.protocols));
      }
      { int o = b.reserve((int)kt.size()); for (size_t i = 0; i < kt.size(); i++) b.w[(size_t)o + i] = kt[i]; rec[AR_OFF_KEYS] = o; }
      // When the only key is 0 and its first (largest min_agents) protocol needs no agents, get_current_protocol
      // (assembler.hpp:378-408) returns that protocol whatever surrounds the assembler: skip the neighbour scan.
      rec[AR_SIMPLE_PO] = -1;
      if (a.keys.size() == 1 && a.keys[0].key == 0 && !a.keys[0].protocols.empty() &&
          a.protocols[(size_t)a.keys[0].protocols[0]].min_agents == 0)
        rec[AR_SIMPLE_PO] = pt[(size_t)a.keys[0].protocols[0]];
      int o = b.reserve(AR_WORDS);
      for (int i = 0; i < AR_WORDS; i++) b.w[(size_t)o + (size_t)i] = rec[i];
      table.push_back(o);
    }
    int o = b.reserve((int)table.size());
    for (size_t i = 0; i < table.size(); i++) b.w[(size_t)o + i] = table[i];
    H(DC_OFF_ASMS) = o;
  }
  {  // clipper: the protocol a clipped station offers is one of these (clipper.hpp:186-193, assembler.hpp:411-421)
    H(DC_CLIPPER) = c.clipper; H(DC_CLIP_LENGTH) = c.clip_length_scale; H(DC_CLIP_CUTOFF) = c.clip_cutoff;
    H(DC_CLIP_PERIOD) = c.clip_period; H(DC_N_UNCLIP) = (int32_t)c.unclip.size();
    std::vector<int32_t> tb;
    for (size_t i = 0; i < c.unclip.size(); i++) {
      const MgProtocol& p = c.unclip[i];
      int32_t pr[PR_WORDS];
      pr[PR_MIN_AGENTS] = p.min_agents; pr[PR_COOLDOWN] = p.cooldown; pr[PR_OFF_VIBES] = b.put_list(p.vibes);
      pr[PR_OFF_IN] = b.put_kv(p.inputs); pr[PR_OFF_OUT] = b.put_kv(p.outputs);
      pr[PR_OFF_IN_S] = b.put_list(p.inputs_scaled); pr[PR_OFF_OUT_S] = b.put_list(p.outputs_scaled);
      int o = b.reserve(PR_WORDS);
      for (int q = 0; q < PR_WORDS; q++) b.w[(size_t)o + (size_t)q] = pr[q];
      tb.push_back(o);
      tb.push_back((int32_t)(uint32_t)(c.unclip_key[i] & 0xFFFFFFFFull));
      tb.push_back((int32_t)(uint32_t)(c.unclip_key[i] >> 32));
    }
    int o = b.reserve((int)tb.size());
    for (size_t i = 0; i < tb.size(); i++) b.w[(size_t)o + i] = tb[i];
    H(DC_OFF_UNCLIP) = o;
  }
  {  // (inside the LDS-staged prefix: the fused kernel's agent scatter reads it per observer)
    std::vector<int32_t> pat = mg_pattern(c.obs_h, c.obs_w);
    // (an even window's offsets still run over -h/2 .. h/2: the table is (h | 1) x (w | 1))
    const int ih = c.obs_h | 1, iw = c.obs_w | 1;
    std::vector<uint8_t> inv((size_t)ih * iw + 3, 0);
    for (size_t p = 0; p < pat.size(); p++) {
      const int dr = (pat[p] & 0xFF) - 8, dcc = ((pat[p] >> 8) & 0xFF) - 8;
      inv[(size_t)(dr + c.obs_h / 2) * iw + (size_t)(dcc + c.obs_w / 2)] = (uint8_t)p;
    }
    const int nw = (ih * iw + 3) / 4;
    int o = b.reserve(nw);
    for (int i = 0; i < nw; i++)
      b.w[(size_t)o + i] = (int32_t)((uint32_t)inv[4 * i] | ((uint32_t)inv[4 * i + 1] << 8) | ((uint32_t)inv[4 * i + 2] << 16) | ((uint32_t)inv[4 * i + 3] << 24));
    H(DC_OFF_IPAT) = o;
  }
  H(DC_OBS_WORDS) = (int32_t)((b.w.size() + 3) & ~(size_t)3);
  while (b.w.size() % 4) b.w.push_back(0);
  // ---- tables read through the global pointer only (KArgs::gdc): the observation pattern (once per wave), the cell
  // table (host), and what only the action phase reads ----
  {
    std::vector<int32_t> pat = mg_pattern(c.obs_h, c.obs_w);
    H(DC_PATTERN_N) = (int32_t)pat.size();
    int o = b.reserve((int)pat.size());
    for (size_t i = 0; i < pat.size(); i++) b.w[(size_t)o + i] = pat[i];
    H(DC_OFF_PATTERN) = o;
  }
  {
    int o = b.reserve((int)c.cells.size() * 2);
    for (size_t i = 0; i < c.cells.size(); i++) {
      b.w[(size_t)o + 2 * i] = c.cells[i].kind;
      b.w[(size_t)o + 2 * i + 1] = c.cells[i].idx;
    }
    H(DC_OFF_CELLS) = o;
  }
  {
    int o = b.reserve((int)c.actions.size() * 2);
    for (size_t i = 0; i < c.actions.size(); i++) {
      b.w[(size_t)o + 2 * i] = c.actions[i].kind;
      b.w[(size_t)o + 2 * i + 1] = c.actions[i].arg;
    }
    H(DC_OFF_ACTIONS) = o;
  }
  for (int h = 0; h < MG_NUM_HANDLERS; h++) {
    { int o = b.put_kv(c.handler[h].required); H(DC_OFF_HANDLERS + 2 * h) = o; }
    { int o = b.put_kv(c.handler[h].consumed); H(DC_OFF_HANDLERS + 2 * h + 1) = o; }
  }
  { int o = b.put_list(c.vibe_handler); H(DC_OFF_VIBE_HANDLER) = o + 1; }
  { int o = b.put_kv(c.atk_consumed); H(DC_OFF_ATK_CONSUMED) = o; }
  { int o = b.put_kv(c.atk_defense); H(DC_OFF_ATK_DEFENSE) = o; }
  { int o = b.put_kv(c.atk_armor); H(DC_OFF_ATK_ARMOR) = o; }
  { int o = b.put_kv(c.atk_weapon); H(DC_OFF_ATK_WEAPON) = o; }
  { int o = b.put_kv(c.atk_actor_delta); H(DC_OFF_ATK_ACTOR) = o; }
  { int o = b.put_kv(c.atk_target_delta); H(DC_OFF_ATK_TARGET) = o; }
  { int o = b.put_list(c.atk_loot); H(DC_OFF_ATK_LOOT) = o; }
  { int o = b.put_list(c.atk_vibe_bonus); H(DC_OFF_ATK_VIBE_BONUS) = o + 1; }
  { int o = b.put_list(c.vibe_resource); H(DC_OFF_VIBE_RESOURCE) = o + 1; }
  { int o = b.put_kv(c.xfer_required); H(DC_OFF_XFER_REQUIRED) = o; }
  {
    std::vector<int32_t> rec;
    for (const auto& e : c.xfer_effects) {
      int oa = b.put_kv(e.actor), ot = b.put_kv(e.target);
      rec.push_back(e.vibe);
      rec.push_back(oa);
      rec.push_back(ot);
    }
    int o = b.reserve(1 + (int)rec.size());
    b.w[(size_t)o] = (int32_t)c.xfer_effects.size();
    for (size_t i = 0; i < rec.size(); i++) b.w[(size_t)o + 1 + i] = rec[i];
    H(DC_OFF_XFER_EFFECTS) = o;
  }
  return b.w;
}
