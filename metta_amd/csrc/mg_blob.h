// mg_blob.h -- the packed game-configuration "blob" exchanged across the C-ABI (header-only parser).
//
// The reference hands its C++ core a tree of pybind11 config objects (GameConfig / AgentConfig / AssemblerConfig /
// Protocol ...: packages/mettagrid/cpp/include/mettagrid/config/mettagrid_config.hpp:31-59).  A C-ABI cannot carry that,
// so the host language flattens the same information into one int32 stream (metta_amd/compile.py writes it; any
// other host language can) and the library parses it here.  Everything that depends on libstdc++'s
// unordered_map iteration order in the reference is already *ordered* in the stream: lists
// below are in the exact order the reference would iterate them.
//
// Stream grammar (all int32; floats are bit-casts; LIST(x) = n, x*n; KV = key,value):
//   see MgConfig::parse below -- the reader IS the grammar.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#define MG_BLOB_MAGIC 0x4D474231  // 'MGB1'
#define MG_BLOB_VERSION 5

enum MgActionKind { MG_ACT_NOOP = 0, MG_ACT_MOVE = 1, MG_ACT_CHANGE_VIBE = 2 };
enum MgCellKind { MG_CELL_EMPTY = 0, MG_CELL_WALL = 1, MG_CELL_AGENT = 2, MG_CELL_ASSEMBLER = 3, MG_CELL_CHEST = 4 };
enum MgVibeHandler { MG_VH_NONE = 0, MG_VH_ATTACK = 1, MG_VH_TRANSFER = 2 };

// Fixed feature slots in MgConfig::fid (reference ids come from PY/config/id_map.py:98-141).
enum MgFid {
  MG_F_GROUP = 0, MG_F_FROZEN, MG_F_COMPLETION, MG_F_LAST_ACTION, MG_F_LAST_REWARD, MG_F_GOAL, MG_F_VIBE,
  MG_F_COMPASS, MG_F_TAG, MG_F_COOLDOWN, MG_F_CLIPPED, MG_F_REMAINING_USES, MG_F_COUNT
};

// ---------------------------------------------------------------------------------------------------------
// Stat slots.  The reference keeps unordered_map<string,float> per agent and per game
// (CPP/systems/stats_tracker.hpp:16-54); here every key the step path can touch has a fixed slot, and a
// "touched" bit reproduces "a key exists only once touched".  metta_amd/stats.py mirrors these formulas to map
// slots back to the reference's exact key strings.
//   R = #resources, G = #distinct agent group names, H = 3 handlers (noop, move, change_vibe).
// ---------------------------------------------------------------------------------------------------------
#define MG_NUM_HANDLERS 3
#define MG_INVALID_VARIANTS 9  // action.invalid_index.<id> for id = -1, n_actions .. n_actions+6, "other"
// Any other id: the reference makes one key per id (mettagrid_c.cpp:565-570).  The agent's stat row ends with
// MG_INVALID_TABLE (id, count) pairs -- the id as raw int32 bits in the f32 slot -- filled first come first kept; a
// further distinct id of one agent in one episode counts under "action.invalid_index.other".
#define MG_INVALID_TABLE 4
struct MgAgentSlots {
  int R, G;
  constexpr int success(int h) const { return h; }
  constexpr int failed(int h) const { return 3 + h; }
  constexpr int failed_any() const { return 6; }
  constexpr int invalid() const { return 7; }
  constexpr int invalid_variant(int v) const { return 8 + v; }
  constexpr int swap() const { return 17; }
  constexpr int frozen_ticks() const { return 18; }
  constexpr int frozen_ticks_group() const { return 19; }
  constexpr int max_steps_without_motion() const { return 20; }
  constexpr int diversity() const { return 21; }
  constexpr int damage_triggered() const { return 22; }
  constexpr int damage_items_lost() const { return 23; }
  constexpr int transfer_count() const { return 24; }
  constexpr int attack_friendly_fire() const { return 25; }
  constexpr int base_r() const { return 26; }
  constexpr int gained(int r) const { return base_r() + 0 * R + r; }
  constexpr int lost(int r) const { return base_r() + 1 * R + r; }
  constexpr int amount(int r) const { return base_r() + 2 * R + r; }
  constexpr int damaged(int r) const { return base_r() + 3 * R + r; }
  constexpr int chest_deposited(int r) const { return base_r() + 4 * R + r; }
  constexpr int chest_withdrawn(int r) const { return base_r() + 5 * R + r; }
  constexpr int transfer_self(int r) const { return base_r() + 6 * R + r; }
  constexpr int base_g() const { return base_r() + 7 * R; }
  constexpr int attack_hit(int g) const { return base_g() + 0 * G + g; }
  constexpr int attack_hit_by(int g) const { return base_g() + 1 * G + g; }
  constexpr int attack_blocked_by(int g) const { return base_g() + 2 * G + g; }
  constexpr int transfer_to(int g, int r) const { return base_g() + 3 * G + g * R + r; }
  constexpr int invalid_table_id(int k) const { return base_g() + 3 * G + G * R + 2 * k; }
  constexpr int invalid_table_count(int k) const { return base_g() + 3 * G + G * R + 2 * k + 1; }
  constexpr int count() const { return base_g() + 3 * G + G * R + 2 * MG_INVALID_TABLE; }
};
struct MgGameSlots {
  int R, C;  // C = number of map cell codes (objects.<cell name>)
  constexpr int tokens_written() const { return 0; }
  constexpr int tokens_dropped() const { return 1; }
  constexpr int tokens_free_space() const { return 2; }
  constexpr int objects(int code) const { return 3 + (code - 1); }
  constexpr int base_r() const { return 3 + C; }
  constexpr int assembler_created(int r) const { return base_r() + 0 * R + r; }
  constexpr int chest_deposited(int r) const { return base_r() + 1 * R + r; }
  constexpr int chest_withdrawn(int r) const { return base_r() + 2 * R + r; }
  constexpr int chest_amount(int r) const { return base_r() + 3 * R + r; }
  constexpr int count() const { return base_r() + 4 * R; }
};

struct MgKV {
  int32_t k, v;
};
typedef std::vector<MgKV> MgKVList;

struct MgLimit {
  int32_t base;
  std::vector<int32_t> resources;
  MgKVList modifiers;
};
struct MgInventoryCfg {
  std::vector<MgLimit> limits;  // creation order == std::set<SharedInventoryLimit*> order (heap-address order)
  MgKVList limits_map_order;    // (item, limit index) in `_limits` iteration order (inventory.cpp:141)
  MgKVList initial;             // (item, amount) in initial_inventory iteration order
};
struct MgStatReward {
  int32_t agent_slot, game_slot;
  float weight;
  int32_t has_max;
  float max;
  // what `getCollective()->stats.get(stat_name)` can find (agent.cpp:115-117; collective.hpp:46-52,101-108):
  // 0 nothing, 1 "aligned.agent", 2 "collective.<coll_res>.deposited"
  int32_t coll_kind, coll_res;
};
struct MgGroup {
  int32_t group_id, group_name_idx, freeze_duration, initial_vibe;
  std::vector<int32_t> tags;
  MgInventoryCfg inv;
  std::vector<MgStatReward> stat_rewards;  // iteration order of the string-keyed map (agent.cpp:110)
  std::vector<int32_t> collectives;        // collective ids named by the "collective:<name>" tags, in tag order: the
                                           // constructor calls setCollective for each (mettagrid_c.cpp:102-127), the last stays
  std::vector<int32_t> goal_values;        // goal token values, precomputed as BIND/mettagrid_c.cpp:363-395
  std::vector<std::pair<int32_t, MgKVList>> regen;  // vibe -> (item, delta) list
  std::vector<int32_t> diversity;
  MgKVList damage_threshold, damage_resources;
};
struct MgProtocol {
  int32_t min_agents, cooldown;
  std::vector<int32_t> vibes;
  MgKVList inputs, outputs;                             // pybind-converted map order
  std::vector<int32_t> inputs_scaled, outputs_scaled;   // key order of the scaled copy (assembler.hpp:427-453)
};
struct MgProtocolKey {
  uint64_t key;
  std::vector<int32_t> protocols;  // indices, min_agents descending
};
struct MgAssembler {
  int32_t type_id, vibe, allow_partial_usage, max_uses, clip_immune, start_clipped, chest_search_distance;
  std::vector<int32_t> tags;
  std::vector<MgProtocol> protocols;
  std::vector<MgProtocolKey> keys;
};
struct MgWall {
  int32_t type_id, vibe;
  std::vector<int32_t> tags;
};
struct MgChest {
  int32_t type_id, vibe;
  std::vector<int32_t> tags;
  MgInventoryCfg inv;
  std::vector<std::pair<int32_t, MgKVList>> vibe_transfers;  // vibe -> (item, delta) list (map order)
};
struct MgCell {
  int32_t kind, idx;
};
struct MgAction {
  int32_t kind, arg;
};
struct MgHandlerRes {
  MgKVList required, consumed;
};
struct MgTransferEffect {
  int32_t vibe;
  MgKVList actor, target;
};

struct MgConfig {
  std::vector<MgKVList> collective_initial;  // per collective: (item, amount)
  // Clipper (systems/clipper.hpp:14-46, clipper_config.hpp:11-27)
  int32_t clipper = 0, clip_length_scale = 0, clip_cutoff = 0, clip_period = 0;
  std::vector<MgProtocol> unclip;     // unclipping protocols, configuration order
  std::vector<uint64_t> unclip_key;   // sorted vibes, 8 bits each (assembler.hpp:326-331)
  int32_t R = 0, A = 0, obs_w = 0, obs_h = 0, T = 0, token_base = 256, max_steps = 0, episode_truncates = 0;
  int32_t g_completion = 1, g_last_action = 1, g_last_reward = 1, g_compass = 0, g_goal = 0;
  int32_t protocol_details_obs = 1, regen_interval = 0, init_obs_passes = 1, n_vibes = 0, n_group_names = 0;
  int32_t fid[MG_F_COUNT];
  int32_t n_inv_tokens = 2;
  std::vector<int32_t> fid_inv;  // [R][n_inv_tokens]
  std::vector<int32_t> fid_in, fid_out;
  std::vector<MgAction> actions;
  MgHandlerRes handler[MG_NUM_HANDLERS];
  int32_t max_priority = 1;
  std::vector<int32_t> vibe_handler;  // [256]
  // attack
  MgKVList atk_consumed, atk_defense, atk_armor, atk_weapon, atk_actor_delta, atk_target_delta;
  std::vector<int32_t> atk_loot;
  int32_t atk_freeze = 0;
  std::vector<int32_t> atk_vibe_bonus;    // [256]
  std::vector<int32_t> vibe_resource;     // [256] resource id whose name equals the vibe's name, or -1
  // transfer
  MgKVList xfer_required;
  std::vector<MgTransferEffect> xfer_effects;
  std::vector<MgGroup> groups;
  std::vector<MgCell> cells;  // index = cell code (0 unused/empty)
  std::vector<MgWall> walls;
  std::vector<MgAssembler> assemblers;
  std::vector<MgChest> chests;
  int32_t n_agent_slots = 0, n_game_slots = 0;

  MgAgentSlots aslots() const { return MgAgentSlots{R, n_group_names}; }
  MgGameSlots gslots() const { return MgGameSlots{R, (int)cells.size() - 1}; }

  // ---- reader ------------------------------------------------------------------------------------------
  struct Reader {
    const int32_t* p;
    size_t n, i = 0;
    int32_t i32() {
      if (i >= n) throw std::runtime_error("mg config blob: truncated");
      return p[i++];
    }
    float f32() {
      int32_t v = i32();
      float f;
      std::memcpy(&f, &v, 4);
      return f;
    }
    std::vector<int32_t> list() {
      int32_t c = i32();
      if (c < 0 || (size_t)c > n) throw std::runtime_error("mg config blob: bad list length");
      std::vector<int32_t> v((size_t)c);
      for (auto& x : v) x = i32();
      return v;
    }
    MgKVList kvlist() {
      int32_t c = i32();
      if (c < 0 || (size_t)c > n) throw std::runtime_error("mg config blob: bad kv-list length");
      MgKVList v((size_t)c);
      for (auto& x : v) {
        x.k = i32();
        x.v = i32();
      }
      return v;
    }
  };

  static MgInventoryCfg read_inv(Reader& r) {
    MgInventoryCfg c;
    int32_t nl = r.i32();
    c.limits.resize((size_t)nl);
    for (auto& l : c.limits) {
      l.base = r.i32();
      l.resources = r.list();
      l.modifiers = r.kvlist();
    }
    c.limits_map_order = r.kvlist();
    c.initial = r.kvlist();
    return c;
  }

  static MgConfig parse(const int32_t* words, size_t n_words) {
    Reader r{words, n_words};
    MgConfig c;
    if (r.i32() != MG_BLOB_MAGIC) throw std::runtime_error("mg config blob: bad magic");
    if (r.i32() != MG_BLOB_VERSION) throw std::runtime_error("mg config blob: version mismatch");
    c.R = r.i32(); c.A = r.i32(); c.obs_w = r.i32(); c.obs_h = r.i32(); c.T = r.i32(); c.token_base = r.i32();
    c.max_steps = r.i32(); c.episode_truncates = r.i32();
    c.g_completion = r.i32(); c.g_last_action = r.i32(); c.g_last_reward = r.i32(); c.g_compass = r.i32();
    c.g_goal = r.i32(); c.protocol_details_obs = r.i32(); c.regen_interval = r.i32(); c.init_obs_passes = r.i32();
    c.n_vibes = r.i32(); c.n_group_names = r.i32();
    for (int i = 0; i < MG_F_COUNT; i++) c.fid[i] = r.i32();
    c.n_inv_tokens = r.i32();
    c.fid_inv = r.list();
    c.fid_in = r.list();
    c.fid_out = r.list();
    {
      int32_t na = r.i32();
      c.actions.resize((size_t)na);
      for (auto& a : c.actions) { a.kind = r.i32(); a.arg = r.i32(); }
    }
    for (int h = 0; h < MG_NUM_HANDLERS; h++) { c.handler[h].required = r.kvlist(); c.handler[h].consumed = r.kvlist(); }
    c.max_priority = r.i32();
    c.vibe_handler = r.list();
    c.atk_consumed = r.kvlist(); c.atk_defense = r.kvlist(); c.atk_armor = r.kvlist(); c.atk_weapon = r.kvlist();
    c.atk_actor_delta = r.kvlist(); c.atk_target_delta = r.kvlist(); c.atk_loot = r.list(); c.atk_freeze = r.i32();
    c.atk_vibe_bonus = r.list(); c.vibe_resource = r.list();
    c.xfer_required = r.kvlist();
    {
      int32_t ne = r.i32();
      c.xfer_effects.resize((size_t)ne);
      for (auto& e : c.xfer_effects) { e.vibe = r.i32(); e.actor = r.kvlist(); e.target = r.kvlist(); }
    }
    {
      int32_t ng = r.i32();
      c.groups.resize((size_t)ng);
      for (auto& g : c.groups) {
        g.group_id = r.i32(); g.group_name_idx = r.i32(); g.freeze_duration = r.i32(); g.initial_vibe = r.i32();
        g.tags = r.list();
        g.inv = read_inv(r);
        int32_t ns = r.i32();
        g.stat_rewards.resize((size_t)ns);
        for (auto& s : g.stat_rewards) {
          s.agent_slot = r.i32(); s.game_slot = r.i32(); s.weight = r.f32(); s.has_max = r.i32(); s.max = r.f32();
          s.coll_kind = r.i32(); s.coll_res = r.i32();
        }
        g.collectives = r.list();
        g.goal_values = r.list();
        int32_t nr = r.i32();
        g.regen.resize((size_t)nr);
        for (auto& v : g.regen) { v.first = r.i32(); v.second = r.kvlist(); }
        g.diversity = r.list();
        g.damage_threshold = r.kvlist();
        g.damage_resources = r.kvlist();
      }
    }
    {
      int32_t nc = r.i32();
      c.cells.resize((size_t)nc);
      for (auto& x : c.cells) { x.kind = r.i32(); x.idx = r.i32(); }
    }
    {
      int32_t nw = r.i32();
      c.walls.resize((size_t)nw);
      for (auto& w : c.walls) { w.type_id = r.i32(); w.vibe = r.i32(); w.tags = r.list(); }
    }
    {
      int32_t na = r.i32();
      c.assemblers.resize((size_t)na);
      for (auto& a : c.assemblers) {
        a.type_id = r.i32(); a.vibe = r.i32(); a.allow_partial_usage = r.i32(); a.max_uses = r.i32();
        a.clip_immune = r.i32(); a.start_clipped = r.i32(); a.chest_search_distance = r.i32();
        a.tags = r.list();
        int32_t np = r.i32();
        a.protocols.resize((size_t)np);
        for (auto& p : a.protocols) {
          p.min_agents = r.i32(); p.cooldown = r.i32(); p.vibes = r.list(); p.inputs = r.kvlist(); p.outputs = r.kvlist();
          p.inputs_scaled = r.list(); p.outputs_scaled = r.list();
        }
        int32_t nk = r.i32();
        a.keys.resize((size_t)nk);
        for (auto& k : a.keys) {
          uint32_t lo = (uint32_t)r.i32(), hi = (uint32_t)r.i32();
          k.key = ((uint64_t)hi << 32) | lo;
          k.protocols = r.list();
        }
      }
    }
    {
      int32_t nc = r.i32();
      c.chests.resize((size_t)nc);
      for (auto& ch : c.chests) {
        ch.type_id = r.i32(); ch.vibe = r.i32(); ch.tags = r.list();
        ch.inv = read_inv(r);
        int32_t nv = r.i32();
        ch.vibe_transfers.resize((size_t)nv);
        for (auto& v : ch.vibe_transfers) { v.first = r.i32(); v.second = r.kvlist(); }
      }
    }
    {  // collectives (collective.hpp:33-42): initial inventory, in the configuration map's iteration order
      int32_t nc = r.i32();
      c.collective_initial.resize((size_t)nc);
      for (auto& kv : c.collective_initial) kv = r.kvlist();
    }
    c.clipper = r.i32();
    if (c.clipper) {
      c.clip_length_scale = r.i32(); c.clip_cutoff = r.i32(); c.clip_period = r.i32();
      int32_t np = r.i32();
      if (np < 1) throw std::runtime_error("Clipper config provided but unclipping_protocols is empty");
      c.unclip.resize((size_t)np);
      c.unclip_key.resize((size_t)np);
      for (size_t i = 0; i < (size_t)np; i++) {
        MgProtocol& p = c.unclip[i];
        p.min_agents = r.i32(); p.cooldown = r.i32(); p.vibes = r.list(); p.inputs = r.kvlist(); p.outputs = r.kvlist();
        p.inputs_scaled = r.list(); p.outputs_scaled = r.list();
        uint32_t lo = (uint32_t)r.i32(), hi = (uint32_t)r.i32();
        c.unclip_key[i] = ((uint64_t)hi << 32) | lo;
      }
    }
    c.n_agent_slots = r.i32();
    c.n_game_slots = r.i32();
    if (r.i32() != MG_BLOB_MAGIC) throw std::runtime_error("mg config blob: bad trailer");
    if (c.n_agent_slots != c.aslots().count() || c.n_game_slots != c.gslots().count())
      throw std::runtime_error("mg config blob: stat slot layout mismatch between host language and library");
    if (c.R < 0 || c.R > 255 || c.A < 1 || c.T < 1 || c.obs_w > 15 || c.obs_h > 15)
      throw std::runtime_error("mg config blob: parameter out of range (observation window must be <= 15x15)");
    return c;
  }
};
