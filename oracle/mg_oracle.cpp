// mg_oracle.cpp -- CPU restatement of the reference MettaGrid step path.   *** TEST INFRASTRUCTURE ONLY ***
//
// This file is the parity oracle of the repository.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it; the product (metta_amd/ + libmettagrid_hip.so) never does.
//
// It restates, as plain sequential C++ over flat arrays, what packages/mettagrid/cpp computes for
// reset/step/observations (paths below are relative to packages/mettagrid/cpp):
//   MettaGrid ctor / _init_grid / _init_buffers   bindings/mettagrid_c.cpp:46-155,159-245,264-289
//   MettaGrid::_step                               bindings/mettagrid_c.cpp:572-678
//   ActionHandler::handle_action                   include/mettagrid/actions/action_handler.hpp:99-153
//   Move / Attack / Transfer / ChangeVibe / Noop   include/mettagrid/actions/{move,attack,transfer,change_vibe,noop}.hpp
//   Inventory / shared_update / transfer_resources src/mettagrid/objects/{inventory,has_inventory}.cpp
//   Agent (stats hooks, rewards, damage, obs)      src/mettagrid/objects/agent.cpp
//   Assembler / Chest / Wall                       include/mettagrid/objects/{assembler,chest,wall}.hpp
//   observation encoding                           bindings/mettagrid_c.cpp:397-563, systems/observation_encoder.hpp,
//                                                  systems/packed_coordinate.hpp:74-143
// and the libstdc++ behaviour the reference inherits: std::mt19937, std::shuffle's
// paired-draw Fisher-Yates with the Lemire bounded sampler, std::discrete_distribution, and
// unordered_map<uint8_t,...> iteration order (for inventories).
//
// Pinning: tests/test_oracle_golden.py (test_oracle_vs_reference_core_fuzz) compares it, step by step and bit for bit, with the reference's own
// compiled core (oracle/_ref, built by oracle/Makefile from /root/reference) and tests/golden/* holds vectors
// emitted by that reference.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "mg_blob.h"

namespace {

// ------------------------------------------------------------------------------------------------------------
// std::mt19937 + libstdc++ uniform_int_distribution (Lemire) + std::shuffle        
// ------------------------------------------------------------------------------------------------------------
struct MT19937 {
  uint32_t mt[624];
  int idx;
  uint64_t draws = 0;
  void seed(uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  void twist() {
    for (int i = 0; i < 624; i++) {
      uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
      mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    idx = 0;
  }
  uint32_t next() {
    if (idx >= 624) twist();
    uint32_t y = mt[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    draws++;
    return y;
  }
  // uniform integer in [0, range), range <= 2^32 - 1   (/usr/include/c++/11/bits/uniform_int_dist.h:258-269)
  uint32_t bounded(uint32_t range) {
    uint64_t product = (uint64_t)next() * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
      uint32_t threshold = (0u - range) % range;
      while (low < threshold) {
        product = (uint64_t)next() * (uint64_t)range;
        low = (uint32_t)product;
      }
    }
    return (uint32_t)(product >> 32);
  }
  // generate_canonical<double,53> for a 32-bit engine: two draws.
  double canonical() {
    double a = (double)next();
    double b = (double)next();
    double r = (a + b * 4294967296.0) / 18446744073709551616.0;
    if (r >= 1.0) r = std::nextafter(1.0, 0.0);
    return r;
  }
};

// std::shuffle(first, last, mt19937)  (/usr/include/c++/11/bits/stl_algo.h:3726-3792)
template <class T>
void std_shuffle(std::vector<T>& v, MT19937& g) {
  size_t n = v.size();
  if (n == 0) return;
  if ((0xFFFFFFFFull / n) >= n) {
    size_t i = 1;
    if ((n % 2) == 0) {
      uint32_t j = g.bounded(2);
      std::swap(v[1], v[j]);
      i = 2;
    }
    while (i != n) {
      uint32_t s = (uint32_t)i + 1;
      uint32_t x = g.bounded(s * (s + 1));
      std::swap(v[i], v[x / (s + 1)]);
      std::swap(v[i + 1], v[x % (s + 1)]);
      i += 2;
    }
  } else {
    for (size_t i = 1; i < n; i++) std::swap(v[i], v[g.bounded((uint32_t)i + 1)]);
  }
}

// ------------------------------------------------------------------------------------------------------------
// unordered_map<uint8_t, uint16_t> with libstdc++ iteration order                      
// ------------------------------------------------------------------------------------------------------------
static const int kFastBkt[14] = {2, 2, 2, 3, 5, 5, 7, 7, 11, 11, 11, 11, 13, 13};
static const int kPrimes[] = {17,  19,  23,  29,  31,  37,  41,  43,  47,  53,  59,  61,  67,  71,  73,  79,  83,
                              89,  97,  103, 109, 113, 127, 137, 139, 149, 157, 167, 179, 193, 199, 211, 227, 241,
                              257, 277, 293, 313, 337, 359, 383, 409, 439, 467, 503, 541, 577, 619};
struct U8Map {
  std::vector<uint8_t> order;  // iteration order, front = begin()
  uint16_t val[256];
  bool has[256];
  int nb = 1, next_resize = 0;
  U8Map() {
    std::memset(val, 0, sizeof(val));
    std::memset(has, 0, sizeof(has));
  }
  int next_bkt(int n) {
    if (n < 14) {
      if (n == 0) return 1;
      next_resize = kFastBkt[n];
      return kFastBkt[n];
    }
    for (int p : kPrimes)
      if (p >= n) {
        next_resize = p;
        return p;
      }
    throw std::runtime_error("U8Map: too large");
  }
  void insert_node(uint8_t k) {
    int b = k % nb;
    for (size_t i = 0; i < order.size(); i++)
      if (order[i] % nb == b) {
        order.insert(order.begin() + (long)i, k);
        return;
      }
    order.insert(order.begin(), k);
  }
  void rehash(int m) {
    std::vector<uint8_t> old;
    old.swap(order);
    nb = m;
    for (uint8_t k : old) insert_node(k);
  }
  // `map[k]` -- inserts a zero entry when absent (inventory.cpp:38 relies on that)
  uint16_t& at_or_insert(uint8_t k) {
    if (!has[k]) {
      int n_elt = (int)order.size();
      if (n_elt + 1 > next_resize) {
        int min_bkts = std::max(n_elt + 1, next_resize ? 0 : 11);
        if (min_bkts >= nb)
          rehash(next_bkt(std::max(min_bkts + 1, nb * 2)));
        else
          next_resize = nb;
      }
      insert_node(k);
      has[k] = true;
      val[k] = 0;
    }
    return val[k];
  }
  void erase(uint8_t k) {
    if (!has[k]) return;
    has[k] = false;
    val[k] = 0;
    order.erase(std::find(order.begin(), order.end(), k));
  }
  bool contains(uint8_t k) const { return has[k]; }
  uint16_t get(uint8_t k) const { return has[k] ? val[k] : 0; }
};

// ObservationPattern (packed_coordinate.hpp:74-143): offsets by (manhattan distance, dr, dc).  The iterator walks
// shells d = 0,1,..; within a shell dr = -d..d, and for each dr first -dc then +dc, keeping offsets inside
// [-h/2, h/2] x [-w/2, w/2] (C++ int division) until h*w offsets were emitted.
static std::vector<std::pair<int, int>> observation_pattern(int h, int w) {
  std::vector<std::pair<int, int>> out;
  int emitted = 0, max_emitted = h * w;
  int row_min = -(h / 2), row_max = h / 2, col_min = -(w / 2), col_max = w / 2;
  for (int d = 0; emitted < max_emitted && d <= h + w + 2; d++)
    for (int dr = -d; dr <= d && emitted < max_emitted; dr++) {
      int dc = d - std::abs(dr);
      for (int step = 0; step < (dc == 0 ? 1 : 2) && emitted < max_emitted; step++) {
        int rr = dr, cc = (step == 0 ? -dc : dc);
        if (rr >= row_min && rr <= row_max && cc >= col_min && cc <= col_max) {
          out.push_back({rr, cc});
          emitted++;
        }
      }
    }
  return out;
}

// ------------------------------------------------------------------------------------------------------------
struct Stats {
  std::vector<float> v;
  std::vector<uint8_t> touched;
  void init(int n) {
    v.assign((size_t)n, 0.f);
    touched.assign((size_t)n, 0);
  }
  void add(int s, float a) {
    v[(size_t)s] += a;
    touched[(size_t)s] = 1;
  }
  void set(int s, float a) {
    v[(size_t)s] = a;
    touched[(size_t)s] = 1;
  }
  float get(int s) const { return s < 0 ? 0.f : v[(size_t)s]; }
};

struct Env;

// Inventory (inventory.hpp:36-67, inventory.cpp:13-158)
struct Inventory {
  const MgInventoryCfg* cfg = nullptr;
  U8Map items;
  std::vector<int> limit_of;        // item -> limit index or -1
  std::vector<uint16_t> limit_amt;  // per limit: tracked amount (uint16 arithmetic, inventory.cpp:68)
  std::vector<uint8_t> modifier;    // item -> is a modifier of some limit
  Env* env = nullptr;
  int owner_kind = 0, owner_idx = -1;  // MG_CELL_AGENT / MG_CELL_CHEST / 0 = nobody

  void init(const MgInventoryCfg* c, int R, Env* e, int kind, int idx) {
    cfg = c;
    env = e;
    owner_kind = kind;
    owner_idx = idx;
    limit_of.assign(256, -1);
    modifier.assign(256, 0);
    limit_amt.assign(c->limits.size(), 0);
    for (const auto& kv : c->limits_map_order) limit_of[(size_t)kv.k] = kv.v;
    // is_modifier() walks `_limits` (one entry per resource), so only limits still referenced there count.
    for (const auto& kv : c->limits_map_order)
      for (const auto& m : c->limits[(size_t)kv.v].modifiers) modifier[(size_t)m.k] = 1;
    (void)R;
  }
  uint16_t effective_limit(int li) const {
    const MgLimit& l = cfg->limits[(size_t)li];
    int eff = (uint16_t)l.base;
    for (const auto& m : l.modifiers)
      if (items.contains((uint8_t)m.k)) eff += (int)items.get((uint8_t)m.k) * (int)(uint16_t)m.v;
    if (eff < 0) eff = 0;
    if (eff > 65535) eff = 65535;
    return (uint16_t)eff;
  }
  uint16_t amount(int item) const { return items.get((uint8_t)item); }
  uint16_t free_space(int item) const {
    int li = limit_of[(size_t)item];
    if (li < 0) return (uint16_t)(65535 - amount(item));
    uint16_t used = limit_amt[(size_t)li];
    uint16_t eff = effective_limit(li);
    return eff > used ? (uint16_t)(eff - used) : 0;
  }
  int update(int item, int attempted_delta, bool ignore_limits = false);
  void enforce_all_limits();
};

struct Agent {
  int r, c, prev_r, prev_c;
  int group;      // index into cfg.groups
  int16_t frozen;
  uint8_t vibe;
  uint32_t steps_without_motion = 0;
  float current_stat_reward = 0.f;
  uint32_t obj_id;
  Inventory inv;
  Stats stats;
  std::vector<uint8_t> presence;
  size_t diversity = 0;
  int collective = -1;  // Alignable::_collective
};
// Collective (collective.hpp:19-109): the part of it the step path can observe is its StatsTracker, read by
// Agent::compute_stat_rewards: "aligned.<type_name>" kept by addMember / removeMember, and
// "collective.<resource>.deposited" written by on_inventory_change while the constructor fills the inventory.
struct CollectiveInst {
  std::vector<int> members;          // agent indices
  int aligned_agents = 0;            // _aligned_counts["agent"]
  float stat_aligned_agent = 0.f;    // stats "aligned.agent"
  std::vector<float> stat_deposited; // stats "collective.<r>.deposited"
};
struct AssemblerInst {
  int r, c, cfg;
  uint32_t obj_id;
  bool is_clipped;
  uint32_t cooldown_end = 0, cooldown_duration = 0, uses_count = 0;
  int unclip = -1;          // index of the unclipping protocol a Clipper installed (become_clipped, assembler.hpp:411-421)
  bool by_clipper = false;  // Assembler::clipper_ptr != nullptr
};
// Clipper (systems/clipper.hpp:14-239).  `border_assemblers` / `unclipped_assemblers` are std::set<Assembler*>, iterated
// in POINTER order (clipper.hpp:24-26,207-229); here: creation order (row-major), which is the order glibc malloc
// returned the assemblers in for every map of the fixtures generated with the reference (tests/golden/feature_12x.npz).
// The adjacency lists (clipper.hpp:130-172) only feed commutative weight updates and set insertions, so they are not
// materialised: j is adjacent to i iff both are members, i != j and L_inf distance <= cutoff * length_scale.
// A border member whose weight has dropped to 0 can never be picked (clipper.hpp:221-227), so the border set is
// "member, not clipped, weight > 0".
struct ClipperInst {
  bool on = false;
  int length_scale = 0;
  uint32_t cutoff = 0, period = 0;
  MT19937 rng;
  std::vector<uint32_t> weight;  // assembler_infection_weight, per assembler index
  std::vector<char> member;      // not clip_immune
};
// clipper.hpp:65-98: auto length scale from the sparsity of the (non-immune) assemblers.
static int clipper_auto_length_scale(int width, int height, size_t n_members) {
  uint32_t sparsity = (uint32_t)(((size_t)(width * height)) / n_members);
  int32_t root = 1, next = 10;
  for (int i = 0; i < 10; i++) {
    if (next == root || root == 0) break;
    root = next;
    next = (int32_t)((sparsity + (uint32_t)(root * root) + (uint32_t)(2 * root - 2)) / (uint32_t)(2 * root));
  }
  return std::max(root / 2, 1);
}
// clipper.hpp:113-120
static uint32_t clipper_infection_weight(int dist, int length_scale, uint32_t cutoff) {
  uint32_t scaled = (uint32_t)dist / (uint32_t)length_scale;
  if (scaled > cutoff) return 0;
  return 1u << (cutoff - scaled);
}
struct ChestInst {
  int r, c, cfg;
  uint32_t obj_id;
  Inventory inv;
};
struct Obj {
  int kind, idx, r, c;
};

struct Env {
  MgConfig cfg;
  MgAgentSlots AS;
  MgGameSlots GS;
  int H = 0, W = 0;
  std::vector<int32_t> grid;  // object id per cell, 0 = empty
  std::vector<Obj> objects;   // index = object id
  std::vector<Agent> agents;
  std::vector<AssemblerInst> assemblers;
  std::vector<ChestInst> chests;
  std::vector<CollectiveInst> collectives;
  ClipperInst clipper;
  Stats game;
  MT19937 rng;
  uint32_t current_step = 0;
  std::vector<uint8_t> obs, terminals, truncations, action_success;
  std::vector<float> rewards, episode_rewards;
  std::vector<std::pair<int, int>> pattern;

  int32_t& cell(int r, int c) { return grid[(size_t)r * (size_t)W + (size_t)c]; }
  bool valid(int r, int c) const { return r >= 0 && c >= 0 && r < H && c < W; }

  void init(const int32_t* blob, size_t n, const uint8_t* map, int h, int w, uint32_t seed);
  void step(const int32_t* actions);
  void compute_observations(const std::vector<int32_t>& executed);
  void compute_observation(size_t ai, int action);
  size_t append(std::vector<uint8_t>& feats, uint8_t* dst, size_t room, uint8_t loc);

  // hooks
  void on_inventory_change(Inventory& inv, int item, int delta);
  // actions
  bool handle_action(Agent& a, const MgAction& act);
  bool do_move(Agent& a, int orientation);
  bool try_attack(Agent& a, int target_obj);
  bool try_transfer(Agent& a, int target_obj);
  bool assembler_use(AssemblerInst& as, Agent& actor);
  bool chest_use(ChestInst& ch, Agent& actor);
  const MgProtocol* current_protocol(const AssemblerInst& as);
  void object_features(int obj_id, std::vector<uint8_t>& out);
  void append_inventory_tokens(std::vector<uint8_t>& out, int item, int amount);
  void check_and_apply_damage(Agent& a);
  void compute_stat_rewards(size_t ai);
  void surrounding_positions(const AssemblerInst& as, std::vector<std::pair<int, int>>& out);
  // clipper
  void clipper_init(uint32_t seed);
  bool clipper_adjacent(size_t i, size_t j, uint32_t* w) const;
  void clip_assembler(size_t i);
  void on_unclip_assembler(size_t i);
  int pick_assembler_to_clip();
  void maybe_clip_new_assembler();
};

int Inventory::update(int item, int attempted_delta, bool ignore_limits) {
  uint16_t initial = items.at_or_insert((uint8_t)item);
  int new_amount = (int)initial + attempted_delta;
  int mn = 0, mx = 65535;
  int li = limit_of[(size_t)item];
  if (!ignore_limits && li >= 0) {
    uint16_t eff = effective_limit(li);
    int used_by_others = (int)limit_amt[(size_t)li] - (int)initial;
    if (used_by_others < 0) used_by_others = 0;
    int max_int = (int)eff - used_by_others;
    if (max_int < 0) max_int = 0;
    mx = (uint16_t)max_int;
  }
  // std::clamp(v, lo, hi): (v < lo) ? lo : (hi < v) ? hi : v
  uint16_t clamped = (uint16_t)((new_amount < mn) ? mn : (mx < new_amount) ? mx : new_amount);
  if (clamped == 0)
    items.erase((uint8_t)item);
  else
    items.at_or_insert((uint8_t)item) = clamped;
  if (li >= 0) limit_amt[(size_t)li] = (uint16_t)(limit_amt[(size_t)li] + (clamped - initial));
  int clamped_delta = (int)clamped - (int)initial;
  if (owner_kind != 0 && clamped_delta != 0) env->on_inventory_change(*this, item, clamped_delta);
  if (clamped_delta < 0 && modifier[(size_t)item]) enforce_all_limits();
  return clamped_delta;
}

void Inventory::enforce_all_limits() {
  // std::set<SharedInventoryLimit*>: pointer order == allocation order of the limit_defs (inventory.cpp:14-22).
  std::vector<uint8_t> live(cfg->limits.size(), 0);
  for (const auto& kv : cfg->limits_map_order) live[(size_t)kv.v] = 1;
  for (size_t li = 0; li < cfg->limits.size(); li++) {
    if (!live[li]) continue;
    int excess = (int)limit_amt[li] - (int)effective_limit((int)li);
    if (excess <= 0) continue;
    for (const auto& kv : cfg->limits_map_order) {
      if ((size_t)kv.v != li) continue;
      int current = amount(kv.k);
      int to_drop = std::min(current, excess);
      if (to_drop > 0) {
        update(kv.k, -to_drop);
        excess = (int)limit_amt[li] - (int)effective_limit((int)li);
      }
      if (excess <= 0) break;
    }
  }
}

// HasInventory::shared_update (has_inventory.cpp:7-73)
int shared_update(std::vector<Inventory*> inventories, int item, int delta) {
  if (inventories.empty()) return 0;
  int delta_remaining = delta;
  std::vector<Inventory*> consider, next = inventories;
  int num_remaining = (int)next.size();
  int per = delta_remaining / num_remaining;
  do {
    consider = next;
    next.clear();
    for (Inventory* inv : consider) {
      bool now;
      if (delta_remaining > 0)
        now = (int)inv->free_space(item) <= per;
      else
        now = (int)inv->amount(item) <= -per;
      if (now) {
        delta_remaining -= inv->update(item, per);
        num_remaining--;
        if (num_remaining > 0) per = delta_remaining / num_remaining;
      } else {
        next.push_back(inv);
      }
    }
  } while (consider.size() != next.size());
  if (num_remaining == 0) return delta - delta_remaining;
  for (int i = (int)consider.size() - 1; i >= 0; i--) {
    int d = delta_remaining / (i + 1);
    int actual = consider[(size_t)i]->update(item, d);
    delta_remaining -= actual;
  }
  return delta - delta_remaining;
}

// HasInventory::transfer_resources (has_inventory.cpp:76-108)
int transfer_resources(Inventory& source, Inventory& target, int item, int delta, bool destroy_untransferred) {
  if (delta <= 0) return 0;
  int max_give = std::min((int)source.amount(item), delta);
  int max_recv = (int)target.free_space(item);
  int amount = std::min(max_give, max_recv);
  int loss = destroy_untransferred ? max_give : amount;
  source.update(item, -loss);
  target.update(item, amount);
  return amount;
}

void Env::on_inventory_change(Inventory& inv, int item, int delta) {
  if (inv.owner_kind == MG_CELL_AGENT) {
    // Agent::on_inventory_change (agent.cpp:74-101)
    Agent& a = agents[(size_t)inv.owner_idx];
    uint16_t amount = a.inv.amount(item);
    if (delta != 0) {
      if (delta > 0)
        a.stats.add(AS.gained(item), (float)delta);
      else
        a.stats.add(AS.lost(item), (float)(-delta));
      a.stats.set(AS.amount(item), (float)amount);
    }
    const MgGroup& g = cfg.groups[(size_t)a.group];
    bool tracked = std::find(g.diversity.begin(), g.diversity.end(), item) != g.diversity.end();
    if (tracked) {
      bool had = a.presence[(size_t)item] != 0, has = amount > 0;
      if (had != has) {
        a.presence[(size_t)item] = has ? 1 : 0;
        a.diversity += has ? 1 : (size_t)-1;
        a.stats.set(AS.diversity(), (float)a.diversity);
      }
    }
  } else if (inv.owner_kind == MG_CELL_CHEST) {
    // Chest::on_inventory_change (chest.hpp:56-63)
    ChestInst& ch = chests[(size_t)inv.owner_idx];
    if (delta > 0)
      game.add(GS.chest_deposited(item), (float)delta);
    else if (delta < 0)
      game.add(GS.chest_withdrawn(item), (float)(-delta));
    game.set(GS.chest_amount(item), (float)ch.inv.amount(item));
  }
}

void Env::init(const int32_t* blob, size_t n, const uint8_t* map, int h, int w, uint32_t seed) {
  cfg = MgConfig::parse(blob, n);
  AS = cfg.aslots();
  GS = cfg.gslots();
  H = h;
  W = w;
  rng.seed(seed);
  current_step = 0;
  grid.assign((size_t)h * (size_t)w, 0);
  objects.clear();
  objects.push_back(Obj{0, 0, 0, 0});
  game.init(cfg.n_game_slots);
  agents.reserve((size_t)cfg.A + 8);
  // _init_grid (mettagrid_c.cpp:181-244): row-major; ids in creation order; agent_id in order of appearance.
  size_t n_agents = 0, n_chests = 0;
  for (int i = 0; i < h * w; i++) {
    int code = map[i];
    if (code == 0) continue;
    if ((size_t)code >= cfg.cells.size()) throw std::runtime_error("Unknown object type: cell code out of range");
    if (cfg.cells[(size_t)code].kind == MG_CELL_AGENT) n_agents++;
    if (cfg.cells[(size_t)code].kind == MG_CELL_CHEST) n_chests++;
  }
  agents.resize(n_agents);
  chests.resize(n_chests);
  size_t ai = 0, ci = 0;
  for (int r = 0; r < h; r++)
    for (int c = 0; c < w; c++) {
      int code = map[(size_t)r * (size_t)w + (size_t)c];
      if (code == 0) continue;
      const MgCell& cell_cfg = cfg.cells[(size_t)code];
      uint32_t id = (uint32_t)objects.size();
      switch (cell_cfg.kind) {
        case MG_CELL_WALL:
          objects.push_back(Obj{MG_CELL_WALL, cell_cfg.idx, r, c});
          game.add(GS.objects(code), 1.f);
          break;
        case MG_CELL_AGENT: {
          Agent& a = agents[ai];
          const MgGroup& g = cfg.groups[(size_t)cell_cfg.idx];
          a.r = a.prev_r = r;
          a.c = a.prev_c = c;
          a.group = cell_cfg.idx;
          a.frozen = 0;
          a.vibe = (uint8_t)g.initial_vibe;
          a.obj_id = id;
          a.stats.init(cfg.n_agent_slots);
          a.presence.assign((size_t)std::max(cfg.R, 1), 0);
          a.inv.init(&g.inv, cfg.R, this, MG_CELL_AGENT, (int)ai);
          objects.push_back(Obj{MG_CELL_AGENT, (int)ai, r, c});
          // populate_initial_inventory (agent.cpp:48-52): ignore_limits, stats hooks fire
          for (const auto& kv : g.inv.initial) a.inv.update(kv.k, kv.v, true);
          ai++;
          break;
        }
        case MG_CELL_ASSEMBLER: {
          AssemblerInst as;
          as.r = r;
          as.c = c;
          as.cfg = cell_cfg.idx;
          as.obj_id = id;
          as.is_clipped = cfg.assemblers[(size_t)cell_cfg.idx].start_clipped != 0;
          assemblers.push_back(as);
          objects.push_back(Obj{MG_CELL_ASSEMBLER, (int)assemblers.size() - 1, r, c});
          game.add(GS.objects(code), 1.f);
          break;
        }
        case MG_CELL_CHEST: {
          ChestInst& ch = chests[ci];
          ch.r = r;
          ch.c = c;
          ch.cfg = cell_cfg.idx;
          ch.obj_id = id;
          const MgChest& cc = cfg.chests[(size_t)cell_cfg.idx];
          ch.inv.init(&cc.inv, cfg.R, this, MG_CELL_CHEST, (int)ci);
          objects.push_back(Obj{MG_CELL_CHEST, (int)ci, r, c});
          // chest.hpp:83-87: initial inventory BEFORE the objects.<name> stat, amount > 0 only
          for (const auto& kv : cc.inv.initial)
            if (kv.v > 0) ch.inv.update(kv.k, kv.v, true);
          game.add(GS.objects(code), 1.f);
          ci++;
          break;
        }
        default:
          throw std::runtime_error("Unable to create object: bad cell kind");
      }
      cell(r, c) = (int32_t)id;
    }
  // collectives (mettagrid_c.cpp:95-128): built after the grid; objects in id order call setCollective once per
  // "collective:<name>" tag, in tag order (alignable.cpp:5-16: leave the old one, join the new one)
  collectives.assign(cfg.collective_initial.size(), CollectiveInst{});
  for (size_t k = 0; k < collectives.size(); k++) {
    collectives[k].stat_deposited.assign((size_t)std::max(cfg.R, 1), 0.f);
    for (const auto& kv : cfg.collective_initial[k])
      if (kv.v > 0)  // collective.hpp:36-41: ignore_limits; Inventory clamps to uint16 (inventory.cpp:41-56)
        collectives[k].stat_deposited[(size_t)kv.k] += (float)std::min(kv.v, 65535);
  }
  for (size_t id = 1; id < objects.size(); id++) {
    if (objects[id].kind != MG_CELL_AGENT) continue;  // only agents are Alignable (agent.hpp:21)
    const int ai2 = objects[id].idx;
    Agent& a = agents[(size_t)ai2];
    for (int k : cfg.groups[(size_t)a.group].collectives) {
      if (a.collective >= 0) {  // removeMember (collective.hpp:57-70)
        CollectiveInst& old = collectives[(size_t)a.collective];
        auto it = std::find(old.members.begin(), old.members.end(), ai2);
        if (it != old.members.end()) {
          old.members.erase(it);
          old.aligned_agents--;
          old.stat_aligned_agent = old.aligned_agents <= 0 ? 0.0f : (float)old.aligned_agents;
        }
      }
      a.collective = k;
      CollectiveInst& c2 = collectives[(size_t)k];  // addMember (collective.hpp:46-54)
      if (std::find(c2.members.begin(), c2.members.end(), ai2) == c2.members.end()) {
        c2.members.push_back(ai2);
        c2.aligned_agents++;
        c2.stat_aligned_agent = (float)c2.aligned_agents;
      }
    }
  }
  size_t A = agents.size();
  obs.assign(A * (size_t)cfg.T * 3, 0xFF);
  terminals.assign(A, 0);
  truncations.assign(A, 0);
  action_success.assign(A, 0);
  rewards.assign(A, 0.f);
  episode_rewards.assign(A, 0.f);
  pattern = observation_pattern(cfg.obs_h, cfg.obs_w);
  // _init_buffers: initial observations with last_action = 0 (mettagrid_c.cpp:264-289); the Puffer path runs it
  // twice (ctor + set_buffers), which double-counts the tokens_* stats.
  std::vector<int32_t> zero(A, 0);
  for (int pass = 0; pass < std::max(1, cfg.init_obs_passes); pass++) {
    std::fill(obs.begin(), obs.end(), 0xFF);
    compute_observations(zero);
    // The constructor builds the Clipper AFTER _make_buffers has computed the first observations (mettagrid_c.cpp:
    // 139-154): a station that starts clipped shows `clipped` without protocol tokens in that pass, with them in
    // the pass set_buffers runs later.
    if (pass == 0 && cfg.clipper) clipper_init(seed);
  }
}

void Env::append_inventory_tokens(std::vector<uint8_t>& out, int item, int amount) {
  // observation_encoder.hpp:165-180
  unsigned base = (unsigned)cfg.token_base;
  out.push_back((uint8_t)cfg.fid_inv[(size_t)item * (size_t)cfg.n_inv_tokens]);
  out.push_back((uint8_t)((unsigned)amount % base));
  unsigned remaining = (unsigned)amount / base;
  for (int p = 1; p < cfg.n_inv_tokens && remaining > 0; p++) {
    out.push_back((uint8_t)cfg.fid_inv[(size_t)item * (size_t)cfg.n_inv_tokens + (size_t)p]);
    out.push_back((uint8_t)(remaining % base));
    remaining /= base;
  }
}

void Env::surrounding_positions(const AssemblerInst& as, std::vector<std::pair<int, int>>& out) {
  out.clear();
  for (int i = -1; i <= 1; i++)
    for (int j = -1; j <= 1; j++) {
      if (i == 0 && j == 0) continue;
      // GridCoord arithmetic is uint16: r+i wraps to 65535 at the border and fails is_valid_location.
      int rr = (uint16_t)(as.r + i), cc = (uint16_t)(as.c + j);
      if (rr < H && cc < W) out.push_back({rr, cc});
    }
}

const MgProtocol* Env::current_protocol(const AssemblerInst& as) {
  // Assembler::get_current_protocol (assembler.hpp:378-408); clipped assemblers use unclip protocols, which
  // only a Clipper can install -- an assembler that *starts* clipped has none, hence no protocol.
  const MgAssembler& ac = cfg.assemblers[(size_t)as.cfg];
  std::vector<std::pair<int, int>> pos;
  surrounding_positions(as, pos);
  std::vector<uint8_t> vibes;
  size_t n_agents = 0;
  for (auto& p : pos) {
    int id = cell(p.first, p.second);
    if (id && objects[(size_t)id].kind == MG_CELL_AGENT) {
      n_agents++;
      uint8_t v = agents[(size_t)objects[(size_t)id].idx].vibe;
      if (v != 0) vibes.push_back(v);
    }
  }
  std::sort(vibes.begin(), vibes.end());
  uint64_t key = 0;
  for (uint8_t v : vibes) key = (key << 8) | v;
  if (as.is_clipped) {
    // unclip_protocols = build_protocol_map({the one protocol the Clipper drew}) (clipper.hpp:186-193): found under the
    // local vibe or under 0
    if (as.unclip < 0) return nullptr;
    const MgProtocol& p = cfg.unclip[(size_t)as.unclip];
    const uint64_t pk = cfg.unclip_key[(size_t)as.unclip];
    if ((pk == key || pk == 0) && (size_t)p.min_agents <= n_agents) return &p;
    return nullptr;
  }
  for (int attempt = 0; attempt < 2; attempt++) {
    uint64_t k = attempt == 0 ? key : 0;
    for (const auto& pk : ac.keys)
      if (pk.key == k)
        for (int pi : pk.protocols)
          if ((size_t)ac.protocols[(size_t)pi].min_agents <= n_agents) return &ac.protocols[(size_t)pi];
  }
  return nullptr;
}

void Env::object_features(int obj_id, std::vector<uint8_t>& f) {
  const Obj& o = objects[(size_t)obj_id];
  auto push = [&](int feat, int val) {
    f.push_back((uint8_t)feat);
    f.push_back((uint8_t)val);
  };
  switch (o.kind) {
    case MG_CELL_WALL: {  // wall.hpp:26-38
      const MgWall& wc = cfg.walls[(size_t)o.idx];
      for (int t : wc.tags) push(cfg.fid[MG_F_TAG], t);
      if (wc.vibe != 0) push(cfg.fid[MG_F_VIBE], wc.vibe);
      break;
    }
    case MG_CELL_AGENT: {  // agent.cpp:187-213
      const Agent& a = agents[(size_t)o.idx];
      const MgGroup& g = cfg.groups[(size_t)a.group];
      push(cfg.fid[MG_F_GROUP], g.group_id);
      push(cfg.fid[MG_F_FROZEN], a.frozen != 0 ? 1 : 0);
      if (a.vibe != 0) push(cfg.fid[MG_F_VIBE], a.vibe);
      for (uint8_t item : a.inv.items.order) append_inventory_tokens(f, item, a.inv.items.get(item));
      for (int t : g.tags) push(cfg.fid[MG_F_TAG], t);
      break;
    }
    case MG_CELL_ASSEMBLER: {  // assembler.hpp:528-578
      const AssemblerInst& as = assemblers[(size_t)o.idx];
      const MgAssembler& ac = cfg.assemblers[(size_t)as.cfg];
      uint32_t remaining = as.cooldown_end <= current_step ? 0 : as.cooldown_end - current_step;
      remaining = std::min(remaining, 255u);
      if (remaining > 0) push(cfg.fid[MG_F_COOLDOWN], (int)remaining);
      if (as.is_clipped) push(cfg.fid[MG_F_CLIPPED], 1);
      if (ac.max_uses > 0) {
        uint32_t ru = as.uses_count < (uint32_t)ac.max_uses ? (uint32_t)ac.max_uses - as.uses_count : 0;
        push(cfg.fid[MG_F_REMAINING_USES], (int)std::min(ru, 255u));
      }
      if (cfg.protocol_details_obs) {
        const MgProtocol* p = current_protocol(as);
        if (p) {
          for (const auto& kv : p->inputs)
            if (kv.v > 0) push(cfg.fid_in[(size_t)kv.k], kv.v);
          for (const auto& kv : p->outputs)
            if (kv.v > 0) push(cfg.fid_out[(size_t)kv.k], kv.v);
        }
      }
      for (int t : ac.tags) push(cfg.fid[MG_F_TAG], t);
      if (ac.vibe != 0) push(cfg.fid[MG_F_VIBE], ac.vibe);
      break;
    }
    case MG_CELL_CHEST: {  // chest.hpp:124-146
      const ChestInst& ch = chests[(size_t)o.idx];
      const MgChest& cc = cfg.chests[(size_t)ch.cfg];
      if (cc.vibe != 0) push(cfg.fid[MG_F_VIBE], cc.vibe);
      for (uint8_t item : ch.inv.items.order) append_inventory_tokens(f, item, ch.inv.items.get(item));
      for (int t : cc.tags) push(cfg.fid[MG_F_TAG], t);
      break;
    }
  }
}

// append_tokens_if_room_available (observation_encoder.hpp:107-117): write min(room, n), report n.
size_t Env::append(std::vector<uint8_t>& feats, uint8_t* dst, size_t room, uint8_t loc) {
  size_t n = feats.size() / 2;
  size_t nw = std::min(room, n);
  for (size_t i = 0; i < nw; i++) {
    dst[i * 3 + 0] = loc;
    dst[i * 3 + 1] = feats[i * 2];
    dst[i * 3 + 2] = feats[i * 2 + 1];
  }
  return n;
}

void Env::compute_observation(size_t ai, int action) {
  // mettagrid_c.cpp:397-555
  const Agent& a = agents[ai];
  const MgGroup& g = cfg.groups[(size_t)a.group];
  size_t T = (size_t)cfg.T;
  uint8_t* row = &obs[ai * T * 3];
  int hr = cfg.obs_h >> 1, wr = cfg.obs_w >> 1;
  int r_start = std::max(a.r - hr, 0), c_start = std::max(a.c - wr, 0);
  int r_end = std::min(a.r + hr + 1, H), c_end = std::min(a.c + wr + 1, W);
  size_t attempted = 0, written = 0;
  std::vector<uint8_t> f;
  auto push = [&](int feat, int val) {
    f.push_back((uint8_t)feat);
    f.push_back((uint8_t)val);
  };
  if (cfg.g_completion) {
    uint8_t pct = 0;
    if (cfg.max_steps > 0) {
      if (current_step >= (uint32_t)cfg.max_steps)
        pct = 255;
      else
        pct = (uint8_t)(256u * current_step / (uint32_t)cfg.max_steps);
    }
    push(cfg.fid[MG_F_COMPLETION], pct);
  }
  if (cfg.g_last_action) push(cfg.fid[MG_F_LAST_ACTION], (uint8_t)action);
  if (cfg.g_last_reward) {
    // rewards are zeroed before, and filled after, the observation pass => always round(0*100) = 0 in a step;
    // kept as the reference computes it.
    uint8_t ri = (uint8_t)std::round(rewards[ai] * 100.0f);
    push(cfg.fid[MG_F_LAST_REWARD], ri);
  }
  if (cfg.g_goal)
    for (int v : g.goal_values) push(cfg.fid[MG_F_GOAL], v);
  uint8_t center = (uint8_t)((hr << 4) | (wr & 0x0F));
  attempted += append(f, row + written * 3, T - written, center);
  written = std::min(attempted, T);
  if (cfg.g_compass) {
    int dr = H / 2 - a.r, dc = W / 2 - a.c;
    int sr = dr != 0 ? (dr > 0 ? 1 : -1) : 0, sc = dc != 0 ? (dc > 0 ? 1 : -1) : 0;
    if (sr != 0 || sc != 0) {
      int orr = hr + sr, occ = wr + sc;
      if (orr >= 0 && orr < cfg.obs_h && occ >= 0 && occ < cfg.obs_w) {
        f.clear();
        push(cfg.fid[MG_F_COMPASS], 1);
        attempted += append(f, row + written * 3, T - written, (uint8_t)((orr << 4) | (occ & 0x0F)));
        written = std::min(attempted, T);
      }
    }
  }
  for (auto& off : pattern) {
    int r = a.r + off.first, c = a.c + off.second;
    if (r < r_start || r >= r_end || c < c_start || c >= c_end) continue;
    int id = cell(r, c);
    if (!id) continue;
    f.clear();
    object_features(id, f);
    int orr = r - a.r + hr, occ = c - a.c + wr;
    attempted += append(f, row + written * 3, T - written, (uint8_t)((orr << 4) | (occ & 0x0F)));
    written = std::min(attempted, T);
  }
  game.add(GS.tokens_written(), (float)written);
  game.add(GS.tokens_dropped(), (float)(attempted - written));
  game.add(GS.tokens_free_space(), (float)(T - written));
}

void Env::compute_observations(const std::vector<int32_t>& executed) {
  for (size_t i = 0; i < agents.size(); i++) compute_observation(i, executed[i]);
}

bool Env::try_attack(Agent& actor, int target_obj) {
  // attack.hpp:93-122, 139-253
  if (!target_obj) return false;
  if (objects[(size_t)target_obj].kind != MG_CELL_AGENT) return false;
  Agent& target = agents[(size_t)objects[(size_t)target_obj].idx];
  if (target.frozen > 0) return false;
  for (const auto& kv : cfg.atk_consumed)
    if ((int)actor.inv.amount(kv.k) < (int)(uint16_t)kv.v) return false;
  int ag = cfg.groups[(size_t)actor.group].group_name_idx, tg = cfg.groups[(size_t)target.group].group_name_idx;
  auto weapon_power = [&]() {
    int p = 0;
    for (const auto& kv : cfg.atk_weapon) p += (int)actor.inv.amount(kv.k) * (int)(uint16_t)kv.v;
    return p;
  };
  auto armor_power = [&]() {
    int p = 0;
    for (const auto& kv : cfg.atk_armor) {
      int amount = target.inv.amount(kv.k);
      bool vibing = target.vibe != 0 && (int)target.vibe < cfg.n_vibes && kv.k < cfg.R &&
                    cfg.vibe_resource[target.vibe] == kv.k;
      if (vibing) amount += cfg.atk_vibe_bonus[target.vibe];
      p += amount * (int)(uint16_t)kv.v;
    }
    return p;
  };
  bool blocked = false;
  if (!cfg.atk_defense.empty()) {
    int bonus = std::max(weapon_power() - armor_power(), 0);
    bool can = true;
    for (const auto& kv : cfg.atk_defense) {
      int required = (int)(uint16_t)kv.v + bonus;
      if (target.inv.amount(kv.k) < (uint16_t)required) {
        can = false;
        break;
      }
    }
    if (can) {
      int bonus2 = std::max(weapon_power() - armor_power(), 0);
      for (const auto& kv : cfg.atk_defense) target.inv.update(kv.k, -((int)(uint16_t)kv.v + bonus2));
      actor.stats.add(AS.attack_blocked_by(tg), 1.f);
      blocked = true;
    }
  }
  if (!blocked) {
    if (cfg.atk_freeze > 0) target.frozen = (int16_t)cfg.atk_freeze;
    for (const auto& kv : cfg.atk_actor_delta) actor.inv.update(kv.k, kv.v);
    for (const auto& kv : cfg.atk_target_delta) target.inv.update(kv.k, kv.v);
    for (int item : cfg.atk_loot) {
      int amount = target.inv.amount(item);
      if (amount > 0) {
        int stolen = actor.inv.update(item, amount);
        target.inv.update(item, -stolen);
      }
    }
    if (ag == tg) {
      actor.stats.add(AS.attack_friendly_fire(), 1.f);
    } else {
      actor.stats.add(AS.attack_hit(tg), 1.f);
      target.stats.add(AS.attack_hit_by(ag), 1.f);
    }
  }
  for (const auto& kv : cfg.atk_consumed)
    if (kv.v > 0) actor.inv.update(kv.k, -(int)(uint16_t)kv.v);
  return true;
}

bool Env::try_transfer(Agent& actor, int target_obj) {
  // transfer.hpp:72-150
  for (const auto& kv : cfg.xfer_required)
    if ((int)actor.inv.amount(kv.k) < (int)(uint16_t)kv.v) return false;
  if (!target_obj) return false;
  if (objects[(size_t)target_obj].kind != MG_CELL_AGENT) return false;
  Agent& target = agents[(size_t)objects[(size_t)target_obj].idx];
  if (target.frozen > 0) return false;
  const MgTransferEffect* eff = nullptr;
  for (const auto& e : cfg.xfer_effects)
    if (e.vibe == actor.vibe) eff = &e;
  if (!eff) return false;
  int tg = cfg.groups[(size_t)target.group].group_name_idx;
  for (const auto& kv : eff->actor)
    if (kv.v < 0 && (int)actor.inv.amount(kv.k) < -kv.v) return false;
  for (const auto& kv : eff->target)
    if (kv.v < 0 && (int)target.inv.amount(kv.k) < -kv.v) return false;
  for (const auto& kv : eff->actor)
    if (kv.v > 0 && kv.v > (int)actor.inv.free_space(kv.k)) return false;
  for (const auto& kv : eff->target)
    if (kv.v > 0 && kv.v > (int)target.inv.free_space(kv.k)) return false;
  for (const auto& kv : eff->actor)
    if (kv.v != 0) {
      int actual = actor.inv.update(kv.k, kv.v);
      if (actual != 0) actor.stats.add(AS.transfer_self(kv.k), (float)actual);
    }
  for (const auto& kv : eff->target)
    if (kv.v != 0) {
      int actual = target.inv.update(kv.k, kv.v);
      if (actual != 0) actor.stats.add(AS.transfer_to(tg, kv.k), (float)actual);
    }
  actor.stats.add(AS.transfer_count(), 1.f);
  return true;
}

bool Env::assembler_use(AssemblerInst& as, Agent& actor) {
  // Assembler::onUse (assembler.hpp:455-526)
  const MgAssembler& ac = cfg.assemblers[(size_t)as.cfg];
  if (ac.max_uses > 0 && as.uses_count >= (uint32_t)ac.max_uses) return false;
  uint32_t remaining = as.cooldown_end <= current_step ? 0 : as.cooldown_end - current_step;
  if (remaining > 0 && !ac.allow_partial_usage) return false;
  const MgProtocol* orig = current_protocol(as);
  if (!orig) return false;
  // protocol_to_use: (item, amount) lists in the iteration order of the copy actually used
  MgKVList inputs = orig->inputs, outputs = orig->outputs;
  auto positive_output = [](const MgKVList& o) {
    for (const auto& kv : o)
      if (kv.v > 0) return true;
    return false;
  };
  if (remaining > 0 && ac.allow_partial_usage) {
    uint32_t duration = as.cooldown_duration, elapsed = duration - remaining;
    auto amount_of = [](const MgKVList& l, int k) {
      for (const auto& kv : l)
        if (kv.k == k) return kv.v;
      return 0;
    };
    inputs.clear();
    outputs.clear();
    for (int k : orig->inputs_scaled)
      inputs.push_back(MgKV{k, (int32_t)(uint16_t)(((uint32_t)amount_of(orig->inputs, k) * elapsed + duration - 1) / duration)});
    for (int k : orig->outputs_scaled)
      outputs.push_back(MgKV{k, (int32_t)(uint16_t)(((uint32_t)amount_of(orig->outputs, k) * elapsed) / duration)});
    if (!positive_output(outputs) && positive_output(orig->outputs) && !as.is_clipped) return false;
  }
  // get_surrounding_agents(&actor): positions rotated to start at the actor (assembler.hpp:48-92)
  std::vector<std::pair<int, int>> pos;
  surrounding_positions(as, pos);
  int start = -1;
  for (size_t i = 0; i < pos.size(); i++)
    if (pos[i].first == actor.r && pos[i].second == actor.c) {
      start = (int)i;
      break;
    }
  if (start < 0) throw std::runtime_error("Starting agent is not in a surrounding position of the assembler");
  std::rotate(pos.begin(), pos.begin() + start, pos.end());
  std::vector<Agent*> near;
  for (auto& p : pos) {
    int id = cell(p.first, p.second);
    if (id && objects[(size_t)id].kind == MG_CELL_AGENT) near.push_back(&agents[(size_t)objects[(size_t)id].idx]);
  }
  std::vector<Inventory*> input_invs;
  for (Agent* a : near) input_invs.push_back(&a->inv);
  if (ac.chest_search_distance > 0) {
    int d = ac.chest_search_distance;
    for (int dr = -d; dr <= d; dr++)
      for (int dc = -d; dc <= d; dc++) {
        if (dr == 0 && dc == 0) continue;
        int rr = (uint16_t)(as.r + dr), cc = (uint16_t)(as.c + dc);
        if (rr < H && cc < W) {
          int id = cell(rr, cc);
          if (id && objects[(size_t)id].kind == MG_CELL_CHEST) input_invs.push_back(&chests[(size_t)objects[(size_t)id].idx].inv);
        }
      }
  }
  // can_afford_protocol: totals accumulate in uint16 (assembler.hpp:124-137)
  {
    uint16_t total[256];
    std::memset(total, 0, sizeof(total));
    for (Inventory* inv : input_invs)
      for (uint8_t item : inv->items.order) total[item] = (uint16_t)(total[item] + inv->items.get(item));
    for (const auto& kv : inputs)
      if (total[kv.k] < (uint16_t)kv.v) return false;
  }
  // get_output_inventories (assembler.hpp:198-223)
  std::vector<Inventory*> output_invs;
  if (orig->vibes.size() <= 1) {
    output_invs.push_back(&actor.inv);
  } else {
    int required_counts[256];
    std::memset(required_counts, 0, sizeof(required_counts));
    bool present[256];
    std::memset(present, 0, sizeof(present));
    for (int v : orig->vibes) {
      required_counts[v]++;
      present[v] = true;
    }
    size_t required = orig->vibes.size();
    for (Agent* a : near) {
      if (a->vibe == 0) continue;
      if (!present[a->vibe] || required_counts[a->vibe] <= 0) continue;
      output_invs.push_back(&a->inv);
      required_counts[a->vibe]--;
      if (output_invs.size() >= required) break;
    }
    if (output_invs.empty()) output_invs.push_back(&actor.inv);
  }
  // can_receive_output (assembler.hpp:143-171): free space sums in uint16
  {
    bool ok = true;
    if (positive_output(outputs)) {
      ok = false;
      if (!output_invs.empty())
        for (const auto& kv : outputs)
          if (kv.v > 0) {
            uint16_t total_free = 0;
            for (Inventory* inv : output_invs) total_free = (uint16_t)(total_free + inv->free_space(kv.k));
            if (total_free >= 1) {
              ok = true;
              break;
            }
          }
    }
    if (!ok && !as.is_clipped) return false;
  }
  for (const auto& kv : inputs) shared_update(input_invs, kv.k, -kv.v);
  for (const auto& kv : outputs) {
    int distributed = shared_update(output_invs, kv.k, kv.v);
    if (distributed > 0) game.add(GS.assembler_created(kv.k), (float)distributed);
  }
  as.cooldown_duration = (uint32_t)orig->cooldown;
  as.cooldown_end = current_step + as.cooldown_duration;
  if (as.is_clipped) {  // become_unclipped (assembler.hpp:583-592)
    as.is_clipped = false;
    as.unclip = -1;
    if (as.by_clipper) on_unclip_assembler((size_t)(&as - assemblers.data()));
    as.by_clipper = false;
  } else {
    as.uses_count++;
  }
  return true;
}

// ---- Clipper (systems/clipper.hpp) ------------------------------------------------------------------------------
void Env::clipper_init(uint32_t seed) {
  ClipperInst& cl = clipper;
  cl.on = true;
  cl.length_scale = cfg.clip_length_scale;
  cl.cutoff = (uint32_t)cfg.clip_cutoff;
  cl.period = (uint32_t)cfg.clip_period;
  cl.rng.seed(seed);  // a copy of _rng, which nothing has drawn from yet (mettagrid_c.cpp:57,153)
  cl.weight.assign(assemblers.size(), 0);
  cl.member.assign(assemblers.size(), 0);
  size_t n_members = 0;
  std::vector<size_t> starting;
  for (size_t i = 0; i < assemblers.size(); i++) {  // object id order = creation order
    const MgAssembler& ac = cfg.assemblers[(size_t)assemblers[i].cfg];
    if (ac.clip_immune) continue;
    cl.member[i] = 1;
    n_members++;
    if (ac.start_clipped) starting.push_back(i);
  }
  if (cl.length_scale <= 0 && n_members > 0) cl.length_scale = clipper_auto_length_scale(W, H, n_members);
  for (size_t i : starting) clip_assembler(i);
}

bool Env::clipper_adjacent(size_t i, size_t j, uint32_t* w) const {
  const ClipperInst& cl = clipper;
  if (cl.period == 0) return false;  // compute_adjacencies only runs for an active clipper (clipper.hpp:101-105)
  if (i == j || !cl.member[i] || !cl.member[j]) return false;
  const AssemblerInst &a = assemblers[i], &b = assemblers[j];
  const int dist = std::max(std::abs(a.r - b.r), std::abs(a.c - b.c));
  if ((uint32_t)dist > cl.cutoff * (uint32_t)cl.length_scale) return false;
  *w = clipper_infection_weight(dist, cl.length_scale, cl.cutoff);
  return true;
}

void Env::clip_assembler(size_t i) {  // clipper.hpp:174-194
  ClipperInst& cl = clipper;
  for (size_t j = 0; j < assemblers.size(); j++) {
    uint32_t w;
    if (clipper_adjacent(i, j, &w)) cl.weight[j] += w;
  }
  const uint32_t k = cl.rng.bounded((uint32_t)cfg.unclip.size());  // uniform_int_distribution<size_t>(0, n - 1)
  AssemblerInst& as = assemblers[i];
  as.is_clipped = true;  // become_clipped (assembler.hpp:411-421)
  as.unclip = (int)k;
  as.by_clipper = true;
  as.cooldown_end = current_step;
  as.cooldown_duration = 0;
}

void Env::on_unclip_assembler(size_t i) {  // clipper.hpp:196-205
  for (size_t j = 0; j < assemblers.size(); j++) {
    uint32_t w;
    if (clipper_adjacent(i, j, &w)) clipper.weight[j] -= w;
  }
}

int Env::pick_assembler_to_clip() {  // clipper.hpp:207-229
  ClipperInst& cl = clipper;
  // unclipped_assemblers: every member the clipper has not clipped (one that starts clipped is clipped by the
  // constructor, so: every member that is not clipped now)
  std::vector<size_t> unclipped;
  for (size_t i = 0; i < assemblers.size(); i++)
    if (cl.member[i] && !assemblers[i].is_clipped) unclipped.push_back(i);
  if (unclipped.empty()) return -1;
  uint32_t total = 0;
  for (size_t i : unclipped) total += cl.weight[i];
  if (total == 0) return (int)unclipped[cl.rng.bounded((uint32_t)unclipped.size())];
  uint32_t random_weight = 1 + cl.rng.bounded(total);  // uniform_int_distribution<>(1, total)
  for (size_t i : unclipped) {
    if (cl.weight[i] == 0) continue;
    if (random_weight <= cl.weight[i]) return (int)i;
    random_weight -= cl.weight[i];
  }
  throw std::runtime_error("Failed to pick an assembler to clip");
}

void Env::maybe_clip_new_assembler() {  // clipper.hpp:231-238
  ClipperInst& cl = clipper;
  bool hit;
  if (cl.period > 0) {
    hit = cl.rng.bounded(cl.period) == 0;  // uniform_int_distribution<uint32_t>(1, period)(rng) == 1
  } else {
    // (1, 0): libstdc++ computes the range in 64 bits, b - a = 2^64 - 1 > the engine's range, and composes two draws
    // (uniform_int_dist.h:283-300): ret = 2^32 x1 + x2, truncated to 32 bits and + 1  ==  1  iff  x2 == 0
    cl.rng.next();
    hit = cl.rng.next() == 0;
  }
  if (!hit) return;
  int i = pick_assembler_to_clip();
  if (i >= 0) clip_assembler((size_t)i);
}

bool Env::chest_use(ChestInst& ch, Agent& actor) {
  // Chest::onUse / transfer_resources (chest.hpp:31-54,103-122)
  const MgChest& cc = cfg.chests[(size_t)ch.cfg];
  if (cc.vibe_transfers.empty()) return false;
  const MgKVList* deltas = nullptr;
  for (const auto& vt : cc.vibe_transfers)
    if (vt.first == actor.vibe) deltas = &vt.second;
  if (!deltas) return false;
  bool any = false;
  for (const auto& kv : *deltas) {
    if (kv.v > 0) {
      int t = transfer_resources(actor.inv, ch.inv, kv.k, kv.v, true);
      if (t > 0) {
        any = true;
        actor.stats.add(AS.chest_deposited(kv.k), (float)t);
      }
    } else if (kv.v < 0) {
      int t = transfer_resources(ch.inv, actor.inv, kv.k, -kv.v, true);
      if (t > 0) {
        any = true;
        actor.stats.add(AS.chest_withdrawn(kv.k), (float)t);
      }
    }
  }
  return any;
}

bool Env::do_move(Agent& actor, int orientation) {
  // Move::_handle_action (move.hpp:74-141); deltas orientation.hpp:26-48
  static const int DX[8] = {0, 0, -1, 1, -1, 1, -1, 1};
  static const int DY[8] = {-1, 1, 0, 0, -1, -1, 1, 1};
  int tr = (uint16_t)(actor.r + DY[orientation]), tc = (uint16_t)(actor.c + DX[orientation]);
  if (!(tr < H && tc < W)) return false;
  int target = cell(tr, tc);
  int vh = cfg.vibe_handler[actor.vibe];
  if (vh == MG_VH_ATTACK) {
    if (try_attack(actor, target)) return true;
  } else if (vh == MG_VH_TRANSFER) {
    if (try_transfer(actor, target)) return true;
  }
  if (target == 0) {
    cell(tr, tc) = (int32_t)actor.obj_id;
    cell(actor.r, actor.c) = 0;
    actor.r = tr;
    actor.c = tc;
    objects[actor.obj_id].r = tr;
    objects[actor.obj_id].c = tc;
    return true;
  }
  Obj& to = objects[(size_t)target];
  if (to.kind == MG_CELL_AGENT) {
    Agent& ta = agents[(size_t)to.idx];
    if (ta.frozen > 0) {
      cell(actor.r, actor.c) = (int32_t)ta.obj_id;
      cell(ta.r, ta.c) = (int32_t)actor.obj_id;
      std::swap(actor.r, ta.r);
      std::swap(actor.c, ta.c);
      objects[actor.obj_id].r = actor.r;
      objects[actor.obj_id].c = actor.c;
      objects[ta.obj_id].r = ta.r;
      objects[ta.obj_id].c = ta.c;
      actor.stats.add(AS.swap(), 1.f);
      return true;
    }
    return false;  // Agent::onUse (agent.cpp:179-185)
  }
  if (to.kind == MG_CELL_ASSEMBLER) return assembler_use(assemblers[(size_t)to.idx], actor);
  if (to.kind == MG_CELL_CHEST) return chest_use(chests[(size_t)to.idx], actor);
  return false;  // walls are not Usable
}

bool Env::handle_action(Agent& actor, const MgAction& act) {
  // ActionHandler::handle_action (action_handler.hpp:99-153)
  if (actor.frozen != 0) {
    actor.stats.add(AS.frozen_ticks(), 1.f);
    actor.stats.add(AS.frozen_ticks_group(), 1.f);
    if (actor.frozen > 0) actor.frozen -= 1;
    return false;
  }
  const MgHandlerRes& hr = cfg.handler[act.kind];
  bool has = true;
  for (const auto& kv : hr.required)
    if ((int)actor.inv.amount(kv.k) < (int)(uint16_t)kv.v) {
      has = false;
      break;
    }
  bool success = false;
  if (has) {
    switch (act.kind) {
      case MG_ACT_NOOP: success = true; break;
      case MG_ACT_MOVE: success = do_move(actor, act.arg); break;
      case MG_ACT_CHANGE_VIBE:
        actor.vibe = (uint8_t)act.arg;
        success = true;
        break;
    }
  }
  if (actor.r == actor.prev_r && actor.c == actor.prev_c) {
    actor.steps_without_motion += 1;
    if ((float)actor.steps_without_motion > actor.stats.get(AS.max_steps_without_motion()))
      actor.stats.set(AS.max_steps_without_motion(), (float)actor.steps_without_motion);
  } else {
    actor.steps_without_motion = 0;
  }
  actor.prev_r = actor.r;
  actor.prev_c = actor.c;
  if (success) {
    actor.stats.add(AS.success(act.kind), 1.f);
    for (const auto& kv : hr.consumed)
      if (kv.v > 0) actor.inv.update(kv.k, -(int)(uint16_t)kv.v);
  } else {
    actor.stats.add(AS.failed(act.kind), 1.f);
    actor.stats.add(AS.failed_any(), 1.f);
  }
  return success;
}

void Env::check_and_apply_damage(Agent& a) {
  // Agent::check_and_apply_damage (agent.cpp:134-177)
  const MgGroup& g = cfg.groups[(size_t)a.group];
  if (g.damage_threshold.empty() || g.damage_resources.empty()) return;
  for (const auto& kv : g.damage_threshold)
    if (a.inv.amount(kv.k) < (uint16_t)kv.v) return;
  for (const auto& kv : g.damage_threshold) a.inv.update(kv.k, -kv.v);
  std::vector<int> avail;
  std::vector<double> weights;
  for (const auto& kv : g.damage_resources) {
    int removable = (int)a.inv.amount(kv.k) - kv.v;
    if (removable > 0) {
      avail.push_back(kv.k);
      weights.push_back((double)removable);
    }
  }
  if (!avail.empty()) {
    size_t sel = 0;
    if (weights.size() >= 2) {  // std::discrete_distribution (random.tcc:2656-2713)
      double sum = 0.0;
      for (double x : weights) sum += x;
      std::vector<double> cp;
      double acc = 0.0;
      for (double x : weights) {
        acc += x / sum;
        cp.push_back(acc);
      }
      cp.back() = 1.0;
      double p = rng.canonical();
      sel = (size_t)(std::lower_bound(cp.begin(), cp.end(), p) - cp.begin());
    }
    int item = avail[sel];
    a.inv.update(item, -1);
    a.stats.add(AS.damage_items_lost(), 1.f);
    a.stats.add(AS.damaged(item), 1.f);
  }
  a.stats.add(AS.damage_triggered(), 1.f);
}

void Env::compute_stat_rewards(size_t ai) {
  // Agent::compute_stat_rewards (agent.cpp:103-132)
  Agent& a = agents[ai];
  const MgGroup& g = cfg.groups[(size_t)a.group];
  if (g.stat_rewards.empty()) return;
  float total = 0.f;
  for (const auto& sr : g.stat_rewards) {
    float v = a.stats.get(sr.agent_slot);
    v += game.get(sr.game_slot);
    if (a.collective >= 0) {  // agent.cpp:115-117
      const CollectiveInst& c2 = collectives[(size_t)a.collective];
      float cv = 0.f;
      if (sr.coll_kind == 1) cv = c2.stat_aligned_agent;
      else if (sr.coll_kind == 2) cv = c2.stat_deposited[(size_t)sr.coll_res];
      v += cv;
    }
    float rew = v * sr.weight;
    if (sr.has_max) rew = std::min(rew, sr.max);
    total += rew;
  }
  float delta = total - a.current_stat_reward;
  if (delta != 0.0f) {
    rewards[ai] += delta;
    a.current_stat_reward = total;
  }
}

void Env::step(const int32_t* actions) {
  // MettaGrid::_step (mettagrid_c.cpp:572-678)
  size_t A = agents.size();
  std::fill(rewards.begin(), rewards.end(), 0.f);
  std::fill(obs.begin(), obs.end(), 0xFF);
  std::fill(action_success.begin(), action_success.end(), 0);
  current_step++;
  std::vector<size_t> order(A);
  for (size_t i = 0; i < A; i++) order[i] = i;
  std_shuffle(order, rng);
  std::vector<int32_t> executed(A, 0);
  for (int off = 0; off <= cfg.max_priority; off++) {
    int prio = cfg.max_priority - off;
    for (size_t ai : order) {
      int32_t act = actions[ai];
      if (act < 0 || (size_t)act >= cfg.actions.size()) {
        Agent& a = agents[ai];
        a.stats.add(AS.invalid(), 1.f);
        int nA = (int)cfg.actions.size();
        if (act == -1 || (act >= nA && act < nA + 7)) {
          a.stats.add(AS.invalid_variant(act == -1 ? 0 : 1 + (act - nA)), 1.f);
        } else {
          // "action.invalid_index.<id>" for any other id: (id, count) pairs at the end of the slot array, the id as
          // raw int32 bits (MG_INVALID_TABLE in mg_blob.h; metta_amd/stats.py names the keys)
          bool placed = false;
          for (int k = 0; k < MG_INVALID_TABLE && !placed; k++) {
            const int si = AS.invalid_table_id(k), sc = AS.invalid_table_count(k);
            int32_t held;
            std::memcpy(&held, &a.stats.v[(size_t)si], 4);
            if (a.stats.v[(size_t)sc] == 0.f) {
              std::memcpy(&a.stats.v[(size_t)si], &act, 4);
              a.stats.add(sc, 1.f);
              placed = true;
            } else if (held == act) {
              a.stats.add(sc, 1.f);
              placed = true;
            }
          }
          if (!placed) a.stats.add(AS.invalid_variant(8), 1.f);
        }
        action_success[ai] = 0;
        continue;
      }
      if (prio != 0) continue;  // every real action handler has priority 0; only Attack (no actions) has 1
      bool ok = handle_action(agents[ai], cfg.actions[(size_t)act]);
      action_success[ai] = ok ? 1 : 0;
      if (ok) executed[ai] = act;
    }
  }
  if (cfg.regen_interval > 0 && current_step % (uint32_t)cfg.regen_interval == 0) {
    for (auto& a : agents) {
      const MgGroup& g = cfg.groups[(size_t)a.group];
      if (g.regen.empty()) continue;
      const MgKVList* lst = nullptr;
      for (const auto& v : g.regen)
        if (v.first == a.vibe) lst = &v.second;
      if (!lst)
        for (const auto& v : g.regen)
          if (v.first == 0) lst = &v.second;
      if (lst)
        for (const auto& kv : *lst) a.inv.update(kv.k, kv.v);
    }
  }
  for (auto& a : agents) check_and_apply_damage(a);
  if (clipper.on) maybe_clip_new_assembler();  // mettagrid_c.cpp:647-650
  compute_observations(executed);
  for (size_t i = 0; i < A; i++) compute_stat_rewards(i);
  for (size_t i = 0; i < A; i++) episode_rewards[i] += rewards[i];
  if (cfg.max_steps > 0 && current_step >= (uint32_t)cfg.max_steps) {
    if (cfg.episode_truncates)
      std::fill(truncations.begin(), truncations.end(), 1);
    else
      std::fill(terminals.begin(), terminals.end(), 1);
  }
}

thread_local std::string g_error;

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// C interface for ctypes (tests only)
// ------------------------------------------------------------------------------------------------------------
extern "C" {

const char* mgo_last_error() { return g_error.c_str(); }

void* mgo_create(const int32_t* blob, size_t n_words, const uint8_t* map, int h, int w, uint32_t seed) {
  try {
    auto e = std::make_unique<Env>();
    e->init(blob, n_words, map, h, w, seed);
    if ((int)e->agents.size() != e->cfg.A) {
      g_error = "map holds " + std::to_string(e->agents.size()) + " agents but config.num_agents = " + std::to_string(e->cfg.A);
      return nullptr;
    }
    return e.release();
  } catch (const std::exception& ex) {
    g_error = ex.what();
    return nullptr;
  }
}
void mgo_destroy(void* h) { delete (Env*)h; }
int mgo_step(void* h, const int32_t* actions) {
  try {
    ((Env*)h)->step(actions);
    return 0;
  } catch (const std::exception& ex) {
    g_error = ex.what();
    return 1;
  }
}
// Step many independent environments `n_steps` times with actions [n_steps][n_envs][A] (cpu_baseline timing).
int mgo_step_many(void** hs, int n_envs, const int32_t* actions, int n_steps) {
  try {
    for (int t = 0; t < n_steps; t++)
      for (int e = 0; e < n_envs; e++) {
        Env* env = (Env*)hs[e];
        size_t A = env->agents.size();
        env->step(actions + ((size_t)t * (size_t)n_envs + (size_t)e) * A);
      }
    return 0;
  } catch (const std::exception& ex) {
    g_error = ex.what();
    return 1;
  }
}
int mgo_num_agents(void* h) { return (int)((Env*)h)->agents.size(); }
uint8_t* mgo_obs(void* h) { return ((Env*)h)->obs.data(); }
float* mgo_rewards(void* h) { return ((Env*)h)->rewards.data(); }
float* mgo_episode_rewards(void* h) { return ((Env*)h)->episode_rewards.data(); }
uint8_t* mgo_terminals(void* h) { return ((Env*)h)->terminals.data(); }
uint8_t* mgo_truncations(void* h) { return ((Env*)h)->truncations.data(); }
uint8_t* mgo_action_success(void* h) { return ((Env*)h)->action_success.data(); }
uint32_t mgo_current_step(void* h) { return ((Env*)h)->current_step; }
uint64_t mgo_rng_draws(void* h) { return ((Env*)h)->rng.draws; }
void mgo_end_episode(void* h) {  // Simulation.end_episode (simulator.py:134-136)
  auto& t = ((Env*)h)->truncations;
  std::fill(t.begin(), t.end(), 1);
}
int mgo_agent_stats(void* h, int agent, float* values, uint8_t* touched) {
  Env* e = (Env*)h;
  const Stats& s = e->agents[(size_t)agent].stats;
  std::memcpy(values, s.v.data(), s.v.size() * 4);
  std::memcpy(touched, s.touched.data(), s.touched.size());
  return (int)s.v.size();
}
int mgo_game_stats(void* h, float* values, uint8_t* touched) {
  Env* e = (Env*)h;
  std::memcpy(values, e->game.v.data(), e->game.v.size() * 4);
  std::memcpy(touched, e->game.touched.data(), e->game.touched.size());
  return (int)e->game.v.size();
}
// Per agent: r, c, prev_r, prev_c, frozen, vibe, steps_without_motion, obj_id, n_items, then R x (item, amount)
// in inventory iteration order (unused tail = -1).  Stride = 9 + 2R.
int mgo_agent_state(void* h, int32_t* out) {
  Env* e = (Env*)h;
  int R = e->cfg.R, stride = 9 + 2 * R;
  for (size_t i = 0; i < e->agents.size(); i++) {
    const Agent& a = e->agents[i];
    int32_t* o = out + i * (size_t)stride;
    o[0] = a.r; o[1] = a.c; o[2] = a.prev_r; o[3] = a.prev_c; o[4] = a.frozen; o[5] = a.vibe;
    o[6] = (int32_t)a.steps_without_motion; o[7] = (int32_t)a.obj_id; o[8] = (int32_t)a.inv.items.order.size();
    for (int k = 0; k < 2 * R; k++) o[9 + k] = -1;
    for (size_t k = 0; k < a.inv.items.order.size() && (int)k < R; k++) {
      o[9 + 2 * k] = a.inv.items.order[k];
      o[9 + 2 * k + 1] = a.inv.items.get(a.inv.items.order[k]);
    }
  }
  return stride;
}
int mgo_num_assemblers(void* h) { return (int)((Env*)h)->assemblers.size(); }
// Per assembler: r, c, cfg, obj_id, is_clipped, cooldown_end, cooldown_duration, uses_count (stride 8)
void mgo_assembler_state(void* h, int32_t* out) {
  Env* e = (Env*)h;
  for (size_t i = 0; i < e->assemblers.size(); i++) {
    const AssemblerInst& a = e->assemblers[i];
    int32_t* o = out + i * 8;
    o[0] = a.r; o[1] = a.c; o[2] = a.cfg; o[3] = (int32_t)a.obj_id;
    o[4] = a.is_clipped ? (a.unclip >= 0 ? 2 + a.unclip : 1) : 0;  // (the device's encoding of DAsm::clipped)
    o[5] = (int32_t)a.cooldown_end; o[6] = (int32_t)a.cooldown_duration; o[7] = (int32_t)a.uses_count;
  }
}
// Clipper state: infection weight per assembler (0 for clip-immune ones); returns the length scale in use, -1 = no clipper
int mgo_clipper_state(void* h, uint32_t* weights) {
  Env* e = (Env*)h;
  if (!e->clipper.on) return -1;
  for (size_t i = 0; i < e->assemblers.size(); i++) weights[i] = e->clipper.weight[i];
  return e->clipper.length_scale;
}
uint64_t mgo_clipper_rng_draws(void* h) { return ((Env*)h)->clipper.rng.draws; }
// the two pure functions the reference's own known-answer tests pin (tests/test_clipper.cpp:56-92,303-356)
int mgo_clipper_auto_length_scale(int width, int height, int n_members) {
  return clipper_auto_length_scale(width, height, (size_t)n_members);
}
uint32_t mgo_clipper_infection_weight(int dist, int length_scale, uint32_t cutoff) {
  return clipper_infection_weight(dist, length_scale, cutoff);
}
int mgo_num_chests(void* h) { return (int)((Env*)h)->chests.size(); }
// Per chest: r, c, cfg, obj_id, n_items, (item, amount) pairs in map order, -1 padded (stride 5 + 2R)
void mgo_chest_state(void* h, int32_t* out) {
  Env* e = (Env*)h;
  int R = e->cfg.R, stride = 5 + 2 * R;
  for (size_t i = 0; i < e->chests.size(); i++) {
    const ChestInst& c = e->chests[i];
    int32_t* o = out + i * (size_t)stride;
    o[0] = c.r; o[1] = c.c; o[2] = c.cfg; o[3] = (int32_t)c.obj_id; o[4] = (int32_t)c.inv.items.order.size();
    for (int k = 0; k < 2 * R; k++) o[5 + k] = -1;
    for (size_t k = 0; k < c.inv.items.order.size() && (int)k < R; k++) {
      o[5 + 2 * k] = c.inv.items.order[k];
      o[5 + 2 * k + 1] = c.inv.items.get(c.inv.items.order[k]);
    }
  }
}
// Object-id grid [H][W] (0 = empty), ids as the reference assigns them (row-major creation order).
void mgo_grid(void* h, int32_t* out) {
  Env* e = (Env*)h;
  std::memcpy(out, e->grid.data(), e->grid.size() * 4);
}
// Agent::set_inventory (agent.cpp:54-72); `items`/`amounts` in the iteration order of the converted map.
void mgo_set_inventory(void* h, int agent, int n, const int32_t* items, const int32_t* amounts) {
  Env* e = (Env*)h;
  if (agent < 0 || (size_t)agent >= e->agents.size()) return;
  Agent& a = e->agents[(size_t)agent];
  std::vector<uint8_t> existing = a.inv.items.order;
  for (uint8_t item : existing) {
    int cur = a.inv.amount(item);
    a.inv.update(item, -cur);
    a.stats.set(e->AS.amount(item), 0.f);
  }
  for (int i = 0; i < n; i++) a.inv.update(items[i], amounts[i] - (int)a.inv.amount(items[i]));
}
int mgo_test_pattern(int h, int w, int32_t* out) {
  auto p = observation_pattern(h, w);
  for (size_t i = 0; i < p.size(); i++) {
    out[2 * i] = p[i].first;
    out[2 * i + 1] = p[i].second;
  }
  return (int)p.size();
}
// Known-answer hooks for the libstdc++ restatements.
void mgo_test_shuffle(uint32_t seed, int n, int rounds, int32_t* out, uint32_t* next_raw) {
  MT19937 g;
  g.seed(seed);
  for (int k = 0; k < rounds; k++) {
    std::vector<int32_t> v((size_t)n);
    for (int i = 0; i < n; i++) v[(size_t)i] = i;
    std_shuffle(v, g);
    std::memcpy(out + (size_t)k * (size_t)n, v.data(), (size_t)n * 4);
  }
  *next_raw = g.next();
}
int mgo_test_shared_update(int n_inv, const int32_t* amounts, const int32_t* limits, int delta, int32_t* out_amounts) {
  // n_inv single-resource inventories with the given starting amounts and limits (test_mettagrid.cpp:1408-1650)
  std::vector<MgInventoryCfg> cfgs((size_t)n_inv);
  std::vector<Inventory> invs((size_t)n_inv);
  std::vector<Inventory*> ptrs;
  for (int i = 0; i < n_inv; i++) {
    MgLimit l;
    l.base = limits[i];
    l.resources = {0};
    cfgs[(size_t)i].limits.push_back(l);
    cfgs[(size_t)i].limits_map_order.push_back(MgKV{0, 0});
    invs[(size_t)i].init(&cfgs[(size_t)i], 1, nullptr, 0, -1);
    invs[(size_t)i].update(0, amounts[i], true);
    ptrs.push_back(&invs[(size_t)i]);
  }
  int consumed = shared_update(ptrs, 0, delta);
  for (int i = 0; i < n_inv; i++) out_amounts[i] = invs[(size_t)i].amount(0);
  return consumed;
}
}
