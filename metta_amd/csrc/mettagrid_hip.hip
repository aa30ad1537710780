// mettagrid_hip.hip -- kernels + C-ABI of the MI355X-native batched MettaGrid step path (gfx950).
//
// Replaces, for the environment-step path, the reference's pybind11 core `MettaGrid`
// (packages/mettagrid/cpp/bindings/mettagrid_c.cpp).  Layout and kernel design: DESIGN.md.
//
// This file is the host side (handle, map pool upload, launches, pool services, inspection).  The kernels live in their
// own translation units (shared device code: mg_kernels.h, mg_sim.h):
//   k_fused.hip        mg_step_fused_kernel  the whole step of one environment in one 2-wave workgroup (attack-free
//   k_fused_nomod.hip                        configs, A <= 64: arena, navigation); the second file is the same source
//   k_fused_nomod_o7.hip                     built without inventory-limit modifiers, the third that for 7 waves per SIMD;
//   k_fused_o6.hip, k_fused_nomod_w1.hip     with modifiers for 6 waves per SIMD; one wave per environment (<= 8 agents)
//   k_act_fast.hip     mg_act_fast_kernel    three-launch variant of the same class (MG_NO_FUSED=1): wave-parallel actions,
//   k_uses.hip         mg_uses_kernel        the assembler uses handed over by it,
//   k_obs.hip          mg_obs_kernel         token lists, observation rows, rewards, flags (also the general path's second launch)
//   k_act_*.hip        mg_act_kernel         general action kernel: rank-ordered parallel rounds (attack / transfer / chests /
//                                            damage / regen, any A); BIG = grids and records in HBM
//   k_reset.hip        mg_reset_kernel       MettaGrid construction + step-0 observations; builds the episode-start templates
//   k_mapgen.hip       mg_generate_maps_kernel  device map pool (border + uniform shuffle per room) and its refresh
//   k_clipper.hip      mg_clipper_kernel        the Clipper's turn between the action and the observation kernel
//   advantage.hip      mg_advantage_kernel   V-trace / GAE row scan (mg_compute_advantage)
//
// There is no CPU fallback: every entry point runs HIP kernels or fails.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mettagrid_hip.h"
#include "mg_blob.h"
#include "mg_kernels.h"
#include "mg_mapgen.h"

__global__ void mg_end_episode_kernel(KArgs a, int env) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= a.E * a.A) return;
  int e = g / a.A;
  if (env >= 0 && e != env) return;
  a.trunc[g] = 1;
  if (g % a.A == 0) a.env[e].done = 1;
}

// Agent::set_inventory (agent.cpp:54-72) for one agent, on HBM state (single thread; inspection/test API).
__global__ void mg_set_inventory_kernel(KArgs a, int e, int agent, int n, const int* items, const int* amounts) {
  extern __shared__ __align__(16) unsigned char smem[];
#ifdef MG_ISA_LITE  // ISA inspection builds (tools/isa.sh): keep only the kernel under study
  return;
#endif
  if (threadIdx.x != 0) return;
  mg_use_map(a, a.env[e].map_id);  // (per-slot configurations: the environment's own stream)
  DAgent* ag = (DAgent*)smem;
  uint32_t* dirty = (uint32_t*)(smem + mg_align16(sizeof(DAgent)));
  *ag = a.agents[(size_t)e * a.A + agent];
  dirty[0] = dirty[1] = 0;
  Ctx cx;
  cx.dc = a.dc; cx.grid = nullptr; cx.ag = ag; cx.as = nullptr; cx.asm_info = nullptr; cx.gd = nullptr;
  cx.astats = a.astats + ((size_t)e * a.A + agent) * a.SA;  // record 0 of this context is `agent`
  cx.atouched = a.atouched + ((size_t)e * a.A + agent) * a.TW;
  cx.TW = a.TW;
  cx.tmp = nullptr;
  cx.dirty_amt = &dirty[0]; cx.dirty_misc = &dirty[1];
  cx.H = a.H; cx.W = a.W; cx.A = 1; cx.NA = 0; cx.R = a.R; cx.SA = a.SA; cx.SG = a.SG; cx.NG = a.NG; cx.step = 0;
  cx.asm_dirty = 0; cx.AS = MgAgentSlots{a.R, a.NG}; cx.GS = MgGameSlots{a.R, a.n_cells - 1}; cx.cc = CellCodec{a.NW, a.A, a.NAmax};
  cx.ch = nullptr; cx.chest_info = nullptr; cx.NC = 0; cx.gsetv = nullptr; cx.gsetm = nullptr; cx.overflow = 0;
  cx.gset64 = nullptr; cx.seq = 0;
  float* as_ = cx.astats;
  uint32_t* at_ = cx.atouched;
  uint8_t existing[MGD_MAX_R];
  int ne = ag->n_items;
  for (int i = 0; i < ne; i++) existing[i] = ag->order[i];
  uint32_t zeroed = 0;
  for (int i = 0; i < ne; i++) {
    int item = existing[i];
    inv_update(cx, 0, item, -(int)ag->amt[item]);
    zeroed |= 1u << item;  // stats.set(<item>.amount, 0)
  }
  for (int i = 0; i < n; i++) inv_update(cx, 0, items[i], amounts[i] - (int)ag->amt[items[i]]);
  uint32_t m = dirty[0] | zeroed;
  while (m) {
    int r = __ffs(m) - 1;
    m &= m - 1;
    int s = cx.AS.amount(r);
    as_[s] = (float)ag->amt[r];
    atomicOr(&at_[s >> 5], 1u << (s & 31));
  }
  if (dirty[1] & 1u) {
    int s = cx.AS.diversity();
    as_[s] = (float)ag->diversity;
    atomicOr(&at_[s >> 5], 1u << (s & 31));
  }
  a.agents[(size_t)e * a.A + agent] = *ag;
}

// Gather the finished-episode snapshots of the listed environments into one contiguous block (mg_finished_batch):
// per env [A*SA astats | A*TW atouched | SG gstats | GW gtouched | A episode rewards | steps].
__global__ void mg_gather_finished_kernel(KArgs a, const int* envs, uint32_t* out, int row_words) {
  const int e = envs[blockIdx.x];
  uint32_t* dst = out + (size_t)blockIdx.x * row_words;
  const int n0 = a.A * a.SA, n1 = a.A * a.TW, n2 = a.SG, n3 = a.GW, n4 = a.A;
  const uint32_t* s0 = (const uint32_t*)(a.fin_astats + (size_t)e * n0);
  const uint32_t* s1 = a.fin_atouched + (size_t)e * n1;
  const uint32_t* s2 = (const uint32_t*)(a.fin_gstats + (size_t)e * n2);
  const uint32_t* s3 = a.fin_gtouched + (size_t)e * n3;
  const uint32_t* s4 = (const uint32_t*)(a.fin_ep_rewards + (size_t)e * n4);
  for (int i = threadIdx.x; i < n0; i += blockDim.x) dst[i] = s0[i];
  dst += n0;
  for (int i = threadIdx.x; i < n1; i += blockDim.x) dst[i] = s1[i];
  dst += n1;
  for (int i = threadIdx.x; i < n2; i += blockDim.x) dst[i] = s2[i];
  dst += n2;
  for (int i = threadIdx.x; i < n3; i += blockDim.x) dst[i] = s3[i];
  dst += n3;
  for (int i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = s4[i];
  if (threadIdx.x == 0) dst[n4] = a.fin_steps[e];
}

// mg_write_config_slots: stream of slot slots[b] <- registered stream ids[b] (stride words each, a multiple of 4)
__global__ void mg_copy_config_kernel(int* gdc, const int* reg, const int* slots, const int* ids, int stride) {
  const uint4* __restrict__ src = (const uint4*)(reg + (size_t)ids[blockIdx.x] * stride);
  uint4* dst = (uint4*)(gdc + (size_t)slots[blockIdx.x] * stride);
  for (int i = threadIdx.x; i < (stride >> 2); i += blockDim.x) dst[i] = src[i];
}

__global__ void mg_fill_u8(uint8_t* p, size_t n, uint8_t v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(const std::string& m) {
  g_err = m;
  return 1;
}
int mg_fail_msg(const std::string& m) { return fail(m); }  // for the library's other translation units
#define HIPCHK(x)                                                                                      \
  do {                                                                                                 \
    hipError_t _e = (x);                                                                               \
    if (_e != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(_e));                 \
  } while (0)

// env.error / status[0] bits (kernels: mg_act_fast_kernel, mg_act_kernel)
static std::string mg_error_text(uint32_t bits) {
  std::string t;
  if (bits & 1u) t += "RNG window overflow (more MT19937 outputs needed in one step than staged); ";
  if (bits & 4u) t += "frozen agent on the wave-parallel act path (configuration misclassified as attack-free); ";
  if (bits & 8u) t += "more than 24 chests within an assembler's search distance (work list overflow); ";
  if (bits & 16u) t += "device map generator produced a map that does not match the pool (agents / assembler / chest counts); ";
  if (bits & ~29u) t += "unknown error bits " + std::to_string(bits & ~29u) + "; ";
  if (!t.empty()) t.resize(t.size() - 2);
  return t;
}

struct mg_handle {
  MgConfig cfg;
  int device = 0;
  int E = 0;
  KArgs k;
  std::vector<void*> allocs;
  std::vector<void*> map_allocs;
  size_t lds_bytes = 0, lds_act = 0, lds_obs = 0, lds_uses = 0, lds_fused = 0, lds_fused_w1 = 0;
  bool fused_w1 = false;  // the one-wave-per-environment build of the fused kernel (few agents, no modifiers)
  bool fused_w18 = false; // ... its 8-waves-per-SIMD build (32 workgroups per CU)
  FusedLds fl2, fl1;      // LDS layouts of the two-wave builds / the one-wave build (handed over in the kernel arguments)
  bool dc_lds = false;
  bool fused = false;              // fast configs: the whole step in one launch (k_fused.hip)
  bool fused_dcl = false, cells8 = false;
  bool no_modifiers = false;       // no group / chest has inventory-limit modifiers: the leaner build of the fused kernel
  bool slot_has_modifiers = false; // sticky: a task with limit modifiers was written into a configuration slot (slots outlive the pool)
  int* d_map_max_steps = nullptr;  // [P] per-map episode lengths (lives with the pool; allocated on first use)
  bool fused_o7 = false;           // ... the builds with a larger register budget (7 waves per SIMD; 6 with modifiers), see below
  bool big = false;                // map tile too large for LDS: grid (and, for obs/reset, agent records) in HBM
  int n_split = 1;                 // concurrent env groups per step (MG_SPLIT env var / mg_set_split)
  hipStream_t aux[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool maps_set = false;
  bool needs_reset = true;     // a new map pool invalidates every environment's map id: mg_reset(all) comes first
  int profiling = 0;  // 0 off, 1 whole step, 2 observation kernel only
  int prof_stride = 1;         // record every prof_stride-th step (the event pair itself costs ~7 % of a 90 us step)
  unsigned long long prof_tick = 0;
  std::vector<hipEvent_t> ev;  // profiling ring: pairs (start, stop) per step
  size_t ev_count = 0;         // steps recorded since profiling was enabled
  std::vector<std::vector<int4>> host_spawn, host_asm;  // per map (inspection)
  std::vector<int2> host_dims;                          // per map (height, width)
  std::vector<std::vector<int4>> host_chest;            // per map (inspection)
  int* d_list = nullptr;       // scratch for reset lists [3][E]
  int* d_inv = nullptr;        // scratch for set_inventory [2][MGD_MAX_R]
  uint8_t* masks = nullptr;    // all ones (reference: mettagrid_c.cpp:983-995)
  // episode-start template builds (mg_set_maps, mg_refresh_maps): the reset kernel's arguments aimed at the templates
  KArgs kt;
  int* d_iota = nullptr;       // [P] 0 .. P-1
  // device map generator (mg_set_generated_maps / mg_refresh_maps)
  bool generated = false;
  bool gen_ready = false;      // gen holds the pool's writable pointers and the slot bookkeeping (mg_mapgen_setup)
  MapGenArgs gen;
  uint32_t* d_inuse = nullptr;
  bool in_set_generated = false;
  // per-slot configurations (mg_set_config_slots): K streams of dc_stride words; map slot m plays stream map_cfg[m]
  int n_cfg_slots = 0;
  int base_dc_words = 0, base_obs_words = 0;
  std::vector<int32_t> base_dcfg;
  std::vector<int32_t> host_map_cfg;  // [P]
  // registered configurations (mg_register_config): streams kept on the device, copied into slots by one kernel
  int* d_reg = nullptr;               // [reg_cap][dc_stride]
  int reg_cap = 0, reg_n = 0;
  int* d_wlist = nullptr;             // [2][n_cfg_slots] staging of (slots, ids) for mg_write_config_slots
  int* d_map_cfg = nullptr;           // [P] (lives with the pool)
  int* d_tlist = nullptr;             // [P] staging for template rebuild lists (lives with the pool)
  int ts[4] = {0, 0, 0, 0};           // token-list strides TSW, TSA, TSS, TSC the pool was sized for
  int ts_need[4] = {0, 0, 0, 0};      // the largest strides among the tasks written before the pool existed
  // pinned staging ring for the small host -> device uploads of the pool services (task streams, template lists):
  // an upload never waits for the stream, only for the ring slot's own previous upload
  int32_t* ring = nullptr;
  size_t ring_words = 0;
  int ring_next = 0;
  std::vector<hipEvent_t> ring_ev;
  std::vector<char> ring_used;
  uint32_t* d_fin_stage = nullptr;  // [MG_FIN_CHUNK][row] device staging of mg_finished_batch
  int* d_fin_envs = nullptr;        // [MG_FIN_CHUNK]
};

// ---- static scratch of the allocation-free step (mg_step_noalloc) ---------------------------------------------
// The hand-over buffers between the action kernel(s) and the observation kernel of one step -- executed action per
// agent, per-env mode word, pending assembler uses -- as one file-scope device array instead of the handle's
// hipMalloc'ed ones.  Every GPU the library runs on holds its own copy until the process ends; the array's name is
// only a host-side stand-in, the device address is looked up once per device (mg_scratch_init) and kept.
#define MG_NOALLOC_MAX_AGENTS 131072  /* E * A */
#define MG_NOALLOC_MAX_ENVS 16384
#define MG_SCRATCH_PEND_BYTES ((size_t)MG_NOALLOC_MAX_AGENTS * 32)
#define MG_SCRATCH_EXEC_BYTES ((size_t)MG_NOALLOC_MAX_AGENTS * 4)
#define MG_SCRATCH_MODE_BYTES ((size_t)MG_NOALLOC_MAX_ENVS * 4)
#define MG_SCRATCH_BYTES (MG_SCRATCH_PEND_BYTES + MG_SCRATCH_EXEC_BYTES + MG_SCRATCH_MODE_BYTES)  /* 4.56 MiB */
static __device__ __align__(256) unsigned char mg_scratch[MG_SCRATCH_BYTES];
#define MG_MAX_DEVICES 64
static unsigned char* g_scratch_ptr[MG_MAX_DEVICES];  // per device: address of that GPU's copy of mg_scratch
static std::mutex g_scratch_mu;

// Once per device, with that device current (called by mg_create).
static int mg_scratch_init(int device) {
  if (device < 0 || device >= MG_MAX_DEVICES) return fail("mg_create: device id out of range");
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (g_scratch_ptr[device]) return 0;
  void* p = nullptr;
  HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(mg_scratch)));
  if (!p) return fail("mg_create: hipGetSymbolAddress(mg_scratch) returned NULL");
  g_scratch_ptr[device] = (unsigned char*)p;
  return 0;
}

static bool mg_has_damage(const MgConfig& c) {
  for (const auto& g : c.groups)
    if (!g.damage_threshold.empty() && !g.damage_resources.empty()) return true;
  return false;
}

static int mg_check_supported(const MgConfig& c, const char* who) {
  if (c.R > MGD_MAX_R)
    return fail(std::string(who) + ": " + std::to_string(c.R) + " resources, this build of the library holds " + std::to_string(MGD_MAX_R) +
                (MGD_MAX_R < 32 ? " (load the wide-inventory build libmettagrid_hip_r32.so: up to 32)" : " (the device path's ceiling)"));
  if (c.A > 8191) return fail(std::string(who) + ": too many agents per environment");
  for (const auto& a : c.assemblers)
    for (const auto& p : a.protocols)
      if (p.vibes.size() > 8) return fail(std::string(who) + ": protocols with more than 8 vibes are not supported");
  for (const auto& p : c.unclip)
    if (p.vibes.size() > 8) return fail(std::string(who) + ": protocols with more than 8 vibes are not supported");
  if (c.clipper && (c.clip_cutoff < 0 || c.clip_cutoff > 30)) return fail(std::string(who) + ": clipper scaled_cutoff_distance must be 0..30");
  return 0;
}

// fast (wave-parallel) act kernel eligibility: see mg_act_fast_kernel
static bool mg_fast_eligible(const MgConfig& c) {
  bool ok = c.A <= 64 && c.regen_interval == 0 && !mg_has_damage(c) && c.chests.empty() && !c.clipper;
  for (int v : c.vibe_handler) ok = ok && v == 0;
  for (int hnd = 0; hnd < MG_NUM_HANDLERS; hnd++) ok = ok && c.handler[hnd].required.empty() && c.handler[hnd].consumed.empty();
  return ok;
}

static bool mg_has_modifiers(const MgConfig& c) {
  for (const auto& g : c.groups)
    for (const auto& lim : g.inv.limits)
      if (!lim.modifiers.empty()) return true;
  return false;
}

// token-list strides: the most tokens an object of each kind can show (wall.hpp:26-38, agent.cpp:187-213,
// assembler.hpp:528-578, chest.hpp:124-146), rounded to 4 (lists are read 4 tokens = 8 bytes at a time)
static void mg_token_strides(const MgConfig& c, int ts[4]) {
  int TSW = 1, TSA = 1, TSS = 1, TSC = 1;
  for (const auto& wl : c.walls) TSW = std::max(TSW, (int)wl.tags.size() + 1);
  for (const auto& g : c.groups) TSA = std::max(TSA, 3 + c.R * c.n_inv_tokens + (int)g.tags.size());
  for (const auto& a : c.assemblers) {
    int io = 0;
    for (const auto& p : a.protocols) io = std::max(io, (int)p.inputs.size() + (int)p.outputs.size());
    if (!a.clip_immune)  // a clipped station shows its unclipping protocol instead
      for (const auto& p : c.unclip) io = std::max(io, (int)p.inputs.size() + (int)p.outputs.size());
    TSS = std::max(TSS, 3 + io + (int)a.tags.size() + 1);
  }
  for (const auto& ch : c.chests) TSC = std::max(TSC, 1 + c.R * c.n_inv_tokens + (int)ch.tags.size());
  ts[0] = TSW; ts[1] = TSA; ts[2] = TSS; ts[3] = TSC;
}

template <class T>
static int dalloc(mg_handle* h, T** p, size_t n, bool map_alloc = false) {
  void* q = nullptr;
  size_t bytes = (n > 0 ? n : 1) * sizeof(T);
  HIPCHK(hipMalloc(&q, bytes));
  HIPCHK(hipMemset(q, 0, bytes));
  (map_alloc ? h->map_allocs : h->allocs).push_back(q);
  *p = (T*)q;
  return 0;
}

extern "C" {

const char* mg_last_error(void) { return g_err.c_str(); }
const char* mg_version(void) { return MGD_MAX_R == 16 ? "mettagrid_hip 0.1 (gfx950; blob v5; 16 resources)" : "mettagrid_hip 0.1 (gfx950; blob v5; 32 resources)"; }
int mg_max_resources(void) { return MGD_MAX_R; }

int mg_create(const int32_t* cfg_blob, size_t cfg_words, int num_envs, int device_id, mg_handle** out) {
  if (!out) return fail("mg_create: out is NULL");
  *out = nullptr;
  if (num_envs < 1) return fail("mg_create: num_envs must be >= 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail("mg_create: no HIP device available (this library has no CPU fallback)");
  // every early `return` below releases what was created so far (device buffers, streams, events)
  struct Destroy {
    void operator()(mg_handle* p) const { (void)mg_destroy(p); }
  };
  std::unique_ptr<mg_handle, Destroy> h(new mg_handle());
  try {
    h->cfg = MgConfig::parse(cfg_blob, cfg_words);
  } catch (const std::exception& ex) {
    return fail(std::string("mg_create: ") + ex.what());
  }
  const MgConfig& c = h->cfg;
  if (mg_check_supported(c, "mg_create")) return 1;
  const bool has_damage = mg_has_damage(c);
  if (device_id >= 0) {
    HIPCHK(hipSetDevice(device_id));
    h->device = device_id;
  } else {
    HIPCHK(hipGetDevice(&h->device));
  }
  if (mg_scratch_init(h->device)) return 1;
  h->E = num_envs;
  KArgs& k = h->k;
  memset(&k, 0, sizeof(k));
  k.E = num_envs; k.A = c.A; k.T = c.T; k.SA = c.n_agent_slots; k.SG = c.n_game_slots; k.R = c.R; k.NG = c.n_group_names;
  k.TW = (k.SA + 31) / 32; k.GW = (k.SG + 31) / 32; k.n_cells = (int)c.cells.size();
  // MT19937 outputs one step may consume: the shuffle's draws (+ rare redraws) and, with damage configured, one
  // canonical double (two outputs) per agent (agent.cpp:134-177)
  const int max_draws = c.A / 2 + 2 + (has_damage ? 2 * c.A : 0);
  k.NWIN = std::min(624, max_draws + MG_WIN_MARGIN);
  if (max_draws > 624) return fail("mg_create: too many agents for the RNG window");
  k.ROWB = (int)mg_align16((size_t)c.T * 3) + 16;  // + the row's 16-byte phase in HBM (mg_encode_all)
  k.auto_reset = 1; k.map_stride = 1;
  k.max_steps = k.max_steps_cfg = c.max_steps;
  k.map_max_steps = nullptr;
  {
    bool ok = mg_fast_eligible(c);
    const char* dg = getenv("MG_DIAG");
    k.diag = dg ? atoi(dg) : 0;
    const char* force = getenv("MG_FORCE_GENERAL");
    if (force && force[0] == '1') ok = false;
    k.fast = ok ? 1 : 0;
  }
  std::vector<int32_t> dcfg = mg_build_dcfg(c);
  while (dcfg.size() % 4) dcfg.push_back(0);  // staged into LDS with 16-byte copies
  int* d_dc = nullptr;
  if (dalloc(h.get(), &d_dc, dcfg.size())) return 1;
  HIPCHK(hipMemcpy(d_dc, dcfg.data(), dcfg.size() * 4, hipMemcpyHostToDevice));
  k.dc = d_dc;
  k.gdc = d_dc;
  k.dc_words = (int)dcfg.size();
  k.dc_obs_words = dcfg[DC_OBS_WORDS];
  k.dc_stride = k.dc_words;
  k.map_cfg = nullptr;
  h->base_dcfg = dcfg;
  h->base_dc_words = k.dc_words;
  h->base_obs_words = k.dc_obs_words;
  size_t N = (size_t)num_envs * c.A;
  if (dalloc(h.get(), &k.env, (size_t)num_envs)) return 1;
  if (dalloc(h.get(), &k.mt, (size_t)num_envs * 624)) return 1;
  if (dalloc(h.get(), &k.mt_init, (size_t)num_envs * 624)) return 1;
  if (c.clipper) {  // (the weights are sized with the pool: mg_set_maps)
    if (dalloc(h.get(), &k.clip_mt, (size_t)num_envs * 624)) return 1;
    if (dalloc(h.get(), &k.clip_state, (size_t)num_envs)) return 1;
  }
  if (dalloc(h.get(), &k.agents, N)) return 1;
  if (dalloc(h.get(), &k.astats, N * k.SA)) return 1;
  if (dalloc(h.get(), &k.atouched, N * k.TW)) return 1;
  if (dalloc(h.get(), &k.gstats, (size_t)num_envs * k.SG)) return 1;
  if (dalloc(h.get(), &k.gtouched, (size_t)num_envs * k.GW)) return 1;
  if (dalloc(h.get(), &k.fin_astats, N * k.SA)) return 1;
  if (dalloc(h.get(), &k.fin_atouched, N * k.TW)) return 1;
  if (dalloc(h.get(), &k.fin_gstats, (size_t)num_envs * k.SG)) return 1;
  if (dalloc(h.get(), &k.fin_gtouched, (size_t)num_envs * k.GW)) return 1;
  if (dalloc(h.get(), &k.fin_ep_rewards, N)) return 1;
  if (dalloc(h.get(), &k.fin_steps, (size_t)num_envs)) return 1;
  if (dalloc(h.get(), &k.summary, (size_t)(3 + k.SG))) return 1;
  if (dalloc(h.get(), &k.status, 4)) return 1;
  {
    uint32_t cap = 1024;  // (8 B an entry)
    while (cap < 16u * (uint32_t)num_envs) cap <<= 1;
    if (dalloc(h.get(), &k.fin_ring, (size_t)cap)) return 1;
    k.fin_ring_mask = cap - 1u;
  }
  if (dalloc(h.get(), &k.obs, N * c.T * 3)) return 1;
  if (dalloc(h.get(), &k.rewards, N)) return 1;
  if (dalloc(h.get(), &k.term, N)) return 1;
  if (dalloc(h.get(), &k.trunc, N)) return 1;
  {
    int* acts = nullptr;
    if (dalloc(h.get(), &acts, N)) return 1;
    k.actions = acts;
  }
  if (dalloc(h.get(), &k.ep_rewards, N)) return 1;
  if (dalloc(h.get(), &k.success, N)) return 1;
  if (dalloc(h.get(), &k.exec, N)) return 1;
  if (dalloc(h.get(), &k.mode, (size_t)num_envs)) return 1;
  if (dalloc(h.get(), &k.pend, N * 32)) return 1;
  if (dalloc(h.get(), &h->d_list, (size_t)num_envs * 3)) return 1;
  if (dalloc(h.get(), &h->d_inv, (size_t)2 * MGD_MAX_R)) return 1;
  HIPCHK(hipMemset(k.obs, 0xFF, N * c.T * 3));
  {
    const char* sp = getenv("MG_SPLIT");
    h->n_split = sp ? std::max(1, std::min(8, atoi(sp))) : 1;
    HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < 7; i++) {
      HIPCHK(hipStreamCreateWithFlags(&h->aux[i], hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
    }
  }
  if (dalloc(h.get(), &h->masks, N)) return 1;
  HIPCHK(hipMemset(h->masks, 1, N));
  *out = h.release();
  return 0;
}

static int mg_set_maps_impl(mg_handle* h, const uint8_t* maps, int num_maps, const int32_t* heights, const int32_t* widths) {
  if (!h) return fail("mg_set_maps: NULL handle");
  if (!maps || !heights || !widths) return fail("mg_set_maps: NULL argument");
  if (num_maps < 1) return fail("mg_set_maps: bad dimensions");
  int HWp = 0, Hmax = 0, Wmax = 0;
  std::vector<size_t> src_off((size_t)num_maps);
  {
    size_t off = 0;
    for (int m = 0; m < num_maps; m++) {
      if (heights[m] < 1 || widths[m] < 1) return fail("mg_set_maps: bad dimensions");
      if (heights[m] > 65535 || widths[m] > 65535) return fail("mg_set_maps: map too large");
      if ((long long)heights[m] * widths[m] > (1 << 24)) return fail("mg_set_maps: map too large");
      src_off[(size_t)m] = off;
      off += (size_t)heights[m] * widths[m];
      HWp = std::max(HWp, (heights[m] * widths[m] + 1) & ~1);  // keep each map 4-byte aligned
      Hmax = std::max(Hmax, heights[m]);
      Wmax = std::max(Wmax, widths[m]);
    }
  }
  HIPCHK(hipSetDevice(h->device));
  // A second call on a live handle replaces the pool: every environment keeps a stale map id until mg_reset, so the
  // handle refuses to step in between (and stays unusable if this call fails half way).
  HIPCHK(hipDeviceSynchronize());
  h->maps_set = false;
  const MgConfig& c = h->cfg;
  KArgs& k = h->k;
  // (the assembler / chest slot counts are known only after the scan: codes are patched below)
  CellCodec codec{(int)c.walls.size(), c.A, 0};
  std::vector<uint16_t> cells((size_t)num_maps * HWp, 0);
  std::vector<int4> spawn((size_t)num_maps * c.A);
  std::vector<std::vector<int4>> asm_lists((size_t)num_maps), chest_lists((size_t)num_maps);
  std::vector<float> objcount((size_t)num_maps * c.cells.size(), 0.f);
  h->host_spawn.assign((size_t)num_maps, {});
  std::vector<int2> dims((size_t)num_maps);
  for (int m = 0; m < num_maps; m++) {
    const uint8_t* src = maps + src_off[(size_t)m];
    const int height = heights[m], width = widths[m];
    dims[(size_t)m] = make_int2(height, width);
    uint32_t next_id = 1;
    int n_agents = 0;
    for (int r = 0; r < height; r++)
      for (int cc = 0; cc < width; cc++) {
        int code = src[(size_t)r * width + cc];
        if (code == 0) continue;
        if ((size_t)code >= c.cells.size())
          return fail("Unknown object type: cell code " + std::to_string(code) + " in map " + std::to_string(m));
        const MgCell& cell = c.cells[(size_t)code];
        uint32_t id = next_id++;
        uint16_t* dst = &cells[(size_t)m * HWp + (size_t)r * width + cc];
        switch (cell.kind) {
          case MG_CELL_WALL:
            *dst = codec.wall(cell.idx);
            objcount[(size_t)m * c.cells.size() + code] += 1.f;
            break;
          case MG_CELL_AGENT:
            if (n_agents < c.A) spawn[(size_t)m * c.A + n_agents] = make_int4(r, cc, cell.idx, (int)id);
            n_agents++;
            break;
          case MG_CELL_ASSEMBLER:
            if ((size_t)c.walls.size() + c.A + asm_lists[(size_t)m].size() + 1 >= (size_t)MG_MAX_OBJECTS)
              return fail("mg_set_maps: too many objects for uint16 cells");
            *dst = codec.assembler((int)asm_lists[(size_t)m].size());
            asm_lists[(size_t)m].push_back(make_int4(r, cc, cell.idx, (int)id));
            objcount[(size_t)m * c.cells.size() + code] += 1.f;
            break;
          case MG_CELL_CHEST:
            chest_lists[(size_t)m].push_back(make_int4(r, cc, cell.idx, (int)id));  // cell code patched below
            objcount[(size_t)m * c.cells.size() + code] += 1.f;
            break;
          default:
            return fail("mg_set_maps: unsupported object kind in map");
        }
      }
    if (n_agents != c.A)
      return fail("mg_set_maps: map " + std::to_string(m) + " holds " + std::to_string(n_agents) +
                  " agents but the config says num_agents = " + std::to_string(c.A));
    h->host_spawn[(size_t)m].assign(spawn.begin() + (size_t)m * c.A, spawn.begin() + (size_t)(m + 1) * c.A);
  }
  int NAmax = 1;
  for (auto& v : asm_lists) NAmax = std::max(NAmax, (int)v.size());
  std::vector<int4> asm_flat((size_t)num_maps * NAmax, make_int4(0, 0, 0, 0));
  std::vector<int> nasm((size_t)num_maps);
  for (int m = 0; m < num_maps; m++) {
    nasm[(size_t)m] = (int)asm_lists[(size_t)m].size();
    for (size_t i = 0; i < asm_lists[(size_t)m].size(); i++) asm_flat[(size_t)m * NAmax + i] = asm_lists[(size_t)m][i];
  }
  h->host_asm = asm_lists;
  int NCmax = 0;
  for (auto& v : chest_lists) NCmax = std::max(NCmax, (int)v.size());
  codec.NA = NAmax;
  if ((size_t)c.walls.size() + c.A + NAmax + NCmax + 1 >= (size_t)MG_MAX_OBJECTS)
    return fail("mg_set_maps: too many objects for uint16 cells");
  std::vector<int4> chest_flat((size_t)num_maps * std::max(NCmax, 1), make_int4(0, 0, 0, 0));
  std::vector<int> nchest((size_t)num_maps);
  for (int m = 0; m < num_maps; m++) {
    nchest[(size_t)m] = (int)chest_lists[(size_t)m].size();
    for (size_t i = 0; i < chest_lists[(size_t)m].size(); i++) {
      const int4 ci = chest_lists[(size_t)m][i];
      chest_flat[(size_t)m * NCmax + i] = ci;
      cells[(size_t)m * HWp + (size_t)ci.x * widths[m] + ci.y] = codec.chest((int)i);
    }
  }
  h->host_chest = chest_lists;
  // LDS budget
  // token-list strides: the most tokens an object of each kind can show (wall.hpp:26-38, agent.cpp:187-213,
  // assembler.hpp:528-578)
  int ts_[4];
  mg_token_strides(c, ts_);
  if (h->n_cfg_slots > 0) {  // room for tasks that show a few more tokens per object than the base configuration
    ts_[1] += 4; ts_[2] += 4; ts_[3] += 4;
  }
  for (int i = 0; i < 4; i++) ts_[i] = std::max(ts_[i], h->ts_need[i]);
  for (int i = 0; i < 4; i++)  // the encoder's scan adds 64 counts in 16 bits (mg_encode_agent)
    if (ts_[i] > 1000) return fail("mg_set_maps: an object kind shows " + std::to_string(ts_[i]) + " tokens (limit 1000)");
  int TSW = ts_[0], TSA = ts_[1], TSS = ts_[2], TSC = ts_[3];
  if (TSW > 255 || TSA > 255 || TSS > 255 || TSC > 255) return fail("mg_set_maps: an object can show more than 255 tokens");
  TSW = (TSW + 3) & ~3; TSA = (TSA + 3) & ~3; TSS = (TSS + 3) & ~3; TSC = (TSC + 3) & ~3;  // lists are read 4 tokens (8 bytes) at a time
  h->ts[0] = TSW; h->ts[1] = TSA; h->ts[2] = TSS; h->ts[3] = TSC;
  const int NOBJ = (int)c.walls.size() + c.A + NAmax + NCmax;
  const int HEAP = (int)c.walls.size() * TSW + c.A * TSA + NAmax * TSS + NCmax * TSC;
  const int NP = c.obs_h * c.obs_w;
  k.NWINCELLS = NP;
  k.NW = (int)c.walls.size(); k.NOBJ = NOBJ; k.TSW = TSW; k.TSA = TSA; k.TSS = TSS; k.TSC = TSC; k.NCmax = NCmax; k.HEAP = HEAP;
  {
    int mx = std::max(TSW, std::max(TSA, TSS));
    k.CB = 1;
    while ((1 << k.CB) <= mx) k.CB++;
  }
  if (HEAP > 65535) return fail("mg_set_maps: token lists of one environment exceed 65535 entries");
  if (num_maps > (1 << 20)) return fail("mg_set_maps: more than 2^20 map slots");
  // LDS plan.  Normal: map tile + records + token lists in LDS.  BIG (the map does not fit): the grid tile of each
  // env lives in HBM, and the observation / reset kernels read the agent records in HBM too.
  const size_t LDS_MAX = 160 * 1024;
  bool big = false;
  size_t lds = mg_carve<false>(nullptr, nullptr, HWp, c.A, NAmax, k.SA, k.SG, k.NWIN, k.ROWB, MG_THREADS / 64, 0, NOBJ, HEAP, NP, -1, NCmax);
  if (lds > LDS_MAX) {
    big = true;
    lds = mg_carve<false>(nullptr, nullptr, 0, c.A, NAmax, k.SA, k.SG, k.NWIN, k.ROWB, MG_THREADS / 64, 0, NOBJ, HEAP, NP, 0, NCmax);
    if (lds > LDS_MAX)
      return fail("mg_set_maps: environment does not fit the 160 KiB LDS of one CU (" + std::to_string(lds) + " bytes)");
  }
  for (void* p : h->map_allocs) (void)hipFree(p);
  h->map_allocs.clear();
  h->d_map_max_steps = nullptr;  // (lived in map_allocs)
  k.map_max_steps = nullptr;
  uint16_t* d_cells; int4* d_spawn; int4* d_asm; int* d_nasm; float* d_obj; DAsm* d_asms; int2* d_dims;
  if (dalloc(h, &d_dims, dims.size(), true)) return 1;
  HIPCHK(hipMemcpy(d_dims, dims.data(), dims.size() * sizeof(int2), hipMemcpyHostToDevice));
  k.map_dims = d_dims;
  h->host_dims = dims;
  if (dalloc(h, &d_cells, cells.size(), true)) return 1;
  if (dalloc(h, &d_spawn, spawn.size(), true)) return 1;
  if (dalloc(h, &d_asm, asm_flat.size(), true)) return 1;
  if (dalloc(h, &d_nasm, nasm.size(), true)) return 1;
  if (dalloc(h, &d_obj, objcount.size(), true)) return 1;
  if (dalloc(h, &d_asms, (size_t)h->E * NAmax, true)) return 1;
  int4* d_chest; int* d_nchest; DAgent* d_chests;
  if (dalloc(h, &d_chest, chest_flat.size(), true)) return 1;
  if (dalloc(h, &d_nchest, nchest.size(), true)) return 1;
  if (dalloc(h, &d_chests, (size_t)h->E * std::max(NCmax, 1), true)) return 1;
  HIPCHK(hipMemcpy(d_chest, chest_flat.data(), chest_flat.size() * sizeof(int4), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_nchest, nchest.data(), nchest.size() * 4, hipMemcpyHostToDevice));
  k.map_chest = d_chest; k.map_nchest = d_nchest; k.chests = d_chests;
  HIPCHK(hipMemcpy(d_cells, cells.data(), cells.size() * 2, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_spawn, spawn.data(), spawn.size() * sizeof(int4), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_asm, asm_flat.data(), asm_flat.size() * sizeof(int4), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_nasm, nasm.data(), nasm.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_obj, objcount.data(), objcount.size() * 4, hipMemcpyHostToDevice));
  k.map_cells = d_cells; k.map_spawn = d_spawn; k.map_asm = d_asm; k.map_nasm = d_nasm; k.map_objcount = d_obj;
  k.asms = d_asms;
  k.P = num_maps; k.H = Hmax; k.W = Wmax; k.HW = HWp; k.NAmax = NAmax;
  h->lds_bytes = lds;
  h->big = big;
  k.big_grid = nullptr;
  k.grid_valid = nullptr;
  k.GPAD = 0;
  k.BHW = HWp;
  if (big) {
    uint16_t* bg = nullptr;
    k.GPAD = c.obs_w >> 1;
    {
      size_t m = 0;
      for (int i = 0; i < num_maps; i++) m = std::max(m, (size_t)heights[i] * (size_t)(widths[i] + k.GPAD));
      k.BHW = (int)((m + 7) & ~(size_t)7);
    }
    if (dalloc(h, &bg, (size_t)std::max(h->E, num_maps) * (size_t)k.BHW, true)) return 1;
    k.big_grid = bg;
    if (dalloc(h, &k.grid_valid, (size_t)h->E, true)) return 1;  // (zeroed: no grid holds an episode yet)
    HIPCHK(hipMemset(k.grid_valid, 0, (size_t)h->E));
  }
  const int hw_lds = big ? 0 : HWp, a_rec_obs = big ? 0 : c.A;
  h->dc_lds = !big && k.dc_words <= MG_DC_LDS_MAX_WORDS;
  int dcw = h->dc_lds ? k.dc_words : 0;
  int dcw_obs = h->dc_lds ? k.dc_obs_words : 0;
  k.NAmax = NAmax; k.NCmax = NCmax; k.HW = HWp;
  h->lds_act = mg_act_lds_bytes(k, dcw != 0, big);
  const int obs_waves = (big ? MG_OBS_BIG_THREADS : MG_THREADS) / 64;
  h->lds_obs = mg_carve_obs<false>(nullptr, nullptr, hw_lds, c.A, NAmax, k.ROWB, obs_waves, dcw_obs, NOBJ, HEAP, NP, a_rec_obs, NCmax);
  if (h->lds_act > LDS_MAX - 1024 || h->lds_obs > LDS_MAX) {
    h->dc_lds = false;
    h->lds_act = mg_act_lds_bytes(k, false, big);
    h->lds_obs = mg_carve_obs<false>(nullptr, nullptr, hw_lds, c.A, NAmax, k.ROWB, obs_waves, 0, NOBJ, HEAP, NP, a_rec_obs, NCmax);
  }
  if (h->lds_act > LDS_MAX - 1024 || h->lds_obs > LDS_MAX)
    return fail("mg_set_maps: environment does not fit the 160 KiB LDS of one CU (act " + std::to_string(h->lds_act) +
                ", obs " + std::to_string(h->lds_obs) + " bytes)");
  h->lds_uses = mg_carve_uses<false>(nullptr, nullptr, c.A, NAmax, k.SG, h->dc_lds ? k.dc_words : 0);
  // Fused step (fast configs whose environment fits LDS): one launch per step.  The byte copy of the pool halves the
  // tile where every object code fits a byte.
  h->fused = false;
  k.map_cells8 = nullptr;
  // the byte copy of the pool (fused kernel): rows of W + PADC bytes, the PADC bytes after each row zero, so that a
  // window cell beside the map reads "empty" without a bounds test (mg_window_lookup_direct)
  k.PADC = c.obs_w >> 1;
  {
    size_t hw8 = 0;
    for (int m = 0; m < num_maps; m++) hw8 = std::max(hw8, (size_t)heights[m] * (size_t)(widths[m] + k.PADC));
    k.HW8 = (int)mg_align16(hw8);
  }
  {
    const char* nf = getenv("MG_NO_FUSED");
    if (k.fast && !big && !(nf && nf[0] == '1')) {
      h->cells8 = NOBJ <= 255;  // the kernel reads the static layer from the byte copy of the pool
      h->fused_dcl = k.dc_obs_words <= MG_DC_LDS_MAX_WORDS;
      if (const char* d = getenv("MG_FUSED_DCL")) h->fused_dcl = h->fused_dcl && d[0] != '0';  // (experiments: no config prefix in LDS)
      h->lds_fused = mg_fused_lds_bytes(k, h->fused_dcl);
      if (h->lds_fused > LDS_MAX / 2 && h->fused_dcl) {  // keep at least two workgroups per CU: drop the config copy first
        h->fused_dcl = false;
        h->lds_fused = mg_fused_lds_bytes(k, false);
      }
      h->fused = h->cells8 && h->lds_fused <= LDS_MAX;
      if (h->fused) {
        std::vector<uint8_t> c8((size_t)num_maps * k.HW8, 0);
        for (int m = 0; m < num_maps; m++)
          for (int r = 0; r < heights[m]; r++)
            for (int cc = 0; cc < widths[m]; cc++)
              c8[(size_t)m * k.HW8 + (size_t)r * (widths[m] + k.PADC) + cc] = (uint8_t)cells[(size_t)m * HWp + (size_t)r * widths[m] + cc];
        uint8_t* d8 = nullptr;
        if (dalloc(h, &d8, c8.size(), true)) return 1;
        HIPCHK(hipMemcpy(d8, c8.data(), c8.size(), hipMemcpyHostToDevice));
        k.map_cells8 = d8;
      }
      if (h->fused) {
        // (tasks written into configuration slots survive a new pool: a task with modifiers keeps the handle on the
        // build that honours them, whatever the handle's own configuration says)
        h->no_modifiers = !mg_has_modifiers(c) && !h->slot_has_modifiers;
        HIPCHK(mg_raise_lds_cap_fused((int)LDS_MAX));
        HIPCHK(mg_raise_lds_cap_fused_nomod((int)LDS_MAX));
        HIPCHK(mg_raise_lds_cap_fused_nomod_o7((int)LDS_MAX));
        HIPCHK(mg_raise_lds_cap_fused_o6((int)LDS_MAX));
        HIPCHK(mg_raise_lds_cap_fused_nomod_w1((int)LDS_MAX));
        HIPCHK(mg_raise_lds_cap_fused_nomod_w18((int)LDS_MAX));
        // A few agents per environment (navigation: 4): one wave does the whole step; a second one would repeat the
        // scalar work and wait through the action and use phases.  MG_FUSED_W1=0/1 overrides.
        h->lds_fused_w1 = mg_fused_lds_bytes_nomod_w1(k, h->fused_dcl);
        {
          const int dcw_f = h->fused_dcl ? k.dc_obs_words : 0;
          const size_t b2 = mg_carve_fused(&h->fl2, k.A, k.SG, k.ROWB, 2, dcw_f, k.NOBJ, k.HEAP, k.NWINCELLS);
          const size_t b1 = mg_carve_fused(&h->fl1, k.A, k.SG, k.ROWB, 1, dcw_f, k.NOBJ, k.HEAP, k.NWINCELLS);
          if (b2 != h->lds_fused || b1 != h->lds_fused_w1) return fail("mg_set_maps: fused LDS layouts disagree (internal)");
        }
        h->fused_w1 = h->no_modifiers && c.A <= 8 && h->lds_fused_w1 <= LDS_MAX;
        h->fused_w18 = false;
        {
          // One launch steps all E environments.  When they do not fit one round of resident workgroups, the ones left
          // over start as the first finish and run through a CU that empties around them: 4096 arena environments on
          // 256 CUs take 0.060 ms where 3584 (one round of 14 per CU) take 0.046.  A build that holds them all at once
          // wins then, although it is the slower one at every other count (arena as ONE wave per environment, 16 per
          // CU by LDS: 0.056 ms; navigation at 8 waves per SIMD, 32 per CU: 0.029 against 0.033 ms for 8192).
          hipDeviceProp_t prop;
          HIPCHK(hipGetDeviceProperties(&prop, h->device));
          const long cus = prop.multiProcessorCount, lds_cu = 160 * 1024;
          auto round = [&](size_t lds, int waves_per_wg, int occ) {
            return cus * std::min<long>(4 * occ / waves_per_wg, lds ? (long)(lds_cu / lds) : 64);
          };
          if (h->no_modifiers && h->lds_fused_w1 <= LDS_MAX) {
            if (c.A > 8) {
              if (h->E > round(h->lds_fused, 2, 7) && h->E <= round(h->lds_fused_w1, 1, 7)) h->fused_w1 = true;
            } else if (h->E > round(h->lds_fused_w1, 1, 7) && h->E <= round(h->lds_fused_w1, 1, 8)) {
              h->fused_w18 = true;
            }
          }
        }
        if (const char* w1 = getenv("MG_FUSED_W1")) {  // 0 / 1 / 8: two waves, one wave, one wave at 8 waves per SIMD
          h->fused_w1 = h->no_modifiers && (w1[0] == '1' || w1[0] == '8') && h->lds_fused_w1 <= LDS_MAX;
          h->fused_w18 = h->fused_w1 && w1[0] == '8';
        }
        // The builds with the larger register budget (7 waves per SIMD, 72 registers, 92 scalar registers; 6 waves with
        // limit modifiers) are the default.  The 8-wave builds (64 registers and, on this hardware, 76 scalar ones) spill
        // in every phase: they won in round 2 for exactly one round of 16 workgroups per CU, and lose everywhere since the
        // round-3 kernel (one launch over 4096 environments: 0.0595 against 0.0645 ms).  MG_FUSED_O7=0 selects them.
        h->fused_o7 = true;
        if (const char* o7 = getenv("MG_FUSED_O7")) h->fused_o7 = o7[0] == '1';
      }
    }
  }
  if (getenv("MG_VERBOSE"))
    fprintf(stderr, "mettagrid_hip: LDS bytes per workgroup: act %zu, obs %zu, uses %zu, fused %zu%s (dc words %d, obs prefix %d, heap %d tokens, %d objects)\n",
            h->lds_act, h->lds_obs, h->lds_uses, h->lds_fused, h->fused ? (h->cells8 ? " [byte tile]" : " [u16 tile]") : " [off]",
            k.dc_words, k.dc_obs_words, HEAP, NOBJ);
  {
    // the cap on dynamic LDS is a per-function setting shared by every handle of the process: raise it once
    const int cap = (int)LDS_MAX;
    HIPCHK(mg_raise_lds_cap_act(cap - 1024));  // (the kernel also holds a little static LDS: block-wide votes)
    HIPCHK(mg_raise_lds_cap_uses(cap));
    HIPCHK(mg_raise_lds_cap_obs(cap));
    HIPCHK(mg_raise_lds_cap_reset(cap));
  }
  // per-slot configurations: every map slot starts on configuration slot 0
  k.map_cfg = nullptr;
  h->d_map_cfg = nullptr;
  h->host_map_cfg.assign((size_t)num_maps, 0);
  if (dalloc(h, &h->d_tlist, (size_t)num_maps, true)) return 1;
  if (h->n_cfg_slots > 0) {
    if (dalloc(h, &h->d_map_cfg, (size_t)num_maps, true)) return 1;
    k.map_cfg = h->d_map_cfg;
  }
  k.clip_w = nullptr;
  if (c.clipper && dalloc(h, &k.clip_w, (size_t)h->E * (size_t)std::max(NAmax, 1), true)) return 1;
  // ---- episode-start templates: run the reset kernel once per map into template buffers ------------------
  {
    const size_t P = (size_t)num_maps, PA = P * c.A;
    DAgent* t_chests;
    if (dalloc(h, &t_chests, P * (size_t)std::max(NCmax, 1), true)) return 1;
    DAgent* t_agents; DAsm* t_asms; float* t_astats; uint32_t* t_atouched; float* t_gstats; uint32_t* t_gtouched;
    if (dalloc(h, &t_agents, PA, true)) return 1;
    if (dalloc(h, &t_asms, P * NAmax, true)) return 1;
    if (dalloc(h, &t_astats, PA * k.SA, true)) return 1;
    if (dalloc(h, &t_atouched, PA * k.TW, true)) return 1;
    if (dalloc(h, &t_gstats, P * k.SG, true)) return 1;
    if (dalloc(h, &t_gtouched, P * k.GW, true)) return 1;
    // (the reset kernel's other outputs -- observations, flags, RNG -- land in throw-away buffers kept with the pool,
    // so that mg_refresh_maps can rebuild templates without allocating)
    k.t_asm_tok = nullptr;
    k.t_asm_n = nullptr;
    k.build_templates = 0;
    if (h->fused) {  // static token lists of the assemblers, per map (built by the template pass below)
      if (dalloc(h, &k.t_asm_tok, P * (size_t)NAmax * 2 * (size_t)TSS, true)) return 1;
      if (dalloc(h, &k.t_asm_n, P * (size_t)NAmax, true)) return 1;
    }
    KArgs kt = k;
    kt.build_templates = 1;
    kt.E = (int)P;
    uint32_t* d_seeds = nullptr;
    int* d_maps = nullptr;
    if (dalloc(h, &kt.env, P, true)) return 1;
    if (dalloc(h, &kt.mt, P * 624, true)) return 1;
    kt.mt_init = nullptr;
    if (c.clipper) {  // the templates' own clipper: what it draws is redrawn per environment at a restart
      if (dalloc(h, &kt.clip_mt, P * 624, true)) return 1;
      if (dalloc(h, &kt.clip_state, P, true)) return 1;
      if (dalloc(h, &kt.clip_w, P * (size_t)k.NAmax, true)) return 1;
    }
    kt.chests = t_chests;
    kt.agents = t_agents; kt.asms = t_asms; kt.astats = t_astats; kt.atouched = t_atouched;
    kt.gstats = t_gstats; kt.gtouched = t_gtouched;
    if (dalloc(h, &kt.obs, PA * c.T * 3, true)) return 1;
    if (dalloc(h, &kt.rewards, PA, true)) return 1;
    if (dalloc(h, &kt.ep_rewards, PA, true)) return 1;
    if (dalloc(h, &kt.term, PA, true)) return 1;
    if (dalloc(h, &kt.trunc, PA, true)) return 1;
    if (dalloc(h, &kt.success, PA, true)) return 1;
    if (dalloc(h, &d_seeds, P, true)) return 1;
    if (dalloc(h, &d_maps, P, true)) return 1;
    if (big) {  // template builds must not touch the live grids of the environments
      uint16_t* tg = nullptr;
      if (dalloc(h, &tg, P * (size_t)k.BHW, true)) return 1;
      kt.big_grid = tg;
      kt.grid_valid = nullptr;
    }
    {
      std::vector<int> iota((size_t)P);
      for (size_t i = 0; i < P; i++) iota[i] = (int)i;
      HIPCHK(hipMemcpy(d_maps, iota.data(), P * 4, hipMemcpyHostToDevice));
      kt.reset_envs = nullptr;
      kt.reset_count = nullptr;
      kt.reset_seeds = d_seeds;
      kt.reset_maps = d_maps;
      mg_launch_reset(kt, (unsigned)P, h->lds_bytes, h->big, 0);
      hipError_t e2 = hipGetLastError();
      hipError_t e3 = hipDeviceSynchronize();
      if (e2 != hipSuccess || e3 != hipSuccess)
        return fail(std::string("mg_set_maps: template build failed: ") + hipGetErrorString(e3 != hipSuccess ? e3 : e2));
    }
    h->kt = kt;
    h->d_iota = d_maps;
    k.t_chests = t_chests;
    k.t_agents = t_agents; k.t_asms = t_asms; k.t_astats = t_astats; k.t_atouched = t_atouched;
    k.t_gstats = t_gstats; k.t_gtouched = t_gtouched;
  }
  k.map_max_steps = nullptr;  // (freed with the old pool)
  h->d_map_max_steps = nullptr;
  if (!h->in_set_generated) h->generated = false;
  h->gen_ready = false;
  h->maps_set = true;
  h->needs_reset = true;
  if (k.map_stride >= num_maps) k.map_stride %= num_maps;
  return 0;
}

int mg_set_maps(mg_handle* h, const uint8_t* maps, int num_maps, int height, int width) {
  if (num_maps < 1) return fail("mg_set_maps: bad dimensions");
  std::vector<int32_t> hs((size_t)num_maps, height), ws((size_t)num_maps, width);
  return mg_set_maps_impl(h, maps, num_maps, hs.data(), ws.data());
}

int mg_set_maps_ragged(mg_handle* h, const uint8_t* maps, int num_maps, const int32_t* heights, const int32_t* widths) {
  return mg_set_maps_impl(h, maps, num_maps, heights, widths);
}

// ---- device map generator ---------------------------------------------------------------------------------------
static int mg_use_device(mg_handle* h);
// The pool's writable pointers and the slot bookkeeping of the generator kernel (pools whose maps share one size).
static int mg_mapgen_setup(mg_handle* h, int height, int width) {
  KArgs& k = h->k;
  const size_t P = (size_t)k.P;
  int *d_list, *d_count;
  uint32_t *d_state, *d_inuse;
  uint8_t* d_bytes;
  if (dalloc(h, &d_list, P, true) || dalloc(h, &d_count, 1, true) || dalloc(h, &d_state, P, true) || dalloc(h, &d_inuse, P, true) ||
      dalloc(h, &d_bytes, P * mg_align16((size_t)k.HW), true))
    return 1;
  MapGenArgs& g = h->gen;
  memset(&g, 0, sizeof(g));
  g.gdc = k.gdc;
  g.height = height; g.width = width;
  g.map_cells = const_cast<uint16_t*>(k.map_cells);
  g.map_cells8 = const_cast<uint8_t*>(k.map_cells8);
  g.map_bytes = d_bytes;
  g.map_spawn = const_cast<int4*>(k.map_spawn);
  g.map_asm = const_cast<int4*>(k.map_asm);
  g.map_nasm = const_cast<int*>(k.map_nasm);
  g.map_chest = const_cast<int4*>(k.map_chest);
  g.map_nchest = const_cast<int*>(k.map_nchest);
  g.map_objcount = const_cast<float*>(k.map_objcount);
  g.HW = k.HW; g.HW8 = k.HW8; g.HWB = (int)mg_align16((size_t)k.HW); g.PADC = k.PADC; g.A = k.A; g.NW = k.NW; g.NAmax = k.NAmax; g.NCmax = k.NCmax; g.n_cells = k.n_cells;
  g.slot_state = d_state; g.inuse = d_inuse; g.gen = 0;
  g.list = d_list; g.count = d_count; g.status = k.status;
  g.direct_slot = -1;
  h->d_inuse = d_inuse;
  HIPCHK(mg_raise_lds_cap_mapgen(160 * 1024 - 4096));
  h->gen_ready = true;
  return 0;
}

static int mg_refresh_impl(mg_handle* h, int force, hipStream_t st) {
  MapGenArgs g = h->gen;
  g.gen = ++h->gen.gen;
  g.force = force;
  HIPCHK(hipMemsetAsync(g.count, 0, sizeof(int), st));
  if (!force) mg_launch_mark_maps(h->k.env, h->E, h->d_inuse, g.gen, st);
  mg_launch_generate_maps(g, (unsigned)h->k.P, st);
  HIPCHK(hipGetLastError());
  // episode-start templates of the regenerated slots (blocks past the device-side count exit at once)
  KArgs kt = h->kt;
  kt.map_max_steps = h->k.map_max_steps;
  kt.map_cfg = h->k.map_cfg;
  kt.reset_envs = g.list;
  kt.reset_maps = g.list;
  kt.reset_count = g.count;
  mg_launch_reset(kt, (unsigned)h->k.P, h->lds_bytes, h->big, st);
  HIPCHK(hipGetLastError());
  return 0;
}

int mg_set_generated_maps(mg_handle* h, int height, int width, int fill_code, int n_rooms, const int32_t* rooms,
                          const int32_t* room_sym_begin, const int32_t* sym_code, const int32_t* sym_count, int num_maps,
                          uint64_t seed) {
  if (!h) return fail("mg_set_generated_maps: NULL handle");
  if (!rooms || !room_sym_begin || (!sym_code && room_sym_begin[n_rooms > 0 ? n_rooms : 0] > 0) || n_rooms < 1)
    return fail("mg_set_generated_maps: NULL argument or no rooms");
  if (num_maps < 1 || height < 1 || width < 1) return fail("mg_set_generated_maps: bad dimensions");
  if ((long long)height * width > MG_MAPGEN_MAX_CELLS)
    return fail("mg_set_generated_maps: map larger than " + std::to_string(MG_MAPGEN_MAX_CELLS) + " cells (the generator builds a slot in LDS)");
  const MgConfig& c = h->cfg;
  auto code_ok = [&](int code) { return code >= 0 && (size_t)code < c.cells.size() && code < 256; };
  if (!code_ok(fill_code) || (fill_code != 0 && c.cells[(size_t)fill_code].kind != MG_CELL_WALL))
    return fail("mg_set_generated_maps: the border object must be empty or a wall type");
  if (room_sym_begin[0] != 0) return fail("mg_set_generated_maps: room_sym_begin[0] must be 0");
  std::vector<uint8_t> canon((size_t)height * width, (uint8_t)fill_code);
  long long n_agents = 0;
  for (int r = 0; r < n_rooms; r++) {
    const int r0 = rooms[4 * r], c0 = rooms[4 * r + 1], rh = rooms[4 * r + 2], rw = rooms[4 * r + 3];
    if (r0 < 0 || c0 < 0 || rh < 1 || rw < 1 || r0 + rh > height || c0 + rw > width)
      return fail("mg_set_generated_maps: room " + std::to_string(r) + " lies outside the map");
    for (int q = 0; q < r; q++) {
      const int a0 = rooms[4 * q], b0 = rooms[4 * q + 1], ah = rooms[4 * q + 2], aw = rooms[4 * q + 3];
      if (r0 < a0 + ah && a0 < r0 + rh && c0 < b0 + aw && b0 < c0 + rw)
        return fail("mg_set_generated_maps: rooms " + std::to_string(q) + " and " + std::to_string(r) + " overlap");
    }
    if (room_sym_begin[r + 1] < room_sym_begin[r]) return fail("mg_set_generated_maps: room_sym_begin must not decrease");
    long long total = 0;
    for (int j = room_sym_begin[r]; j < room_sym_begin[r + 1]; j++) {
      if (!code_ok(sym_code[j]) || sym_code[j] == 0) return fail("Unknown object type: cell code " + std::to_string(sym_code[j]) + " in the generator's symbol list");
      if (sym_count[j] < 0) return fail("mg_set_generated_maps: negative symbol count");
      total += sym_count[j];
      if (c.cells[(size_t)sym_code[j]].kind == MG_CELL_AGENT) n_agents += sym_count[j];
    }
    if (total > (long long)rh * rw)
      return fail("mg_set_generated_maps: room " + std::to_string(r) + " holds " + std::to_string(total) + " symbols but only " +
                  std::to_string((long long)rh * rw) + " cells (halve the counts as the reference's builders do)");
    // canonical layout: the room's symbols in row-major order (sizes the pool: assembler / chest slots per map)
    for (int i = 0; i < rh * rw; i++) canon[(size_t)(r0 + i / rw) * width + c0 + i % rw] = 0;
    int p = 0;
    for (int j = room_sym_begin[r]; j < room_sym_begin[r + 1]; j++)
      for (int q = 0; q < sym_count[j]; q++, p++) canon[(size_t)(r0 + p / rw) * width + c0 + p % rw] = (uint8_t)sym_code[j];
  }
  if (n_agents != c.A)
    return fail("mg_set_generated_maps: the rooms hold " + std::to_string(n_agents) + " agents but the config says num_agents = " + std::to_string(c.A));
  std::vector<uint8_t> pool((size_t)num_maps * canon.size());
  for (int m = 0; m < num_maps; m++) memcpy(&pool[(size_t)m * canon.size()], canon.data(), canon.size());
  h->in_set_generated = true;
  h->generated = false;
  int rc = mg_set_maps(h, pool.data(), num_maps, height, width);
  h->in_set_generated = false;
  if (rc) return rc;
  h->maps_set = false;  // (until the first generation below has succeeded)
  const int n_sym = room_sym_begin[n_rooms];
  int *d_rooms, *d_begin, *d_code, *d_cnt;
  if (dalloc(h, &d_rooms, (size_t)n_rooms * 4, true) || dalloc(h, &d_begin, (size_t)n_rooms + 1, true) ||
      dalloc(h, &d_code, (size_t)n_sym, true) || dalloc(h, &d_cnt, (size_t)n_sym, true))
    return 1;
  if (mg_mapgen_setup(h, height, width)) return 1;
  HIPCHK(hipMemcpy(d_rooms, rooms, (size_t)n_rooms * 16, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_begin, room_sym_begin, ((size_t)n_rooms + 1) * 4, hipMemcpyHostToDevice));
  if (n_sym > 0) {
    HIPCHK(hipMemcpy(d_code, sym_code, (size_t)n_sym * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_cnt, sym_count, (size_t)n_sym * 4, hipMemcpyHostToDevice));
  }
  MapGenArgs& g = h->gen;
  g.fill_code = fill_code; g.n_rooms = n_rooms;
  g.rooms = d_rooms; g.room_sym_begin = d_begin; g.sym_code = d_code; g.sym_count = d_cnt;
  g.seed = seed;
  if (mg_refresh_impl(h, 1, 0)) return 1;
  HIPCHK(hipDeviceSynchronize());
  h->generated = true;
  h->maps_set = true;
  h->needs_reset = true;
  return 0;
}

int mg_refresh_maps(mg_handle* h, int force, void* stream) {
  if (!h) return fail("mg_refresh_maps: NULL handle");
  if (!h->maps_set || !h->generated) return fail("mg_refresh_maps: the pool was not made by mg_set_generated_maps");
  if (mg_use_device(h)) return 1;
  if (mg_refresh_impl(h, force ? 1 : 0, (hipStream_t)stream)) return 1;
  if (force) h->needs_reset = true;  // maps changed under running episodes
  return 0;
}

int mg_write_map_slot(mg_handle* h, int slot, const uint8_t* cells, void* stream) {
  if (!h || !cells) return fail("mg_write_map_slot: NULL argument");
  if (!h->maps_set) return fail("mg_write_map_slot: call mg_set_maps first");
  KArgs& k = h->k;
  if (slot < 0 || slot >= k.P) return fail("mg_write_map_slot: slot out of range");
  const int height = h->host_dims[0].x, width = h->host_dims[0].y;
  for (const auto& d : h->host_dims)
    if (d.x != height || d.y != width) return fail("mg_write_map_slot: the pool's maps differ in size");
  if ((long long)height * width > MG_MAPGEN_MAX_CELLS) return fail("mg_write_map_slot: map too large for the slot builder");
  // the same checks mg_set_maps makes, before anything is written
  const MgConfig& c = h->cfg;
  int n_agents = 0, n_asm = 0, n_chest = 0;
  for (int i = 0; i < height * width; i++) {
    const int code = cells[i];
    if (code == 0) continue;
    if ((size_t)code >= c.cells.size()) return fail("Unknown object type: cell code " + std::to_string(code) + " in the map for slot " + std::to_string(slot));
    const int kind = c.cells[(size_t)code].kind;
    n_agents += kind == MG_CELL_AGENT;
    n_asm += kind == MG_CELL_ASSEMBLER;
    n_chest += kind == MG_CELL_CHEST;
  }
  if (n_agents != c.A)
    return fail("mg_write_map_slot: the map holds " + std::to_string(n_agents) + " agents but the config says num_agents = " + std::to_string(c.A));
  if (n_asm > k.NAmax || n_chest > k.NCmax)
    return fail("mg_write_map_slot: the map holds " + std::to_string(n_asm) + " assemblers / " + std::to_string(n_chest) +
                " chests, the pool was sized for " + std::to_string(k.NAmax) + " / " + std::to_string(k.NCmax));
  if (mg_use_device(h)) return 1;
  if (!h->gen_ready && mg_mapgen_setup(h, height, width)) return 1;
  hipStream_t st = (hipStream_t)stream;
  MapGenArgs g = h->gen;
  HIPCHK(hipMemcpyAsync(g.map_bytes + (size_t)slot * g.HWB, cells, (size_t)height * width, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));  // (`cells` is the caller's)
  g.direct_slot = slot;
  HIPCHK(hipMemsetAsync(g.count, 0, sizeof(int), st));
  mg_launch_generate_maps(g, 1, st);
  HIPCHK(hipGetLastError());
  KArgs kt = h->kt;
  kt.map_max_steps = h->k.map_max_steps;
  kt.map_cfg = h->k.map_cfg;
  kt.reset_envs = g.list;
  kt.reset_maps = g.list;
  kt.reset_count = g.count;
  mg_launch_reset(kt, 1, h->lds_bytes, h->big, st);
  HIPCHK(hipGetLastError());
  return 0;
}

int mg_get_map(mg_handle* h, int map_id, uint8_t* out, void* stream) {
  if (!h || !out) return fail("mg_get_map: NULL argument");
  if (!h->maps_set || !h->generated) return fail("mg_get_map: the pool was not made by mg_set_generated_maps");
  if (map_id < 0 || map_id >= h->k.P) return fail("mg_get_map: map id out of range");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  HIPCHK(hipMemcpy(out, h->gen.map_bytes + (size_t)map_id * h->gen.HWB, (size_t)h->gen.height * h->gen.width, hipMemcpyDeviceToHost));
  return 0;
}

// Inspection helpers read the object lists of a map from the host mirror; a generated pool lives on the device only.
static int mg_sync_lists(mg_handle* h, uint32_t map_id) {
  if (!h->generated && !h->gen_ready) return 0;  // (gen_ready: slots may have been rewritten by mg_write_map_slot)
  const KArgs& k = h->k;
  const size_t m = map_id;
  int na = 0, nc = 0;
  HIPCHK(hipMemcpy(&na, k.map_nasm + m, 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&nc, k.map_nchest + m, 4, hipMemcpyDeviceToHost));
  h->host_spawn[m].resize((size_t)k.A);
  h->host_asm[m].resize((size_t)na);
  h->host_chest[m].resize((size_t)nc);
  HIPCHK(hipMemcpy(h->host_spawn[m].data(), k.map_spawn + m * k.A, (size_t)k.A * sizeof(int4), hipMemcpyDeviceToHost));
  if (na) HIPCHK(hipMemcpy(h->host_asm[m].data(), k.map_asm + m * k.NAmax, (size_t)na * sizeof(int4), hipMemcpyDeviceToHost));
  if (nc) HIPCHK(hipMemcpy(h->host_chest[m].data(), k.map_chest + m * k.NCmax, (size_t)nc * sizeof(int4), hipMemcpyDeviceToHost));
  return 0;
}

// ---- per-slot configurations (curriculum tasks) -----------------------------------------------------------------
int mg_set_config_slots(mg_handle* h, int num_slots, int headroom_words) {
  if (!h) return fail("mg_set_config_slots: NULL handle");
  if (h->maps_set) return fail("mg_set_config_slots: call before mg_set_maps (the pool is sized for the slots)");
  if (num_slots < 1 || num_slots > (1 << 20)) return fail("mg_set_config_slots: bad slot count");
  if (headroom_words < 0) headroom_words = h->base_dc_words / 4 + 64;
  HIPCHK(hipSetDevice(h->device));
  KArgs& k = h->k;
  const int stride = (int)((h->base_dc_words + headroom_words + 3) & ~3);
  const int obs_words = (int)std::min<long long>(stride, ((long long)h->base_obs_words + headroom_words + 3) & ~3LL);
  int* d = nullptr;
  if (dalloc(h, &d, (size_t)num_slots * stride)) return 1;
  std::vector<int32_t> all((size_t)num_slots * stride, 0);
  for (int s = 0; s < num_slots; s++) memcpy(&all[(size_t)s * stride], h->base_dcfg.data(), h->base_dcfg.size() * 4);
  HIPCHK(hipMemcpy(d, all.data(), all.size() * 4, hipMemcpyHostToDevice));
  k.dc = d;
  k.gdc = d;
  k.dc_stride = stride;
  k.dc_words = stride;          // LDS staging copies the slot's whole stride / the largest prefix
  k.dc_obs_words = obs_words;
  h->n_cfg_slots = num_slots;
  return 0;
}

#define MG_RING_SLOTS 32
// Upload `words` int32 from `src` to `dst` on `st` through the pinned ring (falls back to a synchronous copy when the
// block is larger than a ring slot).
static int mg_upload(mg_handle* h, int32_t* dst, const int32_t* src, size_t words, hipStream_t st) {
  if (!h->ring) {
    h->ring_words = std::max<size_t>(256, (size_t)h->k.dc_stride);
    HIPCHK(hipHostMalloc((void**)&h->ring, h->ring_words * 4 * MG_RING_SLOTS, hipHostMallocDefault));
    h->ring_ev.resize(MG_RING_SLOTS);
    h->ring_used.assign(MG_RING_SLOTS, 0);
    for (auto& e : h->ring_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  if (words > h->ring_words) {
    HIPCHK(hipMemcpyAsync(dst, src, words * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
  }
  const int i = h->ring_next;
  h->ring_next = (i + 1) % MG_RING_SLOTS;
  if (h->ring_used[(size_t)i]) HIPCHK(hipEventSynchronize(h->ring_ev[(size_t)i]));
  int32_t* stage = h->ring + (size_t)i * h->ring_words;
  memcpy(stage, src, words * 4);
  HIPCHK(hipMemcpyAsync(dst, stage, words * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipEventRecord(h->ring_ev[(size_t)i], st));
  h->ring_used[(size_t)i] = 1;
  return 0;
}

static int mg_rebuild_templates(mg_handle* h, const std::vector<int>& maps, hipStream_t st) {
  if (maps.empty()) return 0;
  if (mg_upload(h, h->d_tlist, maps.data(), maps.size(), st)) return 1;
  KArgs kt = h->kt;
  kt.map_max_steps = h->k.map_max_steps;
  kt.map_cfg = h->k.map_cfg;
  kt.reset_envs = h->d_tlist;
  kt.reset_maps = h->d_tlist;
  kt.reset_count = nullptr;
  mg_launch_reset(kt, (unsigned)maps.size(), h->lds_bytes, h->big, st);
  HIPCHK(hipGetLastError());
  return 0;
}

// A task's blob -> its device configuration stream, padded to the slot stride: parsed, checked against what the handle's
// buffers and observation layout were sized for, and the handle's kernel choice updated (a task outside the fast class
// moves the handle to the general kernels; a task with limit modifiers to the builds that honour them).
static int mg_task_stream(mg_handle* h, const int32_t* cfg_blob, size_t cfg_words, const char* who, std::vector<int32_t>& dcfg) {
  const std::string W = who;
  MgConfig v;
  try {
    v = MgConfig::parse(cfg_blob, cfg_words);
  } catch (const std::exception& ex) {
    return fail(W + ": " + ex.what());
  }
  if (mg_check_supported(v, who)) return 1;
  const MgConfig& c = h->cfg;
  // what sizes buffers, LDS and the observation layout must match the configuration the handle was created with
  auto differs = [&](const char* what) { return fail(W + ": the task's configuration differs from the handle's in " + what); };
  if (v.R != c.R) return differs("the number of resources");
  if (v.A != c.A) return differs("num_agents");
  if (v.T != c.T || v.obs_w != c.obs_w || v.obs_h != c.obs_h || v.token_base != c.token_base) return differs("the observation shape");
  if (v.n_agent_slots != c.n_agent_slots || v.n_game_slots != c.n_game_slots || v.n_group_names != c.n_group_names) return differs("the statistics layout");
  if (v.n_vibes != c.n_vibes || v.n_inv_tokens != c.n_inv_tokens) return differs("vibes / inventory token encoding");
  if (v.actions.size() != c.actions.size()) return differs("the action space");
  for (size_t i = 0; i < c.actions.size(); i++)
    if (v.actions[i].kind != c.actions[i].kind || v.actions[i].arg != c.actions[i].arg) return differs("the action space");
  if (v.cells.size() != c.cells.size()) return differs("the map cell table");
  for (size_t i = 0; i < c.cells.size(); i++)
    if (v.cells[i].kind != c.cells[i].kind || v.cells[i].idx != c.cells[i].idx) return differs("the map cell table");
  if (v.groups.size() != c.groups.size() || v.walls.size() != c.walls.size() || v.assemblers.size() != c.assemblers.size() ||
      v.chests.size() != c.chests.size())
    return differs("the object types");
  for (int i = 0; i < MG_F_COUNT; i++)
    if (v.fid[i] != c.fid[i]) return differs("the observation feature ids");
  if (v.fid_inv != c.fid_inv || v.fid_in != c.fid_in || v.fid_out != c.fid_out) return differs("the observation feature ids");
  if (v.init_obs_passes != c.init_obs_passes) return differs("init_obs_passes");
  if ((v.clipper != 0) != (c.clipper != 0)) return differs("having a clipper (its buffers belong to the handle)");
  if (mg_has_damage(v) && !mg_has_damage(c)) return differs("damage (the RNG window was sized without it)");
  int ts[4];
  mg_token_strides(v, ts);
  if (h->maps_set && (ts[0] > h->ts[0] || ts[1] > h->ts[1] || ts[2] > h->ts[2] || ts[3] > h->ts[3]))
    return differs("tokens per object (more than the pool was sized for)");
  if (!h->maps_set)  // the pool is sized later: for the widest task seen until then
    for (int i = 0; i < 4; i++) h->ts_need[i] = std::max(h->ts_need[i], ts[i]);
  dcfg = mg_build_dcfg(v);
  while (dcfg.size() % 4) dcfg.push_back(0);
  KArgs& k = h->k;
  if ((int)dcfg.size() > k.dc_stride || dcfg[DC_OBS_WORDS] > k.dc_obs_words)
    return fail(W + ": the task's configuration stream (" + std::to_string(dcfg.size()) + " words, prefix " +
                std::to_string(dcfg[DC_OBS_WORDS]) + ") exceeds the slot (" + std::to_string(k.dc_stride) + ", prefix " +
                std::to_string(k.dc_obs_words) + "): raise headroom_words of mg_set_config_slots");
  dcfg.resize((size_t)k.dc_stride, 0);
  // a task outside the wave-parallel / modifier-free fast paths moves the whole handle to the general kernels
  if (k.fast && !mg_fast_eligible(v)) {
    k.fast = 0;
    h->kt.fast = 0;
    h->fused = false;
  }
  if (mg_has_modifiers(v)) {
    h->slot_has_modifiers = true;
    h->no_modifiers = false;
    h->fused_w1 = h->fused_w18 = false;  // (the one-wave builds exist without modifiers only)
  }
  return 0;
}

int mg_write_config_slot(mg_handle* h, int slot, const int32_t* cfg_blob, size_t cfg_words, void* stream) {
  if (!h || !cfg_blob) return fail("mg_write_config_slot: NULL argument");
  if (h->n_cfg_slots < 1) return fail("mg_write_config_slot: call mg_set_config_slots first");
  if (slot < 0 || slot >= h->n_cfg_slots) return fail("mg_write_config_slot: slot out of range");
  std::vector<int32_t> dcfg;
  if (mg_task_stream(h, cfg_blob, cfg_words, "mg_write_config_slot", dcfg)) return 1;
  KArgs& k = h->k;
  if (mg_use_device(h)) return 1;
  hipStream_t st = (hipStream_t)stream;
  if (mg_upload(h, const_cast<int*>(k.gdc) + (size_t)slot * k.dc_stride, dcfg.data(), dcfg.size(), st)) return 1;
  if (h->maps_set) {
    std::vector<int> maps;
    for (size_t m = 0; m < h->host_map_cfg.size(); m++)
      if (h->host_map_cfg[m] == slot) maps.push_back((int)m);
    if (mg_rebuild_templates(h, maps, st)) return 1;
  }
  return 0;
}

int mg_register_config(mg_handle* h, const int32_t* cfg_blob, size_t cfg_words, int32_t* config_id) {
  if (!h || !cfg_blob || !config_id) return fail("mg_register_config: NULL argument");
  if (h->n_cfg_slots < 1) return fail("mg_register_config: call mg_set_config_slots first");
  std::vector<int32_t> dcfg;
  if (mg_task_stream(h, cfg_blob, cfg_words, "mg_register_config", dcfg)) return 1;
  if (mg_use_device(h)) return 1;
  const size_t stride = (size_t)h->k.dc_stride;
  if (h->reg_n == h->reg_cap) {  // grow the registry (rare, synchronous: steps in flight may be reading the old one)
    const int cap = h->reg_cap ? 2 * h->reg_cap : 16;
    int* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, (size_t)cap * stride * 4));
    HIPCHK(hipDeviceSynchronize());
    if (h->reg_n) HIPCHK(hipMemcpy(d, h->d_reg, (size_t)h->reg_n * stride * 4, hipMemcpyDeviceToDevice));
    if (h->d_reg) {
      for (auto& p : h->allocs)
        if (p == (void*)h->d_reg) p = d;
      HIPCHK(hipFree(h->d_reg));
    } else {
      h->allocs.push_back(d);
    }
    h->d_reg = d;
    h->reg_cap = cap;
  }
  HIPCHK(hipMemcpy(h->d_reg + (size_t)h->reg_n * stride, dcfg.data(), stride * 4, hipMemcpyHostToDevice));
  *config_id = h->reg_n++;
  return 0;
}

int mg_write_config_slots(mg_handle* h, int n, const int32_t* slots, const int32_t* config_ids, void* stream) {
  if (!h || (n > 0 && (!slots || !config_ids))) return fail("mg_write_config_slots: NULL argument");
  if (h->n_cfg_slots < 1) return fail("mg_write_config_slots: call mg_set_config_slots first");
  if (n <= 0) return 0;
  if (n > h->n_cfg_slots) return fail("mg_write_config_slots: more writes than slots");
  for (int i = 0; i < n; i++) {
    if (slots[i] < 0 || slots[i] >= h->n_cfg_slots) return fail("mg_write_config_slots: slot out of range");
    if (config_ids[i] < 0 || config_ids[i] >= h->reg_n) return fail("mg_write_config_slots: unknown configuration id (mg_register_config)");
  }
  if (mg_use_device(h)) return 1;
  hipStream_t st = (hipStream_t)stream;
  KArgs& k = h->k;
  if (!h->d_wlist && dalloc(h, &h->d_wlist, 2 * (size_t)h->n_cfg_slots)) return 1;
  // (slots, ids) go up through the pinned ring in pieces of one ring slot; every piece is one copy kernel
  const int piece = (int)std::max<size_t>(64, h->ring_words ? h->ring_words / 2 : (size_t)k.dc_stride / 2);
  std::vector<int32_t> both;
  for (int lo = 0; lo < n; lo += piece) {
    const int m = std::min(piece, n - lo);
    both.assign(slots + lo, slots + lo + m);
    both.insert(both.end(), config_ids + lo, config_ids + lo + m);
    int* d = h->d_wlist + 2 * (size_t)lo;
    if (mg_upload(h, d, both.data(), both.size(), st)) return 1;
    hipLaunchKernelGGL(mg_copy_config_kernel, dim3((unsigned)m), dim3(256), 0, st, const_cast<int*>(k.gdc), h->d_reg, d, d + m, k.dc_stride);
  }
  HIPCHK(hipGetLastError());
  if (h->maps_set) {
    std::vector<char> hit((size_t)h->n_cfg_slots, 0);
    for (int i = 0; i < n; i++) hit[(size_t)slots[i]] = 1;
    std::vector<int> maps;
    for (size_t m = 0; m < h->host_map_cfg.size(); m++)
      if (hit[(size_t)h->host_map_cfg[m]]) maps.push_back((int)m);
    if (mg_rebuild_templates(h, maps, st)) return 1;
  }
  return 0;
}

int mg_assign_map_configs(mg_handle* h, int n, const int32_t* map_ids, const int32_t* cfg_slots, void* stream) {
  if (!h || (n > 0 && (!map_ids || !cfg_slots))) return fail("mg_assign_map_configs: NULL argument");
  if (!h->maps_set || h->n_cfg_slots < 1 || !h->d_map_cfg) return fail("mg_assign_map_configs: call mg_set_config_slots and mg_set_maps first");
  std::vector<int> maps;
  for (int i = 0; i < n; i++) {
    if (map_ids[i] < 0 || map_ids[i] >= h->k.P) return fail("mg_assign_map_configs: map id out of range");
    if (cfg_slots[i] < 0 || cfg_slots[i] >= h->n_cfg_slots) return fail("mg_assign_map_configs: configuration slot out of range");
  }
  if (mg_use_device(h)) return 1;
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < n; i++) {
    h->host_map_cfg[(size_t)map_ids[i]] = cfg_slots[i];
    maps.push_back(map_ids[i]);
  }
  // (P ints: re-upload the table; stream-ordered after the steps already queued)
  if (mg_upload(h, h->d_map_cfg, h->host_map_cfg.data(), h->host_map_cfg.size(), st)) return 1;
  return mg_rebuild_templates(h, maps, st);
}

int mg_set_map_max_steps(mg_handle* h, const int32_t* max_steps) {
  if (!h) return fail("mg_set_map_max_steps: NULL handle");
  if (!h->maps_set) return fail("mg_set_map_max_steps: call mg_set_maps first");
  HIPCHK(hipSetDevice(h->device));
  if (!max_steps) {
    h->k.map_max_steps = nullptr;  // (kernel arguments are captured per launch: steps already queued keep theirs)
    return 0;
  }
  for (int m = 0; m < h->k.P; m++)
    if (max_steps[m] < 0) return fail("mg_set_map_max_steps: negative episode length");
  // one [P] array per pool, allocated on first use and reused by later calls (it lives and dies with the pool)
  if (!h->d_map_max_steps && dalloc(h, &h->d_map_max_steps, (size_t)h->k.P, true)) return 1;
  HIPCHK(hipDeviceSynchronize());  // steps in flight read the array
  HIPCHK(hipMemcpy(h->d_map_max_steps, max_steps, (size_t)h->k.P * 4, hipMemcpyHostToDevice));
  h->k.map_max_steps = h->d_map_max_steps;
  return 0;
}

int mg_map_dims(mg_handle* h, int map_id, int32_t* height, int32_t* width) {
  if (!h || !height || !width) return fail("mg_map_dims: NULL argument");
  if (!h->maps_set || map_id < 0 || map_id >= (int)h->host_dims.size()) return fail("mg_map_dims: map id out of range");
  *height = h->host_dims[(size_t)map_id].x;
  *width = h->host_dims[(size_t)map_id].y;
  return 0;
}

int mg_reset(mg_handle* h, const int32_t* env_ids, int n, const uint32_t* seeds, const int32_t* map_ids, void* stream) {
  if (!h) return fail("mg_reset: NULL handle");
  if (!h->maps_set) return fail("mg_reset: call mg_set_maps first");
  if (!seeds || !map_ids) return fail("mg_reset: seeds and map_ids are required");
  if (h->needs_reset && env_ids) return fail("mg_reset: the first reset after mg_set_maps must cover every environment (env_ids = NULL)");
  HIPCHK(hipSetDevice(h->device));
  if (!env_ids) n = h->E;
  if (n < 1 || n > h->E) return fail("mg_reset: bad environment count");
  for (int i = 0; i < n; i++) {
    if (env_ids && (env_ids[i] < 0 || env_ids[i] >= h->E)) return fail("mg_reset: environment id out of range");
    if (map_ids[i] < 0 || map_ids[i] >= h->k.P) return fail("mg_reset: map id out of range");
  }
  hipStream_t s = (hipStream_t)stream;
  int* d = h->d_list;
  HIPCHK(hipStreamSynchronize(s));
  if (env_ids) HIPCHK(hipMemcpy(d, env_ids, (size_t)n * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d + h->E, seeds, (size_t)n * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d + 2 * (size_t)h->E, map_ids, (size_t)n * 4, hipMemcpyHostToDevice));
  KArgs k = h->k;
  k.reset_envs = env_ids ? d : nullptr;
  k.reset_seeds = (const uint32_t*)(d + h->E);
  k.reset_maps = d + 2 * (size_t)h->E;
  mg_launch_reset(k, (unsigned)n, h->lds_bytes, h->big, s);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s));  // the scratch lists are reused by the next call
  if (!env_ids) h->needs_reset = false;
  return 0;
}

// The hot entry points launch on the handle's device whatever the caller's current device is; the switch is only
// made when it differs (hipGetDevice is a thread-local read).
static int mg_use_device(mg_handle* h) {
  int cur = -1;
  HIPCHK(hipGetDevice(&cur));
  if (cur != h->device) HIPCHK(hipSetDevice(h->device));
  return 0;
}

#define MG_PROFILE_RING 4096
static int mg_step_impl(mg_handle* h, const int32_t* actions, void* stream, bool static_scratch = false) {
  if (!h) return fail("mg_step: NULL handle");
  if (!h->maps_set || h->needs_reset) return fail("mg_step: call mg_set_maps and mg_reset first");
  if (int rc = mg_use_device(h)) return rc;
  hipStream_t s = (hipStream_t)stream;
  KArgs k = h->k;
  if (actions) k.actions = actions;
  if (static_scratch) {
    // same kernels, the hand-over buffers in this GPU's static array (the pointer kept by mg_scratch_init, never
    // the array's name).  Every word the kernels read there is written earlier in the same step: nothing to clear.
    if ((size_t)h->E * (size_t)h->cfg.A > MG_NOALLOC_MAX_AGENTS || h->E > MG_NOALLOC_MAX_ENVS)
      return fail("mg_step_noalloc: at most " + std::to_string(MG_NOALLOC_MAX_AGENTS) + " agents (envs x agents) and " +
                  std::to_string(MG_NOALLOC_MAX_ENVS) + " environments per handle; this handle has " +
                  std::to_string((size_t)h->E * (size_t)h->cfg.A) + " agents in " + std::to_string(h->E) + " environments");
    unsigned char* sp = g_scratch_ptr[h->device];
    if (!sp) return fail("mg_step_noalloc: static scratch of this device was not initialised");
    k.pend = sp;
    k.exec = (int*)(sp + MG_SCRATCH_PEND_BYTES);
    k.mode = (uint32_t*)(sp + MG_SCRATCH_PEND_BYTES + MG_SCRATCH_EXEC_BYTES);
  }
  size_t slot = 0;
  bool prof = h->profiling && h->ev_count < MG_PROFILE_RING && (h->prof_tick++ % (unsigned)h->prof_stride) == 0;
  const bool obs_only = prof && h->profiling == 2 && h->n_split <= 1;
  if (prof) {
    slot = h->ev_count;
    if (!obs_only) (void)hipEventRecord(h->ev[2 * slot], s);
  }
  // The three kernels of a step are latency bound, not throughput bound: splitting the environments into groups
  // that run on concurrent streams lets one group's observation kernel overlap another group's action kernels.
  const int G = std::max(1, std::min(h->n_split, h->E));
  if (G > 1) (void)hipEventRecord(h->ev_fork, s);
  for (int gi = 0; gi < G; gi++) {
    hipStream_t st = gi == 0 ? s : h->aux[gi - 1];
    if (gi > 0) (void)hipStreamWaitEvent(st, h->ev_fork, 0);
    KArgs kg = k;
    kg.env_base = (int)((long long)h->E * gi / G);
    const unsigned n = (unsigned)((long long)h->E * (gi + 1) / G) - (unsigned)kg.env_base;
    if (h->fused) {
      if (obs_only) (void)hipEventRecord(h->ev[2 * slot], s);
      kg.fl = (h->no_modifiers && h->fused_w1) ? h->fl1 : h->fl2;
      if (h->no_modifiers && h->fused_w1 && h->fused_w18) mg_launch_fused_nomod_w18(kg, n, h->lds_fused_w1, h->fused_dcl, st);
      else if (h->no_modifiers && h->fused_w1) mg_launch_fused_nomod_w1(kg, n, h->lds_fused_w1, h->fused_dcl, st);
      else if (h->no_modifiers && h->fused_o7) mg_launch_fused_nomod_o7(kg, n, h->lds_fused, h->fused_dcl, st);
      else if (h->no_modifiers) mg_launch_fused_nomod(kg, n, h->lds_fused, h->fused_dcl, st);
      else if (h->fused_o7) mg_launch_fused_o6(kg, n, h->lds_fused, h->fused_dcl, st);
      else mg_launch_fused(kg, n, h->lds_fused, h->fused_dcl, st);
    } else {
      if (k.fast) {
        mg_launch_act_fast(kg, n, st);
        mg_launch_uses(kg, n, h->lds_uses, h->dc_lds, st);
      } else {
        mg_launch_act(kg, n, h->lds_act, h->dc_lds, h->big, st);
        if (kg.clip_w) mg_launch_clipper(kg, n, st);
      }
      if (obs_only) (void)hipEventRecord(h->ev[2 * slot], s);
      mg_launch_obs(kg, n, h->lds_obs, h->dc_lds, h->big, st);
    }
    if (gi > 0) {
      (void)hipEventRecord(h->ev_join[gi - 1], st);
      (void)hipStreamWaitEvent(s, h->ev_join[gi - 1], 0);
    }
  }
  if (prof) {
    (void)hipEventRecord(h->ev[2 * slot + 1], s);
    h->ev_count++;
  }
  HIPCHK(hipGetLastError());
  return 0;
}
int mg_step(mg_handle* h, void* stream) { return mg_step_impl(h, nullptr, stream); }
int mg_step_actions(mg_handle* h, const int32_t* device_actions, void* stream) {
  if (!device_actions) return fail("mg_step_actions: NULL action buffer");
  return mg_step_impl(h, device_actions, stream);
}

int mg_step_groups(int n, mg_handle* const* hs, const int32_t* const* device_actions, void* const* streams) {
  if (n < 0 || (n > 0 && (!hs || !device_actions || !streams))) return fail("mg_step_groups: NULL argument");
  for (int i = 0; i < n; i++) {
    if (!device_actions[i]) return fail("mg_step_groups: NULL action buffer");
    if (int rc = mg_step_impl(hs[i], device_actions[i], streams[i])) return rc;
  }
  return 0;
}

int mg_step_noalloc(mg_handle* h, const int32_t* device_actions, void* stream) {
  return mg_step_impl(h, device_actions, stream, true);  // NULL actions = the handle's own action buffer
}

int mg_set_auto_reset(mg_handle* h, int enabled, int map_stride) {
  if (!h) return fail("mg_set_auto_reset: NULL handle");
  if (map_stride < 0) return fail("mg_set_auto_reset: map_stride must be >= 0");
  h->k.auto_reset = enabled ? 1 : 0;
  h->k.map_stride = h->maps_set && h->k.P > 0 ? map_stride % h->k.P : map_stride;
  return 0;
}

int mg_set_early_reset(mg_handle* h, const uint32_t* steps) {
  if (!h) return fail("mg_set_early_reset: NULL handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipDeviceSynchronize());
  std::vector<DEnv> envs((size_t)h->E);
  HIPCHK(hipMemcpy(envs.data(), h->k.env, envs.size() * sizeof(DEnv), hipMemcpyDeviceToHost));
  for (int e = 0; e < h->E; e++) envs[(size_t)e].early_reset = steps ? steps[e] : 0;
  HIPCHK(hipMemcpy(h->k.env, envs.data(), envs.size() * sizeof(DEnv), hipMemcpyHostToDevice));
  return 0;
}

int mg_get_buffers(mg_handle* h, mg_buffers* out) {
  if (!h || !out) return fail("mg_get_buffers: NULL argument");
  static_assert(sizeof(int) == 4, "int32");
  out->observations = h->k.obs;
  out->rewards = h->k.rewards;
  out->terminals = h->k.term;
  out->truncations = h->k.trunc;
  out->actions = (int32_t*)h->k.actions;
  out->masks = h->masks;
  out->episode_rewards = h->k.ep_rewards;
  out->action_success = h->k.success;
  out->num_envs = h->E;
  out->num_agents = h->cfg.A;
  out->num_tokens = h->cfg.T;
  out->num_actions = (int32_t)h->cfg.actions.size();
  return 0;
}

int mg_end_episode(mg_handle* h, int env, void* stream) {
  if (!h) return fail("mg_end_episode: NULL handle");
  if (env >= h->E) return fail("mg_end_episode: environment id out of range");
  if (int rc = mg_use_device(h)) return rc;
  int N = h->E * h->cfg.A;
  hipLaunchKernelGGL(mg_end_episode_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->k, env);
  HIPCHK(hipGetLastError());
  return 0;
}

int mg_stat_slots(mg_handle* h, int* n_agent_slots, int* n_game_slots) {
  if (!h) return fail("mg_stat_slots: NULL handle");
  if (n_agent_slots) *n_agent_slots = h->k.SA;
  if (n_game_slots) *n_game_slots = h->k.SG;
  return 0;
}

int mg_episode_stats(mg_handle* h, int env, int last_finished, float* agent_values, uint8_t* agent_touched,
                     float* game_values, uint8_t* game_touched, void* stream) {
  if (!h) return fail("mg_episode_stats: NULL handle");
  if (env < 0 || env >= h->E) return fail("mg_episode_stats: environment id out of range");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  const KArgs& k = h->k;
  size_t na = (size_t)k.A * k.SA, nta = (size_t)k.A * k.TW;
  std::vector<uint32_t> at(nta), gt((size_t)k.GW);
  HIPCHK(hipMemcpy(agent_values, (last_finished ? k.fin_astats : k.astats) + (size_t)env * na, na * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(at.data(), (last_finished ? k.fin_atouched : k.atouched) + (size_t)env * nta, nta * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(game_values, (last_finished ? k.fin_gstats : k.gstats) + (size_t)env * k.SG, (size_t)k.SG * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(gt.data(), (last_finished ? k.fin_gtouched : k.gtouched) + (size_t)env * k.GW, (size_t)k.GW * 4, hipMemcpyDeviceToHost));
  if (!last_finished) {  // merge the record-resident counters (see DAgent)
    std::vector<DAgent> ag((size_t)k.A);
    HIPCHK(hipMemcpy(ag.data(), k.agents + (size_t)env * k.A, ag.size() * sizeof(DAgent), hipMemcpyDeviceToHost));
    const MgAgentSlots AS = h->cfg.aslots();
    for (int a = 0; a < k.A; a++) {
      float* row = agent_values + (size_t)a * k.SA;
      uint32_t fa = 0;
      for (int q = 0; q < 3; q++) {
        row[AS.success(q)] = (float)ag[(size_t)a].succ[q];
        row[AS.failed(q)] = (float)ag[(size_t)a].fail[q];
        fa += ag[(size_t)a].fail[q];
      }
      row[AS.failed_any()] = (float)fa;
      row[AS.max_steps_without_motion()] = (float)ag[(size_t)a].max_swm;
    }
  }
  // a key exists once touched: explicit bit (set-type slots) or a non-zero value (counters only grow from 0)
  for (int a = 0; a < k.A; a++)
    for (int s = 0; s < k.SA; s++) {
      uint32_t bit = (at[(size_t)a * k.TW + (s >> 5)] >> (s & 31)) & 1u;
      agent_touched[(size_t)a * k.SA + s] = (bit || agent_values[(size_t)a * k.SA + s] != 0.f) ? 1 : 0;
    }
  for (int s = 0; s < k.SG; s++) game_touched[s] = (gt[(size_t)(s >> 5)] >> (s & 31)) & 1u;
  return 0;
}

int mg_env_info(mg_handle* h, int env, uint32_t* current_step, int32_t* map_id, uint32_t* seed, uint32_t* episodes,
                void* stream) {
  if (!h) return fail("mg_env_info: NULL handle");
  if (env < 0 || env >= h->E) return fail("mg_env_info: environment id out of range");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  DEnv d;
  HIPCHK(hipMemcpy(&d, h->k.env + env, sizeof(d), hipMemcpyDeviceToHost));
  if (d.error) return fail("environment " + std::to_string(env) + " flagged an internal error: " + mg_error_text(d.error));
  if (current_step) *current_step = d.step;
  if (map_id) *map_id = (int32_t)d.map_id;
  if (seed) *seed = d.seed;
  if (episodes) *episodes = d.episodes;
  return 0;
}

int mg_debug_inject_error(mg_handle* h, int env, uint32_t bits) {
  if (!h) return fail("mg_debug_inject_error: NULL handle");
  if (env < 0 || env >= h->E) return fail("mg_debug_inject_error: environment id out of range");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(&h->k.env[env].error, &bits, 4, hipMemcpyHostToDevice));
  return 0;
}

const char* mg_step_path(mg_handle* h) {
  if (!h || !h->maps_set) return "";
  if (h->fused)
    return h->no_modifiers ? (h->fused_w1 ? (h->fused_w18 ? "mg_step_fused_nomod_w18_kernel" : "mg_step_fused_nomod_w1_kernel") : h->fused_o7 ? "mg_step_fused_nomod_o7_kernel" : "mg_step_fused_nomod_kernel")
                           : (h->fused_o7 ? "mg_step_fused_o6_kernel" : "mg_step_fused_kernel");
  if (h->k.fast) return "mg_act_fast_kernel + mg_uses_kernel + mg_obs_kernel";
  return "mg_act_kernel + mg_obs_kernel";
}

int mg_status_buffer(mg_handle* h, uint32_t** device_ptr, int32_t* n) {
  if (!h || !device_ptr || !n) return fail("mg_status_buffer: NULL argument");
  *device_ptr = h->k.status;
  *n = 4;
  return 0;
}

int mg_check_errors(mg_handle* h, void* stream) {
  if (!h) return fail("mg_check_errors: NULL handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  uint32_t st[4] = {0, 0, 0, 0};
  HIPCHK(hipMemcpy(st, h->k.status, sizeof(st), hipMemcpyDeviceToHost));
  if (st[0] == 0) return 0;
  // name the first offending environment
  std::vector<DEnv> envs((size_t)h->E);
  HIPCHK(hipMemcpy(envs.data(), h->k.env, envs.size() * sizeof(DEnv), hipMemcpyDeviceToHost));
  int first = -1, count = 0;
  for (int e = 0; e < h->E; e++)
    if (envs[(size_t)e].error) {
      if (first < 0) first = e;
      count++;
    }
  return fail("mg_check_errors: " + std::to_string(count) + " environment(s) left reference parity (first: env " +
              std::to_string(first) + "): " + mg_error_text(st[0]));
}

int mg_agent_state(mg_handle* h, int env, int32_t* out, void* stream) {
  if (!h) return fail("mg_agent_state: NULL handle");
  if (env < 0 || env >= h->E) return fail("mg_agent_state: environment id out of range");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  const KArgs& k = h->k;
  std::vector<DAgent> ag((size_t)k.A);
  DEnv d;
  HIPCHK(hipMemcpy(ag.data(), k.agents + (size_t)env * k.A, ag.size() * sizeof(DAgent), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&d, k.env + env, sizeof(d), hipMemcpyDeviceToHost));
  if (mg_sync_lists(h, d.map_id)) return 1;
  int R = k.R, stride = 9 + 2 * R;
  for (int i = 0; i < k.A; i++) {
    const DAgent& a = ag[(size_t)i];
    int32_t* o = out + (size_t)i * stride;
    o[0] = a.r; o[1] = a.c; o[2] = a.prev_r; o[3] = a.prev_c; o[4] = a.frozen; o[5] = a.vibe; o[6] = (int32_t)a.swm;
    o[7] = h->host_spawn[(size_t)d.map_id][(size_t)i].w;
    o[8] = a.n_items;
    for (int q = 0; q < 2 * R; q++) o[9 + q] = -1;
    for (int q = 0; q < a.n_items && q < R; q++) {
      o[9 + 2 * q] = a.order[q];
      o[9 + 2 * q + 1] = a.amt[a.order[q]];
    }
  }
  return 0;
}

int mg_agent_stat_reward(mg_handle* h, int env, float* out, void* stream) {
  if (!h || !out) return fail("mg_agent_stat_reward: NULL argument");
  if (env < 0 || env >= h->E) return fail("mg_agent_stat_reward: environment id out of range");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  const KArgs& k = h->k;
  std::vector<DAgent> ag((size_t)k.A);
  HIPCHK(hipMemcpy(ag.data(), k.agents + (size_t)env * k.A, ag.size() * sizeof(DAgent), hipMemcpyDeviceToHost));
  for (int i = 0; i < k.A; i++) out[i] = ag[(size_t)i].cur_reward;
  return 0;
}

int mg_chest_state(mg_handle* h, int env, int32_t* out, int max_chests, int32_t* n_chests, void* stream) {
  if (!h || !n_chests) return fail("mg_chest_state: NULL argument");
  if (env < 0 || env >= h->E) return fail("mg_chest_state: environment id out of range");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  const KArgs& k = h->k;
  DEnv d;
  HIPCHK(hipMemcpy(&d, k.env + env, sizeof(d), hipMemcpyDeviceToHost));
  if (mg_sync_lists(h, d.map_id)) return 1;
  const auto& lst = h->host_chest[(size_t)d.map_id];
  *n_chests = (int32_t)lst.size();
  if (!out || lst.empty()) return 0;
  std::vector<DAgent> ch((size_t)k.NCmax);
  HIPCHK(hipMemcpy(ch.data(), k.chests + (size_t)env * k.NCmax, ch.size() * sizeof(DAgent), hipMemcpyDeviceToHost));
  const int R = k.R, stride = 5 + 2 * R;
  for (size_t i = 0; i < lst.size() && (int)i < max_chests; i++) {
    const DAgent& a = ch[i];
    int32_t* o = out + i * (size_t)stride;
    o[0] = lst[i].x; o[1] = lst[i].y; o[2] = lst[i].z; o[3] = lst[i].w; o[4] = a.n_items;
    for (int q = 0; q < 2 * R; q++) o[5 + q] = -1;
    for (int q = 0; q < a.n_items && q < R; q++) {
      o[5 + 2 * q] = a.order[q];
      o[5 + 2 * q + 1] = a.amt[a.order[q]];
    }
  }
  return 0;
}

int mg_num_assemblers(mg_handle* h, int env) {
  if (!h || env < 0 || env >= h->E) return -1;
  DEnv d;
  if (hipMemcpy(&d, h->k.env + env, sizeof(d), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (mg_sync_lists(h, d.map_id)) return -1;
  return (int)h->host_asm[(size_t)d.map_id].size();
}

int mg_assembler_state(mg_handle* h, int env, int32_t* out, void* stream) {
  if (!h) return fail("mg_assembler_state: NULL handle");
  if (env < 0 || env >= h->E) return fail("mg_assembler_state: environment id out of range");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  const KArgs& k = h->k;
  DEnv d;
  HIPCHK(hipMemcpy(&d, k.env + env, sizeof(d), hipMemcpyDeviceToHost));
  if (mg_sync_lists(h, d.map_id)) return 1;
  const auto& lst = h->host_asm[(size_t)d.map_id];
  std::vector<DAsm> as((size_t)k.NAmax);
  HIPCHK(hipMemcpy(as.data(), k.asms + (size_t)env * k.NAmax, as.size() * sizeof(DAsm), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < lst.size(); i++) {
    int32_t* o = out + i * 8;
    o[0] = lst[i].x; o[1] = lst[i].y; o[2] = lst[i].z; o[3] = lst[i].w; o[4] = (int32_t)as[i].clipped;
    o[5] = (int32_t)as[i].cooldown_end; o[6] = (int32_t)as[i].cooldown_duration; o[7] = (int32_t)as[i].uses_count;
  }
  return 0;
}

int mg_clipper_state(mg_handle* h, int env, uint32_t* weights, int32_t* length_scale, void* stream) {
  if (!h || !weights || !length_scale) return fail("mg_clipper_state: NULL argument");
  if (env < 0 || env >= h->E) return fail("mg_clipper_state: environment id out of range");
  if (!h->k.clip_w) return fail("mg_clipper_state: the configuration has no clipper (or mg_set_maps has not run)");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  const KArgs& k = h->k;
  uint2 st;
  HIPCHK(hipMemcpy(&st, k.clip_state + env, sizeof(st), hipMemcpyDeviceToHost));
  DEnv d;
  HIPCHK(hipMemcpy(&d, k.env + env, sizeof(d), hipMemcpyDeviceToHost));
  int n = 0;
  HIPCHK(hipMemcpy(&n, k.map_nasm + d.map_id, 4, hipMemcpyDeviceToHost));
  if (n > 0) HIPCHK(hipMemcpy(weights, k.clip_w + (size_t)env * k.NAmax, (size_t)n * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) weights[i] &= 0x7FFFFFFFu;  // (bit 31 is the kernel's own bookkeeping)
  *length_scale = (int32_t)st.y;
  return 0;
}

int mg_set_inventory(mg_handle* h, int env, int agent, int n, const int32_t* items, const int32_t* amounts, void* stream) {
  if (!h) return fail("mg_set_inventory: NULL handle");
  if (env < 0 || env >= h->E) return fail("mg_set_inventory: environment id out of range");
  if (agent < 0 || agent >= h->cfg.A) return 0;  // reference ignores unknown agents (mettagrid_c.cpp:961)
  if (n < 0 || n > MGD_MAX_R) return fail("mg_set_inventory: too many items");
  for (int i = 0; i < n; i++)
    if (items[i] < 0 || items[i] >= h->cfg.R) return fail("mg_set_inventory: item id out of range");
  HIPCHK(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipStreamSynchronize(s));
  if (n > 0) {
    HIPCHK(hipMemcpy(h->d_inv, items, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_inv + MGD_MAX_R, amounts, (size_t)n * 4, hipMemcpyHostToDevice));
  }
  size_t lds = mg_align16(sizeof(DAgent)) + 16;
  hipLaunchKernelGGL(mg_set_inventory_kernel, dim3(1), dim3(64), lds, s, h->k, env, agent, n, h->d_inv, h->d_inv + MGD_MAX_R);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}

int mg_finished_summary(mg_handle* h, double* out, int clear, void* stream) {
  if (!h || !out) return fail("mg_finished_summary: NULL argument");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  size_t n = (size_t)(3 + h->k.SG);
  HIPCHK(hipMemcpy(out, h->k.summary, n * 8, hipMemcpyDeviceToHost));
  if (clear) HIPCHK(hipMemset(h->k.summary, 0, n * 8));
  return 0;
}

int mg_summary_buffer(mg_handle* h, double** device_ptr, int32_t* n) {
  if (!h || !device_ptr || !n) return fail("mg_summary_buffer: NULL argument");
  *device_ptr = h->k.summary;
  *n = 3 + h->k.SG;
  return 0;
}

int mg_set_profiling_stride(mg_handle* h, int stride) {
  if (!h) return fail("mg_set_profiling_stride: NULL handle");
  if (stride < 1) return fail("mg_set_profiling_stride: stride must be >= 1");
  h->prof_stride = stride;
  h->prof_tick = 0;
  return 0;
}
int mg_set_profiling(mg_handle* h, int enabled) {
  if (!h) return fail("mg_set_profiling: NULL handle");
  HIPCHK(hipSetDevice(h->device));
  if (enabled && h->ev.empty()) {
    h->ev.resize(2 * MG_PROFILE_RING);
    for (auto& e : h->ev) HIPCHK(hipEventCreate(&e));
  }
  h->profiling = enabled == 2 ? 2 : (enabled != 0 ? 1 : 0);
  h->ev_count = 0;
  h->prof_tick = 0;
  return 0;
}

int mg_profile_read(mg_handle* h, float* ms, int max_steps, int* n_steps) {
  if (!h || !ms || !n_steps) return fail("mg_profile_read: NULL argument");
  HIPCHK(hipSetDevice(h->device));
  int n = (int)std::min<size_t>(h->ev_count, (size_t)std::max(max_steps, 0));
  for (int i = 0; i < n; i++) {
    HIPCHK(hipEventSynchronize(h->ev[2 * (size_t)i + 1]));
    HIPCHK(hipEventElapsedTime(&ms[i], h->ev[2 * (size_t)i], h->ev[2 * (size_t)i + 1]));
  }
  *n_steps = n;
  h->ev_count = 0;
  return 0;
}

int mg_env_table(mg_handle* h, uint32_t* out, void* stream) {
  if (!h || !out) return fail("mg_env_table: NULL argument");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  static_assert(sizeof(DEnv) == 32, "DEnv is 8 words");
  HIPCHK(hipMemcpy(out, h->k.env, (size_t)h->E * sizeof(DEnv), hipMemcpyDeviceToHost));
  return 0;
}

int mg_finished_rewards(mg_handle* h, int env, float* rewards, uint32_t* steps, void* stream) {
  if (!h || !rewards) return fail("mg_finished_rewards: NULL argument");
  if (env < 0 || env >= h->E) return fail("mg_finished_rewards: environment id out of range");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  HIPCHK(hipMemcpy(rewards, h->k.fin_ep_rewards + (size_t)env * h->k.A, (size_t)h->k.A * 4, hipMemcpyDeviceToHost));
  if (steps) HIPCHK(hipMemcpy(steps, h->k.fin_steps + env, 4, hipMemcpyDeviceToHost));
  return 0;
}

// The last finished episode of several environments at once: one gather launch and one device -> host copy per chunk of
// environments instead of six blocking copies per environment (mg_episode_stats(last_finished) + mg_finished_rewards).
#define MG_FIN_CHUNK 256
int mg_finished_ring_capacity(mg_handle* h) { return h ? (int)(h->k.fin_ring_mask + 1u) : 0; }

int mg_finished_ring(mg_handle* h, uint32_t from_count, uint32_t to_count, int32_t* env_ids, int32_t* map_ids, void* stream) {
  if (!h || !env_ids || !map_ids) return fail("mg_finished_ring: NULL argument");
  const uint32_t n = to_count - from_count, cap = h->k.fin_ring_mask + 1u;
  if (n == 0) return 0;
  if (n > cap) return fail("mg_finished_ring: " + std::to_string(n) + " episodes finished since the last read, the ring holds " +
                           std::to_string(cap) + " (the host fell behind)");
  if (mg_use_device(h)) return 1;
  std::vector<uint2> buf((size_t)n);
  const uint32_t lo = from_count & h->k.fin_ring_mask, first = std::min(n, cap - lo);
  // (entries up to a counter value the host has SEEN were written by kernels that finished before that value was copied)
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemcpyAsync(buf.data(), h->k.fin_ring + lo, (size_t)first * sizeof(uint2), hipMemcpyDeviceToHost, st));
  if (first < n) HIPCHK(hipMemcpyAsync(buf.data() + first, h->k.fin_ring, (size_t)(n - first) * sizeof(uint2), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (uint32_t i = 0; i < n; i++) {
    if ((buf[i].y >> 20) != ((from_count + i) & 0xFFFu))
      return fail("mg_finished_ring: entry " + std::to_string(from_count + i) + " was overwritten while the range was read (the device "
                  "ran more than a ring ahead of the host): read again from a later counter value");
    env_ids[i] = (int32_t)buf[i].x;
    map_ids[i] = (int32_t)(buf[i].y & 0xFFFFFu);
  }
  return 0;
}

int mg_finished_batch(mg_handle* h, int n, const int32_t* env_ids, float* agent_values, uint8_t* agent_touched, float* game_values,
                      uint8_t* game_touched, float* rewards, uint32_t* steps, void* stream) {
  if (!h || (n > 0 && (!env_ids || !agent_values || !agent_touched || !game_values || !game_touched || !rewards || !steps)))
    return fail("mg_finished_batch: NULL argument");
  for (int i = 0; i < n; i++)
    if (env_ids[i] < 0 || env_ids[i] >= h->E) return fail("mg_finished_batch: environment id out of range");
  if (n <= 0) return 0;
  if (mg_use_device(h)) return 1;
  const KArgs& k = h->k;
  hipStream_t st = (hipStream_t)stream;
  const size_t n0 = (size_t)k.A * k.SA, n1 = (size_t)k.A * k.TW, n2 = (size_t)k.SG, n3 = (size_t)k.GW, n4 = (size_t)k.A;
  const size_t row = n0 + n1 + n2 + n3 + n4 + 1;
  if (!h->d_fin_stage) {
    if (dalloc(h, &h->d_fin_stage, row * MG_FIN_CHUNK)) return 1;
    if (dalloc(h, &h->d_fin_envs, (size_t)MG_FIN_CHUNK)) return 1;
  }
  std::vector<uint32_t> host(row * (size_t)std::min(n, MG_FIN_CHUNK));
  for (int c0 = 0; c0 < n; c0 += MG_FIN_CHUNK) {
    const int m = std::min(MG_FIN_CHUNK, n - c0);
    HIPCHK(hipMemcpyAsync(h->d_fin_envs, env_ids + c0, (size_t)m * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(mg_gather_finished_kernel, dim3((unsigned)m), dim3(256), 0, st, k, h->d_fin_envs, h->d_fin_stage, (int)row);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host.data(), h->d_fin_stage, row * (size_t)m * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < m; i++) {
      const uint32_t* src = host.data() + row * (size_t)i;
      const size_t o = (size_t)(c0 + i);
      float* av = agent_values + o * n0;
      uint8_t* at = agent_touched + o * n0;
      memcpy(av, src, n0 * 4);
      const uint32_t* tb = src + n0;
      // a key exists once touched: explicit bit (set-type slots) or a non-zero value (counters only grow from 0)
      for (int a = 0; a < k.A; a++)
        for (int s = 0; s < k.SA; s++) {
          const uint32_t bit = (tb[(size_t)a * k.TW + (s >> 5)] >> (s & 31)) & 1u;
          at[(size_t)a * k.SA + s] = (bit || av[(size_t)a * k.SA + s] != 0.f) ? 1 : 0;
        }
      memcpy(game_values + o * n2, src + n0 + n1, n2 * 4);
      const uint32_t* gb = src + n0 + n1 + n2;
      for (int s = 0; s < k.SG; s++) game_touched[o * n2 + (size_t)s] = (gb[(size_t)(s >> 5)] >> (s & 31)) & 1u;
      memcpy(rewards + o * n4, src + n0 + n1 + n2 + n3, n4 * 4);
      steps[o] = src[n0 + n1 + n2 + n3 + n4];
    }
  }
  return 0;
}

// Diagnostic: per-environment cycle stamps of the observation kernel's phases (s_memtime, 100 MHz units on gfx950).
int mg_debug_stamps(mg_handle* h, unsigned long long* out, int enable) {
  if (!h) return fail("mg_debug_stamps: NULL handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipDeviceSynchronize());
  if (enable && !h->k.dbg) {
    if (dalloc(h, &h->k.dbg, (size_t)h->E * 16)) return 1;
  }
  if (out && h->k.dbg) HIPCHK(hipMemcpy(out, h->k.dbg, (size_t)h->E * 16 * 8, hipMemcpyDeviceToHost));
  return 0;
}

int mg_destroy(mg_handle* h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (void* p : h->allocs) (void)hipFree(p);
  for (void* p : h->map_allocs) (void)hipFree(p);
  for (auto& e : h->ev) (void)hipEventDestroy(e);
  for (auto& e : h->ring_ev) (void)hipEventDestroy(e);
  if (h->ring) (void)hipHostFree(h->ring);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  for (int i = 0; i < 7; i++) {
    if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    if (h->aux[i]) (void)hipStreamDestroy(h->aux[i]);
  }
  delete h;
  return 0;
}

}  // extern "C"

