// mettagrid_hip.hip -- kernels + C-ABI of the MI355X-native batched MettaGrid step path (gfx950).
//
// Replaces, for the environment-step path, the reference's pybind11 core `MettaGrid`
// (packages/mettagrid/cpp/bindings/mettagrid_c.cpp).  Layout and kernel design: DESIGN.md.
//
//   mg_act_kernel   one wave per environment: stage the static map tile (shared, L2-resident pool) and the
//                   agent/assembler records in LDS; resolve the actions in the reference's shuffled sequential
//                   order; flush stat deltas; write the records back.  Restarts finished episodes first.
//   mg_obs_kernel   one workgroup (4 waves) per environment: each wave encodes whole agents' token rows into an LDS
//                   row buffer (wave prefix-sum compaction) and streams the 0xFF-padded 600-byte rows to HBM with
//                   coalesced dword stores; then rewards, termination flags, token stats.
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
#include "mg_device_cfg.h"
#include "mg_sim.h"

#define MG_THREADS 256
#ifndef MG_OBS_WAVES
#define MG_OBS_WAVES 8
#endif
#define MG_WIN_MARGIN 32
#define MG_STAMP(slot)                                                     \
  do {                                                                    \
    if (a.dbg && threadIdx.x == 0) a.dbg[((size_t)blockIdx.x + a.env_base) * 16 + (slot)] = clock64(); \
  } while (0)
#define MG_DC_LDS_MAX_WORDS 6144  // 24 KiB

struct KArgs {
  const int* dc;
  int dc_obs_words;  // prefix of the stream the observation kernel stages
  int dc_words;  // size of the device config stream; staged in LDS when it fits (MG_DC_LDS_MAX_WORDS)
  const uint16_t* map_cells;  // [P][HW]
  const int4* map_spawn;      // [P][A]  (r, c, group, obj_id)
  const int4* map_asm;        // [P][NAmax] (r, c, cfg, obj_id)
  const int* map_nasm;        // [P]
  const int4* map_chest;      // [P][NCmax] (r, c, cfg, obj_id)
  const int* map_nchest;      // [P]
  DAgent* chests;             // [E][NCmax] chest records (inventory holders, same layout as agents)
  const DAgent* t_chests;     // [P][NCmax] episode-start templates
  int NCmax, TSC;             // chest slots per map; token-list stride of a chest
  const float* map_objcount;  // [P][n_cells]
  uint16_t* big_grid;         // [max(E, P)][HW] per-env grid in HBM for maps that do not fit LDS (BIG kernels), else null
  const int2* map_dims;       // [P] (height, width) of each map; H, W below are the current environment's
  int P, H, W, HW, NAmax, n_cells;  // HW = cell stride of one map in the pool (the largest map, even)
  int NW, NOBJ;            // wall types, objects per env (NW + A + NAmax)
  int CB;                  // bits needed for the largest per-object token count
  int TSW, TSA, TSS, HEAP; // token-list strides of walls / agents / assemblers; heap size in tokens
  DEnv* env;
  uint32_t* mt;  // [E][624]
  DAgent* agents;
  DAsm* asms;
  float* astats;       // [E][A][SA]
  uint32_t* atouched;  // [E][A][TW]
  float* gstats;       // [E][SG]
  uint32_t* gtouched;  // [E][GW]
  float* fin_astats;
  uint32_t* fin_atouched;
  float* fin_gstats;
  uint32_t* fin_gtouched;
  float* fin_ep_rewards;  // [E][A]
  uint32_t* fin_steps;    // [E]
  double* summary;        // [3 + SG]
  uint32_t* status;       // [4]: OR of every env's error bits | finished episodes (count) | 2 spare words
  uint8_t* obs;
  float* rewards;
  uint8_t* term;
  uint8_t* trunc;
  const int* actions;
  float* ep_rewards;
  uint8_t* success;
  int* exec;  // [E][A] executed action of this step (last_action token)
  uint32_t* mode;  // [E] fast path hand-off: 0 = step done, 1 | n << 8 = n pending assembler uses, 2 = general path
  uint8_t* pend;   // [E][A][32] pending assembler uses (fast path -> general kernel)
  int fast;        // fast (wave-parallel) act kernel in use
  // Episode-start templates, one per map (built once by mg_set_maps with the reset kernel): restarting an
  // environment inside a step is a handful of coalesced copies.
  const DAgent* t_agents;     // [P][A]
  const DAsm* t_asms;         // [P][NAmax]
  const float* t_astats;      // [P][A][SA]
  const uint32_t* t_atouched; // [P][A][TW]
  const float* t_gstats;      // [P][SG]
  const uint32_t* t_gtouched; // [P][GW]
  uint32_t* mt_init;          // [E][624] std::mt19937(seed) state of each env's seed
  int env_base;               // first environment of this launch (a step may be split over concurrent streams)
  int diag;                   // diagnostic ablation bits (MG_DIAG env var): 1 = no obs stores, 8 = serial shuffle redraw, 16 / 32 / 64 = skip encode / token lists / rewards in the obs kernel
  unsigned long long* dbg;    // optional [E][16] cycle stamps (diagnostic builds of the phases; null = off)
  int E, A, T, SA, SG, R, NG, TW, GW, NWIN, ROWB;
  int auto_reset, map_stride;
  const int* reset_envs;  // reset kernel only
  const uint32_t* reset_seeds;
  const int* reset_maps;
};

struct Lds {
  int* dc;
  int* tmp;  // [8][96] scratch ints: one slot per concurrently acting lane
  uint16_t* grid;
  DAgent* ag;
  DAgent* ch;       // chest records (directly after the agent records)
  int* gsetv;       // [16] pending chest.<r>.amount values; gsetm[0] = which
  uint32_t* gsetm;
  DAsm* as;
  int* gd;
  uint32_t* dirty_amt;
  uint32_t* dirty_misc;
  uint16_t* order;
  int* exec;
  int* tok_att;
  uint32_t* win;
  uint32_t* mtnext;
  float* tokacc;  // [3] tokens_written / dropped / free_space accumulators
  int* flags;     // [4]
  uint8_t* rows;
  uint32_t* ocnt;   // [-1 .. NOBJ) per object this step: heap offset | token count << 16; ocnt[-1] = 0 (the empty cell)
  uint16_t* heap;   // token lists (feature | value << 8): [NW][TSW] walls, [A][TSA] agents, [NA][TSS] assemblers
  int4* ainfo;      // [NA] static assembler info of the map (r, c, cfg, object id)
  uint32_t* pat;    // [P] (dr+8) | (dc+8) << 8 | location byte << 16
};

__host__ __device__ inline size_t mg_align16(size_t x) { return (x + 15) & ~(size_t)15; }

// Maps of one pool may differ in size (navigation suite): the kernel's copy of the arguments takes the dimensions of
// the map its environment plays.
__device__ __forceinline__ void mg_use_map(KArgs& a, uint32_t map_id) {
  const int2 d = a.map_dims[map_id];
  a.H = d.x;
  a.W = d.y;
}

// FILL = false: sizes only (host).  A compile-time switch rather than `if (l)`: a null test on the kernel's Lds
// object can survive to the back end and then keeps the whole struct in scratch memory (every LDS access turns flat).
template <bool FILL>
__host__ __device__ inline size_t mg_carve(unsigned char* base, Lds* l, int HW, int A, int NA, int SA, int SG, int NWIN,
                                           int ROWB, int nwaves, int dcw, int NOBJ, int HEAP, int NP, int a_rec = -1,
                                           int NC = 0) {
  if (a_rec < 0) a_rec = A;  // agent records staged in LDS (0: the kernel works on the records in HBM)
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = mg_align16(o + bytes);
    return at;
  };
  size_t dco = take((size_t)dcw * 4);
  size_t tmpo = take(8 * 96 * 4);
  if (FILL) {
    l->dc = (int*)(base + dco);
    l->tmp = (int*)(base + tmpo);
  }
  size_t g = take((size_t)HW * 2), ag = take((size_t)(a_rec + (a_rec ? NC : 0)) * sizeof(DAgent)), as = take((size_t)(NA > 0 ? NA : 1) * sizeof(DAsm));
  size_t gsv = take(16 * 4 + 16);
  size_t gd = take((size_t)SG * 4), da = take((size_t)A * 4), dm = take((size_t)A * 4);
  size_t ord = take((size_t)A * 2), ex = take((size_t)A * 4), ta = take((size_t)A * 4), win = take((size_t)NWIN * 4);
  size_t mtn = take(624 * 4), tk = take(16), fl = take(16), rows = take((size_t)ROWB * nwaves);
  size_t oc = take(nwaves ? (size_t)(NOBJ + 1) * 4 : 0), hp = take(nwaves ? (size_t)HEAP * 2 : 0), pt = take(nwaves ? (size_t)NP * 4 : 0);
  size_t ai = take(nwaves ? (size_t)(NA > 0 ? NA : 1) * 16 : 0);
  if (FILL) {
    l->ocnt = (uint32_t*)(base + oc) + 1; l->heap = (uint16_t*)(base + hp); l->pat = (uint32_t*)(base + pt);
    l->ainfo = (int4*)(base + ai);
    l->grid = (uint16_t*)(base + g); l->ag = (DAgent*)(base + ag); l->as = (DAsm*)(base + as);
    l->ch = l->ag + a_rec; l->gsetv = (int*)(base + gsv); l->gsetm = (uint32_t*)(base + gsv + 64);
    l->gd = (int*)(base + gd); l->dirty_amt = (uint32_t*)(base + da); l->dirty_misc = (uint32_t*)(base + dm);
    l->order = (uint16_t*)(base + ord); l->exec = (int*)(base + ex); l->tok_att = (int*)(base + ta);
    l->win = (uint32_t*)(base + win); l->mtnext = (uint32_t*)(base + mtn); l->tokacc = (float*)(base + tk);
    l->flags = (int*)(base + fl); l->rows = (uint8_t*)(base + rows);
  }
  return o;
}

// LDS of the observation kernel: grid, agent/assembler records, per-wave row buffers.
template <bool FILL>
__host__ __device__ inline size_t mg_carve_obs(unsigned char* base, Lds* l, int HW, int A, int NA, int ROWB, int nwaves,
                                               int dcw, int NOBJ, int HEAP, int NP, int a_rec = -1, int NC = 0) {
  if (a_rec < 0) a_rec = A;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = mg_align16(o + bytes);
    return at;
  };
  size_t dco = take((size_t)dcw * 4);
  if (FILL) l->dc = (int*)(base + dco);
  size_t g = take((size_t)HW * 2), ag = take((size_t)(a_rec + (a_rec ? NC : 0)) * sizeof(DAgent)), as = take((size_t)(NA > 0 ? NA : 1) * sizeof(DAsm));
  size_t ex = take((size_t)A * 4), ta = take((size_t)A * 4), tk = take(16), rows = take((size_t)ROWB * nwaves);
  size_t oc = take((size_t)(NOBJ + 1) * 4), hp = take((size_t)HEAP * 2), pt = take((size_t)NP * 4);
  // the static assembler info is only read while the token lists are built, the row buffers only afterwards
  size_t ai = rows;
  if ((size_t)(NA > 0 ? NA : 1) * 16 > (size_t)ROWB * nwaves) ai = take((size_t)(NA > 0 ? NA : 1) * 16);
  if (FILL) {
    l->ocnt = (uint32_t*)(base + oc) + 1; l->heap = (uint16_t*)(base + hp); l->pat = (uint32_t*)(base + pt);
    l->ainfo = (int4*)(base + ai);
    l->grid = (uint16_t*)(base + g); l->ag = (DAgent*)(base + ag); l->as = (DAsm*)(base + as);
    l->ch = l->ag + a_rec;
    l->exec = (int*)(base + ex); l->tok_att = (int*)(base + ta); l->tokacc = (float*)(base + tk);
    l->rows = (uint8_t*)(base + rows);
  }
  return o;
}

// LDS of the assembler-use kernel (fast mode): config stream, records, stat deltas, the hand-over records.
template <bool FILL>
__host__ __device__ inline size_t mg_carve_uses(unsigned char* base, Lds* l, int A, int NA, int SG, int dcw) {
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = mg_align16(o + bytes);
    return at;
  };
  size_t dco = take((size_t)dcw * 4), tmpo = take(8 * 96 * 4);
  size_t ag = take((size_t)A * sizeof(DAgent)), as = take((size_t)(NA > 0 ? NA : 1) * sizeof(DAsm));
  size_t gd = take((size_t)SG * 4), da = take((size_t)A * 4), dm = take((size_t)A * 4), ord = take((size_t)A * 2);
  size_t fl = take(16), rows = take((size_t)A * 32), ai = take((size_t)(NA > 0 ? NA : 1) * 16);
  if (FILL) {
    l->ainfo = (int4*)(base + ai);
    l->dc = (int*)(base + dco); l->tmp = (int*)(base + tmpo);
    l->ag = (DAgent*)(base + ag); l->as = (DAsm*)(base + as); l->gd = (int*)(base + gd);
    l->dirty_amt = (uint32_t*)(base + da); l->dirty_misc = (uint32_t*)(base + dm); l->order = (uint16_t*)(base + ord);
    l->flags = (int*)(base + fl); l->rows = (uint8_t*)(base + rows);
    l->grid = nullptr;
  }
  return o;
}

// ------------------------------------------------------------------------------------------------------------
// cooperative pieces (all threads of the workgroup)
// ------------------------------------------------------------------------------------------------------------
__device__ inline void mg_load_grid(const KArgs& a, const Lds& l, int map_id, int tid, int nt) {
  const uint16_t* src = a.map_cells + (size_t)map_id * a.HW;
  int n2 = a.HW >> 1;
  const uint32_t* s32 = (const uint32_t*)src;  // HW*2 bytes per map; maps start 4-byte aligned (HW even or padded)
  uint32_t* d32 = (uint32_t*)l.grid;
  if ((a.HW & 1) == 0) {
    for (int i = tid; i < n2; i += nt) d32[i] = s32[i];
  } else {
    for (int i = tid; i < a.HW; i += nt) l.grid[i] = src[i];
  }
}

__device__ inline void mg_overlay_agents(const KArgs& a, const Lds& l, int tid, int nt) {
  const CellCodec cc{a.NW, a.A, a.NAmax};
  for (int i = tid; i < a.A; i += nt) l.grid[(int)l.ag[i].r * a.W + l.ag[i].c] = cc.agent(i);
}

__device__ __forceinline__ int mg_heap_off(const KArgs& a, int obj) {
  return obj < a.NW ? obj * a.TSW
                    : (obj < a.NW + a.A
                           ? a.NW * a.TSW + (obj - a.NW) * a.TSA
                           : (obj < a.NW + a.A + a.NAmax
                                  ? a.NW * a.TSW + a.A * a.TSA + (obj - a.NW - a.A) * a.TSS
                                  : a.NW * a.TSW + a.A * a.TSA + a.NAmax * a.TSS + (obj - a.NW - a.A - a.NAmax) * a.TSC));
}

// Build every object's token list once (they are observer independent) and the pattern table.
// Requires l.ainfo (the map's static assembler info) staged + a barrier before.
__device__ inline void mg_build_tokens(const KArgs& a, const Lds& l, int map_id, uint32_t step, int tid, int nt) {
  const int* dc = a.dc;
  const CellCodec cc{a.NW, a.A, a.NAmax};
  const int4* asm_info = l.ainfo;
  const int nasm = a.map_nasm[map_id];
  const int nchest = a.NCmax > 0 ? a.map_nchest[map_id] : 0;
  const int4* chest_info = a.map_chest + (size_t)map_id * a.NCmax;
  const int first_chest = a.NW + a.A + a.NAmax;
  for (int obj = tid; obj < a.NOBJ; obj += nt) {
    const int ho = mg_heap_off(a, obj);
    PairOut out{l.heap + ho, 0};
    if (obj < a.NW + a.A + nasm || (obj >= first_chest && obj < first_chest + nchest))
      mg_object_tokens(dc, l.grid, l.ag, l.as, asm_info, cc, a.H, a.W, step, (uint16_t)(obj + 1), out, l.ch, chest_info);
    l.ocnt[obj] = (uint32_t)ho | ((uint32_t)out.n << 16);
  }
  if (tid == 0) l.ocnt[-1] = 0;  // cell code 0 (empty) looks up "no tokens" without a branch
  const int P = dc[DC_PATTERN_N], op = dc[DC_OFF_PATTERN], hr = dc[DC_OBS_H] >> 1, wr = dc[DC_OBS_W] >> 1;
  for (int p = tid; p < P; p += nt) {
    uint32_t packed = (uint32_t)dc[op + p];
    int dr = (int)(packed & 0xFF) - 8, dcc = (int)((packed >> 8) & 0xFF) - 8;
    l.pat[p] = packed | ((uint32_t)(((dr + hr) << 4) | ((dcc + wr) & 0x0F)) << 16);
  }
}

struct EncCfg {  // per-kernel constants of the encoder, read from the config stream once (not per agent)
  int oh, ow, hr, wr, P, T, H, W;
  uint32_t center;
  int g_goal, g_compass;
  uint32_t f_goal, f_compass;
};
__device__ inline EncCfg mg_enc_cfg(const KArgs& a) {
  const int* dc = a.dc;
  EncCfg c;
  c.oh = dc[DC_OBS_H]; c.ow = dc[DC_OBS_W]; c.hr = c.oh >> 1; c.wr = c.ow >> 1; c.P = dc[DC_PATTERN_N]; c.T = a.T;
  c.H = a.H; c.W = a.W;
  c.center = (uint32_t)((c.hr << 4) | (c.wr & 0x0F));
  c.g_goal = dc[DC_G_GOAL]; c.g_compass = dc[DC_G_COMPASS];
  c.f_goal = (uint32_t)dc[DC_FID0 + MG_F_GOAL];
  c.f_compass = (uint32_t)dc[DC_FID0 + MG_F_COMPASS];
  return c;
}
// The fixed global tokens, one per lane (lane i < n holds the i-th): episode_completion_pct, last_action,
// last_reward in the reference's order (bindings/mettagrid_c.cpp:416-447).
struct GlobalTok {
  int n;
  uint32_t f, v;
  bool is_action;
};
__device__ inline GlobalTok mg_global_tok(const KArgs& a, uint32_t step, int lane) {
  const int* dc = a.dc;
  GlobalTok g;
  g.f = 0; g.v = 0; g.is_action = false;
  int n = 0;
  if (dc[DC_G_COMPLETION]) {
    const int ms = dc[DC_MAX_STEPS];
    int pct = 0;
    if (ms > 0) pct = step >= (uint32_t)ms ? 255 : (int)((256u * step / (uint32_t)ms) & 0xFF);
    if (lane == n) { g.f = (uint32_t)dc[DC_FID0 + MG_F_COMPLETION]; g.v = (uint32_t)pct; }
    n++;
  }
  if (dc[DC_G_LAST_ACTION]) {
    if (lane == n) { g.f = (uint32_t)dc[DC_FID0 + MG_F_LAST_ACTION]; g.is_action = true; }
    n++;
  }
  if (dc[DC_G_LAST_REWARD]) {
    // rewards are zero whenever observations are computed (zeroed at step start, filled after the encode)
    if (lane == n) { g.f = (uint32_t)dc[DC_FID0 + MG_F_LAST_REWARD]; g.v = 0; }
    n++;
  }
  g.n = n;
  return g;
}

#define MG_MAX_ROUNDS 4  // ceil(15 * 15 / 64): observation windows are at most 15 x 15

// One agent's observation row, encoded by one wave into `row` (LDS, pre-filled with 0xFF).
// MettaGrid::_compute_observation (bindings/mettagrid_c.cpp:397-555).  Returns the attempted token count.
// Window lookups of one agent, all rounds (independent LDS reads: they overlap, and the caller issues them for both
// agents of a pair before either row is written).  Window offsets never exceed the radius, so "inside the window and
// the map" is just "inside the map".  meta = heap offset | token count << 16 per window cell.
template <int NR, bool HOT>
__device__ __forceinline__ void mg_window_lookup(const KArgs& a, const Lds& l, const EncCfg& ec, const DAgent* me,
                                                 const uint32_t (&pk)[MG_MAX_ROUNDS], int lane, uint32_t (&meta)[NR],
                                                 uint32_t& rc) {
  const int H = ec.H, W = ec.W, P = ec.P;
  rc = *(const uint32_t*)me;  // r | c << 16 (first two fields of the record)
  const int ar = (int)(rc & 0xFFFF), ac = (int)(rc >> 16);
#pragma unroll
  for (int q = 0; q < NR; q++) {
    int p = q * 64 + lane;
    int r = ar + (int)(pk[q] & 0xFF) - 8, c = ac + (int)((pk[q] >> 8) & 0xFF) - 8;
    const bool ok = p < P && (unsigned)r < (unsigned)H && (unsigned)c < (unsigned)W && (HOT || !(a.diag & 512));
    // no branches: cells outside the map read cell 0 and are masked; empty cells read the zero entry ocnt[-1]
    const uint32_t cv = l.grid[ok ? __mul24(r, W) + c : 0];
    meta[q] = l.ocnt[(int)(ok ? cv : 0u) - 1];
  }
}

template <int NR, bool HOT = false>
__device__ __forceinline__ int mg_encode_agent(const KArgs& a, const Lds& l, const EncCfg& ec, const GlobalTok& gt,
                                      const DAgent* me, int action, uint8_t* __restrict__ row,
                                      const uint32_t (&pk)[MG_MAX_ROUNDS], int lane, const uint32_t (&meta)[NR],
                                      const uint32_t rc) {
  const int T = ec.T, hr = ec.hr, wr = ec.wr, H = ec.H, W = ec.W;
  const int ar = (int)(rc & 0xFFFF), ac = (int)(rc >> 16);
  // Global tokens at the centre.  The fixed ones (episode completion, last action, last reward) sit one per lane,
  // prepared once per wave (GlobalTok); goal and compass tokens depend on the agent and follow from lane 0.
  int base = gt.n;
  if (lane < gt.n && lane < T) {
    row[lane * 3 + 0] = (uint8_t)ec.center;
    row[lane * 3 + 1] = (uint8_t)gt.f;
    row[lane * 3 + 2] = (uint8_t)(gt.is_action ? action : gt.v);
  }
  if ((ec.g_goal || ec.g_compass) && lane == 0) {
    const uint8_t loc = (uint8_t)ec.center;
    auto put = [&](uint8_t lc, uint32_t f, int v) {
      if (base < T) {
        row[base * 3 + 0] = lc;
        row[base * 3 + 1] = (uint8_t)f;
        row[base * 3 + 2] = (uint8_t)v;
      }
      base++;
    };
    if (ec.g_goal) {
      const int* dc = a.dc;
      int og = group_rec(dc, me->group)[GR_OFF_GOAL], ng = L_n(dc, og);
      for (int i = 0; i < ng; i++) put(loc, ec.f_goal, L_at(dc, og, i));
    }
    if (ec.g_compass) {
      int dr = H / 2 - ar, dcc = W / 2 - ac;
      int sr = dr != 0 ? (dr > 0 ? 1 : -1) : 0, sc = dcc != 0 ? (dcc > 0 ? 1 : -1) : 0;
      int orr = hr + sr, occ = wr + sc;
      if ((sr != 0 || sc != 0) && orr >= 0 && orr < ec.oh && occ >= 0 && occ < ec.ow)
        put((uint8_t)((orr << 4) | (occ & 0x0F)), ec.f_compass, 1);
    }
  }
  base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
  for (int q = 0; q < NR; q++) {
    const int cnt = (int)(meta[q] >> 16);
    // exclusive prefix sum of cnt over the wave: six DPP steps (row shifts, then the row broadcasts).  (A ballot +
    // mbcnt count of the single-token objects plus a uniform loop over the multi-token ones cost 4 us more per launch.)
    int off, total;
    {
      int x = cnt;
      x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
      x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
      x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
      x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
      x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
      x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
      off = x - cnt;
      total = __builtin_amdgcn_readlane(x, 63);
    }
    off += base;
    if (cnt > 0 && off < T && (HOT || !(a.diag & 128))) {
      // token lists start 8-byte aligned and are padded to x4 tokens: read 4 tokens per LDS access
      const uint2* __restrict__ src = (const uint2*)(l.heap + (meta[q] & 0xFFFF));
      const uint8_t loc = (uint8_t)(pk[q] >> 16);
      const int n = min(cnt, T - off);
      uint8_t* __restrict__ dst = row + off * 3;
      uint2 v = src[0];
      {  // most objects (walls) have one token: it goes out without any further test
        const uint32_t t0 = v.x & 0xFFFF;
        dst[0] = loc; dst[1] = (uint8_t)t0; dst[2] = (uint8_t)(t0 >> 8);
      }
      if (n > 1) {
        for (int k = 0; k < n; k += 4) {
          if (k) v = src[k >> 2];
          uint32_t t0 = v.x & 0xFFFF, t1 = v.x >> 16, t2 = v.y & 0xFFFF, t3 = v.y >> 16;
          if (k) { dst[k * 3 + 0] = loc; dst[k * 3 + 1] = (uint8_t)t0; dst[k * 3 + 2] = (uint8_t)(t0 >> 8); }
          if (k + 1 < n) { dst[k * 3 + 3] = loc; dst[k * 3 + 4] = (uint8_t)t1; dst[k * 3 + 5] = (uint8_t)(t1 >> 8); }
          if (k + 2 < n) { dst[k * 3 + 6] = loc; dst[k * 3 + 7] = (uint8_t)t2; dst[k * 3 + 8] = (uint8_t)(t2 >> 8); }
          if (k + 3 < n) { dst[k * 3 + 9] = loc; dst[k * 3 + 10] = (uint8_t)t3; dst[k * 3 + 11] = (uint8_t)(t3 >> 8); }
        }
      }
    }
    base += total;
  }
  return base;
}

// All agents' observations; optionally streamed to HBM.  `exec` holds each agent's executed action (or null = 0).
// Requires mg_build_tokens + a barrier before.
template <int NR, bool PAIRS>
__device__ __forceinline__ void mg_encode_all_t(const KArgs& a, const Lds& l, int e, uint32_t step, const int* exec,
                                       bool write_global, int tid, int nt) {
  // (the wave index stays a vector value on purpose: made scalar with readfirstlane, the per-agent address arithmetic
  // moves to the scalar unit, SGPR spills go from 74 to 99 and the kernel gets slower, 62.5 us against 58.7)
  const int lane = tid & 63, wave = tid >> 6, nwaves = nt >> 6;
  uint8_t* rowbuf = l.rows + (size_t)wave * a.ROWB;
  const int row_bytes = a.T * 3;
  const EncCfg ec = mg_enc_cfg(a);
  const GlobalTok gt = mg_global_tok(a, step, lane);
  uint32_t pk[MG_MAX_ROUNDS];
#pragma unroll
  for (int q = 0; q < MG_MAX_ROUNDS; q++) pk[q] = (q < NR && (q * 64 + lane) < ec.P) ? l.pat[q * 64 + lane] : 0u;
  // Rows whose length is 8 mod 16 (T = 200: 600 bytes) alternate between phase 0 and 8 in HBM.  A wave then takes
  // two consecutive agents: the even row keeps its last 8 bytes back, and they leave at the front of the odd row,
  // so that every store is a full aligned 16-byte chunk (8-byte partial stores of two waves into one chunk cost
  // more than the whole rest of the row).
  uint8_t* env_obs = a.obs + (size_t)e * a.A * (size_t)row_bytes;
  const uint32_t env_phase = (uint32_t)(((size_t)e * a.A * (size_t)row_bytes) & 15);
  constexpr bool pairs = PAIRS;  // write_global && A even && row_bytes % 16 == 8 (checked by the caller)
  constexpr int group = pairs ? 2 : 1;
  const int nfill = a.ROWB >> 4;
  uint2 carry = make_uint2(0, 0);
  // per-agent offsets are carried along instead of multiplied out (v_mul_lo_u32 runs at quarter rate)
  uint32_t roff0 = (uint32_t)(wave * group) * (uint32_t)row_bytes;  // within the env's block of rows (A * T * 3 < 2^32)
  const DAgent* me0 = l.ag + wave * group;
  const uint32_t roff_step = (uint32_t)(nwaves * group) * (uint32_t)row_bytes;
  for (int a0 = wave * group; a0 < a.A; a0 += nwaves * group, roff0 += roff_step, me0 += nwaves * group)
  {
    uint32_t meta[group][NR], rc[group];
#pragma unroll
    for (int h = 0; h < group; h++) mg_window_lookup<NR, PAIRS>(a, l, ec, me0 + h, pk, lane, meta[h], rc[h]);
#pragma unroll
    for (int h = 0; h < group; h++) {
      const int ai = a0 + h;
      // The row sits in LDS at the same 16-byte phase as in HBM, so that it leaves as aligned 16-byte stores.
      const uint32_t roff = roff0 + (uint32_t)(h * row_bytes);
      // paired rows: A * row_bytes is a multiple of 16, the even row starts a chunk, the odd one sits at 8
      const int phase = PAIRS ? h * 8 : (int)((env_phase + roff) & 15u);
      uint8_t* row = rowbuf + phase;
      if (PAIRS || nfill <= 64) {  // the paired shape is only chosen for rows of at most 1024 bytes
        if (lane < nfill) ((uint4*)rowbuf)[lane] = make_uint4(~0u, ~0u, ~0u, ~0u);
      } else {
        for (int i = lane; i < nfill; i += 64) ((uint4*)rowbuf)[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
      }
      if (pairs && h == 1 && lane == 0) *(uint2*)rowbuf = carry;  // phase == 8 here
      __builtin_amdgcn_wave_barrier();
      const int act = exec ? exec[ai] : 0;
      int attempted = mg_encode_agent<NR, PAIRS>(a, l, ec, gt, me0 + h, act, row, pk, lane, meta[h], rc[h]);
      if (lane == 0) l.tok_att[ai] = attempted;
      __builtin_amdgcn_wave_barrier();
      if (write_global && !(a.diag & 1)) {
        uint8_t* gbase = env_obs + roff - phase;  // 16-byte aligned (the buffer itself is 256-byte aligned)
        const int end = phase + row_bytes;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        if (pairs) {
          // odd row: the chunk at 0 holds the even row's tail; even row: the tail waits for the odd row.  Every
          // chunk is full; the rows (at most 1024 bytes in this shape) leave in one store instruction, addressed as
          // uniform base + 32-bit lane offset.
          const int nch = (h == 0 ? end - 8 : end) >> 4;
          if (h == 0) carry = *(const uint2*)(rowbuf + end - 8);
          if (lane < nch) {
            const uint4 v = *(const uint4*)(rowbuf + lane * 16);
            const uint32_t off32 = roff - (uint32_t)phase + (uint32_t)lane * 16u;
            // streamed out, never re-read by the step kernels: keep it from displacing the map pool and records in L2
            __builtin_nontemporal_store((u32x4){v.x, v.y, v.z, v.w}, (u32x4*)(env_obs + off32));
          }
        } else {
          for (int ch = lane; ch * 16 < end; ch += 64) {
            const int b0 = ch * 16;
            if (b0 >= phase && b0 + 16 <= end) {
              const uint4 v = *(const uint4*)(rowbuf + b0);
              __builtin_nontemporal_store((u32x4){v.x, v.y, v.z, v.w}, (u32x4*)(gbase + b0));
            } else {  // the row's ragged head / tail
              const int lo = b0 < phase ? phase : b0, hi = b0 + 16 < end ? b0 + 16 : end;
              if (((lo | hi) & 3) == 0) {
                for (int b = lo; b < hi; b += 4) *(uint32_t*)(gbase + b) = *(const uint32_t*)(rowbuf + b);
              } else {
                for (int b = lo; b < hi; b++) gbase[b] = rowbuf[b];
              }
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// The headline shape (window of at most 128 cells, rows of 8 mod 16 bytes in pairs: 11 x 11 window, 200 tokens) has
// a loop of its own, so that the general loop's state stays out of its registers.
__device__ inline void mg_encode_all(const KArgs& a, const Lds& l, int e, uint32_t step, const int* exec,
                                     bool write_global, int tid, int nt) {
  const bool pairs = write_global && (a.A & 1) == 0 && ((a.T * 3) & 15) == 8 && a.ROWB <= 1024;
  if (pairs && a.dc[DC_PATTERN_N] <= 128) mg_encode_all_t<2, true>(a, l, e, step, exec, write_global, tid, nt);
  else mg_encode_all_t<MG_MAX_ROUNDS, false>(a, l, e, step, exec, write_global, tid, nt);  // any shape
}

// Sequential f32 accumulation of the tokens_* game stats in agent order (bindings/mettagrid_c.cpp:552-554),
// so that rounding above 2^24 matches the reference.  Thread 0 only.
__device__ inline void mg_token_stats(const KArgs& a, const Lds& l, int passes) {
  float w = l.tokacc[0], d = l.tokacc[1], f = l.tokacc[2];
  for (int p = 0; p < passes; p++)
    for (int i = 0; i < a.A; i++) {
      int att = l.tok_att[i];
      int wr = att < a.T ? att : a.T;
      w += (float)wr;
      d += (float)(att - wr);
      f += (float)(a.T - wr);
    }
  l.tokacc[0] = w;
  l.tokacc[1] = d;
  l.tokacc[2] = f;
}

// Fresh episode state in LDS + zeroed global stats (MettaGrid ctor: bindings/mettagrid_c.cpp:46-155,159-245).
__device__ inline void mg_reset_env(const KArgs& a, const Lds& l, Ctx& cx, int e, int map_id, uint32_t seed, int tid,
                                    int nt) {
  const int* dc = a.dc;
  const int4* spawn = a.map_spawn + (size_t)map_id * a.A;
  for (int i = tid; i < a.A; i += nt) {
    int4 s = spawn[i];
    DAgent ag;
    memset(&ag, 0, sizeof(ag));
    ag.r = ag.prev_r = (uint16_t)s.x;
    ag.c = ag.prev_c = (uint16_t)s.y;
    ag.group = (uint8_t)s.z;
    ag.vibe = (uint8_t)group_rec(dc, s.z)[GR_INITIAL_VIBE];
    ag.nb = 1;
    l.ag[i] = ag;
  }
  int nasm = a.map_nasm[map_id];
  const int4* asm_info = a.map_asm + (size_t)map_id * a.NAmax;
  for (int i = tid; i < a.NAmax; i += nt) {
    DAsm s = {0, 0, 0, 0};
    if (i < nasm) s.clipped = (uint32_t)asm_rec(dc, asm_info[i].z)[AR_START_CLIPPED];
    l.as[i] = s;
  }
  for (int i = tid; i < a.SG; i += nt) l.gd[i] = 0;
  for (int i = tid; i < a.A; i += nt) {
    l.dirty_amt[i] = 0;
    l.dirty_misc[i] = 0;
  }
  // zero the running-episode stats in HBM
  float* as_ = a.astats + (size_t)e * a.A * a.SA;
  for (int i = tid; i < a.A * a.SA; i += nt) as_[i] = 0.f;
  uint32_t* at_ = a.atouched + (size_t)e * a.A * a.TW;
  for (int i = tid; i < a.A * a.TW; i += nt) at_[i] = 0u;
  float* gs_ = a.gstats + (size_t)e * a.SG;
  uint32_t* gt_ = a.gtouched + (size_t)e * a.GW;
  for (int i = tid; i < a.GW; i += nt) gt_[i] = 0u;
  __syncthreads();
  // objects.<cell name> counts (bindings/mettagrid_c.cpp:203,224,235)
  const float* oc = a.map_objcount + (size_t)map_id * a.n_cells;
  for (int i = tid; i < a.SG; i += nt) {
    float v = 0.f;
    int code = i - 3 + 1;
    if (code >= 1 && code < a.n_cells && oc[code] > 0.f) {
      v = oc[code];
      atomicOr(&gt_[i >> 5], 1u << (i & 31));
    }
    gs_[i] = v;
  }
  for (int i = tid; i < a.A; i += nt) {
    size_t g = (size_t)e * a.A + i;
    a.rewards[g] = 0.f;
    a.ep_rewards[g] = 0.f;
    a.term[g] = 0;
    a.trunc[g] = 0;
    a.success[g] = 0;
  }
  if (tid == 0) {
    l.tokacc[0] = l.tokacc[1] = l.tokacc[2] = 0.f;
    // std::mt19937(seed): mt[i] = 1812433253 * (mt[i-1] ^ (mt[i-1] >> 30)) + i
    uint32_t* mt = a.mt + (size_t)e * 624;
    uint32_t* mi = a.mt_init ? a.mt_init + (size_t)e * 624 : mt;
    uint32_t x = seed;
    mt[0] = x;
    mi[0] = x;
    for (int i = 1; i < 624; i++) {
      x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
      mt[i] = x;
      mi[i] = x;
    }
  }
  // populate_initial_inventory (agent.cpp:48-52): ignore_limits, stat hooks fire; one thread per agent
  for (int i = tid; i < a.A; i += nt) {
    const int* gr = group_rec(dc, l.ag[i].group);
    int oi = gr[GR_OFF_INITIAL], n = L_n(dc, oi);
    for (int k = 0; k < n; k++) inv_update(cx, i, KV_k(dc, oi, k), KV_v(dc, oi, k), true);
  }
  // chests (chest.hpp:83-87): initial inventory, positive amounts only, limits ignored.  One thread, map order: the
  // chest.<r>.amount game stat is a `set`, the last chest holding <r> wins.
  if (tid == 0) {
    *l.gsetm = 0;
    const int nchest = a.NCmax > 0 ? a.map_nchest[map_id] : 0;
    const int4* chest_info = a.map_chest + (size_t)map_id * a.NCmax;
    for (int ci = 0; ci < a.NCmax; ci++) {
      DAgent rec;
      memset(&rec, 0, sizeof(rec));
      rec.nb = 1;
      if (ci < nchest) {
        const int4 info = chest_info[ci];
        rec.r = rec.prev_r = (uint16_t)info.x;
        rec.c = rec.prev_c = (uint16_t)info.y;
        rec.group = (uint8_t)(dc + dc[dc[DC_OFF_CHESTS] + info.z])[CR_GROUP];
      }
      l.ch[ci] = rec;
      if (ci < nchest) {
        const int oi = group_rec(dc, rec.group)[GR_OFF_INITIAL], n = L_n(dc, oi);
        for (int k = 0; k < n; k++)
          if (KV_v(dc, oi, k) > 0) inv_update(cx, a.A + ci, KV_k(dc, oi, k), KV_v(dc, oi, k), true);
      }
    }
  }
  __threadfence_block();
  __syncthreads();
}

// Push this step's stat deltas to HBM and mark touched keys.
__device__ inline void mg_flush_stats(const KArgs& a, const Lds& l, const Ctx& cx, int e, int tid, int nt) {
  float* as_ = a.astats + (size_t)e * a.A * a.SA;
  uint32_t* at_ = a.atouched + (size_t)e * a.A * a.TW;
  for (int ai = tid; ai < a.A; ai += nt) {
    uint32_t m = l.dirty_amt[ai];
    while (m) {
      int r = __ffs(m) - 1;
      m &= m - 1;
      int s = cx.AS.amount(r);
      as_[ai * a.SA + s] = (float)l.ag[ai].amt[r];
      atomicOr(&at_[ai * a.TW + (s >> 5)], 1u << (s & 31));
    }
    uint32_t dm = l.dirty_misc[ai];
    if (dm & 1u) {
      int s = cx.AS.diversity();
      as_[ai * a.SA + s] = (float)l.ag[ai].diversity;
      atomicOr(&at_[ai * a.TW + (s >> 5)], 1u << (s & 31));
    }
  }
  float* gs_ = a.gstats + (size_t)e * a.SG;
  uint32_t* gt_ = a.gtouched + (size_t)e * a.GW;
  for (int i = tid; i < a.SG; i += nt) {
    int v = l.gd[i];
    if (v != 0) {
      gs_[i] += (float)v;
      atomicOr(&gt_[i >> 5], 1u << (i & 31));
    }
  }
  if (a.NCmax > 0 && tid < a.R && ((*l.gsetm >> tid) & 1u)) {  // StatsTracker::set of chest.<r>.amount
    const int s = cx.GS.chest_amount(tid);
    gs_[s] = (float)l.gsetv[tid];
    atomicOr(&gt_[s >> 5], 1u << (s & 31));
  }
}

// Agent::compute_stat_rewards (agent.cpp:103-132) + episode bookkeeping (bindings/mettagrid_c.cpp:656-677).
__device__ inline void mg_rewards_and_flags(const KArgs& a, const Lds& l, int e, DEnv env, int tid, int nt) {
  const int* dc = a.dc;
  const MgAgentSlots AS{a.R, a.NG};
  bool end_now = false, trunc_now = false;
  {
    int ms = dc[DC_MAX_STEPS];
    if (ms > 0 && env.step >= (uint32_t)ms) {
      end_now = true;
      trunc_now = dc[DC_TRUNCATES] != 0;
    }
  }
  bool early = env.early_reset != 0 && env.step >= env.early_reset;
  const float* as_ = a.astats + (size_t)e * a.A * a.SA;
  const float* gs_ = a.gstats + (size_t)e * a.SG;
  for (int ai = tid; ai < a.A; ai += nt) {
    DAgent* me = &l.ag[ai];
    const int* gr = group_rec(dc, me->group);
    int os = gr[GR_OFF_STAT_REWARDS], ns = dc[os];
    float reward = 0.f;
    size_t g = (size_t)e * a.A + ai;
    if (ns > 0) {
      float total = 0.f;
      for (int q = 0; q < ns; q++) {
        const int* sr = dc + os + 1 + 5 * q;
        int aslot = sr[0], gslot = sr[1];
        float v = 0.f;
        if (aslot >= 0) {
          if (aslot >= AS.amount(0) && aslot < AS.amount(0) + a.R)
            v = (float)me->amt[aslot - AS.amount(0)];
          else if (aslot == AS.max_steps_without_motion())
            v = (float)me->max_swm;
          else if (aslot == AS.diversity())
            v = (float)me->diversity;
          else if (aslot < 3)
            v = (float)me->succ[aslot];
          else if (aslot < 6)
            v = (float)me->fail[aslot - 3];
          else if (aslot == AS.failed_any())
            v = (float)(me->fail[0] + me->fail[1] + me->fail[2]);
          else
            v = as_[ai * a.SA + aslot];
        }
        if (gslot >= 0) v += gslot < 3 ? l.tokacc[gslot] : gs_[gslot];
        float rew = v * __int_as_float(sr[2]);
        if (sr[3]) {
          float mx = __int_as_float(sr[4]);
          rew = (mx < rew) ? mx : rew;
        }
        total += rew;
      }
      float delta = total - me->cur_reward;
      if (delta != 0.0f) {
        reward = delta;
        me->cur_reward = total;
        a.agents[g].cur_reward = total;
      }
    }
    a.rewards[g] = reward;
    a.ep_rewards[g] += reward;
    if (end_now) {
      if (trunc_now) a.trunc[g] = 1; else a.term[g] = 1;
    }
    if (early) a.trunc[g] = 1;
  }
  if (tid == 0 && (end_now || early)) {
    a.env[e].done = 1;
    if (early) a.env[e].early_reset = 0;
  }
}

__device__ inline void mg_make_ctx(const KArgs& a, const Lds& l, Ctx& cx, int e, int map_id, uint32_t step) {
  cx.dc = a.dc;
  cx.grid = l.grid;
  cx.ag = l.ag;
  cx.as = l.as;
  cx.asm_info = a.map_asm + (size_t)map_id * a.NAmax;
  cx.astats = a.astats + (size_t)e * a.A * a.SA;
  cx.atouched = a.atouched + (size_t)e * a.A * a.TW;
  cx.TW = a.TW;
  cx.gd = l.gd;
  cx.dirty_amt = l.dirty_amt;
  cx.tmp = l.tmp;
  cx.dirty_misc = l.dirty_misc;
  cx.H = a.H; cx.W = a.W; cx.A = a.A; cx.NA = a.NAmax; cx.R = a.R; cx.SA = a.SA; cx.SG = a.SG; cx.NG = a.NG;
  cx.step = step;
  cx.asm_dirty = 0;
  cx.AS = MgAgentSlots{a.R, a.NG};
  cx.GS = MgGameSlots{a.R, a.n_cells - 1};
  cx.cc = CellCodec{a.NW, a.A, a.NAmax};
  cx.ch = l.ch;
  cx.chest_info = a.map_chest + (size_t)map_id * a.NCmax;
  cx.NC = a.NCmax;
  cx.gsetv = l.gsetv;
  cx.gsetm = l.gsetm;
  cx.overflow = 0;
}

// Cooperative copy of `bytes` (a multiple of 4) between HBM arrays: 16 bytes per lane and several loads in flight
// when the alignment allows (restarts sit on the critical path of the act kernels).
__device__ inline void mg_copy(void* __restrict__ dst, const void* __restrict__ src, size_t bytes, int tid, int nt) {
  if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 15) == 0) {
    const uint4* __restrict__ s = (const uint4*)src;
    uint4* __restrict__ d = (uint4*)dst;
    const int n = (int)(bytes >> 4);
    int i = tid;
    for (; i + 3 * nt < n; i += 4 * nt) {
      uint4 v0 = s[i], v1 = s[i + nt], v2 = s[i + 2 * nt], v3 = s[i + 3 * nt];
      d[i] = v0; d[i + nt] = v1; d[i + 2 * nt] = v2; d[i + 3 * nt] = v3;
    }
    for (; i < n; i += nt) d[i] = s[i];
  } else {
    const uint32_t* __restrict__ s = (const uint32_t*)src;
    uint32_t* __restrict__ d = (uint32_t*)dst;
    const int n = (int)(bytes >> 2);
    for (int i = tid; i < n; i += nt) d[i] = s[i];
  }
}

// Keep the finished episode's stats for the host and add it to the device-wide summary.
__device__ inline void mg_snapshot_finished(const KArgs& a, int e, uint32_t steps, int tid, int nt) {
  size_t na = (size_t)a.A * a.SA, nta = (size_t)a.A * a.TW;
  float* fas = a.fin_astats + (size_t)e * na;
  mg_copy(fas, a.astats + (size_t)e * na, na * 4, tid, nt);
  mg_copy(a.fin_atouched + (size_t)e * nta, a.atouched + (size_t)e * nta, nta * 4, tid, nt);
  __syncthreads();
  {  // merge the record-resident counters into the finished snapshot
    const MgAgentSlots AS{a.R, a.NG};
    for (int ai = tid; ai < a.A; ai += nt) {
      const DAgent* ag = &a.agents[(size_t)e * a.A + ai];
      float* row = fas + (size_t)ai * a.SA;
      uint32_t fa = 0;
      for (int h = 0; h < 3; h++) {
        row[AS.success(h)] = (float)ag->succ[h];
        row[AS.failed(h)] = (float)ag->fail[h];
        fa += ag->fail[h];
      }
      row[AS.failed_any()] = (float)fa;
      row[AS.max_steps_without_motion()] = (float)ag->max_swm;
    }
  }
  const float* gs_ = a.gstats + (size_t)e * a.SG;
  float* fgs = a.fin_gstats + (size_t)e * a.SG;
  for (int i = tid; i < a.SG; i += nt) {
    float v = gs_[i];
    fgs[i] = v;
    if (v != 0.f) atomicAdd(&a.summary[3 + i], (double)v);
  }
  const uint32_t* gt_ = a.gtouched + (size_t)e * a.GW;
  uint32_t* fgt = a.fin_gtouched + (size_t)e * a.GW;
  for (int i = tid; i < a.GW; i += nt) fgt[i] = gt_[i];
  for (int i = tid; i < a.A; i += nt) {
    float r = a.ep_rewards[(size_t)e * a.A + i];
    a.fin_ep_rewards[(size_t)e * a.A + i] = r;
    if (r != 0.f) atomicAdd(&a.summary[2], (double)r);
  }
  if (tid == 0) {
    a.fin_steps[e] = steps;
    atomicAdd(&a.status[1], 1u);
    atomicAdd(&a.summary[0], 1.0);
    atomicAdd(&a.summary[1], (double)steps);
  }
}

// Stage the window of MT19937 outputs this step may consume; twist cooperatively when it crosses the block end.
__device__ inline void mg_stage_rng(const KArgs& a, const Lds& l, int e, uint32_t mt_idx, int tid, int nt) {
  const uint32_t* mt = a.mt + (size_t)e * 624;
  bool need_twist = (int)mt_idx + a.NWIN > 624;  // block-uniform
  if (need_twist) {
    for (int i = tid; i < 227; i += nt) l.mtnext[i] = mt_mix(mt[i], mt[i + 1], mt[i + 397]);
    __syncthreads();
    for (int i = 227 + tid; i < 454; i += nt) l.mtnext[i] = mt_mix(mt[i], mt[i + 1], l.mtnext[i - 227]);
    __syncthreads();
    for (int i = 454 + tid; i < 623; i += nt) l.mtnext[i] = mt_mix(mt[i], mt[i + 1], l.mtnext[i - 227]);
    __syncthreads();
    if (tid == 0) l.mtnext[623] = mt_mix(mt[623], l.mtnext[0], l.mtnext[396]);
    __syncthreads();
  }
  for (int i = tid; i < a.NWIN; i += nt) {
    int j = (int)mt_idx + i;
    uint32_t raw = j < 624 ? mt[j] : l.mtnext[(j - 624) % 624];
    l.win[i] = mt_temper(raw);
  }
}

// Restart a finished environment from the per-map templates (MettaGridPufferEnv.step -> _new_sim(),
// mettagrid_puffer_env.py:158-164; same seed, next map).  All threads of the workgroup; HBM state only.
__device__ inline void mg_restart_env(const KArgs& a, int e, DEnv& env, int tid, int nt) {
  mg_snapshot_finished(a, e, env.step, tid, nt);
  __syncthreads();
  const int m = (int)(((int)env.map_id + a.map_stride) % a.P);
  env.map_id = (uint32_t)m;
  env.episodes += 1;
  env.step = 0;
  env.mt_idx = 624;
  env.done = 0;
  env.early_reset = 0;  // EarlyResetHandler cuts the first episode only
  mg_copy(a.agents + (size_t)e * a.A, a.t_agents + (size_t)m * a.A, (size_t)a.A * sizeof(DAgent), tid, nt);
  mg_copy(a.asms + (size_t)e * a.NAmax, a.t_asms + (size_t)m * a.NAmax, (size_t)a.NAmax * sizeof(DAsm), tid, nt);
  if (a.NCmax > 0)
    mg_copy(a.chests + (size_t)e * a.NCmax, a.t_chests + (size_t)m * a.NCmax, (size_t)a.NCmax * sizeof(DAgent), tid, nt);
  mg_copy(a.astats + (size_t)e * a.A * a.SA, a.t_astats + (size_t)m * a.A * a.SA, (size_t)a.A * a.SA * 4, tid, nt);
  mg_copy(a.atouched + (size_t)e * a.A * a.TW, a.t_atouched + (size_t)m * a.A * a.TW, (size_t)a.A * a.TW * 4, tid, nt);
  mg_copy(a.gstats + (size_t)e * a.SG, a.t_gstats + (size_t)m * a.SG, (size_t)a.SG * 4, tid, nt);
  mg_copy(a.gtouched + (size_t)e * a.GW, a.t_gtouched + (size_t)m * a.GW, (size_t)a.GW * 4, tid, nt);
  mg_copy(a.mt + (size_t)e * 624, a.mt_init + (size_t)e * 624, 624 * 4, tid, nt);
  for (int i = tid; i < a.A; i += nt) {
    size_t g = (size_t)e * a.A + i;
    a.rewards[g] = 0.f;
    a.ep_rewards[g] = 0.f;
    a.term[g] = 0;
    a.trunc[g] = 0;
    a.success[g] = 0;
  }
  __threadfence();
  __syncthreads();
}

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
#define MG_FAST_THREADS 64
__global__ __launch_bounds__(MG_FAST_THREADS) void mg_act_fast_kernel(KArgs a) {
  __shared__ uint32_t mtnext[624];
  const int lane = threadIdx.x, e = blockIdx.x + a.env_base;
  const int A = a.A;
  const int* dc = a.dc;
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
      const MgAgentSlots AS{a.R, a.NG};
      const int passes = dc[DC_MAX_PRIORITY] + 1;
      float* row = a.astats + g * a.SA;
      int variant = act == -1 ? 0 : (act >= n_actions && act < n_actions + 7) ? 1 + (act - n_actions) : 8;
      row[AS.invalid()] += (float)passes;
      row[AS.invalid_variant(variant)] += (float)passes;
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

// ------------------------------------------------------------------------------------------------------------
// kernel 1b (fast path): assembler uses handed over by mg_act_fast_kernel, in action order, on LDS-staged
// agent/assembler records.  One wave per environment; environments without uses leave at once.  The LDS
// footprint is kept small (no map tile, no RNG window) so that every environment with work is resident at once:
// the kernel's duration is the latency of one environment's chain, not a sum of rounds.
// ------------------------------------------------------------------------------------------------------------
#define MG_USES_THREADS 64
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
      const uint4* __restrict__ s0 = (const uint4*)a.dc;
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
  mg_use_map(a, map_id);
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

// ------------------------------------------------------------------------------------------------------------
// kernel 1: actions.  One wave per environment; thread 0 resolves the actions in the reference's shuffled
// sequential order on LDS state (bindings/mettagrid_c.cpp:572-650).
// ------------------------------------------------------------------------------------------------------------
#define MG_ACT_THREADS 64
// BIG: the map does not fit LDS next to the records (256 x 256, 1024 agents): the env's grid
// lives in an HBM scratch tile (served by L2), everything else as usual.
template <bool DCL, bool BIG>
__global__ __launch_bounds__(MG_ACT_THREADS) void mg_act_kernel(KArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
#ifdef MG_ISA_LITE  // ISA inspection builds (tools/isa.sh): keep only the kernel under study
  return;
#endif
  const int tid = threadIdx.x, nt = blockDim.x, e = blockIdx.x + a.env_base;
  Lds l;
  mg_carve<true>(smem, &l, BIG ? 0 : a.HW, a.A, a.NAmax, a.SA, a.SG, a.NWIN, a.ROWB, 0, DCL ? a.dc_words : 0, 0, 0, 0, -1, a.NCmax);
  if (BIG) l.grid = a.big_grid + (size_t)e * a.HW;
  MG_STAMP(8);
  if (DCL) {
    for (int i = tid; i < (a.dc_words >> 2); i += nt) ((uint4*)l.dc)[i] = ((const uint4*)a.dc)[i];  // padded to x4
    a.dc = l.dc;
    __syncthreads();
  }
  DEnv env = a.env[e];
  const int* dc = a.dc;
  Ctx cx;

  const bool do_reset = a.auto_reset && env.done;  // block-uniform
  if (do_reset) mg_restart_env(a, e, env, tid, nt);
  mg_use_map(a, env.map_id);
  mg_load_grid(a, l, (int)env.map_id, tid, nt);
  const uint4* src = (const uint4*)(a.agents + (size_t)e * a.A);
  uint4* dst = (uint4*)l.ag;
  for (int i = tid; i < a.A * (int)(sizeof(DAgent) / 16); i += nt) dst[i] = src[i];
  const uint4* s2 = (const uint4*)(a.asms + (size_t)e * a.NAmax);
  uint4* d2 = (uint4*)l.as;
  for (int i = tid; i < a.NAmax; i += nt) d2[i] = s2[i];
  {
    const uint4* s3 = (const uint4*)(a.chests + (size_t)e * a.NCmax);
    uint4* d3 = (uint4*)l.ch;
    for (int i = tid; i < a.NCmax * (int)(sizeof(DAgent) / 16); i += nt) d3[i] = s3[i];
    if (tid == 0) *l.gsetm = 0;
  }
  for (int i = tid; i < a.SG; i += nt) l.gd[i] = 0;
  for (int i = tid; i < a.A; i += nt) {
    l.dirty_amt[i] = 0;
    l.dirty_misc[i] = 0;
  }
  __syncthreads();
  mg_overlay_agents(a, l, tid, nt);
  mg_stage_rng(a, l, e, env.mt_idx, tid, nt);
  env.step += 1;
  mg_make_ctx(a, l, cx, e, (int)env.map_id, env.step);
  __syncthreads();

  if (tid == 0) {
    RngWin rng{l.win, a.NWIN, 0, 0};
    mg_shuffle(l.order, a.A, rng);
    const int n_actions = dc[DC_N_ACTIONS], oa = dc[DC_OFF_ACTIONS];
    const int passes = dc[DC_MAX_PRIORITY] + 1;
    const int* acts = a.actions + (size_t)e * a.A;
    for (int i = 0; i < a.A; i++) l.exec[i] = 0;
    for (int k = 0; k < a.A; k++) {
      int ai = l.order[k];
      int act = acts[ai];
      bool ok = false;
      if (act < 0 || act >= n_actions) {
        // _handle_invalid_action runs once per priority level (the check precedes the priority filter)
        mg_stat_add(cx, ai, cx.AS.invalid(), passes);
        int variant = act == -1 ? 0 : (act >= n_actions && act < n_actions + 7) ? 1 + (act - n_actions) : 8;
        mg_stat_add(cx, ai, cx.AS.invalid_variant(variant), passes);
      } else {
        ok = mg_handle_action(cx, ai, dc[oa + 2 * act], dc[oa + 2 * act + 1]);
        if (ok) l.exec[ai] = act;
      }
      a.success[(size_t)e * a.A + ai] = ok ? 1 : 0;
    }
    // inventory regeneration (bindings/mettagrid_c.cpp:625-640)
    int ri = dc[DC_REGEN_INTERVAL];
    if (ri > 0 && env.step % (uint32_t)ri == 0) {
      for (int ai = 0; ai < a.A; ai++) {
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
    if (dc[DC_HAS_DAMAGE])  // bindings/mettagrid_c.cpp:642-650: after regeneration, agent order
      for (int ai = 0; ai < a.A; ai++) mg_check_damage(cx, ai, rng);
    l.flags[0] = rng.used;
    l.flags[1] = rng.overflow | (cx.overflow ? 8 : 0);
    l.flags[2] = cx.asm_dirty;
  }
  __syncthreads();
  for (int i = tid; i < a.A; i += nt) a.exec[(size_t)e * a.A + i] = l.exec[i];
  mg_flush_stats(a, l, cx, e, tid, nt);
  {
    uint4* dst = (uint4*)(a.agents + (size_t)e * a.A);
    const uint4* src = (const uint4*)l.ag;
    for (int i = tid; i < a.A * (int)(sizeof(DAgent) / 16); i += nt) dst[i] = src[i];
    if (l.flags[2] || do_reset) {
      uint4* d2 = (uint4*)(a.asms + (size_t)e * a.NAmax);
      const uint4* s2 = (const uint4*)l.as;
      for (int i = tid; i < a.NAmax; i += nt) d2[i] = s2[i];
    }
    {
      uint4* d3 = (uint4*)(a.chests + (size_t)e * a.NCmax);
      const uint4* s3 = (const uint4*)l.ch;
      for (int i = tid; i < a.NCmax * (int)(sizeof(DAgent) / 16); i += nt) d3[i] = s3[i];
    }
    uint32_t new_idx = env.mt_idx + (uint32_t)l.flags[0];
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
}

// ------------------------------------------------------------------------------------------------------------
// kernel 2: observations + rewards + episode flags.  One workgroup (4 waves) per environment; each wave encodes
// whole agents' token rows in LDS and streams them to HBM (bindings/mettagrid_c.cpp:652-677).
// ------------------------------------------------------------------------------------------------------------
// BIG: grid tile and agent records are read in HBM (through L2); the token lists still live in LDS.
template <bool DCL, bool BIG>
__global__ __launch_bounds__(MG_THREADS, BIG ? 1 : MG_OBS_WAVES) void mg_obs_kernel(KArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, nt = blockDim.x, e = blockIdx.x + a.env_base;
  Lds l;
  mg_carve_obs<true>(smem, &l, BIG ? 0 : a.HW, a.A, a.NAmax, a.ROWB, nt >> 6, DCL ? a.dc_obs_words : 0, a.NOBJ, a.HEAP,
               a.dc[DC_PATTERN_N], BIG ? 0 : a.A, a.NCmax);
  if (BIG) {
    l.grid = a.big_grid + (size_t)e * a.HW;
    l.ag = a.agents + (size_t)e * a.A;
    l.ch = a.chests + (size_t)e * a.NCmax;
  }
  if (DCL) {
    for (int i = tid; i < (a.dc_obs_words >> 2); i += nt) ((uint4*)l.dc)[i] = ((const uint4*)a.dc)[i];  // padded to x4
    a.dc = l.dc;
  }
  MG_STAMP(0);
  DEnv env = a.env[e];
  mg_use_map(a, env.map_id);
  mg_load_grid(a, l, (int)env.map_id, tid, nt);
  {
    if (!BIG) {
      const uint4* src = (const uint4*)(a.agents + (size_t)e * a.A);
      uint4* dst = (uint4*)l.ag;
      for (int i = tid; i < a.A * (int)(sizeof(DAgent) / 16); i += nt) dst[i] = src[i];
      const uint4* s3 = (const uint4*)(a.chests + (size_t)e * a.NCmax);
      uint4* d3 = (uint4*)l.ch;
      for (int i = tid; i < a.NCmax * (int)(sizeof(DAgent) / 16); i += nt) d3[i] = s3[i];
    }
    const uint4* s2 = (const uint4*)(a.asms + (size_t)e * a.NAmax);
    uint4* d2 = (uint4*)l.as;
    for (int i = tid; i < a.NAmax; i += nt) d2[i] = s2[i];
    for (int i = tid; i < a.A; i += nt) l.exec[i] = a.exec[(size_t)e * a.A + i];
    const int4* ai_src = a.map_asm + (size_t)env.map_id * a.NAmax;
    for (int i = tid; i < a.NAmax; i += nt) l.ainfo[i] = ai_src[i];
    if (tid < 3) l.tokacc[tid] = a.gstats[(size_t)e * a.SG + tid];
  }
  __syncthreads();
  MG_STAMP(1);
  mg_overlay_agents(a, l, tid, nt);
  __syncthreads();
  MG_STAMP(2);
  if (!(a.diag & 32)) mg_build_tokens(a, l, (int)env.map_id, env.step, tid, nt);
  __syncthreads();
  MG_STAMP(3);
  if (!(a.diag & 16)) mg_encode_all(a, l, e, env.step, l.exec, true, tid, nt);
  __syncthreads();
  MG_STAMP(4);
  if (a.diag & 64) return;
  // The closing phases read their pointers from the kernel-argument segment again (an opaque pointer, so that the
  // loads are not merged with the ones at kernel entry): otherwise some thirty SGPRs stay live across the encode
  // loop for their sake and are spilled there.
  KArgs b = a;
#if defined(__HIP_DEVICE_COMPILE__)
  {
    const __attribute__((address_space(4))) KArgs* ka =
        (const __attribute__((address_space(4))) KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    b = *ka;
  }
#endif
  b.dc = a.dc; b.H = a.H; b.W = a.W;
  if (tid == 0) {
    mg_token_stats(b, l, 1);
    float* gs_ = b.gstats + (size_t)e * b.SG;
    gs_[0] = l.tokacc[0];
    gs_[1] = l.tokacc[1];
    gs_[2] = l.tokacc[2];
  }
  __syncthreads();
  mg_rewards_and_flags(b, l, e, env, tid, nt);
}

// Reset kernel: `MettaGrid` construction + `_init_buffers` for the listed environments.
template <bool BIG>
__global__ __launch_bounds__(MG_THREADS) void mg_reset_kernel(KArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
#ifdef MG_ISA_LITE  // ISA inspection builds (tools/isa.sh): keep only the kernel under study
  return;
#endif
  const int tid = threadIdx.x, nt = blockDim.x;
  const int e = a.reset_envs ? a.reset_envs[blockIdx.x] : (int)blockIdx.x;
  Lds l;
  mg_carve<true>(smem, &l, BIG ? 0 : a.HW, a.A, a.NAmax, a.SA, a.SG, a.NWIN, a.ROWB, nt >> 6, 0, a.NOBJ, a.HEAP,
           a.dc[DC_PATTERN_N], BIG ? 0 : a.A, a.NCmax);
  if (BIG) {
    l.grid = a.big_grid + (size_t)e * a.HW;
    l.ag = a.agents + (size_t)e * a.A;
    l.ch = a.chests + (size_t)e * a.NCmax;
  }
  DEnv env = a.env[e];
  env.seed = a.reset_seeds[blockIdx.x];
  env.map_id = (uint32_t)a.reset_maps[blockIdx.x];
  env.step = 0;
  env.mt_idx = 624;
  env.done = 0;
  env.error = 0;
  Ctx cx;
  mg_use_map(a, env.map_id);
  mg_make_ctx(a, l, cx, e, (int)env.map_id, 0);
  mg_load_grid(a, l, (int)env.map_id, tid, nt);
  mg_reset_env(a, l, cx, e, (int)env.map_id, env.seed, tid, nt);
  mg_overlay_agents(a, l, tid, nt);
  {
    const int4* ai_src = a.map_asm + (size_t)env.map_id * a.NAmax;
    for (int i = tid; i < a.NAmax; i += nt) l.ainfo[i] = ai_src[i];
  }
  __syncthreads();
  mg_build_tokens(a, l, (int)env.map_id, 0, tid, nt);
  __syncthreads();
  mg_encode_all(a, l, e, 0, nullptr, true, tid, nt);
  __syncthreads();
  if (tid == 0) {
    mg_token_stats(a, l, a.dc[DC_INIT_PASSES]);
    float* gs_ = a.gstats + (size_t)e * a.SG;
    gs_[0] = l.tokacc[0];
    gs_[1] = l.tokacc[1];
    gs_[2] = l.tokacc[2];
    atomicOr(&a.gtouched[(size_t)e * a.GW], 7u);
  }
  __syncthreads();
  mg_flush_stats(a, l, cx, e, tid, nt);
  if (!BIG) {
    uint4* dst = (uint4*)(a.agents + (size_t)e * a.A);
    const uint4* src = (const uint4*)l.ag;
    for (int i = tid; i < a.A * (int)(sizeof(DAgent) / 16); i += nt) dst[i] = src[i];
    uint4* d3 = (uint4*)(a.chests + (size_t)e * a.NCmax);
    const uint4* s3 = (const uint4*)l.ch;
    for (int i = tid; i < a.NCmax * (int)(sizeof(DAgent) / 16); i += nt) d3[i] = s3[i];
  }
  uint4* d2 = (uint4*)(a.asms + (size_t)e * a.NAmax);
  const uint4* s2 = (const uint4*)l.as;
  for (int i = tid; i < a.NAmax; i += nt) d2[i] = s2[i];
  if (tid == 0) a.env[e] = env;
}

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
  if (bits & ~13u) t += "unknown error bits " + std::to_string(bits & ~13u) + "; ";
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
  size_t lds_bytes = 0, lds_act = 0, lds_obs = 0, lds_uses = 0;
  bool dc_lds = false;
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
const char* mg_version(void) { return "mettagrid_hip 0.1 (gfx950; blob v" "3" ")"; }

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
  if (c.R > MGD_MAX_R) return fail("mg_create: more than 16 resources are not supported on the device path yet");
  bool has_damage = false;
  for (const auto& g : c.groups)
    if (!g.damage_threshold.empty() && !g.damage_resources.empty()) has_damage = true;
  if (c.A > 8191) return fail("mg_create: too many agents per environment");
  for (const auto& a : c.assemblers) {
    for (const auto& p : a.protocols)
      if (p.vibes.size() > 8) return fail("mg_create: protocols with more than 8 vibes are not supported");
  }
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
  {
    // fast (wave-parallel) act kernel eligibility: see mg_act_fast_kernel
    bool ok = c.A <= 64 && c.regen_interval == 0 && !has_damage && c.chests.empty();
    for (int v : c.vibe_handler) ok = ok && v == 0;
    for (int hnd = 0; hnd < MG_NUM_HANDLERS; hnd++) ok = ok && c.handler[hnd].required.empty() && c.handler[hnd].consumed.empty();
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
  k.dc_words = (int)dcfg.size();
  k.dc_obs_words = dcfg[DC_OBS_WORDS];
  size_t N = (size_t)num_envs * c.A;
  if (dalloc(h.get(), &k.env, (size_t)num_envs)) return 1;
  if (dalloc(h.get(), &k.mt, (size_t)num_envs * 624)) return 1;
  if (dalloc(h.get(), &k.mt_init, (size_t)num_envs * 624)) return 1;
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
  int TSW = 1, TSA = 1, TSS = 1;
  for (const auto& wl : c.walls) TSW = std::max(TSW, (int)wl.tags.size() + 1);
  for (const auto& g : c.groups) TSA = std::max(TSA, 3 + c.R * c.n_inv_tokens + (int)g.tags.size());
  for (const auto& a : c.assemblers) {
    int io = 0;
    for (const auto& p : a.protocols) io = std::max(io, (int)p.inputs.size() + (int)p.outputs.size());
    TSS = std::max(TSS, 3 + io + (int)a.tags.size() + 1);
  }
  int TSC = 1;
  for (const auto& ch : c.chests) TSC = std::max(TSC, 1 + c.R * c.n_inv_tokens + (int)ch.tags.size());  // chest.hpp:124-146
  if (TSW > 255 || TSA > 255 || TSS > 255 || TSC > 255) return fail("mg_set_maps: an object can show more than 255 tokens");
  TSW = (TSW + 3) & ~3; TSA = (TSA + 3) & ~3; TSS = (TSS + 3) & ~3; TSC = (TSC + 3) & ~3;  // lists are read 4 tokens (8 bytes) at a time
  const int NOBJ = (int)c.walls.size() + c.A + NAmax + NCmax;
  const int HEAP = (int)c.walls.size() * TSW + c.A * TSA + NAmax * TSS + NCmax * TSC;
  const int NP = c.obs_h * c.obs_w;
  k.NW = (int)c.walls.size(); k.NOBJ = NOBJ; k.TSW = TSW; k.TSA = TSA; k.TSS = TSS; k.TSC = TSC; k.NCmax = NCmax; k.HEAP = HEAP;
  {
    int mx = std::max(TSW, std::max(TSA, TSS));
    k.CB = 1;
    while ((1 << k.CB) <= mx) k.CB++;
  }
  if (HEAP > 65535) return fail("mg_set_maps: token lists of one environment exceed 65535 entries");
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
  if (big) {
    uint16_t* bg = nullptr;
    if (dalloc(h, &bg, (size_t)std::max(h->E, num_maps) * HWp, true)) return 1;
    k.big_grid = bg;
  }
  const int hw_lds = big ? 0 : HWp, a_rec_obs = big ? 0 : c.A;
  h->dc_lds = !big && k.dc_words <= MG_DC_LDS_MAX_WORDS;
  int dcw = h->dc_lds ? k.dc_words : 0;
  int dcw_obs = h->dc_lds ? k.dc_obs_words : 0;
  h->lds_act = mg_carve<false>(nullptr, nullptr, hw_lds, c.A, NAmax, k.SA, k.SG, k.NWIN, k.ROWB, 0, dcw, 0, 0, 0, -1, NCmax);
  h->lds_obs = mg_carve_obs<false>(nullptr, nullptr, hw_lds, c.A, NAmax, k.ROWB, MG_THREADS / 64, dcw_obs, NOBJ, HEAP, NP, a_rec_obs, NCmax);
  if (h->lds_act > LDS_MAX || h->lds_obs > LDS_MAX) {
    h->dc_lds = false;
    h->lds_act = mg_carve<false>(nullptr, nullptr, hw_lds, c.A, NAmax, k.SA, k.SG, k.NWIN, k.ROWB, 0, 0, 0, 0, 0, -1, NCmax);
    h->lds_obs = mg_carve_obs<false>(nullptr, nullptr, hw_lds, c.A, NAmax, k.ROWB, MG_THREADS / 64, 0, NOBJ, HEAP, NP, a_rec_obs, NCmax);
  }
  if (h->lds_act > LDS_MAX || h->lds_obs > LDS_MAX)
    return fail("mg_set_maps: environment does not fit the 160 KiB LDS of one CU (act " + std::to_string(h->lds_act) +
                ", obs " + std::to_string(h->lds_obs) + " bytes)");
  h->lds_uses = mg_carve_uses<false>(nullptr, nullptr, c.A, NAmax, k.SG, h->dc_lds ? k.dc_words : 0);
  if (getenv("MG_VERBOSE"))
    fprintf(stderr, "mettagrid_hip: LDS bytes per workgroup: act %zu, obs %zu, uses %zu (dc words %d, obs prefix %d, heap %d tokens, %d objects)\n",
            h->lds_act, h->lds_obs, h->lds_uses, k.dc_words, k.dc_obs_words, HEAP, NOBJ);
  {
    // the cap on dynamic LDS is a per-function setting shared by every handle of the process: raise it once
    const int cap = (int)LDS_MAX;
    HIPCHK(hipFuncSetAttribute((const void*)mg_act_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    HIPCHK(hipFuncSetAttribute((const void*)mg_act_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    HIPCHK(hipFuncSetAttribute((const void*)mg_act_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    HIPCHK(hipFuncSetAttribute((const void*)mg_uses_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    HIPCHK(hipFuncSetAttribute((const void*)mg_uses_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    HIPCHK(hipFuncSetAttribute((const void*)mg_obs_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    HIPCHK(hipFuncSetAttribute((const void*)mg_obs_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    HIPCHK(hipFuncSetAttribute((const void*)mg_obs_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    HIPCHK(hipFuncSetAttribute((const void*)mg_reset_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    HIPCHK(hipFuncSetAttribute((const void*)mg_reset_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
  }
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
    std::vector<void*> tmp;
    auto talloc = [&](size_t bytes) -> void* {
      void* q = nullptr;
      if (hipMalloc(&q, bytes ? bytes : 1) != hipSuccess) return nullptr;
      (void)hipMemset(q, 0, bytes ? bytes : 1);
      tmp.push_back(q);
      return q;
    };
    KArgs kt = k;
    kt.E = (int)P;
    kt.env = (DEnv*)talloc(P * sizeof(DEnv));
    kt.mt = (uint32_t*)talloc(P * 624 * 4);
    kt.mt_init = nullptr;
    kt.chests = t_chests;
    kt.agents = t_agents; kt.asms = t_asms; kt.astats = t_astats; kt.atouched = t_atouched;
    kt.gstats = t_gstats; kt.gtouched = t_gtouched;
    kt.obs = (uint8_t*)talloc(PA * c.T * 3);
    kt.rewards = (float*)talloc(PA * 4);
    kt.ep_rewards = (float*)talloc(PA * 4);
    kt.term = (uint8_t*)talloc(PA);
    kt.trunc = (uint8_t*)talloc(PA);
    kt.success = (uint8_t*)talloc(PA);
    uint32_t* d_seeds = (uint32_t*)talloc(P * 4);
    int* d_maps = (int*)talloc(P * 4);
    bool okalloc = kt.env && kt.mt && kt.obs && kt.rewards && kt.ep_rewards && kt.term && kt.trunc && kt.success && d_seeds && d_maps;
    int rc = 0;
    if (!okalloc) {
      rc = fail("mg_set_maps: out of device memory while building episode templates");
    } else {
      std::vector<int> iota((size_t)P);
      for (size_t i = 0; i < P; i++) iota[i] = (int)i;
      hipError_t e1 = hipMemcpy(d_maps, iota.data(), P * 4, hipMemcpyHostToDevice);
      kt.reset_envs = nullptr;
      kt.reset_seeds = d_seeds;
      kt.reset_maps = d_maps;
      if (h->big) hipLaunchKernelGGL(mg_reset_kernel<true>, dim3((unsigned)P), dim3(MG_THREADS), h->lds_bytes, 0, kt);
      else hipLaunchKernelGGL(mg_reset_kernel<false>, dim3((unsigned)P), dim3(MG_THREADS), h->lds_bytes, 0, kt);
      hipError_t e2 = hipGetLastError();
      hipError_t e3 = hipDeviceSynchronize();
      if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess)
        rc = fail(std::string("mg_set_maps: template build failed: ") + hipGetErrorString(e3 != hipSuccess ? e3 : (e2 != hipSuccess ? e2 : e1)));
    }
    for (void* q : tmp) (void)hipFree(q);
    if (rc) return rc;
    k.t_chests = t_chests;
    k.t_agents = t_agents; k.t_asms = t_asms; k.t_astats = t_astats; k.t_atouched = t_atouched;
    k.t_gstats = t_gstats; k.t_gtouched = t_gtouched;
  }
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
  if (h->big) hipLaunchKernelGGL(mg_reset_kernel<true>, dim3((unsigned)n), dim3(MG_THREADS), h->lds_bytes, s, k);
  else hipLaunchKernelGGL(mg_reset_kernel<false>, dim3((unsigned)n), dim3(MG_THREADS), h->lds_bytes, s, k);
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
    if (k.fast) {
      hipLaunchKernelGGL(mg_act_fast_kernel, dim3(n), dim3(MG_FAST_THREADS), 0, st, kg);
      if (h->dc_lds) hipLaunchKernelGGL(mg_uses_kernel<true>, dim3(n), dim3(MG_USES_THREADS), h->lds_uses, st, kg);
      else hipLaunchKernelGGL(mg_uses_kernel<false>, dim3(n), dim3(MG_USES_THREADS), h->lds_uses, st, kg);
    } else {
      if (h->big) hipLaunchKernelGGL((mg_act_kernel<false, true>), dim3(n), dim3(MG_ACT_THREADS), h->lds_act, st, kg);
      else if (h->dc_lds) hipLaunchKernelGGL((mg_act_kernel<true, false>), dim3(n), dim3(MG_ACT_THREADS), h->lds_act, st, kg);
      else hipLaunchKernelGGL((mg_act_kernel<false, false>), dim3(n), dim3(MG_ACT_THREADS), h->lds_act, st, kg);
    }
    if (obs_only) (void)hipEventRecord(h->ev[2 * slot], s);
    if (h->big) hipLaunchKernelGGL((mg_obs_kernel<false, true>), dim3(n), dim3(MG_THREADS), h->lds_obs, st, kg);
    else if (h->dc_lds) hipLaunchKernelGGL((mg_obs_kernel<true, false>), dim3(n), dim3(MG_THREADS), h->lds_obs, st, kg);
    else hipLaunchKernelGGL((mg_obs_kernel<false, false>), dim3(n), dim3(MG_THREADS), h->lds_obs, st, kg);
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
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  for (int i = 0; i < 7; i++) {
    if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    if (h->aux[i]) (void)hipStreamDestroy(h->aux[i]);
  }
  delete h;
  return 0;
}

}  // extern "C"
