// mg_kernels.h -- shared device code of the step kernels (arguments, LDS layout, encoder, restart, statistics) and the
// host-side launchers of each kernel translation unit.  The kernels are split over several .hip files so that they
// compile in parallel (make -j) and a change to one kernel rebuilds one file.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "mg_device_cfg.h"
#include "mg_sim.h"

#define MG_THREADS 256
#define MG_OBS_BIG_THREADS 1024
#ifndef MG_OBS_WAVES
#define MG_OBS_WAVES 8
#endif
#define MG_WIN_MARGIN 32
#define MG_STAMP(slot)                                                     \
  do {                                                                    \
    if (a.dbg && threadIdx.x == 0) a.dbg[((size_t)blockIdx.x + a.env_base) * 16 + (slot)] = clock64(); \
  } while (0)
#define MG_DC_LDS_MAX_WORDS 6144  // 24 KiB

// LDS layout of the fused kernel (k_fused.hip), byte offsets from the start of the workgroup's dynamic LDS.  Computed
// once on the host (mg_carve_fused) and handed over in the kernel arguments: a phase of the kernel reads the few
// offsets it uses instead of redoing the whole carve (scalar instructions and live scalar registers, five times per step).
struct FusedLds {
  int dc, ag, rows, u, heap, ws, ip, ex, ta, tk, gd, da, dm, ord, fl, su, WS;
};

struct KArgs {
  const int* dc;
  const int* gdc;    // the whole config stream in HBM (dc may be re-pointed at an LDS copy of its prefix)
  int dc_obs_words;  // prefix of the stream the observation kernel stages
  int dc_words;  // size of the device config stream; staged in LDS when it fits (MG_DC_LDS_MAX_WORDS)
  const uint16_t* map_cells;  // [P][HW]
  const uint8_t* map_cells8;  // [P][HW8] the same codes as bytes when every object code fits one (fused kernel), else null
  int HW8;                    // byte stride of one map in map_cells8 (multiple of 16)
  int PADC;                   // map_cells8 rows are W + PADC bytes long, the PADC bytes after each row zero (= obs_w / 2): a
                              // window cell left or right of the map lands on padding and reads "empty" without a test
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
  int GPAD, BHW;              // BIG: rows of big_grid are W + GPAD cells long, the GPAD cells after each row zero (= obs_w / 2,
                              // as PADC for the byte maps); BHW = cell stride of one environment's grid.  0 / HW otherwise
  uint8_t* grid_valid;        // [E] BIG: big_grid[e] holds the running episode's map with the agents where they stand now
                              // (the action kernel moves them in place), so a step neither copies the map nor overlays
                              // the agents again; cleared by whatever replaces the pool, null for template builds
  const int2* map_dims;       // [P] (height, width) of each map; H, W below are the current environment's
  const int* map_max_steps;   // [P] episode length of each map (0 = the configuration's), or null: every map the configuration's
  int max_steps, max_steps_cfg;  // the current environment's / the configuration's max_steps
  // Per-slot configurations (curriculum tasks: a task is a (configuration, map) pair): map slot m plays the stream at
  // gdc + map_cfg[m] * dc_stride; null = one configuration for the whole pool.  dc_words / dc_obs_words are then the
  // largest over the slots (LDS staging copies that many words from the slot's base; every slot is dc_stride long).
  const int* map_cfg;            // [P] configuration slot of each map slot, or null
  int dc_stride;
  int P, H, W, HW, NAmax, n_cells;  // HW = cell stride of one map in the pool (the largest map, even)
  int NW, NOBJ;            // wall types, objects per env (NW + A + NAmax)
  int NWINCELLS;           // cells of the observation window (obs_h * obs_w)
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
  // Finished episodes in the order they ended: entry (count & mask) = (environment, map slot of the finished episode),
  // written when the environment restarts.  The host reads the entries between two values of the counter
  // (mg_finished_ring) instead of scanning every environment's episode counter.
  uint2* fin_ring;
  uint32_t fin_ring_mask;
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
  const int* reset_count;  // or null: blocks at or past *reset_count leave at once (device-side list lengths)
  // Clipper (systems/clipper.hpp), null without one: its own std::mt19937 per environment (a copy of the engine as
  // seeded, mettagrid_c.cpp:153), (next state word, length scale) and the infection weight of every station
  // (bit 31: the clipper clipped it and has not seen it unclipped yet)
  uint32_t* clip_mt;   // [E][624]
  uint2* clip_state;   // [E]
  uint32_t* clip_w;    // [E][NAmax]
  // Static token lists of each map's assemblers (fused kernel): what a station shows while idle and while cooling
  // down never changes during an episode except for the cooldown value, so the template build (reset kernel) keeps both
  // lists per station and a step copies one instead of walking the configuration.  t_asm_n = tokens of the idle list
  // (the cooling list has one more: cooldown_remaining first), 255 = not static (max_uses, clipped, a protocol that
  // depends on the neighbours' vibes): built the long way every step.
  uint16_t* t_asm_tok;  // [P][NAmax][2][TSS]
  uint8_t* t_asm_n;     // [P][NAmax]
  int build_templates;  // reset kernel: this launch builds the per-map templates (writes the two arrays above)
  FusedLds fl;         // fused kernel only: its LDS layout for this launch (two-wave or one-wave build, with / without the config prefix)
};

struct Lds {
  int* dc;
  int* tmp;  // [8][96] scratch ints: one slot per concurrently acting lane
  uint16_t* grid;
  DAgent* ag;
  DAgent* ch;       // chest records (directly after the agent records)
  int* gsetv;       // [MGD_MAX_R] pending chest.<r>.amount values; gsetm[0] = which
  uint32_t* gsetm;
  unsigned long long* gset64;  // [MGD_MAX_R] the same as sequence-keyed maxima (parallel action rounds), or null
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
  uint8_t* pend;    // fused kernel: [A][32] assembler uses found by the action phase
  uint8_t* succ;    // fused kernel: [A] action_success of this step
  uint8_t* wscr;    // fused kernel: [waves][2][WS] agents inside the window being encoded (agent index + 1 per window cell)
  uint8_t* ipat;    // fused kernel: [obs_h * obs_w] window cell -> position in the pattern order
  const uint8_t* gmap;  // fused kernel: the byte map of this environment in the HBM pool (static layer)
  int WS;
  uint32_t* pos;    // observation kernel with the records in HBM (BIG): [A] r | c << 16 of every agent
};

__host__ __device__ inline size_t mg_align16(size_t x) { return (x + 15) & ~(size_t)15; }

#define MG_FUSED_SLOTS 4  // fused kernel: assembler uses resolved concurrently (scratch slots of 96 ints)
// The fused kernel's LDS plan; returns the bytes one workgroup needs.  About 10 KB per environment (arena): nothing
// that other workgroups can share is copied into LDS (see k_fused.hip).
__host__ __device__ inline size_t mg_carve_fused(FusedLds* f, int A, int SG, int ROWB, int nwaves, int dcw, int NOBJ, int HEAP,
                                                 int NP) {
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = mg_align16(o + bytes);
    return at;
  };
  size_t dco = take((size_t)dcw * 4);
  size_t ag = take((size_t)A * sizeof(DAgent));
  // row buffers (encode) <- aliases -> pending assembler uses (P1-P2)
  size_t rows_b = (size_t)ROWB * nwaves;
  if (rows_b < (size_t)A * 32) rows_b = (size_t)A * 32;
  size_t rows = take(rows_b);
  // token lists + per-object counts (P3) <- aliases -> use scratch (P2) <- aliases -> next MT19937 block (P0-P1)
  size_t oc_b = mg_align16((size_t)(NOBJ + 1) * 4), hp_b = mg_align16((size_t)HEAP * 2);
  size_t u_b = oc_b + hp_b;
  if (u_b < (size_t)MG_FUSED_SLOTS * 96 * 4) u_b = (size_t)MG_FUSED_SLOTS * 96 * 4;
  if (u_b < 624 * 4) u_b = 624 * 4;
  size_t u = take(u_b);
  const int WS = NP <= 128 ? 128 : ((NP + 63) & ~63);  // (two full rounds: the encoder reads them without a bounds test)
  size_t ws = take((size_t)nwaves * 2 * WS), ip = take((size_t)NP + 64);  // (ip: (h | 1) x (w | 1) <= h w + h + w + 1 bytes, windows are at most 15 x 15)
  size_t ex = take((size_t)A * 4), ta = take((size_t)A * 4), tk = take(16);
  size_t gd = take((size_t)SG * 4), da = take((size_t)A * 4), dm = take((size_t)A * 4), ord = take((size_t)A);
  size_t fl = take(32), su = take((size_t)A);
  if (f) {
    f->dc = (int)dco; f->ag = (int)ag; f->rows = (int)rows; f->u = (int)u; f->heap = (int)(u + oc_b);
    f->ws = (int)ws; f->ip = (int)ip; f->ex = (int)ex; f->ta = (int)ta; f->tk = (int)tk; f->gd = (int)gd;
    f->da = (int)da; f->dm = (int)dm; f->ord = (int)ord; f->fl = (int)fl; f->su = (int)su; f->WS = WS;
  }
  return o;
}

// Maps of one pool may differ in size (navigation suite): the kernel's copy of the arguments takes the dimensions of
// the map its environment plays.
__device__ __forceinline__ void mg_use_map(KArgs& a, uint32_t map_id) {
  const int2 d = a.map_dims[map_id];
  a.H = d.x;
  a.W = d.y;
  int ms = a.max_steps_cfg;
  if (a.map_cfg) {  // (before any use of a.dc: the callers stage / capture the stream after this call)
    const int* base = a.gdc + (size_t)a.map_cfg[map_id] * (size_t)a.dc_stride;
    a.gdc = base;
    a.dc = base;
    ms = base[DC_MAX_STEPS];
  }
  if (a.map_max_steps) {
    const int m = a.map_max_steps[map_id];
    if (m > 0) ms = m;
  }
  a.max_steps = ms;
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
  size_t gsv = take(MGD_MAX_R * 4 + 16);
  size_t gd = take((size_t)SG * 4), da = take((size_t)A * 4), dm = take((size_t)A * 4);
  size_t ord = take((size_t)A * 2), ex = take((size_t)A * 4), ta = take((size_t)A * 4), win = take((size_t)NWIN * 4);
  size_t mtn = take(624 * 4), tk = take(16), fl = take(16), rows = take((size_t)ROWB * nwaves);
  size_t oc = take(nwaves ? (size_t)(NOBJ + 1) * 4 : 0), hp = take(nwaves ? (size_t)HEAP * 2 : 0), pt = take(nwaves ? (size_t)NP * 4 : 0);
  size_t ai = take(nwaves ? (size_t)(NA > 0 ? NA : 1) * 16 : 0);
  if (FILL) {
    l->ocnt = (uint32_t*)(base + oc) + 1; l->heap = (uint16_t*)(base + hp); l->pat = (uint32_t*)(base + pt);
    l->ainfo = (int4*)(base + ai);
    l->grid = (uint16_t*)(base + g); l->ag = (DAgent*)(base + ag); l->as = (DAsm*)(base + as);
    l->ch = l->ag + a_rec; l->gsetv = (int*)(base + gsv); l->gsetm = (uint32_t*)(base + gsv + MGD_MAX_R * 4); l->gset64 = nullptr;
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
  size_t ps = take(a_rec == 0 ? (size_t)A * 4 : 0);  // (records in HBM: the positions alone are staged)
  if (FILL) l->pos = a_rec == 0 ? (uint32_t*)(base + ps) : nullptr;
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
  if (a.GPAD) {  // a grid kept in HBM: rows padded with GPAD empty cells (window lookups need no column test)
    const int Wp = a.W + a.GPAD, n = a.H * Wp;
    for (int i = tid; i < n; i += nt) {
      const int r = i / Wp, c = i - r * Wp;
      l.grid[i] = c < a.W ? src[r * a.W + c] : (uint16_t)0;
    }
    return;
  }
  int n2 = a.HW >> 1;
  const uint32_t* s32 = (const uint32_t*)src;  // HW*2 bytes per map; maps start 4-byte aligned (HW even or padded)
  uint32_t* d32 = (uint32_t*)l.grid;
  if ((a.HW & 1) == 0) {
    for (int i = tid; i < n2; i += nt) d32[i] = s32[i];
  } else {
    for (int i = tid; i < a.HW; i += nt) l.grid[i] = src[i];
  }
}

// CT = cell type of the LDS grid tile: uint16_t, or uint8_t where every object code fits a byte (fused kernel)
template <class CT = uint16_t>
__device__ inline void mg_overlay_agents(const KArgs& a, const Lds& l, int tid, int nt) {
  const CellCodec cc{a.NW, a.A, a.NAmax};
  CT* grid = (CT*)l.grid;
  for (int i = tid; i < a.A; i += nt) grid[(int)l.ag[i].r * (a.W + a.GPAD) + l.ag[i].c] = (CT)cc.agent(i);
}

__device__ __forceinline__ int mg_heap_off(const KArgs& a, int obj) {
  return obj < a.NW ? obj * a.TSW
                    : (obj < a.NW + a.A
                           ? a.NW * a.TSW + (obj - a.NW) * a.TSA
                           : (obj < a.NW + a.A + a.NAmax
                                  ? a.NW * a.TSW + a.A * a.TSA + (obj - a.NW - a.A) * a.TSS
                                  : a.NW * a.TSW + a.A * a.TSA + a.NAmax * a.TSS + (obj - a.NW - a.A - a.NAmax) * a.TSC));
}

// Pattern entry p: (dr+8) | (dc+8) << 8 | location byte << 16
// (`gdc` = the stream in HBM: the pattern lies behind the LDS-staged prefix)
__device__ __forceinline__ uint32_t mg_pattern_entry(const int* gdc, int p) {
  const int* dc = gdc;
  const int hr = dc[DC_OBS_H] >> 1, wr = dc[DC_OBS_W] >> 1;
  const uint32_t packed = (uint32_t)dc[dc[DC_OFF_PATTERN] + p];
  const int dr = (int)(packed & 0xFF) - 8, dcc = (int)((packed >> 8) & 0xFF) - 8;
  return packed | ((uint32_t)(((dr + hr) << 4) | ((dcc + wr) & 0x0F)) << 16);
}

// Build every object's token list once (they are observer independent) and the pattern table.
// Requires l.ainfo (the map's static assembler info) staged + a barrier before.
template <class CT = uint16_t, bool PAT = true>
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
      mg_object_tokens(dc, (const CT*)l.grid, l.ag, l.as, asm_info, cc, a.H, a.W, a.W + a.GPAD, step, (uint16_t)(obj + 1), out, l.ch, chest_info);
    l.ocnt[obj] = (uint32_t)ho | ((uint32_t)out.n << 16);
  }
  if (tid == 0) l.ocnt[-1] = 0;  // cell code 0 (empty) looks up "no tokens" without a branch
  if (PAT) {
    const int P = dc[DC_PATTERN_N];
    for (int p = tid; p < P; p += nt) l.pat[p] = mg_pattern_entry(a.gdc, p);
  }
}

// The fused kernel's token lists: walls and agents as above; assemblers from the per-map static lists where they
// apply (KArgs::t_asm_tok) -- a 16-byte record load, a list copy and the cooldown value instead of the walk through
// the configuration stream (assembler.hpp:528-578), which was a tenth of the step's instructions.
__device__ inline void mg_build_tokens_fused(const KArgs& a, const Lds& l, int map_id, uint32_t step, int tid, int nt) {
  const int* dc = a.dc;
  const CellCodec cc{a.NW, a.A, a.NAmax};
  const int nasm = a.map_nasm[map_id];
  const int first_asm = a.NW + a.A;
  const uint32_t f_cd = (uint32_t)dc[DC_FID0 + MG_F_COOLDOWN] & 0xFFu;
  for (int obj = tid; obj < a.NOBJ; obj += nt) {
    const int ho = mg_heap_off(a, obj);
    int n = 0;
    bool done = obj >= first_asm + nasm;  // (unused assembler slots: no tokens)
    if (!done && obj >= first_asm) {
      const int idx = obj - first_asm;
      const int ns = a.t_asm_n[(size_t)map_id * a.NAmax + idx];
      const DAsm s = l.as[idx];
      if (ns != 255 && s.clipped == 0) {
        uint32_t remaining = s.cooldown_end <= step ? 0 : s.cooldown_end - step;
        if (remaining > 255u) remaining = 255u;
        const int variant = remaining > 0 ? 1 : 0;
        n = ns + variant;
        const uint2* __restrict__ src = (const uint2*)(a.t_asm_tok + (((size_t)map_id * a.NAmax + idx) * 2 + variant) * a.TSS);
        uint2* dst = (uint2*)(l.heap + ho);
        uint2 v = src[0];
        if (variant) v.x = (v.x & 0xFFFF0000u) | f_cd | (remaining << 8);
        dst[0] = v;
        for (int k = 1; k * 4 < n; k++) dst[k] = src[k];
        done = true;
      }
    }
    if (!done) {
      PairOut out{l.heap + ho, 0};
      mg_object_tokens(dc, (const uint8_t*)nullptr, l.ag, l.as, l.ainfo, cc, a.H, a.W, a.W + a.GPAD, step, (uint16_t)(obj + 1), out, l.ch, a.map_chest);
      n = out.n;
    }
    l.ocnt[obj] = (uint32_t)ho | ((uint32_t)n << 16);
  }
  if (tid == 0) l.ocnt[-1] = 0;  // cell code 0 (empty) looks up "no tokens" without a branch
}

// Template build (reset kernel): the static lists of map slot `m`'s assemblers, see KArgs::t_asm_tok.
template <class CT>
__device__ inline void mg_build_asm_templates(const KArgs& a, const Lds& l, int m, int map_id, int tid, int nt) {
  const int* dc = a.dc;
  const CellCodec cc{a.NW, a.A, a.NAmax};
  const int nasm = a.map_nasm[map_id];
  for (int i = tid; i < a.NAmax; i += nt) {
    uint16_t* dst = a.t_asm_tok + ((size_t)m * a.NAmax + i) * 2 * a.TSS;
    for (int k = 0; k < 2 * a.TSS; k++) dst[k] = 0;
    int n = 255;
    if (i < nasm) {
      const int* rec = asm_rec(dc, l.ainfo[i].z);
      const bool is_static = rec[AR_MAX_USES] == 0 && l.as[i].clipped == 0 && (dc[DC_PROTO_OBS] == 0 || rec[AR_SIMPLE_PO] >= 0);
      if (is_static) {
        DAsm tmp = {0u, 0u, 0u, 0u};
        const DAsm* fake = &tmp - i;  // (mg_object_tokens indexes the record array by the station's index)
        PairOut idle{dst, 0};
        mg_object_tokens(dc, (const CT*)l.grid, l.ag, fake, l.ainfo, cc, a.H, a.W, a.W + a.GPAD, 0u, cc.assembler(i), idle);
        tmp.cooldown_end = 1u;  // step 0: one tick left -- the value is patched in every step
        PairOut cooling{dst + a.TSS, 0};
        mg_object_tokens(dc, (const CT*)l.grid, l.ag, fake, l.ainfo, cc, a.H, a.W, a.W + a.GPAD, 0u, cc.assembler(i), cooling);
        n = (cooling.n == idle.n + 1 && idle.n < 255) ? idle.n : 255;
      }
    }
    a.t_asm_n[(size_t)m * a.NAmax + i] = (uint8_t)n;
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
    const int ms = a.max_steps;  // (the map's, where the pool carries per-map episode lengths)
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
template <int NR, bool HOT, class CT = uint16_t>
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
    const uint32_t cv = ((const CT*)l.grid)[ok ? __mul24(r, W + a.GPAD) + c : 0];
    meta[q] = l.ocnt[(int)(ok ? cv : 0u) - 1];
  }
}

// One map of the byte pool behind a buffer descriptor: `buffer_load_ubyte` takes the cell offset as it is (scalar base
// in the descriptor, no address arithmetic on the vector unit) and returns 0 for offsets outside the map, so cells
// outside it need no address of their own.
struct MgByteMap {
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t rsrc;
#else
  const uint8_t* p;
#endif
};
__device__ __forceinline__ MgByteMap mg_byte_map(const uint8_t* p, int bytes) {
  MgByteMap m;
#if defined(__HIP_DEVICE_COMPILE__)
  m.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);  // raw buffer, gfx9 data format word
#else
  m.p = p;
#endif
  return m;
}
__device__ __forceinline__ uint32_t mg_byte_at(const MgByteMap& m, uint32_t off) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(m.rsrc, (int)off, 0, 0);
#else
  return m.p[off];
#endif
}

// The same lookups without an occupancy tile in LDS (fused kernel): the static layer comes from the byte map of the
// pool in HBM (L2-resident: every environment playing the map reads the same few KB), and the agents of the environment
// that stand inside the window are scattered over the window cells through a small per-wave LDS buffer `ws` (all zero
// between uses).  `posj` = position of agent `lane` (r | c << 16), loaded once per wave.
template <int NR>
__device__ __forceinline__ void mg_window_lookup_direct(const KArgs& a, const Lds& l, const EncCfg& ec, const DAgent* me,
                                                        const int (&dlt)[MG_MAX_ROUNDS], int lane, uint32_t (&meta)[NR],
                                                        uint32_t& rc, const uint32_t posj, uint8_t* ws, const MgByteMap& gm,
                                                        const int Wp) {
  rc = *(const uint32_t*)me;
  const int ar = (int)(rc & 0xFFFF), ac = (int)(rc >> 16);
  // Rows of the byte map are Wp = W + PADC long with zero padding, and the buffer descriptor spans exactly the map's H
  // rows: a window cell left / right of the map reads padding, one above / below it lies outside the descriptor's
  // range (a negative offset is a huge unsigned one) and reads 0 -- "empty" either way, with no bounds test.  `dlt` is
  // the lane's window offset dr * Wp + dc (an out-of-range constant for lanes past the pattern).
  const int base = __mul24(ar, Wp) + ac;
  uint32_t cvs[NR];
#pragma unroll
  for (int q = 0; q < NR; q++) cvs[q] = mg_byte_at(gm, (uint32_t)(base + dlt[q]));
  const int jr = (int)(posj & 0xFFFF) - ar, jc = (int)(posj >> 16) - ac;
  const bool inwin = lane < a.A && jr >= -ec.hr && jr <= ec.hr && jc >= -ec.wr && jc <= ec.wr;
  int pj = 0;
  if (inwin) {
    pj = l.ipat[__mul24(jr + ec.hr, ec.ow | 1) + jc + ec.wr];  // (the table is (h | 1) x (w | 1): DC_OFF_IPAT)
    ws[pj] = (uint8_t)(lane + 1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  uint32_t aj[NR];
#pragma unroll
  for (int q = 0; q < NR; q++) {
    const int p = q * 64 + lane;
    // (the headline template reads two rounds: the buffer holds at least 128 entries, mg_carve_fused)
    aj[q] = (NR <= 2 || p < l.WS) ? ws[p] : 0u;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (inwin) ws[pj] = 0;
#pragma unroll
  for (int q = 0; q < NR; q++) {
    const uint32_t cv = aj[q] ? (uint32_t)a.NW + aj[q] : cvs[q];  // agents stand on empty cells of the static layer
    meta[q] = l.ocnt[(int)cv - 1];
  }
}

// Up to four tokens of one object -- v = two dwords of (feature | value << 8) pairs -- as the row bytes
// loc f v  loc f v ..., written with the widest stores that cover exactly 3 m bytes.  The row sits at any byte phase:
// LDS stores of 2 to 12 bytes need no alignment on this hardware (tools/microbench/unaligned_lds.hip checks every
// phase), and a token costs a fraction of the three byte stores it used to.  loc4 = the location byte, replicated.
typedef uint16_t mg_u16_unaligned __attribute__((aligned(1)));
typedef uint32_t mg_u32_unaligned __attribute__((aligned(1)));
struct __attribute__((packed)) MgPack8 { uint32_t a, b; };
struct __attribute__((packed)) MgPack12 { uint32_t a, b, c; };
__device__ __forceinline__ void mg_put_tokens(uint8_t* __restrict__ dst, const uint2 v, const uint32_t loc4, const int m) {
  const uint32_t d0 = (loc4 & 0xFF0000FFu) | ((v.x & 0xFFFFu) << 8);            // loc f0 v0 loc
  const uint32_t d1 = (v.x >> 16) | (loc4 & 0x00FF0000u) | (v.y << 24);          // f1 v1 loc f2
  const uint32_t d2 = ((v.y >> 8) & 0xFFu) | (loc4 & 0x0000FF00u) | (v.y & 0xFFFF0000u);  // v2 loc f3 v3
#ifdef MG_NARROW_TOKENS  // (A/B switch: byte stores)
  {
    const uint32_t t[4] = {v.x & 0xFFFF, v.x >> 16, v.y & 0xFFFF, v.y >> 16};
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k < m) { dst[k * 3] = (uint8_t)loc4; dst[k * 3 + 1] = (uint8_t)t[k]; dst[k * 3 + 2] = (uint8_t)(t[k] >> 8); }
    return;
  }
#endif
  if (m == 1) {
    *(mg_u16_unaligned*)dst = (uint16_t)d0;
    dst[2] = (uint8_t)(d0 >> 16);
  } else if (m == 2) {
    *(mg_u32_unaligned*)dst = d0;
    *(mg_u16_unaligned*)(dst + 4) = (uint16_t)d1;
  } else if (m == 3) {
    *(MgPack8*)dst = MgPack8{d0, d1};
    dst[8] = (uint8_t)d2;
  } else {
    *(MgPack12*)dst = MgPack12{d0, d1, d2};
  }
}

template <int NR, bool HOT = false>
__device__ __forceinline__ int mg_encode_agent(const KArgs& a, const Lds& l, const EncCfg& ec, const GlobalTok& gt,
                                      const DAgent* me, int action, uint8_t* __restrict__ row,
                                      const uint32_t (&loc4)[MG_MAX_ROUNDS], int lane, const uint32_t (&meta)[NR],
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
  // Two rounds share one scan: both counts ride in one register, 16 bits each.  (An object has at most one token per
  // feature id and one per tag id, under 512 in all; mg_set_maps refuses lists of more than 1000: 64 stay below 2^16.)
  uint32_t ex2 = 0, tot2 = 0;
  if (NR == 2) {
    const uint32_t c = (meta[0] >> 16) | (meta[NR - 1] & 0xFFFF0000u);
    int x = (int)c;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    ex2 = (uint32_t)x - c;
    tot2 = (uint32_t)__builtin_amdgcn_readlane(x, 63);
  }
#pragma unroll
  for (int q = 0; q < NR; q++) {
    const int cnt = (int)(meta[q] >> 16);
    // exclusive prefix sum of cnt over the wave: six DPP steps (row shifts, then the row broadcasts).  (A ballot +
    // mbcnt count of the single-token objects plus a uniform loop over the multi-token ones cost 4 us more per launch.)
    int off, total;
    if (NR == 2) {
      off = (int)(q == 0 ? (ex2 & 0xFFFFu) : (ex2 >> 16));
      total = (int)(q == 0 ? (tot2 & 0xFFFFu) : (tot2 >> 16));
    } else {
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
      const int n = min(cnt, T - off);
      uint8_t* __restrict__ dst = row + off * 3;
#ifdef MG_PUT_TOKENS  // (A/B: wide unaligned stores, or MG_NARROW_TOKENS byte stores through the same helper)
      mg_put_tokens(dst, src[0], loc4[q], n);  // (walls show one token, most objects at most four)
      for (int k = 4; k < n; k += 4) mg_put_tokens(dst + k * 3, src[k >> 2], loc4[q], n - k);
#else
      const uint8_t loc = (uint8_t)loc4[q];
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
#endif
    }
    base += total;
  }
  return base;
}

// All agents' observations; optionally streamed to HBM.  `exec` holds each agent's executed action (or null = 0).
// Requires mg_build_tokens + a barrier before.
// PF (records and grid in HBM, BIG): the positions come from LDS (l.pos) and the grid cells of the NEXT group of rows
// are requested before the current group is encoded, so that their HBM latency runs under the encode instead of in
// front of it (16 waves of one workgroup per CU have little else to hide it with).
template <int NR, bool PAIRS, class CT = uint16_t, bool PAT = true, bool DIRECT = false, bool PF = false>
__device__ __forceinline__ void mg_encode_all_t(const KArgs& a, const Lds& l, int e, uint32_t step, const int* exec,
                                       bool write_global, int tid, int nt) {
  // (the wave index stays a vector value on purpose: made scalar with readfirstlane, the per-agent address arithmetic
  // moves to the scalar unit, SGPR spills go from 74 to 99 and the kernel gets slower, 62.5 us against 58.7)
  const int lane = tid & 63, wave = tid >> 6, nwaves = nt >> 6;
  uint8_t* rowbuf = l.rows + (size_t)wave * a.ROWB;
  const int row_bytes = a.T * 3;
  const EncCfg ec = mg_enc_cfg(a);
  const GlobalTok gt = mg_global_tok(a, step, lane);
  uint32_t pk[MG_MAX_ROUNDS], loc4[MG_MAX_ROUNDS];
  int dlt[MG_MAX_ROUNDS];
  const int Wp = ec.W + a.PADC;  // row stride of the padded byte map (DIRECT)
#pragma unroll
  for (int q = 0; q < MG_MAX_ROUNDS; q++) {
    const bool in_pat = q < NR && (q * 64 + lane) < ec.P;
    pk[q] = in_pat ? (PAT ? l.pat[q * 64 + lane] : mg_pattern_entry(a.gdc, q * 64 + lane)) : 0u;
#ifdef MG_PUT_TOKENS
    loc4[q] = (pk[q] >> 16) * 0x01010101u;
#else
    loc4[q] = pk[q] >> 16;
#endif
    dlt[q] = in_pat ? __mul24((int)(pk[q] & 0xFF) - 8, Wp) + (int)((pk[q] >> 8) & 0xFF) - 8 : 0x40000000;
  }
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
  uint32_t posj = 0;
  if (DIRECT && lane < a.A) posj = *(const uint32_t*)&l.ag[lane];
  const MgByteMap gm = mg_byte_map(DIRECT ? l.gmap : nullptr, DIRECT ? __mul24(ec.H, Wp) : 0);
  // per-agent offsets are carried along instead of multiplied out (v_mul_lo_u32 runs at quarter rate)
  uint32_t roff0 = (uint32_t)(wave * group) * (uint32_t)row_bytes;  // within the env's block of rows (A * T * 3 < 2^32)
  const DAgent* me0 = l.ag + wave * group;
  const uint32_t roff_step = (uint32_t)(nwaves * group) * (uint32_t)row_bytes;
  uint32_t cvn[group][NR], rcn[group];  // PF: the next group's positions and raw grid cells, in flight
  // PF: the environment's grid in HBM behind a buffer descriptor, rows padded by GPAD empty cells: a window cell is
  // position + offset (dlt, in cells of the padded grid), cells above / below the map fall outside the descriptor and
  // read 0, cells left / right of it land on padding -- no row or column test on the vector unit.
#if defined(__HIP_DEVICE_COMPILE__)
  const int Wg = ec.W + a.GPAD;
  __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc((void*)l.grid, 0, PF ? __mul24(ec.H, Wg) * 2 : 0, 0x00020000);
  int dlg[MG_MAX_ROUNDS];
#pragma unroll
  for (int q = 0; q < MG_MAX_ROUNDS; q++) {
    const bool in_pat = q < NR && (q * 64 + lane) < ec.P;
    dlg[q] = in_pat ? (__mul24((int)(pk[q] & 0xFF) - 8, Wg) + (int)((pk[q] >> 8) & 0xFF) - 8) * 2 : 0x40000000;
  }
#endif
  auto request = [&](int first) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int h = 0; h < group; h++) {
      const uint32_t p = first + h < a.A ? l.pos[first + h] : 0u;
      rcn[h] = p;
      const int base = (__mul24((int)(p & 0xFFFF), Wg) + (int)(p >> 16)) * 2;
#pragma unroll
      for (int q = 0; q < NR; q++) cvn[h][q] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(grsrc, base + dlg[q], 0, 0);
    }
#endif
  };
  if (PF) request(wave * group);
  for (int a0 = wave * group; a0 < a.A; a0 += nwaves * group, roff0 += roff_step, me0 += nwaves * group)
  {
    uint32_t meta[group][NR], rc[group];
    if (PF) {
#pragma unroll
      for (int h = 0; h < group; h++) {
        rc[h] = rcn[h];
#pragma unroll
        for (int q = 0; q < NR; q++) meta[h][q] = l.ocnt[(int)cvn[h][q] - 1];
      }
      if (a0 + nwaves * group < a.A) request(a0 + nwaves * group);
    }
#ifndef MG_ENC_SEQ  // (the lookups of both agents of a pair up front; MG_ENC_SEQ: one agent at a time, fewer live registers)
#pragma unroll
    for (int h = 0; h < group && !PF; h++) {
      if (DIRECT)
        mg_window_lookup_direct<NR>(a, l, ec, me0 + h, dlt, lane, meta[h], rc[h], posj, l.wscr + (size_t)(wave * 2 + h) * l.WS, gm, Wp);
      else
        mg_window_lookup<NR, PAIRS, CT>(a, l, ec, me0 + h, pk, lane, meta[h], rc[h]);
    }
#endif
#pragma unroll
    for (int h = 0; h < group; h++) {
#ifdef MG_ENC_SEQ
      if (DIRECT)
        mg_window_lookup_direct<NR>(a, l, ec, me0 + h, dlt, lane, meta[h], rc[h], posj, l.wscr + (size_t)(wave * 2 + h) * l.WS, gm, Wp);
      else
        mg_window_lookup<NR, PAIRS, CT>(a, l, ec, me0 + h, pk, lane, meta[h], rc[h]);
#endif
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
      int attempted = mg_encode_agent<NR, PAIRS>(a, l, ec, gt, me0 + h, act, row, loc4, lane, meta[h], rc[h]);
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
template <class CT = uint16_t, bool PAT = true, bool DIRECT = false, bool PF = false>
__device__ inline void mg_encode_all(const KArgs& a, const Lds& l, int e, uint32_t step, const int* exec,
                                     bool write_global, int tid, int nt) {
  const bool pairs = write_global && (a.A & 1) == 0 && ((a.T * 3) & 15) == 8 && a.ROWB <= 1024;
  if (pairs && a.dc[DC_PATTERN_N] <= 128) mg_encode_all_t<2, true, CT, PAT, DIRECT, PF>(a, l, e, step, exec, write_global, tid, nt);
  else mg_encode_all_t<MG_MAX_ROUNDS, false, CT, PAT, DIRECT, PF>(a, l, e, step, exec, write_global, tid, nt);  // any shape
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

// The tokens_* statistics of the constructor's observation passes (mettagrid_c.cpp:139-154: one pass before the Clipper
// exists, init_obs_passes - 1 after it) for THIS environment as it stands right after a restart.  The episode-start
// templates hold them per MAP; with a Clipper whose start-clipped stations draw their unclipping protocol from the
// environment's seed (clipper.hpp:107-110,186-193) and protocol tokens in the observations, the later passes differ per
// environment whenever the protocols differ in shape.  Needs the grid with the agents overlaid, the records and the
// station records staged (`l`), and 2A + 4 zeroed ints of LDS (`scr`, left zero again); all threads of the workgroup.
template <class CT = uint16_t>
__device__ inline void mg_recount_initial_tokens(const KArgs& a, const Lds& l, int e, int map_id, int* scr, int tid, int nt) {
  const int* dc = a.dc;
  const EncCfg ec = mg_enc_cfg(a);
  const CellCodec cc{a.NW, a.A, a.NAmax};
  const int4* asm_info = a.map_asm + (size_t)map_id * a.NAmax;
  const int4* chest_info = a.map_chest + (size_t)map_id * a.NCmax;
  const CT* grid = (const CT*)l.grid;
  int* att1 = scr, *att2 = scr + a.A, *sums = scr + 2 * a.A;  // attempted tokens of the first / the later passes
  for (int w = tid; w < a.A * ec.P; w += nt) {
    const int ai = w / ec.P, p = w - ai * ec.P;
    const uint32_t pk = mg_pattern_entry(a.gdc, p);
    const int r = (int)l.ag[ai].r + (int)(pk & 0xFF) - 8, c = (int)l.ag[ai].c + (int)((pk >> 8) & 0xFF) - 8;
    if ((unsigned)r >= (unsigned)ec.H || (unsigned)c >= (unsigned)ec.W) continue;
    const uint16_t cv = (uint16_t)grid[r * (ec.W + a.GPAD) + c];
    if (cv == 0) continue;
    CountOut n2{0};
    mg_object_tokens(dc, grid, l.ag, l.as, asm_info, cc, ec.H, ec.W, ec.W + a.GPAD, 0u, cv, n2, l.ch, chest_info);
    int n1 = n2.n;
    if (cc.kind(cv) == MG_CELL_ASSEMBLER && l.as[cc.asm_idx(cv)].clipped >= 2u) {
      // before the Clipper was built the station was plainly clipped: the flag, no protocol tokens
      DAsm before = l.as[cc.asm_idx(cv)];
      before.clipped = 1u;
      CountOut c1{0};
      // (the station list is indexed by the cell's station index: hand over a pointer that puts `before` there)
      mg_object_tokens(dc, grid, l.ag, &before - cc.asm_idx(cv), asm_info, cc, ec.H, ec.W, ec.W + a.GPAD, 0u, cv, c1, l.ch, chest_info);
      n1 = c1.n;
    }
    atomicAdd(&att1[ai], n1);
    atomicAdd(&att2[ai], n2.n);
  }
  __syncthreads();
  const GlobalTok gt = mg_global_tok(a, 0u, 0);
  const int later = dc[DC_INIT_PASSES] - 1;
  for (int ai = tid; ai < a.A; ai += nt) {
    int base = gt.n;
    if (ec.g_goal) base += L_n(dc, group_rec(dc, l.ag[ai].group)[GR_OFF_GOAL]);
    if (ec.g_compass) {
      const int dr = ec.H / 2 - (int)l.ag[ai].r, dcc = ec.W / 2 - (int)l.ag[ai].c;
      const int sr = dr != 0 ? (dr > 0 ? 1 : -1) : 0, sc = dcc != 0 ? (dcc > 0 ? 1 : -1) : 0;
      const int orr = ec.hr + sr, occ = ec.wr + sc;
      if ((sr != 0 || sc != 0) && orr >= 0 && orr < ec.oh && occ >= 0 && occ < ec.ow) base++;
    }
    const int a1 = att1[ai] + base, a2 = att2[ai] + base;
    const int w1 = a1 < a.T ? a1 : a.T, w2 = a2 < a.T ? a2 : a.T;
    atomicAdd(&sums[0], w1 + later * w2);
    atomicAdd(&sums[1], (a1 - w1) + later * (a2 - w2));
    atomicAdd(&sums[2], (a.T - w1) + later * (a.T - w2));
  }
  __syncthreads();
  if (tid < 3) a.gstats[(size_t)e * a.SG + tid] = (float)sums[tid];  // (integers below 2^24: exact in f32 in any order)
  __syncthreads();
  for (int i = tid; i < 2 * a.A + 4; i += nt) scr[i] = 0;
  __syncthreads();
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
    {  // members of the agent's collective on this map ("aligned.agent", collective.hpp:46-52)
      const int coll = group_rec(dc, s.z)[GR_COLLECTIVE];
      int n = 0;
      if (coll >= 0)
        for (int j = 0; j < a.A; j++) n += group_rec(dc, spawn[j].z)[GR_COLLECTIVE] == coll ? 1 : 0;
      ag.aligned = (float)n;
    }
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
  if (a.NCmax > 0 && tid < a.R) {  // StatsTracker::set of chest.<r>.amount
    bool have = false;
    int value = 0;
    if (cx.gset64) {
      const unsigned long long v = cx.gset64[tid];
      have = v != 0ull;
      value = (int)(v & 0xFFFFull);
    } else if ((*l.gsetm >> tid) & 1u) {
      have = true;
      value = l.gsetv[tid];
    }
    if (have) {
      const int s = cx.GS.chest_amount(tid);
      gs_[s] = (float)value;
      atomicOr(&gt_[s >> 5], 1u << (s & 31));
    }
  }
}

// Agent::compute_stat_rewards (agent.cpp:103-132) + episode bookkeeping (bindings/mettagrid_c.cpp:656-677).
// FUSED: the records are written back whole afterwards (no separate store of cur_reward), and stat slots that other
// threads of this launch added to with atomics are read past the L1 (agent-scope loads).
template <bool FUSED = false>
__device__ inline void mg_rewards_and_flags(const KArgs& a, const Lds& l, int e, DEnv env, int tid, int nt) {
  const int* dc = a.dc;
  const MgAgentSlots AS{a.R, a.NG};
  bool end_now = false, trunc_now = false;
  {
    int ms = a.max_steps;
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
        const int* sr = dc + os + 1 + MG_SR_WORDS * q;
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
            v = FUSED ? __hip_atomic_load(&as_[ai * a.SA + aslot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                      : as_[ai * a.SA + aslot];
        }
        if (gslot >= 0)
          v += gslot < 3 ? l.tokacc[gslot]
                         : (FUSED ? __hip_atomic_load(&gs_[gslot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : gs_[gslot]);
        if (gr[GR_COLLECTIVE] >= 0) v += sr[6] ? me->aligned : __int_as_float(sr[5]);  // agent.cpp:115-117
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
        if (!FUSED) a.agents[g].cur_reward = total;
      }
    }
    a.rewards[g] = reward;
    a.ep_rewards[g] += reward;
    if (FUSED) a.success[g] = l.succ[ai];
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
  cx.H = a.H; cx.W = a.W; cx.GW = a.W + a.GPAD; cx.A = a.A; cx.NA = a.NAmax; cx.R = a.R; cx.SA = a.SA; cx.SG = a.SG; cx.NG = a.NG;
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
  cx.gset64 = nullptr;
  cx.seq = 0;
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
__device__ inline void mg_snapshot_finished(const KArgs& a, int e, uint32_t steps, uint32_t map_id, int tid, int nt) {
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
    const uint32_t idx = atomicAdd(&a.status[1], 1u);
    // (the upper 12 bits of the second word carry the entry's position in the sequence: a reader that was lapped while
    // it copied sees the mismatch instead of somebody else's episode; map slots are below 2^20, mg_set_maps checks)
    if (a.fin_ring) a.fin_ring[idx & a.fin_ring_mask] = make_uint2((uint32_t)e, (map_id & 0xFFFFFu) | (idx << 20));
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

#ifdef MG_WITH_CLIPPER
// ---- Clipper (systems/clipper.hpp:14-239) -------------------------------------------------------------------------------
// Its sets of assemblers are std::set<Assembler*>, walked in pointer order (clipper.hpp:24-26,207-229); here: station
// index = creation order, the order of a fresh process of the reference (oracle/mg_oracle.cpp, tests/golden/
// feature_119.. fixtures).  Adjacency lists are not kept: j is adjacent to i iff both are members (not clip-immune),
// i != j and their L_inf distance is within cutoff * length_scale; then the weight is >= 1.  The border set is
// "member, not clipped, weight > 0" (a zero-weight member can never be picked).  All of it runs on one thread: one
// draw per step, a clip event every `clip_period` steps.
struct ClipRng {  // std::mt19937 one word at a time: the same sequence as twisting whole blocks
  uint32_t* mt;
  uint32_t idx;
  __device__ inline uint32_t next() {
    const uint32_t i = idx, i1 = i + 1 == 624 ? 0 : i + 1, im = i + 397 >= 624 ? i + 397 - 624 : i + 397;
    const uint32_t y = mt_mix(mt[i], mt[i1], mt[im]);
    mt[i] = y;
    idx = i1;
    return mt_temper(y);
  }
  __device__ inline uint32_t bounded(uint32_t range) {  // uniform_int_dist.h:258-269
    unsigned long long product = (unsigned long long)next() * range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
      const uint32_t threshold = (0u - range) % range;
      while (low < threshold) {
        product = (unsigned long long)next() * range;
        low = (uint32_t)product;
      }
    }
    return (uint32_t)(product >> 32);
  }
};
__device__ inline bool mg_clip_member(const int* dc, const int4* info, int i) {
  return asm_rec(dc, info[i].z)[AR_CLIP_IMMUNE] == 0;
}
// infection_weight (clipper.hpp:113-128) of two adjacent stations, 0 = not adjacent (clipper.hpp:156-169)
__device__ inline uint32_t mg_clip_weight(const int4 a, const int4 b, uint32_t ls, uint32_t cutoff) {
  const int dr = a.x > b.x ? a.x - b.x : b.x - a.x, dcl = a.y > b.y ? a.y - b.y : b.y - a.y;
  const uint32_t dist = (uint32_t)(dr > dcl ? dr : dcl);
  if (dist > cutoff * ls) return 0u;
  return 1u << (cutoff - dist / ls);
}
// clip_assembler (clipper.hpp:174-194) + become_clipped (assembler.hpp:411-421)
__device__ inline void mg_clip_assembler(const int* dc, DAsm* as, const int4* info, int nasm, uint32_t* w, uint32_t ls,
                                         ClipRng& rng, int i, uint32_t step) {
  const uint32_t cutoff = (uint32_t)dc[DC_CLIP_CUTOFF];
  if (dc[DC_CLIP_PERIOD] > 0)  // (compute_adjacencies only runs for an active clipper, clipper.hpp:101-105)
    for (int j = 0; j < nasm; j++) {
      if (j == i || !mg_clip_member(dc, info, j)) continue;
      const uint32_t wt = mg_clip_weight(info[i], info[j], ls, cutoff);
      if (wt) w[j] += wt;
    }
  const uint32_t k = rng.bounded((uint32_t)dc[DC_N_UNCLIP]);
  as[i].clipped = 2u + k;
  as[i].cooldown_end = step;
  as[i].cooldown_duration = 0;
  w[i] |= 0x80000000u;
}
// The Clipper's constructor (clipper.hpp:35-111) for environment e on map `map_id`: a fresh engine, zero weights, the
// length scale, and the stations that start clipped, in object order.  All threads; `as` = the station records
// (LDS or HBM), `dc` = the configuration the map plays.
__device__ inline void mg_clipper_init(const KArgs& a, const int* dc, DAsm* as, int e, int map_id, int H, int W, int tid,
                                       int nt) {
  if (!a.clip_w || !dc[DC_CLIPPER]) return;  // block-uniform
  uint32_t* cm = a.clip_mt + (size_t)e * 624;
  const uint32_t* src = (a.mt_init ? a.mt_init : a.mt) + (size_t)e * 624;  // std::mt19937(seed), nothing drawn yet
  for (int i = tid; i < 624; i += nt) cm[i] = src[i];
  uint32_t* w = a.clip_w + (size_t)e * a.NAmax;
  for (int i = tid; i < a.NAmax; i += nt) w[i] = 0u;
  __threadfence_block();
  __syncthreads();
  if (tid == 0) {
    const int nasm = a.map_nasm[map_id];
    const int4* info = a.map_asm + (size_t)map_id * a.NAmax;
    int members = 0;
    for (int i = 0; i < nasm; i++) members += mg_clip_member(dc, info, i) ? 1 : 0;
    int ls = dc[DC_CLIP_LENGTH];
    if (ls <= 0 && members > 0) {  // clipper.hpp:65-98: from the sparsity of the stations
      const uint32_t sparsity = (uint32_t)(W * H) / (uint32_t)members;
      int root = 1, next = 10;
      for (int it = 0; it < 10; it++) {
        if (next == root || root == 0) break;
        root = next;
        next = (int)((sparsity + (uint32_t)(root * root) + (uint32_t)(2 * root - 2)) / (uint32_t)(2 * root));
      }
      ls = root / 2 > 1 ? root / 2 : 1;
    }
    if (ls <= 0) ls = 1;  // (no members: never used)
    ClipRng rng{cm, 0u};
    for (int i = 0; i < nasm; i++)
      if (mg_clip_member(dc, info, i) && asm_rec(dc, info[i].z)[AR_START_CLIPPED])
        mg_clip_assembler(dc, as, info, nasm, w, (uint32_t)ls, rng, i, 0u);
    a.clip_state[e] = make_uint2(rng.idx, (uint32_t)ls);
  }
  __threadfence_block();
  __syncthreads();
}
#endif

// Restart a finished environment from the per-map templates (MettaGridPufferEnv.step -> _new_sim(),
// mettagrid_puffer_env.py:158-164; same seed, next map).  All threads of the workgroup; HBM state only.
__device__ inline void mg_restart_env(const KArgs& a, int e, DEnv& env, int tid, int nt) {
  mg_snapshot_finished(a, e, env.step, env.map_id, tid, nt);
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
#ifdef MG_WITH_CLIPPER
  if (a.clip_w) {
    // The template holds what the reset kernel's own clipper drew for the stations that start clipped; which
    // unclipping protocol each of them offers comes from THIS environment's seed (clipper.hpp:107-110,186-193).
    KArgs b = a;
    mg_use_map(b, (uint32_t)m);
    mg_clipper_init(a, b.gdc, a.asms + (size_t)e * a.NAmax, e, m, b.H, b.W, tid, nt);
    __threadfence();
    __syncthreads();
  }
#endif
}


// ---- launchers (one per kernel translation unit) ---------------------------------------------------------------
#define MG_FAST_THREADS 64
#define MG_USES_THREADS 64
void mg_launch_act_fast(const KArgs& k, unsigned n, hipStream_t st);                                   // k_act_fast.hip
void mg_launch_uses(const KArgs& k, unsigned n, size_t lds, bool dcl, hipStream_t st);                // k_uses.hip
void mg_launch_act(const KArgs& k, unsigned n, size_t lds, bool dcl, bool big, hipStream_t st);       // k_act_body.h (k_act_dcl / _nodcl / _big.hip)
size_t mg_act_lds_bytes(const KArgs& k, bool dcl, bool big);
int mg_act_threads(const KArgs& k);
void mg_launch_obs(const KArgs& k, unsigned n, size_t lds, bool dcl, bool big, hipStream_t st);       // k_obs.hip
void mg_launch_reset(const KArgs& k, unsigned n, size_t lds, bool big, hipStream_t st);               // k_reset.hip
void mg_launch_clipper(const KArgs& k, unsigned n, hipStream_t st);                                   // k_clipper.hip
// k_fused.hip: the whole step of one environment in one workgroup (attack-free configs, A <= 64)
void mg_launch_fused(const KArgs& k, unsigned n, size_t lds, bool dcl, hipStream_t st);
size_t mg_fused_lds_bytes(const KArgs& k, bool dcl);
hipError_t mg_raise_lds_cap_fused(int cap);
void mg_launch_fused_nomod(const KArgs& k, unsigned n, size_t lds, bool dcl, hipStream_t st);  // k_fused_nomod.hip
hipError_t mg_raise_lds_cap_fused_nomod(int cap);
void mg_launch_fused_nomod_o7(const KArgs& k, unsigned n, size_t lds, bool dcl, hipStream_t st);  // k_fused_nomod_o7.hip
hipError_t mg_raise_lds_cap_fused_nomod_o7(int cap);
void mg_launch_fused_nomod_w1(const KArgs& k, unsigned n, size_t lds, bool dcl, hipStream_t st);  // k_fused_nomod_w1.hip
size_t mg_fused_lds_bytes_nomod_w1(const KArgs& k, bool dcl);
hipError_t mg_raise_lds_cap_fused_nomod_w1(int cap);
void mg_launch_fused_nomod_w18(const KArgs& k, unsigned n, size_t lds, bool dcl, hipStream_t st);  // k_fused_nomod_w18.hip
hipError_t mg_raise_lds_cap_fused_nomod_w18(int cap);
void mg_launch_fused_o6(const KArgs& k, unsigned n, size_t lds, bool dcl, hipStream_t st);  // k_fused_o6.hip
hipError_t mg_raise_lds_cap_fused_o6(int cap);
// the cap on dynamic LDS is a per-function setting shared by every handle of the process
hipError_t mg_raise_lds_cap_uses(int cap);
hipError_t mg_raise_lds_cap_act(int cap);
hipError_t mg_raise_lds_cap_obs(int cap);
hipError_t mg_raise_lds_cap_reset(int cap);
