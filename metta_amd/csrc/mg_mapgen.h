// mg_mapgen.h -- arguments of the device map generator (k_mapgen.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#define MG_MAPGEN_THREADS 256
#define MG_MAPGEN_MAX_CELLS (150 * 1024)  // the byte map of one slot is built in LDS

struct DEnv;

struct MapGenArgs {
  const int* gdc;  // device config stream (cell code -> kind, idx)
  int height, width, fill_code, n_rooms;
  const int* rooms;           // [n_rooms][4] r0, c0, h, w (disjoint)
  const int* room_sym_begin;  // [n_rooms + 1] into sym_code / sym_count
  const int* sym_code;        // cell codes (the bytes mg_set_maps takes)
  const int* sym_count;
  // pool entries written
  uint16_t* map_cells;
  uint8_t* map_cells8;  // or null; [P][HW8], rows of width + PADC bytes (the padding stays zero)
  uint8_t* map_bytes;   // [P][HWB] the generated maps as cell codes (read back by mg_get_map)
  int4* map_spawn;
  int4* map_asm;
  int* map_nasm;
  int4* map_chest;
  int* map_nchest;
  float* map_objcount;
  int HW, HW8, HWB, PADC, A, NW, NAmax, NCmax, n_cells;
  // slot selection
  uint32_t* slot_state;   // [P] 1 = used by an episode since its last generation
  const uint32_t* inuse;  // [P] == gen where an environment is on the slot now
  uint32_t gen;
  int force;
  int* list;   // [P] regenerated slots (compacted)
  int* count;  // [1]
  uint32_t* status;
  unsigned long long seed;
  int direct_slot;  // >= 0: derive this one slot from the cell codes already in map_bytes (mg_write_map_slot), no drawing
};

void mg_launch_mark_maps(const DEnv* env, int E, uint32_t* inuse, uint32_t gen, hipStream_t st);
void mg_launch_generate_maps(const MapGenArgs& g, unsigned n, hipStream_t st);
hipError_t mg_raise_lds_cap_mapgen(int cap);
