// k_fused_o6.hip -- k_fused.hip (with inventory-limit modifiers) built for 6 waves per SIMD: 80 registers, 4 vector
// spills instead of 49 (see k_fused.hip)
#define MG_FUSED_O6 1
#include "k_fused.hip"
