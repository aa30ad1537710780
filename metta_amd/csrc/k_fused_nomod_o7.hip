// k_fused_nomod_o7.hip -- k_fused_nomod.hip built for 7 waves per SIMD (72 registers, no vector spills; see k_fused.hip)
#define MG_NO_MODIFIERS 1
#define MG_FUSED_O7 1
#include "k_fused.hip"
