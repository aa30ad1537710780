// k_fused_nomod_w18.hip -- the one-wave build (k_fused_nomod_w1.hip) for 8 waves per SIMD: 32 workgroups per CU, chosen
// when a launch holds more environments than one round of the 7-wave build and at most one round of this one
#define MG_NO_MODIFIERS 1
#define MG_FUSED_W1 1
#define MG_FUSED_W18 1
#include "k_fused.hip"
