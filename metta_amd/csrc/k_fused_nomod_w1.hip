// k_fused_nomod_w1.hip -- the fused step kernel without limit modifiers as ONE wave per environment (see k_fused.hip)
#define MG_NO_MODIFIERS 1
#define MG_FUSED_W1 1
#include "k_fused.hip"
