// k_fused_nomod.hip -- the fused step kernel for configurations without inventory-limit modifiers (see k_fused.hip)
#define MG_NO_MODIFIERS 1
#include "k_fused.hip"
