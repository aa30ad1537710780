// k_act_big.hip -- one instantiation of the general action kernel (see k_act_body.h)
#define MG_ACT_VARIANT 2
#include "k_act_body.h"
