// k_act_dcl.hip -- one instantiation of the general action kernel (see k_act_body.h)
#define MG_ACT_VARIANT 0
#include "k_act_body.h"
