// k_act_nodcl.hip -- one instantiation of the general action kernel (see k_act_body.h)
#define MG_ACT_VARIANT 1
#include "k_act_body.h"
