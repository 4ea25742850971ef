// k_attract_pool / k_digit_lifetimes for states of 1 32-bit word (bsx_pool_kernel.h)
#include "bsx_pool_kernel.h"
BSX_POOL_TU(1, )
