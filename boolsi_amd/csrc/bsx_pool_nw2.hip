// k_attract_pool / k_digit_lifetimes for states of 2 32-bit words (bsx_pool_kernel.h)
#include "bsx_pool_kernel.h"
BSX_POOL_TU(2, )
