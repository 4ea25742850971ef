// k_attract_pool / k_digit_lifetimes for states of 4 32-bit words and 1, 3 or 5 predecessor slots (bsx_pool_kernel.h)
#define BSX_POOL_KMASK 0x2A
#include "bsx_pool_kernel.h"
BSX_POOL_TU(4, a)
