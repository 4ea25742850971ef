// k_attract_pool / k_digit_lifetimes for states of 8 32-bit words and 2 or 4 predecessor slots (bsx_pool_kernel.h)
#define BSX_POOL_KMASK 0x14
#include "bsx_pool_kernel.h"
BSX_POOL_TU(8, b)
