// k_attract_pool / k_digit_lifetimes for states of 4 32-bit words and 2, 4 or 6 predecessor slots (bsx_pool_kernel.h)
#define BSX_POOL_KMASK 0x54
#include "bsx_pool_kernel.h"
BSX_POOL_TU(4, b)
