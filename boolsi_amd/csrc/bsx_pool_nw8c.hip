// k_attract_pool / k_digit_lifetimes for states of 8 32-bit words and 6 predecessor slots -- the instantiations that take
// longest to compile (bsx_pool_kernel.h)
#define BSX_POOL_KMASK 0x40
#include "bsx_pool_kernel.h"
BSX_POOL_TU(8, c)
