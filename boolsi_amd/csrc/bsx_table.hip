// HBM attractor table (bsx_kernels_common.h: table_insert): drain kernel.
#include <hip/hip_runtime.h>
#include "bsx_device.h"

namespace bsx {

// Every ready slot goes to out[] (dense, any order) and is cleared for the next call.
__global__ __launch_bounds__(256) void k_table_drain(LogRec* tab, uint64_t slots, LogRec* out, unsigned long long out_cap,
                                                     unsigned long long* cursor) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += stride) {
        if (tab[i].pad == 0u) continue;
        const LogRec r = tab[i];
        const unsigned long long at = atomicAdd(cursor, 1ull);
        if (at < out_cap) { out[at] = r; out[at].pad = 0; }
        LogRec z{};
        tab[i] = z;
    }
}

hipError_t launch_table_drain(LogRec* tab, uint64_t slots, LogRec* out, uint64_t out_cap, unsigned long long* cursor, hipStream_t st) {
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((slots + 255) / 256, 256u * 16u);
    hipLaunchKernelGGL(k_table_drain, dim3(blocks), dim3(256), 0, st, tab, slots, out, (unsigned long long)out_cap, cursor);
    return hipGetLastError();
}

}  // namespace bsx
