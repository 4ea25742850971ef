#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 4) void k(float* p) { extern __shared__ float s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[(threadIdx.x + 1) & 511]; }
__global__ __launch_bounds__(640, 5) void k2(float* p) { extern __shared__ float s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[(threadIdx.x + 1) % 640]; }
__global__ __launch_bounds__(768, 6) void k3(float* p) { extern __shared__ float s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[(threadIdx.x + 1) % 768]; }
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu regsPerBlock %d maxThreadsPerMP %d\n", pr.sharedMemPerBlock, pr.maxSharedMemoryPerMultiProcessor, pr.regsPerBlock, pr.maxThreadsPerMultiProcessor);
    for (size_t sh : {40000, 52000, 53000, 54000, 56000, 64000, 69776, 75000, 77456, 80000, 81000, 82000, 100000}) {
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        int a = 0, b = 0, c = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k, 512, sh);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k2, 640, sh);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, k3, 768, sh);
        printf("shmem %6zu: blocks/CU 512thr %d  640thr %d  768thr %d\n", sh, a, b, c);
    }
    return 0;
}
