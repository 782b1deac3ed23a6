// Maps CU-mask bit -> (XCC, SE, SH, CU) on the box: one stream per mask bit, a kernel records HW_ID / XCC_ID of every block.
// hipcc --offload-arch=gfx950 scripts/cumask_map.hip -o scripts/_bin/cumask_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <tuple>
__global__ void where(uint32_t* out) {
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));        // HW_REG_HW_ID
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
    }
    __builtin_amdgcn_s_sleep(64);
}
int main(int argc, char** argv) {
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    uint32_t def[16] = {0};
    hipError_t e = hipExtStreamGetCUMask(nullptr, 16, def);
    printf("ncu %d default-mask(%s):", ncu, hipGetErrorString(e));
    for (int i = 0; i < 16; ++i) printf(" %08x", def[i]);
    printf("\n");
    const int NB = 512;
    uint32_t* d;
    hipMalloc(&d, NB * 8);
    uint32_t h[NB * 2];
    const int words = argc > 1 ? atoi(argv[1]) : 8;
    const int nbits = argc > 2 ? atoi(argv[2]) : 256;
    for (int b = 0; b < nbits; ++b) {
        uint32_t mask[16] = {0};
        mask[b >> 5] = 1u << (b & 31);
        hipStream_t s;
        e = hipExtStreamCreateWithCUMask(&s, words, mask);
        if (e != hipSuccess) { printf("bit %3d: create failed %s\n", b, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        hipMemsetAsync(d, 0xff, NB * 8, s);
        hipLaunchKernelGGL(where, dim3(NB), dim3(64), 0, s, d);
        hipMemcpyAsync(h, d, NB * 8, hipMemcpyDeviceToHost, s);
        e = hipStreamSynchronize(s);
        std::set<std::tuple<int, int, int, int>> seen;
        for (int i = 0; i < NB; ++i) {
            const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 15;
            seen.insert({(int)xcc, (int)((hw >> 13) & 7), (int)((hw >> 12) & 1), (int)((hw >> 8) & 15)});
        }
        printf("bit %3d:", b);
        for (auto& t : seen) printf(" (xcc%d se%d sh%d cu%d)", std::get<0>(t), std::get<1>(t), std::get<2>(t), std::get<3>(t));
        printf("\n");
        hipStreamDestroy(s);
    }
    return 0;
}
