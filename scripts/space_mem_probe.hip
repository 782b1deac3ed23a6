// Memory-side probe of the space-attention access pattern (scripts/space_probe.py's "memory only" mode, freed from the kernel's LDS and
// register footprint): one workgroup per (clip, frame, head) stages K and V rows (128 B segments at a 6 KB stride) by LDS-DMA into a
// WRAPPED LDS window, reads its Q rows and writes O rows.  The LDS window size sets how many workgroups a CU holds, i.e. the bytes in
// flight per CU: tells whether the kernel's 4.3 TB/s is an occupancy (latency) limit or the pattern's bandwidth ceiling.
// hipcc --offload-arch=gfx950 -O3 scripts/space_mem_probe.hip -o scripts/_bin/space_mem_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__global__ __launch_bounds__(256) void probe(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ out, int T, int n, int heads,
                                             int lds_pieces, int q_first, int head_major, long Mtot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = heads * 64;
        const int N = 1 + T * n;
    int bid = blockIdx.x;
    const int head = bid % heads; bid /= heads;
    const int f = bid % T;
    const int b = bid / T;
    // token-major: element (row, which, head, d) at row * 3D + which * D + head * 64 + d;  head-major planes: ((which * heads + head) * Mtot + row) * 64 + d
    const long ldr = head_major ? 64 : 3L * D, ldp = head_major ? Mtot * 64 : 64;
    const long ld_which = (long)heads * ldp;
    const unsigned short* base = qkv + (long)b * N * ldr + head * ldp;
    const unsigned short* q_ptr = base + (long)(1 + f * n) * ldr;
    const int c = lane & 15, g = lane >> 4;
    u32x4 qa[4], qb[4];
    if (q_first)
        for (int j = 0; j < 4; ++j) {
            const unsigned short* qrow = q_ptr + (long)((wave * 4 + j) * 16 + c) * ldr + 8 * g;
            qa[j] = *(const u32x4*)qrow; qb[j] = *(const u32x4*)(qrow + 32);
        }
    const int pieces = (n + 32) >> 3;
    for (int kv = 1; kv <= 2; ++kv)
        for (int pc = wave; pc < pieces; pc += 4) {
            const int row = pc * 8 + (lane >> 3);
            const unsigned short* src = (row < n) ? q_ptr + (long)row * ldr : base;
            glds16(src + kv * ld_which + (lane & 7) * 8, smem + (pc % lds_pieces) * 1024);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!q_first)
        for (int j = 0; j < 4; ++j) {
            const unsigned short* qrow = q_ptr + (long)((wave * 4 + j) * 16 + c) * ldr + 8 * g;
            qa[j] = *(const u32x4*)qrow; qb[j] = *(const u32x4*)(qrow + 32);
        }
    const u32x4 x = *(const u32x4*)(smem + lane * 16);
    for (int j = 0; j < 4; ++j) {
        unsigned short* op = out + ((long)b * N + 1 + f * n + (wave * 4 + j) * 16 + c) * D + head * 64 + 16 * g;
        *(u32x4*)op = qa[j] ^ x; *(u32x4*)(op + 8) = qb[j];
    }
}
int main() {
    const int B = 32, T = 16, n = 256, heads = 16, D = heads * 64, N = 1 + T * n;
    unsigned short *qkv, *out;
    hipMalloc(&qkv, (size_t)B * N * 3 * D * 2); hipMalloc(&out, (size_t)B * N * D * 2);
    hipMemset(qkv, 0, (size_t)B * N * 3 * D * 2);
    const double bytes = 8.0 * B * N * D;
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int hm = 0; hm < 2; ++hm)
    for (int q_first = 0; q_first < 2; ++q_first)
        for (int kb : {74, 38, 18}) {       // LDS KB per workgroup -> 2, 3, 4, 5, 8, 16 workgroups per CU (wave slots cap it at 8)
            const int lds_pieces = kb;                 // 1 KB pieces
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe, dim3(B * T * heads), dim3(256), kb * 1024, 0, qkv, out, T, n, heads, lds_pieces, q_first, hm, (long)B * N);
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(probe, dim3(B * T * heads), dim3(256), kb * 1024, 0, qkv, out, T, n, heads, lds_pieces, q_first, hm, (long)B * N);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("head_major=%d q_first=%d  LDS %2d KB/workgroup (%2d per CU): %7.1f us  %5.2f TB/s\n", hm, q_first, kb, 160 / kb > 8 ? 8 : 160 / kb, ms * 100, bytes / (ms * 1e-4) / 1e12);
        }
    return 0;
}
