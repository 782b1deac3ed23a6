// Probe: store throughput of the GEMM epilogue's access pattern.  One workgroup of 4 waves per CU (128 KB of LDS claimed), each wave
// issues 32 global_store_dwordx4 per "tile" (1 KB per instruction), 16 tiles, on 1 / 8 / 64 / 256 CUs.  Pattern A = the MFMA-layout
// epilogue (a store covers 16 rows x 64 B, row stride ldc), B = 4 rows x 256 B (what an LDS transpose would give), C = 2 rows x 512 B;
// V = pattern A with ~18 dependent-free VALU instructions between two stores (the real epilogue's arithmetic); D = 8 rows x 128 B;
// H = the attention kernels' output pattern (16 rows, each lane 16 B at a 32-B pitch, the holes filled by the next instruction).
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/_bin/store_probe scripts/store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ __launch_bounds__(256, 1) void probe(char* C, long ldc_bytes, unsigned long long* out, int tiles, float f) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f * (lane + i);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        char* base = C + ((long)(blockIdx.x * tiles + t) * 256) * ldc_bytes;
        u32x4 v = {(unsigned)t, (unsigned)lane, 3u, 4u};
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            long off;
            if (PAT == 0 || PAT == 3) {
                const int idx = i >> 2, j = i & 3;
                const long row = (idx >> 2) * 128 + wr * 64 + (idx & 3) * 16 + (lane & 15);
                const long colb = (wc * 64 + (j & 1) * 32 + (j >> 1) * 128 + 8 * (lane >> 4)) * 2;
                off = row * ldc_bytes + colb;
            } else if (PAT == 1) {
                const long row = wr * 128 + i * 4 + (lane >> 4);
                off = row * ldc_bytes + (wc * 128) * 2 + (lane & 15) * 16;
            } else if (PAT == 2) {
                const long row = wave * 64 + i * 2 + (lane >> 5);
                off = row * ldc_bytes + (lane & 31) * 16;
            } else if (PAT == 4) {   // D: 8 rows x 128 B (one head's slice of 8 token rows)
                const long row = wave * 64 + (i >> 2) * 8 + (lane >> 3);
                off = row * ldc_bytes + (i & 3) * 128 + (lane & 7) * 16;
            } else {                 // H: the attention kernels' pattern: 16 rows, lane g writes 16 B at 32 g (+16 in the second store)
                const long row = wave * 64 + (i >> 3) * 16 + (lane & 15);
                off = row * ldc_bytes + ((i >> 1) & 3) * 128 + (lane >> 4) * 32 + (i & 1) * 16;
            }
            if (PAT == 3) {
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = acc[q] * f + 1.0f;
                v[2] = __float_as_uint(acc[i & 15]);
            }
            *(u32x4*)(base + off) = v;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = t2 - t0; }
    if (tiles < 0) smem[threadIdx.x] = 1;
}
template <int PAT> void run(const char* name, char* C, long ldc, unsigned long long* out, int tiles, int grid) {
    hipFuncSetAttribute((const void*)probe<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    std::vector<unsigned long long> h(2 * grid);
    double bi = 1e30, bd = 1e30;
    for (int r = 0; r < 5; ++r) {
        hipLaunchKernelGGL(probe<PAT>, dim3(grid), dim3(256), 131072, 0, C, ldc, out, tiles, 1.0001f);
        hipMemcpy(h.data(), out, 2 * grid * 8, hipMemcpyDeviceToHost);
        double s = 0, d = 0; for (int i = 0; i < grid; ++i) { s += (double)h[2 * i]; d += (double)h[2 * i + 1]; }
        s /= grid; d /= grid; if (s < bi) bi = s; if (d < bd) bd = d;
    }
    printf("  %-28s %3d CUs: %6.1f cycles per store instruction issued, %6.1f incl. the final drain\n", name, grid, bi / (tiles * 32.0), bd / (tiles * 32.0));
}
int main() {
    const int tiles = 16; const long ldc = 3072 * 2;
    char* C; unsigned long long* out;
    hipMalloc(&C, (size_t)256 * tiles * 256 * ldc); hipMalloc(&out, 2 * 256 * 8);
    hipMemset(C, 0, (size_t)256 * tiles * 256 * ldc);
    for (int grid : {1, 8, 64, 256}) {
        run<0>("A: 16 rows x 64 B", C, ldc, out, tiles, grid);
        run<1>("B: 4 rows x 256 B", C, ldc, out, tiles, grid);
        run<2>("C: 2 rows x 512 B", C, ldc, out, tiles, grid);
        run<3>("V: A + 16 VALU per store", C, ldc, out, tiles, grid);
        run<4>("D: 8 rows x 128 B", C, ldc, out, tiles, grid);
        run<5>("H: 16 rows x 4 x 16 B, holes", C, ldc, out, tiles, grid);
    }
    return 0;
}
