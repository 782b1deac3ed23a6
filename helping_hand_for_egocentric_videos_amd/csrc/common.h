// Shared device/host helpers for libhh (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/hh.h"

typedef __bf16 bf16_t;
typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

void hh_set_error(const char* fmt, ...);
int hh_check_launch(const char* what);
int hh_stream_cu_count(hipStream_t s);
int hh_stream_slot(hipStream_t s);          // 0..31, or -1 when more than 32 streams have launched persistent GEMMs      // CUs the stream may use (runtime.cpp)

// device timing of one kernel launch (runtime.cpp: hh_prof_enable / hh_prof_read); `work` = algorithmic flops or bytes of the launch
struct HHProfScope {
    HHProfScope(int klass, double work, hipStream_t s);
    ~HHProfScope();
    int rec_, gen_;
    hipStream_t stream_;
};

// per-device once-guards for hipFuncSetAttribute (dynamic LDS sizes are a per-device function attribute): `mask` is the call site's static
// bit set of devices already done.  hh_attr_needed: true until hh_attr_done() was called for the CURRENT device (set the bit only after every
// attribute call of the site succeeded).  Thread-safe; devices >= 64 are simply set every time.
#include <atomic>
bool hh_attr_needed(const std::atomic<uint64_t>& mask);
void hh_attr_done(std::atomic<uint64_t>& mask);

void hh_prof_note_kernel(int klass, const char* name);    // runtime.cpp: the kernel a launch site dispatched for a profiled class (hh_prof_kernel_name)

#define HH_REQUIRE(cond, code, ...)                 \
    do {                                            \
        if (!(cond)) {                              \
            hh_set_error(__VA_ARGS__);              \
            return (code);                          \
        }                                           \
    } while (0)

#define HH_ALIGNED16(p) ((((uintptr_t)(p)) & 15) == 0)

__device__ __forceinline__ float bf16_lo_to_f32(unsigned int u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi_to_f32(unsigned int u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned int pack_bf16(float lo, float hi) {
    bf16x2 v;
    v[0] = (bf16_t)lo;
    v[1] = (bf16_t)hi;
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Full-line output stores for the 16-query MFMA output layout (lane (c = lane & 15, g = lane >> 4) holds 32 B of row c: w0 = bytes
// [32 g, 32 g + 16), w1 = the next 16).  Stored as they are, one instruction covers 16 rows x four 16-B pieces -- 16 partial lines,
// ~270 cycles on the CU's store path against ~77 for 8 rows x 128 B (scripts/store_probe.hip).  Here lanes c and c ^ 8 swap one piece
// (DPP row_ror:8, two bank-masked moves per dword): `x` then holds, for row c & 7, piece 2 g + (c >> 3) of that row's 128 B, `y` the same
// for row 8 + (c & 7): two instructions of 8 full 128-B lines each.  All 64 lanes must be active.
__device__ __forceinline__ void hh_fullline_swap(const u32x4& w0, const u32x4& w1, u32x4& x, u32x4& y) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[i] = (unsigned)__builtin_amdgcn_update_dpp((int)w0[i], (int)w1[i], 0x128, 0xF, 0xC, false);     // lanes 8..15 of a row: w1 of lane c - 8
        y[i] = (unsigned)__builtin_amdgcn_update_dpp((int)w1[i], (int)w0[i], 0x128, 0xF, 0x3, false);     // lanes 0..7: w0 of lane c + 8
    }
}
