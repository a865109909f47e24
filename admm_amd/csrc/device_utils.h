// Device-side helpers for gfx950: 64-lane wave reductions, block reductions, vector types.
#pragma once
#include <hip/hip_runtime.h>

namespace admm {

constexpr int kWave = 64;   // CDNA4 wavefront

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        T o = __shfl_xor(v, off, kWave);
        v = o > v ? o : v;
    }
    return v;
}

// Halving butterfly for 8 values per lane over the wave.  `_top` performs the xor-32 / 16 / 8 steps: afterwards lane l
// holds, for value (l >> 3), the sum over the 8 lanes congruent to l mod 8.  `halving_sum8` adds the xor-4 / 2 / 1 steps:
// lane l holds the wave total of value (l >> 3).  Every addition is one that wave_sum performs for that value, pair by
// pair and in the same step order, so the totals are bit-identical to eight wave_sum calls (IEEE addition commutes) at a
// third of the exchanges (7 + 3 instead of 48).
template <typename T>
__device__ __forceinline__ T halving_sum8_top(const T (&v)[8], int lane) {
    T a4[4], a2[2];
    {
        const int b = (lane >> 5) & 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const T keep = b ? v[k + 4] : v[k], send = b ? v[k] : v[k + 4];
            a4[k] = keep + __shfl_xor(send, 32, kWave);
        }
    }
    {
        const int b = (lane >> 4) & 1;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const T keep = b ? a4[k + 2] : a4[k], send = b ? a4[k] : a4[k + 2];
            a2[k] = keep + __shfl_xor(send, 16, kWave);
        }
    }
    const int b = (lane >> 3) & 1;
    const T keep = b ? a2[1] : a2[0], send = b ? a2[0] : a2[1];
    return keep + __shfl_xor(send, 8, kWave);
}
template <typename T>
__device__ __forceinline__ T halving_sum8(const T (&v)[8], int lane) {
    T r = halving_sum8_top(v, lane);
    r += __shfl_xor(r, 4, kWave);
    r += __shfl_xor(r, 2, kWave);
    r += __shfl_xor(r, 1, kWave);
    return r;
}
__device__ __forceinline__ double readlane_f64(double v, int src) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Deterministic block-wide sum of NV values per thread; result valid in every thread.
// scratch: NV * (blockDim/64) elements of LDS.
template <typename T, int NV>
__device__ __forceinline__ void block_sum(T (&v)[NV], T* scratch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[i * nw + wid] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        T s = 0;
        for (int w = 0; w < nw; ++w) s += scratch[i * nw + w];
        v[i] = s;
    }
}

// A small control block (or flag word) that every workgroup of a launch reads first: through the VECTOR memory path, then
// made wave-uniform again with v_readfirstlane.  Measured with in-kernel timestamps (round 2): (1) a scalar s_load of it
// shares its wait counter with the kernel-argument loads, so the compiler waits for it before it can form the first vector
// address -- one more dependent round trip per launch (wide x-update); (2) in the consensus z kernel (391 workgroups) the
// scalar load of the 64-byte block took 15-17 us to arrive in every workgroup but the first (1.6 us there), the same request
// as vector loads 0.8 us everywhere: the kernel went from 22 to 6.6 us.  The zero offset comes from inline asm so that the
// address is not provably uniform.
template <typename C>
__device__ __forceinline__ C load_ctl_vector(const C* p) {
    static_assert(sizeof(C) % 16 == 0, "control blocks are padded to whole 16-byte words");
    constexpr int N = (int)(sizeof(C) / 16);
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    const uint4* cp = reinterpret_cast<const uint4*>(p) + vz;
    uint4 w[N];
#pragma unroll
    for (int k = 0; k < N; ++k) w[k] = cp[k];
    unsigned u[N * 4];
    __builtin_memcpy(u, w, sizeof(C));
#pragma unroll
    for (int k = 0; k < N * 4; ++k) u[k] = (unsigned)__builtin_amdgcn_readfirstlane((int)u[k]);
    C c;
    __builtin_memcpy(&c, u, sizeof(C));
    return c;
}

__device__ __forceinline__ int load_flag_vector(const int* p) {
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    return __builtin_amdgcn_readfirstlane(p[vz]);
}

// 16-byte load with the non-temporal hint (global_load_dwordx4 ... nt): for operands that are streamed once per pass and
// are too large to stay in the 256 MB Infinity Cache anyway.  Measured on one MI355X (round 2, same box): the GEMV streams
// of C4 / C5 gain 10-13 % (C5 LAD 1278 -> 1430 it/s, BP 1375 -> 1525, C4 589 -> 646); operands that DO fit the cache and are
// re-read every iteration (the tall path's 200 MB triangle) lose 8 % with it and keep plain loads.
template <typename V>
__device__ __forceinline__ V load16_nt(const void* p) {
    static_assert(sizeof(V) == 16, "16-byte vectors only");
    typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
    const u4_t raw = __builtin_nontemporal_load(reinterpret_cast<const u4_t*>(p));
    V v;
    __builtin_memcpy(&v, &raw, 16);
    return v;
}

// 16-byte vector of T (float4 / double2) for coalesced 1 KiB-per-wave loads.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    using type = float4;
    static constexpr int N = 4;
    static __device__ __forceinline__ float dot(const float4& a, const float4& b, float acc) {
        acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc);
        acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
        return acc;
    }
    static __device__ __forceinline__ float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
};
template <> struct Vec16<double> {
    using type = double2;
    static constexpr int N = 2;
    static __device__ __forceinline__ double dot(const double2& a, const double2& b, double acc) {
        acc = fma(a.x, b.x, acc); acc = fma(a.y, b.y, acc);
        return acc;
    }
    static __device__ __forceinline__ double2 zero() { return make_double2(0.0, 0.0); }
};

}  // namespace admm
