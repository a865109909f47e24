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
