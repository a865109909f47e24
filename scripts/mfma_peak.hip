// Dev tool (GPU box): what the fp32 matrix cores of THIS box sustain with no memory traffic at all -- the yardstick the Gram
// kernel's 0.87 of the nominal 157 TF/s should be read against (the nominal figure assumes 2.4 GHz; under a chip-wide MFMA load
// the clock settles lower).  Every wave issues independent v_mfma_f32_32x32x2_f32 on four accumulators; 1, 2 and 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/mfma_peak.hip -o scripts/_bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) mfma_kernel(float* out, int iters) {
    floatx16 a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
    float x = (float)threadIdx.x * 1e-3f, y = 1.f - x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
typedef double doublex4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) mfma64_kernel(double* out, int iters) {
    doublex4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    double x = (double)threadIdx.x * 1e-3, y = 1.0 - x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a4, 0, 0, 0);
            a5 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a5, 0, 0, 0);
            a6 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a6, 0, 0, 0);
            a7 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a7, 0, 0, 0);
        }
    }
    double s = 0;
    for (int r = 0; r < 4; ++r) s += a0[r] + a1[r] + a2[r] + a3[r] + a4[r] + a5[r] + a6[r] + a7[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs_per_cu : {1, 2, 4}) {
        const int grid = 256 * wgs_per_cu;
        for (int iters : {20000, 200000}) {                     // ~3 ms and ~30..110 ms: short burst vs sustained
            hipLaunchKernelGGL(mfma_kernel, dim3(grid), dim3(256), 0, 0, out, 1000);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_kernel, dim3(grid), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * 4 * iters * 32.0 * 4096.0;
            std::printf("waves/SIMD %d  iters %6d  %8.3f ms  %7.1f TF/s  (%.3f of 157.3)\n", wgs_per_cu, iters, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3);
        }
    }
    double* out64 = reinterpret_cast<double*>(out);
    for (int wgs_per_cu : {1, 2, 4}) {
        const int grid = 256 * wgs_per_cu / 2 * 2;
        for (int iters : {20000, 200000}) {
            hipLaunchKernelGGL(mfma64_kernel, dim3(grid), dim3(256), 0, 0, out64, 1000);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma64_kernel, dim3(grid), dim3(256), 0, 0, out64, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * 4 * iters * 32.0 * 2048.0;          // 16 x 16 x 4 x 2 flop per instruction
            std::printf("fp64  waves/SIMD %d  iters %6d  %8.3f ms  %7.1f TF/s  (%.3f of 78.6)\n", wgs_per_cu, iters, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 78.6);
        }
    }
    return 0;
}
