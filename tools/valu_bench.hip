// valu_bench.hip — what the FP32 vector pipe of one MI355X sustains for unfused mul/add streams.
// The render path may not contract a*b+c (reference semantics), so its ceiling is the issue rate
// of separate v_mul_f32 / v_add_f32 (or their packed forms), not the FMA peak.  Measures, at
// 1..8 waves per SIMD: scalar mul+add pairs, packed (v_pk_*) mul+add pairs, scalar FMA, packed FMA.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o valu_bench valu_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 4096
#define NACC 16

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float b, float c) {
    float a[NACC];
    v2f p[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = v2f{a[i], a[i] + 1.f}; }
    const v2f pb = {b, b}, pc = {c, c};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (KIND == 0) { asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
            if (KIND == 1) { asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(p[i]) : "v"(pb), "v"(pc)); }
            if (KIND == 2) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
            if (KIND == 3) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc)); }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
double run(int blocks, float* d, const char* name, double flops_per_iter_lane) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001f, 1e-9f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001f, 1e-9f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 256 * ITERS * NACC * flops_per_iter_lane;
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("%-28s blocks/CU=%d  %8.3f ms  %7.2f TFLOP/s\n", name, blocks / 256, ms, tf);
    return tf;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d MHz\n", prop.name, cu, prop.clockRate / 1000);
    float* d; hipMalloc(&d, sizeof(float) * 256 * cu * 8);
    for (int per_cu : {1, 2, 4, 8}) {   // 256-thread blocks per CU = waves per SIMD
        const int blocks = cu * per_cu;
        run<0>(blocks, d, "v_mul_f32+v_add_f32", 2.0);
        run<1>(blocks, d, "v_pk_mul_f32+v_pk_add_f32", 4.0);
        run<2>(blocks, d, "v_fma_f32", 2.0);
        run<3>(blocks, d, "v_pk_fma_f32", 4.0);
    }
    return 0;
}
