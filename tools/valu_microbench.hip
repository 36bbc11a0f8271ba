// Throughput of the vector instructions the tracer is made of, per SIMD, on the GPU at hand.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o valu_microbench tools/valu_microbench.hip
// Each kernel runs ITER iterations of 8 independent chains per lane; grid = 256 CUs x 4 SIMDs x W waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int ITER = 4096;

template <int OP> __device__ __forceinline__ void step(double (&a)[8], double b, double c);
template <int OP> __device__ __forceinline__ void stepf(float (&a)[8], float b, float c);

template <int OP>
__global__ void k64(double* out, double b, double c) {
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + 1e-3 * (threadIdx.x + i);
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_fma(a[i], b, c);
            if (OP == 1) a[i] = a[i] + b;
            if (OP == 2) a[i] = a[i] * b;
            if (OP == 3) a[i] = c / a[i];               // full IEEE division
            if (OP == 4) a[i] = __builtin_sqrt(a[i]) + b;  // full IEEE sqrt (+1 add)
            if (OP == 5) a[i] = __builtin_amdgcn_rcp(a[i]);
            if (OP == 6) a[i] = __builtin_amdgcn_rsq(a[i]);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 12345.678) out[0] = s;
}

typedef float float2_ __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void k32(float* out, float b, float c) {
    float a[8];
    float2_ p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = 1.0f + 1e-3f * (threadIdx.x + i); p[i] = float2_{a[i], a[i] + 1.f}; }
    float2_ pb = {b, b}, pc = {c, c};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_fmaf(a[i], b, c);
            if (OP == 1) a[i] = a[i] + b;
            if (OP == 2) a[i] = a[i] * b;
            if (OP == 3) p[i] = __builtin_elementwise_fma(p[i], pb, pc);   // v_pk_fma_f32
            if (OP == 4) a[i] = __builtin_amdgcn_sqrtf(a[i]) + b;
            if (OP == 5) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (OP == 6) a[i] = __builtin_amdgcn_rsqf(a[i]);
            if (OP == 7) a[i] = c / a[i];                 // IEEE float division
            if (OP == 8) a[i] = fmaxf(a[i], b) ;          // v_max
            if (OP == 9) a[i] = (a[i] > b) ? c : a[i] + 1.0f;   // cmp + cndmask + add
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <typename F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 3;
}

int main() {
    double* d64; float* d32;
    CHECK(hipMalloc(&d64, 64)); CHECK(hipMalloc(&d32, 64));
    const char* n64[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "f64 div (IEEE)", "f64 sqrt(IEEE)+add", "v_rcp_f64", "v_rsq_f64"};
    const char* n32[] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_pk_fma_f32", "v_sqrt_f32+add", "v_rcp_f32", "v_rsq_f32",
                         "f32 div (IEEE)", "v_max_f32", "cmp+cndmask+add f32"};
    for (int waves : {1, 2, 4}) {
        const int blocks = 256 * waves;   // 256-thread blocks: one wave per SIMD each
        printf("== %d wave(s) per SIMD, %d blocks x 256 threads, %d iters x 8 chains\n", waves, blocks, ITER);
        auto report = [&](const char* name, double ms) {
            // SIMD-cycles per wave-level op, assuming 2.4 GHz
            double ops_per_simd = (double)waves * ITER * 8;
            printf("  %-22s %8.3f ms  -> %6.2f cycles/op/SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / ops_per_simd);
        };
#define R64(OP) report(n64[OP], time_ms([&] { hipLaunchKernelGGL(k64<OP>, dim3(blocks), dim3(256), 0, 0, d64, 1.0000001, 1e-9); }));
        R64(0) R64(1) R64(2) R64(3) R64(4) R64(5) R64(6)
#define R32(OP) report(n32[OP], time_ms([&] { hipLaunchKernelGGL(k32<OP>, dim3(blocks), dim3(256), 0, 0, d32, 1.0000001f, 1e-9f); }));
        R32(0) R32(1) R32(2) R32(3) R32(4) R32(5) R32(6) R32(7) R32(8) R32(9)
    }
    return 0;
}
