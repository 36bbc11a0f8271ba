// How many scalar-ALU instructions per cycle does one CU sustain, as waves per CU grow?
// Build: hipcc --offload-arch=gfx950 -O3 -o salu_microbench tools/salu_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int ITER = 2048;

template <int MODE>
__global__ void k(unsigned* out, unsigned seed) {
    unsigned s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0 || MODE == 2) {
            // 16 independent-ish scalar adds
            asm volatile(
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_add_u32 %3, %3, 7\n"
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_add_u32 %3, %3, 7\n"
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_add_u32 %3, %3, 7\n"
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_add_u32 %3, %3, 7\n"
                : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        }
        if (MODE == 1 || MODE == 2) {
            asm volatile(
                "v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n"
                "v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n"
                "v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n"
                "v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n"
                : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        }
    }
    if (s0 + s1 + s2 + s3 + (unsigned)(v0 + v1 + v2 + v3) == 0x12345678u) out[0] = 1;
}

template <int MODE>
double run(int waves_per_cu, unsigned* d) {
    // one block of waves_per_cu*64 threads per CU
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    dim3 grid(256), block(64 * waves_per_cu);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, 1u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5;
}

int main() {
    unsigned* d; CHECK(hipMalloc(&d, 64));
    const char* names[] = {"16 SALU/iter", "16 VALU(f32)/iter", "16 SALU + 16 VALU/iter"};
    for (int wpc : {1, 4, 8, 12, 16}) {
        double t0 = run<0>(wpc, d), t1 = run<1>(wpc, d), t2 = run<2>(wpc, d);
        double n = (double)ITER * 16 * wpc;   // instructions of each kind per CU
        printf("waves/CU %2d: SALU-only %.3f ms (%.2f inst/cyc/CU)  VALU-only %.3f ms (%.2f inst/cyc/CU)  both %.3f ms\n",
               wpc, t0, n / (t0 * 1e-3 * 2.4e9), t1, n / (t1 * 1e-3 * 2.4e9), t2);
    }
    (void)names;
    return 0;
}
