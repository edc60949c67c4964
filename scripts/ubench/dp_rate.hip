// FP64 VALU issue-rate microbenchmark (gfx950): independent chains of v_add_f64 / v_mul_f64 / v_fma_f64 / mixed
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, double a, double b, int iters) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) x[i] = x[i] + b;
                else if (MODE == 1) x[i] = x[i] * b;
                else if (MODE == 2) x[i] = __builtin_fma(x[i], b, a);
                else if (MODE == 3) { x[i] = x[i] + b; x[i] = x[i] * a; }
                else if (MODE == 4) x[i] = (x[i] > b) ? x[i] : b;     // cmp + cndmask
                else if (MODE == 5) x[i] = __builtin_fmax(x[i], b);
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int ops_per_inner) {
    double *d; hipMalloc(&d, 256 * 2048 * 8 * sizeof(double));
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
        int blocks = 256 * waves_per_simd;  // 256-thread blocks: 1 wave per SIMD per block
        k<MODE><<<blocks, 256>>>(d, 1.0, 1.0000001, 10);
        hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(d, 1.0, 1.0000001, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double winst = (double)blocks * 4 * iters * 64 * ops_per_inner;   // wave-instructions
        double per_simd = winst / (256.0 * 4);
        printf("%-10s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f clk @2.4GHz)\n", name,
               waves_per_simd, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    }
    hipFree(d);
}
int main() {
    run<0>("add_f64", 1); run<1>("mul_f64", 1); run<2>("fma_f64", 1); run<3>("add+mul", 2); run<4>("cmp+sel", 1); run<5>("fmax", 1);
    return 0;
}
