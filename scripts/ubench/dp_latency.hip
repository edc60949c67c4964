// FP64 VALU dependent-issue latency on gfx950: NCH independent chains of add->mul->add (the FIR's per-tap pattern),
// 1 or 2 waves per SIMD.  cycles per instruction vs NCH shows how many independent chains hide the pipeline latency.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCH, int MODE>
__global__ void __launch_bounds__(256) k(double *out, double a, double b, int iters) {
    double x[NCH], t[NCH];
    for (int i = 0; i < NCH; ++i) { x[i] = a + threadIdx.x * 1e-9 + i; t[i] = b * i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {            // pure dependent adds
#pragma unroll
                for (int i = 0; i < NCH; ++i) x[i] = x[i] + b;
            } else {                    // tap pattern: s = x + t; s *= w; t += s   (x perturbed so nothing hoists)
                double s[NCH];
#pragma unroll
                for (int i = 0; i < NCH; ++i) s[i] = x[i] + t[i];
#pragma unroll
                for (int i = 0; i < NCH; ++i) s[i] = s[i] * a;
#pragma unroll
                for (int i = 0; i < NCH; ++i) t[i] = t[i] + s[i];
            }
        }
    }
    double s = 0;
    for (int i = 0; i < NCH; ++i) s += x[i] + t[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH, int MODE> void run() {
    double *d; hipMalloc(&d, 256 * 2048 * sizeof(double));
    int iters = 1000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = 256 * wps;
        k<NCH, MODE><<<blocks, 256>>>(d, 1.0, 1.0000001, 10);
        hipEventRecord(e0);
        k<NCH, MODE><<<blocks, 256>>>(d, 1.0, 1.0000001, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double inst = (double)iters * 16 * NCH * (MODE == 0 ? 1 : 3) * wps;     // wave-instructions per SIMD
        printf("mode %d chains %d waves/SIMD %d: %.3f ms -> %.2f ns per instr per SIMD\n", MODE, NCH, wps, ms, ms * 1e6 / inst);
    }
    hipFree(d);
}
int main() {
    run<1, 0>(); run<2, 0>(); run<4, 0>(); run<8, 0>();
    run<1, 1>(); run<2, 1>(); run<4, 1>(); run<8, 1>();
    return 0;
}
