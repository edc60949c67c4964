// Issue cost of the 32-bit float VALU instructions a high-word 3x3 maximum would use, next to v_max_f64 (gfx950, wave64):
// v_max_f32, v_max3_f32, v_cmp_lt_f32 into an SGPR pair (+ s_or), v_or_b32, v_mov_b32 dpp.
//   hipcc -O3 --offload-arch=gfx950 -w f32_ops.hip -o f32_ops
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, float a, int iters) {
    float t[8];
    double dd[8];
    unsigned long long acc = 0;
    for (int i = 0; i < 8; ++i) { t[i] = a + threadIdx.x * 1e-3f + i; dd[i] = t[i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_max_f32 %0, %0, %1" : "+v"(t[i]) : "v"(a));
                if (MODE == 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(t[i]) : "v"(a), "v"(t[(i + 1) & 7]));
                if (MODE == 2) {
                    unsigned long long m;
                    asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(t[i]), "v"(a));
                    acc |= m;
                }
                if (MODE == 3) asm volatile("v_or_b32 %0, %0, %1" : "+v"(t[i]) : "v"(a));
                if (MODE == 4) asm volatile("v_max_f64 %0, %0, %1" : "+v"(dd[i]) : "v"((double)a));
                if (MODE == 5) {
                    unsigned long long m;
                    asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(dd[i]), "v"((double)a));
                    acc |= m;
                }
                if (MODE == 6) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(t[i]), "v"(a) : "vcc");
                if (MODE == 7) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(t[i]));
                if (MODE == 8) asm volatile("v_max_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(t[i]) : "v"(a));
                if (MODE == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(t[i]) : "v"(a) : "vcc");
            }
        }
    }
    float s = (float)(acc & 1);
    for (int i = 0; i < 8; ++i) s += t[i] + (float)dd[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *what) {
    float *d; hipMalloc(&d, 256 * 2048 * sizeof(float));
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 2; ++wps) {
        int blocks = 256 * wps;
        k<MODE><<<blocks, 256>>>(d, 1.0000001f, 10);
        hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(d, 1.0000001f, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double inst = (double)iters * 64 * wps;     // wave-instructions per SIMD
        printf("%-34s waves/SIMD %d: %.3f ms -> %.3f ns per wave-instruction per SIMD\n", what, wps, ms, ms * 1e6 / inst);
    }
    hipFree(d);
}
int main() {
    run<0>("v_max_f32"); run<1>("v_max3_f32"); run<2>("v_cmp_lt_f32 -> sgpr (+s_or)"); run<3>("v_or_b32");
    run<4>("v_max_f64"); run<5>("v_cmp_lt_f64 -> sgpr (+s_or)"); run<6>("v_cmp_lt_f32 -> vcc"); run<7>("v_mov_b32_dpp wave_shr");
    run<8>("v_max_f32_dpp wave_shr"); run<9>("v_cndmask_b32");
    return 0;
}
