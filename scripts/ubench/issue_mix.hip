// Issue cost of non-FP64 VALU instructions next to FP64 ones on gfx950 (wave64): does a 32-bit VALU op (v_mov_b32,
// v_mov_b32 dpp, v_cndmask_b32, v_mov_b64) take a 4-cycle FP64-sized issue slot or a 2-cycle one?
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -w issue_mix.hip -o issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, double a, int iters) {
    double t[8];
    int m[8];
    for (int i = 0; i < 8; ++i) { t[i] = a + threadIdx.x * 1e-9 + i; m[i] = threadIdx.x + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0 || (MODE >= 4 && MODE <= 6)) t[i] = t[i] + a;                                   // v_add_f64
                if (MODE == 1 || MODE == 4) m[i] = __builtin_amdgcn_mov_dpp(m[i], 0x138, 0xf, 0xf, true);   // v_mov_b32 dpp
                if (MODE == 2 || MODE == 5) m[i] = (m[i] ^ 0x55) + 3;                         // 2 x 32-bit int VALU
                if (MODE == 3 || MODE == 6) {                                                  // v_mov_b64 (opaque copy)
                    double x = t[i];
                    asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(t[i]));
                    t[i] = x;
                }
                if (MODE == 7) asm volatile("v_max_f64 %0, %0, %1" : "+v"(t[i]) : "v"(a));
                if (MODE == 8) m[i] += (t[i] > a + m[i]) ? 1 : 0;                              // v_cmp_gt_f64 + v_addc / cndmask
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += t[i] + m[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *what, double per_iter) {
    double *d; hipMalloc(&d, 256 * 2048 * sizeof(double));
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 2; ++wps) {
        int blocks = 256 * wps;
        k<MODE><<<blocks, 256>>>(d, 1.0000001, 10);
        hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(d, 1.0000001, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double inst = (double)iters * 64 * per_iter * wps;     // wave-instructions per SIMD
        printf("%-34s waves/SIMD %d: %.3f ms -> %.3f ns per wave-instruction per SIMD\n", what, wps, ms, ms * 1e6 / inst);
    }
    hipFree(d);
}
int main() {
    run<0>("v_add_f64", 1); run<1>("v_mov_b32_dpp", 1); run<2>("v_xor+v_add (32-bit)", 2); run<3>("v_mov_b64", 1);
    run<4>("v_add_f64 + v_mov_b32_dpp", 2); run<5>("v_add_f64 + 2 x 32-bit", 3); run<6>("v_add_f64 + v_mov_b64", 2);
    run<7>("v_max_f64", 1); run<8>("v_cmp_gt_f64 + cvt/add (>=3 instr)", 3);
    return 0;
}
