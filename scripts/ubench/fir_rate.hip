// How fast does the tap-major FIR body issue on gfx950?  Register-only (no LDS), 8 or 4 chains, radius 14 / 4.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K, int R>
__device__ __forceinline__ void fir_sym(const double (&win)[K + 2 * R], const double (&w)[R + 1], double (&t)[K]) {
#pragma unroll
    for (int k = 0; k < K; ++k) t[k] = win[k + R] * w[0];
#pragma unroll
    for (int j = R; j >= 1; --j) {
        double s[K];
#pragma unroll
        for (int k = 0; k < K; ++k) s[k] = win[k + R - j] + win[k + R + j];
#pragma unroll
        for (int k = 0; k < K; ++k) s[k] = s[k] * w[j];
#pragma unroll
        for (int k = 0; k < K; ++k) t[k] = t[k] + s[k];
    }
}
template <int K, int R, bool WV>
__global__ void __launch_bounds__(256) kern(double *out, const double *in, const double *wg, int iters) {
    double win[K + 2 * R], w[R + 1];
    for (int i = 0; i < K + 2 * R; ++i) win[i] = in[threadIdx.x + 256 * i];
    for (int j = 0; j <= R; ++j) w[j] = WV ? wg[j + (threadIdx.x & 1)] : wg[j];    // WV: weights in VGPRs
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        double t[K];
        fir_sym<K, R>(win, w, t);
#pragma unroll
        for (int k = 0; k < K; ++k) { win[k] = t[k]; acc += t[k]; }   // feed back: nothing hoists out of the loop
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int K, int R, bool WV> void run(const char *name) {
    double *d, *in, *w; hipMalloc(&d, 8 << 20); hipMalloc(&in, 8 << 20); hipMalloc(&w, 4096);
    hipMemset(in, 0, 8 << 20); hipMemset(w, 0, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 400;
    for (int wps = 1; wps <= 2; wps *= 2) {
        int blocks = 256 * wps;
        kern<K, R, WV><<<blocks, 256>>>(d, in, w, 5);
        hipEventRecord(e0);
        kern<K, R, WV><<<blocks, 256>>>(d, in, w, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double dp = (double)iters * K * (3 * R + 1 + 1);          // DP instr per wave (incl. the acc add)
        double per_simd = dp * wps;                                 // waves per SIMD
        printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f ns per DP instr per SIMD\n", name, wps, ms, ms * 1e6 / per_simd);
    }
}
int main() {
    run<8, 14, false>("K8 R14 sgpr-taps"); run<8, 14, true>("K8 R14 vgpr-taps");
    run<4, 14, false>("K4 R14 sgpr-taps"); run<8, 4, false>("K8 R4 sgpr-taps");
    return 0;
}
