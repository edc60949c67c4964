// Does the FP64 matrix pipe of gfx950 run next to the FP64 vector pipe?  v_mfma_f64_16x16x4_f64 with a 0/1 selection matrix
// as A computes, lane-locally, D[v] = B + C[v] (v = 0..3): four exact additions (the three zero products add nothing), i.e.
// a possible second adder for the blur's pair sums -- if and only if MFMA and VALU FP64 instructions overlap.
//   mode 0: 32 v_add_f64 per iteration          mode 1: 8 MFMA per iteration (= 32 lane-local additions)
//   mode 2: both, independent                   mode 3: 32 v_add_f64 + 4 MFMA      mode 4: 4 MFMA
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -w mfma_overlap.hip -o mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, double a, int iters) {
    double t[8];
    double4_t c[8];
    const int lane = threadIdx.x & 63;
    const double sel = ((lane >> 4) == ((lane & 15) >> 2)) ? 1.0 : 0.0;      // A[i][k] = (k == i / 4), i = lane % 16, k = lane / 16
    for (int i = 0; i < 8; ++i) {
        t[i] = a + threadIdx.x * 1e-9 + i;
        c[i] = double4_t{a + i, a + 2 * i, a - i, a * i} + threadIdx.x * 1e-9;
    }
    double b = a * 0.5 + lane * 1e-7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = t[i] + a;
            }
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i) c[2 * u % 8 + i] = __builtin_amdgcn_mfma_f64_16x16x4f64(sel, b, c[2 * u % 8 + i], 0, 0, 0);
            }
            if (MODE == 3 || MODE == 4) c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(sel, b, c[u], 0, 0, 0);
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += t[i] + c[i].x + c[i].y + c[i].z + c[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// exactness: D[v] == B + C[v] bit for bit, lane by lane
__global__ void check(double *bad) {
    const int lane = threadIdx.x & 63;
    const double sel = ((lane >> 4) == ((lane & 15) >> 2)) ? 1.0 : 0.0;
    double n = 0;
    for (int it = 0; it < 2000; ++it) {
        const double b = (1.0 + lane * 0.37 + it * 1.1e-3) * (it % 3 == 0 ? 1e-9 : (it % 3 == 1 ? 1.0 : -3e5));
        double4_t c = {lane * 1.3 - it * 7e-4, -b, 1e-17 * lane, 2.0};
        const double4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(sel, b, c, 0, 0, 0);
        const double e0 = b + c.x, e1 = b + c.y, e2 = b + c.z, e3 = b + c.w;
        n += (d.x != e0) + (d.y != e1) + (d.z != e2) + (d.w != e3);
    }
    bad[threadIdx.x] = n;
}
template <int MODE> void run(const char *what) {
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
        printf("%-44s waves/SIMD %d: %.3f ms = %.1f ns per iteration per wave-slot\n", what, wps, ms, ms * 1e6 / iters / wps);
    }
    hipFree(d);
}
int main() {
    double *bad; hipMalloc(&bad, 256 * sizeof(double));
    check<<<1, 256>>>(bad);
    double h[256]; hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
    double nb = 0; for (int i = 0; i < 256; ++i) nb += h[i];
    printf("MFMA as adder: %g mismatches against v_add_f64 in 256 lanes x 2000 x 4 additions\n", nb);
    run<0>("32 v_add_f64"); run<1>("8 MFMA f64 16x16x4 (32 lane-local adds)"); run<2>("32 v_add_f64 + 8 MFMA (independent)");
    run<4>("4 MFMA"); run<3>("32 v_add_f64 + 4 MFMA (independent)");
    return 0;
}
