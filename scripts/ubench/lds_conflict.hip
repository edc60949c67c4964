// Which LDS access patterns of the fused kernel does SQ_LDS_BANK_CONFLICT count?  (run under rocprofv3 --pmc)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, int iters) {
    extern __shared__ __align__(16) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 9600; i += 256) lds[i] = i;
    __syncthreads();
    double acc = 0;
    const double2 *p2;
    int stride2 = 0;
    p2 = (const double2 *)lds;
    if (MODE == 0) p2 = (const double2 *)(lds + lane * 62 + (tid >> 6) * 8);          // V main: column stride 62 doubles
    if (MODE == 1) p2 = (const double2 *)(lds + (tid % 32) * 94 + (tid / 32) * 8);    // H: row stride 94 doubles
    if (MODE == 2) p2 = (const double2 *)(lds + tid * 2);                             // consecutive 16 B
    if (MODE == 3) p2 = (const double2 *)(lds + lane * 64);                            // worst case: same banks
    if (MODE == 4) {                                                                  // pieces, R = 8: XC = 16 columns
        const int rgp = tid / 16, col = 64 + tid % 16;
        p2 = (const double2 *)(lds + col * 62 + rgp * 4);
    }
    if (MODE == 5) {                                                                  // pieces, R = 5: XC = 10 columns
        const int it = tid % 80, rgp = it / 10, col = 64 + it % 10;
        p2 = (const double2 *)(lds + col * 62 + rgp * 4);
    }
    if (MODE < 6) for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double2 v = p2[q + stride2];
            acc += v.x + v.y;
        }
        asm volatile("" ::: "memory");
    }
    if (MODE == 7 || MODE == 8) {                                                     // pieces, row group fastest: (A)
        const int XC = MODE == 7 ? 16 : 10;
        const int it = tid % (8 * XC), rgp = it & 7, col = 64 + (it >> 3);
        p2 = (const double2 *)(lds + col * 62 + rgp * 4);
        for (int it2 = 0; it2 < iters; ++it2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double2 v = p2[q];
                acc += v.x + v.y;
            }
            asm volatile("" ::: "memory");
        }
    }
    if (MODE == 9) {                                                                  // (A) writes: 4 rows x b64
        const int it = tid % 128, rgp = it & 7, col = 64 + (it >> 3);
        for (int it2 = 0; it2 < iters; ++it2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) lds[(rgp * 4 + q) * 94 + col] = acc + q;
            asm volatile("" ::: "memory");
        }
    }
    if (MODE == 10) {                                                                 // current piece writes
        const int rgp = tid / 16, col = 64 + tid % 16;
        for (int it2 = 0; it2 < iters; ++it2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) lds[((rgp & 7) * 4 + q) * 94 + col] = acc + q;
            asm volatile("" ::: "memory");
        }
    }
    if (MODE >= 11 && MODE <= 16) {                                                   // 16 cols x 4 row groups, row step S doubles
        const int S = MODE == 11 ? 8 : MODE == 12 ? 16 : MODE == 13 ? 2 : MODE == 14 ? 6 : MODE == 15 ? 10 : 12;
        const int rgp = lane / 16, col = 64 + lane % 16;
        p2 = (const double2 *)(lds + col * 62 + rgp * S);
        for (int it2 = 0; it2 < iters; ++it2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double2 v = p2[q];
                acc += v.x + v.y;
            }
            asm volatile("" ::: "memory");
        }
    }
    if (MODE >= 17 && MODE <= 22) {
        int off = 0;
        if (MODE == 17) off = (lane % 16 + 64) * 62 + (lane / 16) * 0;              // same cols, same rows (broadcast)
        if (MODE == 18) off = lane * 62 + (lane / 16) * 4;                          // distinct cols, row shift per 16 lanes
        if (MODE == 19) off = (lane % 32) * 62 + (lane / 32) * 4;                   // 32 cols x 2 row groups
        if (MODE == 20) off = lane * 62 + (lane / 32) * 4;                          // distinct cols, row shift per 32 lanes
        if (MODE == 21) off = lane * 62 + (lane & 1) * 4;                           // distinct cols, alternating row shift
        if (MODE == 22) off = (lane % 8) * 62 + (lane / 8) * 4;                     // 8 cols x 8 row groups
        p2 = (const double2 *)(lds + off);
        for (int it2 = 0; it2 < iters; ++it2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double2 v = p2[q];
                acc += v.x + v.y;
            }
            asm volatile("" ::: "memory");
        }
    }
    if (MODE == 6) {                                                                  // b64 writes, consecutive lanes
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) lds[q * 94 + tid % 64 + (tid >> 6) * 8 * 94] = acc + q;
            asm volatile("" ::: "memory");
        }
    }
    out[blockIdx.x * 256 + tid] = acc + lds[tid];
}
#define RUN(M) k<M><<<512, 256, 9600 * 8>>>(d, 200);
int main() {
    double *d; hipMalloc(&d, 512 * 256 * 8);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22)
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
