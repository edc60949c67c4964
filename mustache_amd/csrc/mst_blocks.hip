// Dense-block construction for the scale-space path (gfx950).
//   mst_scatter_blocks  <- reference mustache/mustache.py:919-924 (COO -> dense block)
//   mst_block_prologue  <- reference mustache/mustache.py:699-706 (nz mask + constant fills)
// Both are pure HBM-bound byte movers: coalesced 16-byte accesses, grid-stride, no LDS needed.
#include "mst_common.h"

namespace {

constexpr int kThreads = 256;

// One thread per COO entry; each entry lands in every block whose [start, start+CH) range holds both x and y.
// Consecutive blocks overlap, so an entry can belong to 2-3 blocks; `starts` is ascending (regulator's tiling),
// which lets us bracket the candidates with two binary searches instead of scanning all B.
__global__ void __launch_bounds__(kThreads)
scatter_kernel(const int64_t *__restrict__ x, const int64_t *__restrict__ y, const double *__restrict__ v,
               int64_t nnz, const int64_t *__restrict__ starts, int B, int CH, double *__restrict__ c) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        const int64_t xi = x[i], yi = y[i];
        const double vi = v[i];
        const int64_t lo_v = (xi < yi ? xi : yi), hi_v = (xi < yi ? yi : xi);
        // blocks with start <= lo_v and start + CH > hi_v  <=>  start in (hi_v - CH, lo_v]
        int lo = 0, hi = B;  // first block with start > hi_v - CH
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (starts[mid] > hi_v - CH) hi = mid; else lo = mid + 1;
        }
        for (int b = lo; b < B && starts[b] <= lo_v; ++b) {
            const int64_t s = starts[b];
            c[((int64_t)b * CH + (xi - s)) * CH + (yi - s)] = vi;
        }
    }
}

// VEC = 4: 4 pixels (32 bytes of c, 4 bytes of nz) per thread per step; VEC = 1: any CH.
template <int VEC>
__global__ void __launch_bounds__(kThreads)
prologue_kernel(double *__restrict__ c, uint8_t *__restrict__ nz, uint32_t *__restrict__ nz_count, int CH,
                int dpx, int intra) {
    const int b = blockIdx.y;
    const int64_t nq = ((int64_t)CH * CH) / VEC;
    double *cb = c + (int64_t)b * CH * CH;
    uint8_t *nb = nz + (int64_t)b * CH * CH;
    uint32_t local = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
        const int64_t p = q * VEC;
        const int row = (int)(p / CH);
        const int col = (int)(p - (int64_t)row * CH);
        double val[VEC];
        if constexpr (VEC == 4) {
            double2 a0 = *reinterpret_cast<const double2 *>(cb + p);
            double2 a1 = *reinterpret_cast<const double2 *>(cb + p + 2);
            val[0] = a0.x; val[1] = a0.y; val[2] = a1.x; val[3] = a1.y;
        } else {
            val[0] = cb[p];
        }
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int off = col + k - row;
            const bool t = (val[k] != 0.0) && (off >= 4);
            bits |= (t ? 1u : 0u) << (8 * k);
            local += t ? 1u : 0u;
            if (off <= 4 || (intra && off >= dpx + 1)) val[k] = 2.0;
        }
        if constexpr (VEC == 4) {
            *reinterpret_cast<double2 *>(cb + p) = make_double2(val[0], val[1]);
            *reinterpret_cast<double2 *>(cb + p + 2) = make_double2(val[2], val[3]);
            *reinterpret_cast<uint32_t *>(nb + p) = bits;
        } else {
            cb[p] = val[0];
            nb[p] = (uint8_t)bits;
        }
    }
    // integer count: wave reduce, then one atomic per wave (exact, order independent)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(nz_count + b, local);
}

}  // namespace

extern "C" int mst_scatter_blocks(const int64_t *x, const int64_t *y, const double *v, int64_t nnz,
                                  const int64_t *starts, int32_t B, int32_t CH, double *c, void *stream) {
    if (!c || !starts || B <= 0 || CH <= 0 || nnz < 0 || (nnz > 0 && (!x || !y || !v)))
        return mst::fail(MST_E_ARG, "mst_scatter_blocks: bad argument");
    for (int b = 1; b < B; ++b)
        if (starts[b] < starts[b - 1]) return mst::fail(MST_E_ARG, "mst_scatter_blocks: starts must ascend");
    hipStream_t s = mst::as_stream(stream);
    MST_HIP(hipMemsetAsync(c, 0, sizeof(double) * (size_t)B * CH * CH, s));
    if (nnz == 0) return MST_OK;
    int64_t *d_starts = nullptr;
    MST_HIP(hipMallocAsync((void **)&d_starts, sizeof(int64_t) * B, s));
    MST_HIP(mst::upload_small(d_starts, starts, sizeof(int64_t) * B, s));
    int64_t want = (nnz + kThreads - 1) / kThreads;
    int grid = (int)(want < 65536 ? want : 65536);
    scatter_kernel<<<grid, kThreads, 0, s>>>(x, y, v, nnz, d_starts, B, CH, c);
    MST_LAUNCH_CHECK();
    MST_HIP(hipFreeAsync(d_starts, s));
    return MST_OK;
}

extern "C" int mst_block_prologue(double *c, uint8_t *nz, uint32_t *nz_count, int32_t B, int32_t CH, int32_t dpx,
                                  int32_t intra, void *stream) {
    if (!c || !nz || !nz_count || B <= 0 || CH <= 0 || dpx < 0)
        return mst::fail(MST_E_ARG, "mst_block_prologue: bad argument");
    hipStream_t s = mst::as_stream(stream);
    MST_HIP(hipMemsetAsync(nz_count, 0, sizeof(uint32_t) * B, s));
    const bool vec = (CH & 3) == 0 && (reinterpret_cast<uintptr_t>(c) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(nz) & 3) == 0;
    int64_t nq = ((int64_t)CH * CH) / (vec ? 4 : 1);
    int64_t want = (nq + kThreads - 1) / kThreads;
    int gx = (int)(want < 4096 ? want : 4096);
    if (vec)
        prologue_kernel<4><<<dim3(gx, B), kThreads, 0, s>>>(c, nz, nz_count, CH, dpx, intra);
    else
        prologue_kernel<1><<<dim3(gx, B), kThreads, 0, s>>>(c, nz, nz_count, CH, dpx, intra);
    MST_LAUNCH_CHECK();
    return MST_OK;
}
