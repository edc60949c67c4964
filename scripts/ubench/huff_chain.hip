// How fast can ONE wavefront walk a Huffman-coded stream?  The serial chain of a deflate decoder -- index = low bits of the bit
// buffer, table entry, shift by the entry's length -- in three forms: (A) wave-uniform code with the table in LDS (ds_read +
// v_readfirstlane), (B) wave-uniform code with the table behind the scalar cache (s_load), (C) lane 0 alone with the table in LDS.
// Each wave decodes `nsym` symbols of its own random stream; 768 waves (three per CU) run at once, like the 745 blocks of the
// bench's chr1 .hic would.      hipcc -O3 --offload-arch=gfx950 -w huff_chain.hip -o huff_chain && ./huff_chain
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kTableBits = 11;
constexpr int kInWords = 8192;            // 32 KB of stream per wave, walked round and round

template <int MODE>
__global__ __launch_bounds__(64) void chain(const uint32_t *__restrict__ table_g, const uint32_t *__restrict__ in_g, int nsym,
                                            uint8_t *out_g, uint64_t *cycles, uint32_t *sink) {
    __shared__ uint32_t table[1 << kTableBits];
    __shared__ uint32_t in[kInWords];
    __shared__ uint8_t window[32768];
    const int lane = threadIdx.x;
    for (int i = lane; i < (1 << kTableBits); i += 64) table[i] = table_g[i];
    const uint32_t *mine = in_g + (size_t)blockIdx.x * kInWords;
    for (int i = lane; i < kInWords; i += 64) in[i] = mine[i];
    __syncthreads();
    uint64_t t0 = clock64();
    uint32_t acc = 0;
    if (MODE != 2 || lane == 0) {
        uint64_t bitbuf = ((uint64_t)in[1] << 32) | in[0];
        int bits = 64, wp = 2;
        uint32_t next = in[wp];
        uint32_t outpos = 0;
        for (int s = 0; s < nsym; ++s) {
            uint32_t idx = (uint32_t)bitbuf & ((1u << kTableBits) - 1);
            if (MODE != 2) idx = __builtin_amdgcn_readfirstlane(idx);
            uint32_t e = MODE == 1 ? table_g[idx] : table[idx];
            if (MODE != 2) e = __builtin_amdgcn_readfirstlane(e);
            const int len = e & 15;
            bitbuf >>= len;
            bits -= len;
            if (e & 0x100) {                                   // "literal": one byte into the window
                window[outpos & 32767] = (uint8_t)(e >> 16);
                outpos += 1;
            } else {                                           // "match": a short copy inside the window
                const uint32_t dist = 1 + ((e >> 16) & 1023), n = 3 + ((e >> 9) & 3);
                for (uint32_t k = 0; k < n; ++k) window[(outpos + k) & 32767] = window[(outpos + k - dist) & 32767];
                outpos += n;
            }
            if (bits <= 32) {
                bitbuf |= (uint64_t)next << bits;
                bits += 32;
                wp = (wp + 1) & (kInWords - 1);
                next = in[wp];
            }
            acc += e;
        }
        acc += outpos;
    }
    uint64_t t1 = clock64();
    __syncthreads();
    if (lane == 0) {
        cycles[blockIdx.x] = t1 - t0;
        sink[blockIdx.x] = acc + window[acc & 32767];
    }
}

int main() {
    const int waves = 768, nsym = 200000;
    std::vector<uint32_t> table(1 << kTableBits), in((size_t)waves * kInWords);
    srand(7);
    for (auto &e : table) {
        const int len = 6 + rand() % 5;                         // 6..10 bits per symbol
        const bool lit = rand() % 9 < 7;                        // 78 % literals, as in the .hic blocks
        e = (uint32_t)len | (lit ? 0x100 : 0) | ((uint32_t)(rand() & 0xFFFF) << 16) | ((uint32_t)(rand() & 3) << 9);
    }
    for (auto &w : in) w = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    uint32_t *d_table, *d_in, *d_sink;
    uint8_t *d_out;
    uint64_t *d_cyc;
    CK(hipMalloc(&d_table, table.size() * 4));
    CK(hipMalloc(&d_in, in.size() * 4));
    CK(hipMalloc(&d_out, 1 << 20));
    CK(hipMalloc(&d_cyc, waves * 8));
    CK(hipMalloc(&d_sink, waves * 4));
    CK(hipMemcpy(d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const char *names[3] = {"A uniform code, table in LDS", "B uniform code, table through the scalar cache", "C lane 0 alone, table in LDS"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a));
            if (mode == 0) chain<0><<<waves, 64>>>(d_table, d_in, nsym, d_out, d_cyc, d_sink);
            if (mode == 1) chain<1><<<waves, 64>>>(d_table, d_in, nsym, d_out, d_cyc, d_sink);
            if (mode == 2) chain<2><<<waves, 64>>>(d_table, d_in, nsym, d_out, d_cyc, d_sink);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
        }
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        std::vector<uint64_t> cyc(waves);
        CK(hipMemcpy(cyc.data(), d_cyc, waves * 8, hipMemcpyDeviceToHost));
        double mean = 0;
        for (auto c : cyc) mean += (double)c;
        mean /= waves;
        printf("%-50s %8.3f ms for %d waves x %d symbols = %6.1f ns per symbol per wave; clock64 ticks per symbol %.1f\n", names[mode], ms,
               waves, nsym, ms * 1e6 / nsym, mean / nsym);
    }
    return 0;
}
