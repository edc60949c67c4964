// Shared device code of the separable Gaussian passes (gfx950): tile geometry, the SciPy-ordered FIR chunk, the axis-0 and
// axis-1 passes out of LDS.  Included by mst_scale_space.hip (the fused sigma-stack kernel) and mst_diff.hip (the DoG of the
// two-sample difference image); see mst_scale_space.hip for the design notes.  Everything lives in an anonymous namespace:
// each translation unit gets its own copy, compiled with that file's flags (-ffp-contract=off in both).
#pragma once
#include <cstdint>
#include "mst_common.h"

// Timing ablations (blur only, no wave reductions, staging only) exist in PROFILE builds only (make PROFILE=1 ->
// libmustache_hip_profile.so); the product library has no run-time switch that could change a result.
#if defined(MST_PROFILE) && defined(MST_ABLATE_CT)     /* the ablation fixed at compile time (variant builds): dead phases cost no registers */
#define MST_VARIANT(bit_) ((MST_ABLATE_CT) & (bit_))
#elif defined(MST_PROFILE)
#define MST_VARIANT(bit_) (variant & (bit_))
#else
#define MST_VARIANT(bit_) 0
#endif

// PROFILE builds can record a phase timeline (s_memtime per wave at the phase boundaries of every level) for a sample of
// workgroups: MST_TRACE=<file> (scripts/trace_timeline.py reads it).  ~+10 % run time while it is on.
#ifdef MST_PROFILE
#define MST_TRACE_WGS 64
#define MST_TRACE_STAMPS 8
#define MST_STAMP(tr_, slot_)                                                        \
    if (tr_) {                                                                       \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                  \
        if ((threadIdx.x & 63) == 0) (tr_)[slot_] = t_;                              \
    }
#else
#define MST_STAMP(tr_, slot_)
#endif

#ifndef MST_KC8_BELOW
#define MST_KC8_BELOW 11      // radii below this use one 8-output window per item, wider ones two 4-output windows
#endif

namespace {

// Pricing experiment of round 6 (variant builds with -DMST_PROFILE -DMST_INJECT=1|2 only; never in the product): a block of
// VALU work that depends on nothing but its own registers -- INJ_DP v_max_f64 and INJ_INT 32-bit operations per thread, 8
// independent chains each, the mix of the max / sieve / statistics phase (scripts/isa_census.py) -- placed either INSIDE the
// axis-0 pass's main basic block (1: where an intra-wave overlap of level l's sieve with level l + 1's axis-0 pass would put
// the sieve's instructions) or in a block of its own behind the level's second barrier (2: where the sieve is today).
// Non-volatile asm: the optimiser cannot fold it, the scheduler may place it anywhere in its basic block.
#if defined(MST_INJECT)
#ifndef MST_INJ_DP
#define MST_INJ_DP 96
#endif
#ifndef MST_INJ_INT
#define MST_INJ_INT 224
#endif
struct Inject {
    double q[8];
    uint32_t u[8];
};
template <int NDP = MST_INJ_DP, int NINT = MST_INJ_INT>
__device__ __forceinline__ void inject_work(Inject &z) {
#pragma unroll
    for (int i = 0; i < NDP; ++i) asm("v_max_f64 %0, %1, %2" : "=v"(z.q[i & 7]) : "v"(z.q[i & 7]), "v"(z.q[(i + 1 + i / 8) & 7]));
#pragma unroll
    for (int i = 0; i < NINT; ++i) asm("v_xor_b32 %0, %1, %2" : "=v"(z.u[i & 7]) : "v"(z.u[i & 7]), "v"(z.u[(i + 3 + i / 8) & 7]));
}
#define MST_INJECT_ARG , Inject &inj
#define MST_INJECT_PASS , inj
#else
#define MST_INJECT_ARG
#define MST_INJECT_PASS
#endif

struct DevLevels {
    int n_octaves;
    int levels_per_octave;
    int radius[MST_MAX_LEVELS];
    int first_level[16];          // per octave: 1, or 3 when its first two levels repeat the previous octave's last two
    double taps[MST_MAX_LEVELS][MST_MAX_RADIUS + 1];
};

constexpr int pitch_for(int need) {
    // Window loads are 16-byte ds_read_b128 (full LDS rate; ds_read2_b64 runs at half).  Measured rule on gfx950
    // (scripts/ubench/lds_conflict.hip, SQ_LDS_BANK_CONFLICT): a b128 access is conflict-free iff the 32 lanes of each
    // half-wave hit 32 distinct 16-byte slots modulo 512 bytes.  With lanes striding whole rows, a pitch of 2 or 30
    // (mod 32) doubles -- pitch/2 odd -- makes 32 consecutive rows land on 32 distinct slots, and keeps the loads
    // 16-byte aligned.  (The V pass's 4-row leftover pieces cannot meet the rule for radii below 9: their half-waves
    // mix up to 8 row groups with < 18 columns, at most 14 + 2r distinct slots -- they run at half LDS rate.)
    int p = need;
    while (p % 32 != 2 && p % 32 != 30) ++p;
    return p;
}
// The rule itself only needs pitch / 2 odd (a unit modulo 32: lane i of a half-wave then lands on slot i * pitch / 2 + c, 32
// distinct values); pitch_for's 2 / 30 (mod 32) is the subset the product tiles were measured with.  Tiles that must fit a
// tighter LDS budget (the experimental radius-7 tile) take the smallest even pitch with an odd half.
constexpr int pitch_min(int need) {
    int p = need + (need & 1);
    while ((p / 2) % 2 == 0) p += 2;
    return p;
}

template <int RGR_, int RGC_, int RMAX_, int K_ = 8, int MINW_ = 1, bool FMA_ = false, bool TIGHT_ = false>
struct Tile {
    // FMA = false: SciPy's exact operation sequence (add, multiply, add -- three roundings per tap pair), DoG values
    //              bit-identical to the reference.  This is the default and what every parity claim refers to.
    // FMA = true : opt-in relaxed arithmetic, the multiply-add of each tap pair fused (two roundings).  DoG values then
    //              differ from the reference by ~1e-16 relative (north_star allows 1e-5); one third fewer FP64
    //              instructions in the blurs.  Never used unless the caller sets MST_FLAG_FMA.
    static constexpr bool FMA = FMA_;
    static constexpr int RGR = RGR_, RGC = RGC_;  // region rows / cols: interior + 1-pixel ring for the 3x3 max
    static constexpr int ITR = RGR_ - 2, ITC = RGC_ - 2;   // interior (owned) pixels
    static constexpr int RMAX = RMAX_;            // largest blur radius this instantiation supports
    static constexpr int K = K_;                  // samples per thread along the filter axis
    static constexpr int MINW = MINW_;            // waves per SIMD the register allocation must allow
    static constexpr int NCG = RGC / K;           // column groups
    static constexpr int NT = RGR * NCG;          // threads per workgroup
    static constexpr int NW = NT / 64;            // waves per workgroup
    static constexpr int CTR = RGR + 2 * RMAX;    // c tile rows / cols held in LDS (stored TRANSPOSED: ct[col][row])
    static constexpr int CTC = RGC + 2 * RMAX;
    static constexpr int CTP = TIGHT_ ? pitch_min(CTR) : pitch_for(CTR);
    static constexpr int VP = TIGHT_ ? pitch_min(RGC + 2 * RMAX) : pitch_for(RGC + 2 * RMAX);
    static constexpr int CT_ELEMS = CTC * CTP;
    static constexpr int VB_ELEMS = RGR * VP;
    static constexpr int DE_ELEMS = NCG * 2 * RGR;           // edge strip: [cg][left/right][row]
    static constexpr int ST_ELEMS = MST_MAX_TESTED * NW * 2;  // per (level, wave) partial {min, sum}
    static constexpr size_t LDS_BYTES = sizeof(double) * (size_t)(CT_ELEMS + VB_ELEMS + DE_ELEMS + ST_ELEMS);
    static_assert(NT % 64 == 0 && RGR % K == 0 && RGC % K == 0, "whole waves, whole groups");
    static_assert(64 % RGR == 0 || RGR % 64 == 0, "a wave holds whole runs of consecutive rows");
    static_assert(CTP % 2 == 0 && VP % 2 == 0 && CT_ELEMS % 2 == 0 && VB_ELEMS % 2 == 0, "16-byte alignment");
};

__device__ __forceinline__ int reflect_idx(int i, int n) {
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

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

// Outputs per register window: 8 for small radii; for the widest kernels the 8 outputs are produced as two windows
// of 4 so window + accumulators + the per-pixel sieve state stay inside the 256-VGPR budget.
template <int K, int R>
struct Chunk {
    static constexpr int KC = (K >= 8 && R < MST_KC8_BELOW) ? 8 : (K >= 4 ? 4 : K);
};

// One FIR chunk: KC outputs whose first tap sits at p[OFF]; p is 16-byte aligned, OFF is 0 or 1.
//   element e of the window = p[e + OFF], e in [0, KC + 2R);   output k:  centre e = R + k, taps e = R + k -+ j.
// SciPy's order per output:  t = x[c]*w0;  for j = R..1:  t += (x[c-j] + x[c+j]) * w[j].
// The window is streamed: 16-byte pairs are loaded (ds_read_b128) in the order the taps first touch them -- the centre
// run, then alternately from the left end inwards and from the right end inwards -- so only ~2*KC samples plus the
// loads in flight are live at a time instead of all KC + 2R, and the LDS latency hides under the FP64 work.
// For each tap the KC adds / muls / accumulates are adjacent in program order: KC independent chains keep the FP64
// pipe issuing (a sample-major order is one serial add->mul->add chain).
#if defined(MST_INJECT)
#define MST_FIR_INJ_TPARAM , int INJ = 0
#define MST_FIR_INJ_PARAM , Inject *inj = nullptr
#else
#define MST_FIR_INJ_TPARAM
#define MST_FIR_INJ_PARAM
#endif
template <int KC, int R, int OFF, bool FMA MST_FIR_INJ_TPARAM>
__device__ __forceinline__ void fir_chunk(const double *__restrict__ p, const double (&w)[R + 1], double (&t)[KC] MST_FIR_INJ_PARAM) {
    constexpr int NP = (KC + 2 * R + OFF + 1) / 2;                       // 16-byte pairs spanned by the window
    constexpr int QC0 = (R + OFF) >> 1, QC1 = (R + KC - 1 + OFF) >> 1;   // pairs holding the centre run
    const double2 *p2 = reinterpret_cast<const double2 *>(p);
    double x[2 * NP];
#if defined(MST_PROFILE) && defined(MST_ABL_NOLDS)   /* timing ablation (PROFILE builds): no LDS window loads */
#define MST_LD(q_)                                        \
    {                                                     \
        double a_ = w[0], b_ = w[R];                      \
        asm volatile("" : "+v"(a_), "+v"(b_));            \
        x[2 * (q_)] = a_;                                 \
        x[2 * (q_) + 1] = b_;                             \
    }
#else
#define MST_LD(q_)                     \
    {                                  \
        const double2 v_ = p2[q_];     \
        x[2 * (q_)] = v_.x;            \
        x[2 * (q_) + 1] = v_.y;        \
    }
#endif
#pragma unroll
    for (int q = QC0; q <= QC1; ++q) MST_LD(q)
#if defined(MST_INJECT)
    if constexpr (INJ == 3) {           // the whole block behind the chunk's first window loads, pinned there
        __builtin_amdgcn_sched_barrier(0);
        inject_work<>(*inj);
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
    int lq_hi = -1, rq_lo = NP;         // left pairs <= lq_hi and right pairs >= rq_lo are loaded (folds at compile time)
#pragma unroll
    for (int k = 0; k < KC; ++k) t[k] = x[R + k + OFF] * w[0];
#pragma unroll
    for (int j = R; j >= 1; --j) {
        const int lq1 = (R - j + KC - 1 + OFF) >> 1, rq0 = (R + j + OFF) >> 1;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (q > lq_hi && q <= lq1 && q < QC0) MST_LD(q)
            if (q < rq_lo && q >= rq0 && q > QC1) MST_LD(q)
        }
        lq_hi = lq1 > lq_hi ? lq1 : lq_hi;
        rq_lo = rq0 < rq_lo ? rq0 : rq_lo;
#if defined(MST_INJECT)
        if constexpr (INJ == 4) {       // an R-th of the block behind every tap's window loads, pinned there
            __builtin_amdgcn_sched_barrier(0);
            inject_work<(MST_INJ_DP + R - 1) / R, (MST_INJ_INT + R - 1) / R>(*inj);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
#if defined(MST_PROFILE) && defined(MST_ABL_NOMATH)  /* timing ablation (PROFILE builds): loads only */
        if (j & 1) t[j % KC] = t[j % KC] + (x[R - j + OFF] + x[R + KC - 1 + j + OFF]);
#else
        double s[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) s[k] = x[R + k - j + OFF] + x[R + k + j + OFF];
        if constexpr (FMA) {
#pragma unroll
            for (int k = 0; k < KC; ++k) t[k] = __builtin_fma(s[k], w[j], t[k]);
        } else {
#pragma unroll
            for (int k = 0; k < KC; ++k) s[k] = s[k] * w[j];
#pragma unroll
            for (int k = 0; k < KC; ++k) t[k] = t[k] + s[k];
        }
#endif
    }
#undef MST_LD
}

// Axis-0 pass for radius R over the (RGR rows) x (RGC + 2R columns) strip the axis-1 pass will need.  The c tile is
// stored transposed (ct[col][row]), so a thread's window of consecutive rows is contiguous in LDS.
// Work split: the first NT items are (8-row group, column) pairs over the first NT*8/RGR columns -- exactly one per
// thread; the remaining 2R-ish columns are cut into 4-row pieces so that the longest thread does 8 + 4 outputs
// instead of 8 + 8 (the strip is 1.1-1.5 x NT*8 outputs, so whole extra 8-row items would leave most lanes idle for
// a full second round).
template <class T, int R>
__device__ __forceinline__ void vpass(const double *__restrict__ ct, double *__restrict__ vb,
                                      const double (&wall)[T::RMAX + 1], int ptid, const double *__restrict__ vsrc,
                                      double *__restrict__ vdst, int variant MST_INJECT_ARG) {
    constexpr int K = T::K, KC = Chunk<T::K, R>::KC;
    constexpr int NC = T::RGC + 2 * R;               // columns to produce
    constexpr int NRG = T::RGR / K;                  // 8-row groups
    constexpr int MAINC = T::NT / NRG < NC ? T::NT / NRG : NC;   // columns covered by one full-length item per thread
    constexpr int OFF = (T::RMAX - R) & 1;           // parity of the first tap's row index (row0, h*KC are even)
    double w[R + 1];
#pragma unroll
    for (int j = 0; j <= R; ++j) w[j] = wall[j];
    static_assert(MAINC == T::NT / NRG, "the main-item mapping below is radius independent");
    if (!MST_VARIANT(16)) {                          // [ablation 16] no main V items
        // vsrc = ct + col * CTP + row0, vdst = vb + row0 * VP + col for this thread's (8-row group, column): computed once
        // per tile; the radius only adds a compile-time constant that folds into the ds_read offset field
        const double *p = vsrc + (T::RMAX - R) * T::CTP + (T::RMAX - R - OFF);
        double *q = vdst;
#pragma unroll
        for (int h = 0; h < K / KC; ++h) {
            double t[KC];
#if defined(MST_INJECT) && (MST_INJECT == 3 || MST_INJECT == 4)
            if (h == 0) fir_chunk<KC, R, OFF, T::FMA, MST_INJECT>(p + h * KC, w, t, &inj);
            else
#endif
            fir_chunk<KC, R, OFF, T::FMA>(p + h * KC, w, t);
#pragma unroll
            for (int k = 0; k < KC; ++k) q[(h * KC + k) * T::VP] = t[k];
        }
#if defined(MST_INJECT) && MST_INJECT == 1
        inject_work<>(inj);
#endif
    }
    if constexpr (MAINC < NC) if (!MST_VARIANT(8)) {  // [ablation 8] no leftover pieces
        constexpr int PR = 4;                        // rows per leftover piece (2-row pieces halve the skew between the waves
                                                     // but double the pieces' window loads: +2 % kernel time, measured)
        constexpr int XC = NC - MAINC;               // leftover columns
        constexpr int XRG = T::RGR / PR;             // pieces per column
        for (int it = ptid; it < XRG * XC; it += T::NT) {
            const int rgp = it / XC;
            const int col = MAINC + (it - rgp * XC);
            const int row0 = rgp * PR;
            const double *p = ct + ((T::RMAX - R) + col) * T::CTP + (row0 + T::RMAX - R - OFF);
            double t[PR];
            fir_chunk<PR, R, OFF, T::FMA>(p, w, t);
            double *q = vb + row0 * T::VP + col;
#pragma unroll
            for (int k = 0; k < PR; ++k) q[k * T::VP] = t[k];
        }
    }
}

// Axis-1 pass: thread (row rr, column group cg) -> g[0..K) = G at region columns cg*K .. cg*K+K-1.
template <class T, int R>
__device__ __forceinline__ void hpass(const double *__restrict__ p, const double (&wall)[T::RMAX + 1],
                                      double (&g)[T::K]) {
    constexpr int K = T::K, KC = Chunk<T::K, R>::KC;
    double w[R + 1];
#pragma unroll
    for (int j = 0; j <= R; ++j) w[j] = wall[j];
#pragma unroll
    for (int h = 0; h < K / KC; ++h) {
        double t[KC];
        fir_chunk<KC, R, 0, T::FMA>(p + h * KC, w, t);
#pragma unroll
        for (int k = 0; k < KC; ++k) g[h * KC + k] = t[k];
    }
}

}  // namespace
