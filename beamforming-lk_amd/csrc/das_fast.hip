// das_fast.hip -- the LDS-tiled, packed-FMA sweep kernels (AWPU_MATH_F32_FAST) for gfx950.
//
// They replace the loop nest of MIMOWorker::update, src/dsp/mimo.cpp:121-151, around delay(),
// src/dsp/delay.cpp:16-26, with the interpolation written as two FMAs per sample:
//     out[i] += f * X[off+i] + g * X[off+i+1],   g = 1 - f
// (the reference computes X[off+i+1] + f * (X[off+i] - X[off+i+1]); the two differ by fp32
// rounding only).  One (pixel, mic) pair = one "item" of 256 samples.
//
// Three kernels, one idea (docs/HISTORY.md 4.2):
//   das_pair_kernel      frames packed two by two, sample-interleaved; the two lanes of every packed
//                        FMA are the two FRAMES; the two pixels of a block are swept in mic-major order
//                        and share a mic's sample reads when their integer delays coincide (vertically
//                        adjacent pixels when the grid's row length is known).  Default for batches.
//                        (bottom of this file)
//   das_fast_db_kernel   one frame per item, 16-wave workgroup per CU, two LDS images filled by
//                        LDS-DMA while the other is swept.  Single-frame calls on full grids.
//   das_fast_kernel      one (or two) frames per item, 8-wave workgroups, two per CU.  Small grids.
// Common to all:
//   * the touched window of a chunk of mics is staged once per workgroup in LDS; the accumulators
//     of a wave's pixels stay in registers across chunks;
//   * an item's (f, g, LDS address) are wave-uniform and arrive by scalar loads (s_load_dwordx16 =
//     4 items) into SGPRs; besides the FMAs the only per-item VALU instruction is the address add;
//   * every inner FMA is a v_pk_fma_f32 -- the only way to reach the fp32 peak on this chip (a
//     plain v_fma_f32 issues at half the lane rate, tools/ubench.hip) -- fed by conflict-free
//     ds_read_b64 (64 lanes x 8 B = every bank once per half-wave);
//   * the inner loops of the two big kernels are hand-scheduled asm (das_fast_trip.inc, generated
//     by tools/gen_trip_asm.py);
//   * the g-terms land one sample low (g * X[t] belongs to out[t-1]); that skew is undone once per
//     pixel with lane shifts, and the 257th sample X[off+256] (owned by no lane) is gathered in a
//     side pass, one lane per mic;
//   * epilogue (mimo.cpp:131-137): MA filter via lane shifts, squares, wave reduction.
// Single-frame kernels only: lane l owns samples {2l,2l+1} and {128+2l,129+2l}, and the window is
// staged TWICE, copy q shifted by q floats, so that any integer delay starts 8-byte aligned in copy
// (off & 1) -- a misaligned ds_read_b64 is a 64-cycle replay:
//         A += f*x   -> out[2l], out[2l+1]          Q += g*x   -> out[2l-1], out[2l]
//         C += f*y   -> out[128+2l], out[129+2l]    R += g*y   -> out[127+2l], out[128+2l]
#include "das_kernels.h"

#include <algorithm>
#include <atomic>

namespace awpu {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte load, 4-byte aligned
typedef float f8 __attribute__((ext_vector_type(8)));

#define AWPU_AS4 __attribute__((address_space(4)))

// Four table entries = 16 dwords = one s_load_dwordx16 from the constant address space.
typedef int i16 __attribute__((ext_vector_type(16)));

struct EntryGroup {
    i16 v;
    __device__ __forceinline__ float f(int k) const { return __int_as_float(v[4 * k + 0]); }
    __device__ __forceinline__ unsigned addr(int k) const { return (unsigned) v[4 * k + 1]; }
    __device__ __forceinline__ float g(int k) const { return __int_as_float(v[4 * k + 2]); }
};

template <int FPI>
struct Reads {
    f2 x[FPI], y[FPI];
};

// Compiler-managed 8-byte LDS reads (so that hipcc tracks lgkmcnt itself), kept apart by an
// empty asm with a memory clobber: without it SILoadStoreOptimizer fuses the x/y pair into one
// ds_read2st64_b64, which the LDS serves at half the bytes per clock of two ds_read_b64.
__device__ __forceinline__ void no_fuse() { asm volatile("" ::: "memory"); }

template <int FPI>
__device__ __forceinline__ void issue_reads(Reads<FPI> &r, unsigned entry_addr, const char *lane_base) {
    constexpr int FS = kFastLdsBytes / FPI;
    const char *p = lane_base + entry_addr;
#pragma unroll
    for (int b = 0; b < FPI; b++) {
        r.x[b] = *(const f2 *) (p + b * FS);
        no_fuse();
        r.y[b] = *(const f2 *) (p + b * FS + 512);
        no_fuse();
    }
}

template <int FPI>
struct Acc {
    f2 A[FPI], Q[FPI], C[FPI], R[FPI];
};

template <int FPI>
__device__ __forceinline__ void accumulate(Acc<FPI> &a, const Reads<FPI> &r, float f, float g) {
    const f2 F = f2{f, f}, G = f2{g, g};
#pragma unroll
    for (int b = 0; b < FPI; b++) {
        a.A[b] = __builtin_elementwise_fma(F, r.x[b], a.A[b]);
        a.Q[b] = __builtin_elementwise_fma(G, r.x[b], a.Q[b]);
        a.C[b] = __builtin_elementwise_fma(F, r.y[b], a.C[b]);
        a.R[b] = __builtin_elementwise_fma(G, r.y[b], a.R[b]);
    }
}

// One group = four items: all reads first, then 16*FPI packed FMAs.
template <int FPI>
__device__ __forceinline__ void sweep_group(Acc<FPI> &acc, const EntryGroup &e, const char *lane_base) {
    Reads<FPI> r0, r1, r2, r3;
    issue_reads<FPI>(r0, e.addr(0), lane_base);
    issue_reads<FPI>(r1, e.addr(1), lane_base);
    issue_reads<FPI>(r2, e.addr(2), lane_base);
    issue_reads<FPI>(r3, e.addr(3), lane_base);
    accumulate<FPI>(acc, r0, e.f(0), e.g(0));
    accumulate<FPI>(acc, r1, e.f(1), e.g(1));
    accumulate<FPI>(acc, r2, e.f(2), e.g(2));
    accumulate<FPI>(acc, r3, e.f(3), e.g(3));
}

#include "das_fast_trip.inc"
static_assert(kFirStaticPlaneBytes == kFirStaticPlaneBytesHost, "das_kernels.h and tools/gen_trip_asm.py disagree on the FIR8 plane pitch");

// The items of one wave for one staged chunk: PPW pixels x ng groups of four, one frame per
// item, each pixel through one hand-scheduled asm block (das_fast_trip.inc).
// The items of one wave for one staged chunk: PPW pixels x ng groups of four, one frame per
// item, through the hand-scheduled asm blocks (das_fast_trip.inc), four pixels per block where
// PPW allows.  Pixels past the grid are swept too: the table has null rows for them.
template <int PPW, bool LOW = false>
__device__ __forceinline__ void sweep_chunk_trips(Acc<1> (&acc)[PPW], const FastEntry *lut, int pix0,
                                                  int usable_pad, int m0, int ng, unsigned lane_addr) {
    const int stride = usable_pad * (int) sizeof(FastEntry);
    if constexpr (PPW % 4 == 0 && LOW) {
#pragma unroll
        for (int q = 0; q < PPW; q += 4) {
            const void *row = lut + (size_t) (pix0 + q) * usable_pad + m0;
            sweep_quad_lo(acc[q].A[0], acc[q].Q[0], acc[q].C[0], acc[q].R[0],
                          acc[q + 1].A[0], acc[q + 1].Q[0], acc[q + 1].C[0], acc[q + 1].R[0],
                          acc[q + 2].A[0], acc[q + 2].Q[0], acc[q + 2].C[0], acc[q + 2].R[0],
                          acc[q + 3].A[0], acc[q + 3].Q[0], acc[q + 3].C[0], acc[q + 3].R[0], row, stride, ng, lane_addr);
        }
    } else if constexpr (PPW % 4 == 0) {
#pragma unroll
        for (int q = 0; q < PPW; q += 4) {
            const void *row = lut + (size_t) (pix0 + q) * usable_pad + m0;
            sweep_quad_hi(acc[q].A[0], acc[q].Q[0], acc[q].C[0], acc[q].R[0],
                          acc[q + 1].A[0], acc[q + 1].Q[0], acc[q + 1].C[0], acc[q + 1].R[0],
                          acc[q + 2].A[0], acc[q + 2].Q[0], acc[q + 2].C[0], acc[q + 2].R[0],
                          acc[q + 3].A[0], acc[q + 3].Q[0], acc[q + 3].C[0], acc[q + 3].R[0], row, stride, ng, lane_addr);
        }
    } else {
#pragma unroll
        for (int pp = 0; pp < PPW; pp++) {
            const void *row = lut + (size_t) (pix0 + pp) * usable_pad + m0;
            sweep_pixel_hi(acc[pp].A[0], acc[pp].Q[0], acc[pp].C[0], acc[pp].R[0], row, stride, ng, lane_addr);
        }
    }
}

template <int PPW, bool LOW = false>
__device__ __forceinline__ void sweep_chunk_stamped(Acc<1> (&acc)[PPW], const FastEntry *lut, int pix0,
                                                    int usable_pad, int m0, int ng, unsigned lane_addr,
                                                    unsigned &t_wait, unsigned &t_all) {
    static_assert(PPW % 4 == 0, "diagnostic build: quads only");
    const int stride = usable_pad * (int) sizeof(FastEntry);
#pragma unroll
    for (int q = 0; q < PPW; q += 4) {
        const void *row = lut + (size_t) (pix0 + q) * usable_pad + m0;
        unsigned dw = 0, da = 0;
        if constexpr (LOW) {
            sweep_quad_lo_stamped(acc[q].A[0], acc[q].Q[0], acc[q].C[0], acc[q].R[0],
                                  acc[q + 1].A[0], acc[q + 1].Q[0], acc[q + 1].C[0], acc[q + 1].R[0],
                                  acc[q + 2].A[0], acc[q + 2].Q[0], acc[q + 2].C[0], acc[q + 2].R[0],
                                  acc[q + 3].A[0], acc[q + 3].Q[0], acc[q + 3].C[0], acc[q + 3].R[0], row, stride, ng,
                                  lane_addr, dw, da);
        } else
        sweep_quad_stamped(acc[q].A[0], acc[q].Q[0], acc[q].C[0], acc[q].R[0],
                           acc[q + 1].A[0], acc[q + 1].Q[0], acc[q + 1].C[0], acc[q + 1].R[0],
                           acc[q + 2].A[0], acc[q + 2].Q[0], acc[q + 2].C[0], acc[q + 2].R[0],
                           acc[q + 3].A[0], acc[q + 3].Q[0], acc[q + 3].C[0], acc[q + 3].R[0], row, stride, ng,
                           lane_addr, dw, da);
        t_wait += dw;
        t_all += da;
    }
}

// Sum over the 64 lanes, returned wave-uniform.  All in the VALU's data-parallel-primitive path (gfx9 DPP modes):
// pairs, quads, half rows and rows by permutations within a row of 16, then row 0 into row 1 and row 2 into row 3
// (row_bcast:15), rows 0-1 into row 3 (row_bcast:31), and lane 63 holds the total -- no LDS crossbar, no waits,
// where the __shfl_xor butterfly compiles to six ds_bpermute_b32 round trips.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_take(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_take<0xB1>(v);        // quad_perm:[1,0,3,2]
    v += dpp_take<0x4E>(v);        // quad_perm:[2,3,0,1]
    v += dpp_take<0x141>(v);       // row_half_mirror
    v += dpp_take<0x140>(v);       // row_mirror: every lane holds its row's sum
    v += dpp_take<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3 (the other rows add 0)
    v += dpp_take<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// EIGHT per-lane partial sums reduced together (round 4: the powers of a quad's four pixels x two frames).  gfx950's
// v_permlane32_swap / v_permlane16_swap exchange half-waves / alternate rows between two registers, so one swap and one add
// fold two registers into one at each of the two upper levels; a row rotation by 8 and a select fold the third, three DPP adds
// finish the 8-lane groups: 18 VALU instructions where eight wave_sum()s take 48 and eight v_readlane_b32 (and what follows --
// division, address, store -- happens once, in vector form, instead of eight times on wave-uniform values).
// Returns, in every lane of the 8-lane group g = lane >> 3, the wave sum of value kWaveSum8Value(g):
//   row r = g >> 1, half h = g & 1:  value index = 4 h + {0, 2, 1, 3}[r]
// (tools/microbench/permlane_swap_check.hip prints what the two swaps do, lane by lane).
__device__ __forceinline__ float fold_halves(float a, float b) {  // lanes 0..31: a's two halves added, lanes 32..63: b's
    // (inline asm, not __builtin_amdgcn_permlane32_swap: hipcc 7.2 folds `r[0] + r[1]` of the builtin's two results into
    // `r[0] + r[0]` -- tools/microbench/permlane_swap_check.hip caught it; the nops cover the VALU-write -> swap-read hazard)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float fold_rows(float a, float b) {  // rows 0, 2: a's rows (0+1), (2+3); rows 1, 3: b's
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float wave_sum8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7, int lane) {
    const float x = fold_rows(fold_halves(v0, v1), fold_halves(v2, v3));  // rows: values 0, 2, 1, 3 (16 lanes each)
    const float y = fold_rows(fold_halves(v4, v5), fold_halves(v6, v7));  // rows: values 4, 6, 5, 7
    const float xs = x + dpp_take<0x128>(x), ys = y + dpp_take<0x128>(y);  // row_ror:8: lane l + lane l ^ 8 of its row
    float z = (lane & 8) ? ys : xs;                                        // 8-lane groups: x's value | y's value of the row
    z += dpp_take<0xB1>(z);   // quad_perm:[1,0,3,2]
    z += dpp_take<0x4E>(z);   // quad_perm:[2,3,0,1]
    z += dpp_take<0x141>(z);  // row_half_mirror: the other quad of the 8
    return z;
}
__device__ __forceinline__ constexpr int kWaveSum8Value(int group) { return 4 * (group & 1) + (((group >> 1) & 1) << 1) + (group >> 2); }
// FOUR partial sums together (a single frame's quad): two fold levels, then the sum over each row of 16 lanes; row r of the
// result holds, in all 16 lanes, the wave sum of value {0, 2, 1, 3}[r] = kWaveSum4Value(r).
__device__ __forceinline__ float wave_sum4(float v0, float v1, float v2, float v3) {
    float z = fold_rows(fold_halves(v0, v1), fold_halves(v2, v3));
    z += dpp_take<0xB1>(z);   // quad_perm:[1,0,3,2]
    z += dpp_take<0x4E>(z);   // quad_perm:[2,3,0,1]
    z += dpp_take<0x141>(z);  // row_half_mirror
    z += dpp_take<0x140>(z);  // row_mirror
    return z;
}
__device__ __forceinline__ constexpr int kWaveSum4Value(int row) { return ((row & 1) << 1) + (row >> 1); }

// The powers of a wave's four pixels x the two frames of its pair, from the lanes' partial sums s[pp] = (frame 2 pair, frame
// 2 pair + 1): the eight wave sums together, then the first lane of each 8-lane group divides and stores its (pixel, frame).
__device__ __forceinline__ void store_powers8(const f2 (&s)[4], const int (&pix)[4], const bool (&live)[4], int pair, int batch,
                                              int pixel_count, float norm, float *power, int lane) {
    const float total = wave_sum8(s[0].x, s[0].y, s[1].x, s[1].y, s[2].x, s[2].y, s[3].x, s[3].y, lane);
    const int value = kWaveSum8Value(lane >> 3), pp = value >> 1, frame = 2 * pair + (value & 1);
    const int p = pp == 0 ? pix[0] : pp == 1 ? pix[1] : pp == 2 ? pix[2] : pix[3];
    const bool on = pp == 0 ? live[0] : pp == 1 ? live[1] : pp == 2 ? live[2] : live[3];
    if ((lane & 7) == 0 && on && frame < batch) power[(size_t) frame * pixel_count + p] = total / norm;
}

// Sum over each group of 8 consecutive lanes, left in all 8 (the tail passes give 8 lanes to a pixel), and the value
// of one lane as a wave-uniform scalar: DPP and v_readlane instead of ds_bpermute_b32.
__device__ __forceinline__ float sum8(float v) {
    v += dpp_take<0xB1>(v);   // quad_perm:[1,0,3,2]
    v += dpp_take<0x4E>(v);   // quad_perm:[2,3,0,1]
    v += dpp_take<0x141>(v);  // row_half_mirror: the other quad of the 8
    return v;
}
__device__ __forceinline__ float sum16(float v) {  // the same over each row of 16 lanes
    v = sum8(v);
    return v + dpp_take<0x140>(v);  // row_mirror: the other half of the row
}
__device__ __forceinline__ float sum32(float v) {  // over each half of the wave; rows 1 and 3 then hold it
    v = sum16(v);
    return v + dpp_take<0x142, 0xa>(v);  // row_bcast:15: row 0's (2's) sum into every lane of row 1 (3)
}
template <int LPP>
__device__ __forceinline__ float sum_lanes(float v) {
    static_assert(LPP == 8 || LPP == 16 || LPP == 32, "groups of 8, 16 or 32 lanes");
    return LPP == 8 ? sum8(v) : LPP == 16 ? sum16(v) : sum32(v);
}
template <int LPP>
__device__ __forceinline__ constexpr int sum_lane_of(int group) {  // a lane that holds group `group`'s sum
    return LPP == 32 ? 32 * group + 16 : LPP * group;
}
__device__ __forceinline__ float lane_value(float v, int lane_index) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane_index));
}

// Whole-wave rotation by one lane in the VALU's data-parallel-primitive path (gfx9's wave_rol / wave_ror): no LDS
// crossbar traffic and no waits, where __shfl_up/_down compile to ds_bpermute_b32.
constexpr int kDppWaveRol1 = 0x134;  // lane l takes lane l+1's value, lane 63 lane 0's
constexpr int kDppWaveRor1 = 0x13C;  // lane l takes lane l-1's value, lane 0 lane 63's
template <int CTRL>
__device__ __forceinline__ float wave_rotate1(float v) {
    // (every lane of a whole-wave rotation receives a value, so `old` is never seen: passing the source itself
    // spares the v_mov_b32 0 that a constant `old` costs before every rotation)
    const int bits = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(bits, bits, CTRL, 0xf, 0xf, false));
}

// out[] of one pixel and frame from the skewed accumulators, then mimo.cpp:131-137.
// `tail` = sum over mics of g * X[off+256], the contribution to out[255].
__device__ __forceinline__ float finish_pixel(f2 A, f2 Q, f2 C, f2 R, float tail, int lane) {
    // neighbour exchanges: whole-wave rotations by one lane (see wave_rotate); a rotation's wrap-around element is
    // the one the other half of the samples needs at the seam
    const float q_rot = wave_rotate1<kDppWaveRol1>(Q.x);  // lane l+1's contribution to out[2l+1]
    const float r_rot = wave_rotate1<kDppWaveRol1>(R.x);  // lane l+1's contribution to out[129+2l]; lane 63: lane 0's, to out[127]
    const float q_next = lane == 63 ? r_rot : q_rot;
    const float r_next = lane == 63 ? tail : r_rot;
    const float o0 = A.x + Q.y;     // out[2l]
    const float o1 = A.y + q_next;  // out[2l+1]
    const float o2 = C.x + R.y;     // out[128+2l]
    const float o3 = C.y + r_next;  // out[129+2l]
    const float o1_prev = wave_rotate1<kDppWaveRor1>(o1);  // out[2l-1]; lane 0: out[127]
    const float o2_next = wave_rotate1<kDppWaveRol1>(o2);  // out[130+2l]; lane 63: out[128]
    const float o0_rot = wave_rotate1<kDppWaveRol1>(o0);   // out[2l+2]
    const float o3_rot = wave_rotate1<kDppWaveRor1>(o3);   // out[127+2l]
    const float o0_next = lane == 63 ? o2_next : o0_rot;
    const float o3_prev = lane == 0 ? o1_prev : o3_rot;
    const float ma0 = o0 * 0.5f - 0.25f * (o1 + o1_prev);  // i = 2l      (valid for l >= 1)
    const float ma1 = o1 * 0.5f - 0.25f * (o0_next + o0);  // i = 2l+1
    const float ma2 = o2 * 0.5f - 0.25f * (o3 + o3_prev);  // i = 128+2l
    const float ma3 = o3 * 0.5f - 0.25f * (o2_next + o2);  // i = 129+2l  (valid for l <= 62)
    float sum = ma1 * ma1 + ma2 * ma2;
    if (lane != 0) sum += ma0 * ma0;
    if (lane != 63) sum += ma3 * ma3;
    return wave_sum(sum);
}

// NW waves per workgroup, PPW pixels per wave, FPI frames per item, WPS waves per SIMD the
// register budget is sized for (two workgroups share a CU and its 160 KiB of LDS).
template <int NW, int PPW, int FPI, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void das_fast_kernel(FastArgs a) {
    static_assert(PPW <= 8, "the tail pass gives 8 lanes to each of at most 8 pixels");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int FS = kFastLdsBytes / FPI;  // byte stride between the frames of a group
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char *lane_base = (const char *) lds + lane * 8;
    // the same address as a 32-bit LDS offset, for the asm trips
    const unsigned lane_addr = (unsigned) (unsigned long long) (const __attribute__((address_space(3))) char *) lane_base;
    const int frame0 = blockIdx.x * FPI;  // frames vary fastest in dispatch order: the
    const int pix0 = (blockIdx.y * NW + wave) * PPW;  // workgroups of a tile share its table rows in L2
    const int wr = a.wr;
    const AWPU_AS4 FastEntry *lut = (const AWPU_AS4 FastEntry *) (unsigned long long) a.lut;
    const AWPU_AS4 int32_t *index = (const AWPU_AS4 int32_t *) (unsigned long long) a.index;
    const AWPU_AS4 int32_t *row_off = (const AWPU_AS4 int32_t *) (unsigned long long) a.row_off;

    Acc<FPI> acc[PPW];
    float tail[FPI];  // lane (8*pp + k): partial tail sum of pixel pp over the mics s = k (mod 8)
#pragma unroll
    for (int b = 0; b < FPI; b++) tail[b] = 0.0f;
#pragma unroll
    for (int pp = 0; pp < PPW; pp++)
#pragma unroll
        for (int b = 0; b < FPI; b++) acc[pp].A[b] = acc[pp].Q[b] = acc[pp].C[b] = acc[pp].R[b] = f2{0.0f, 0.0f};

    // tail pass (the 257th sample of every window): 8 lanes per pixel whatever the shape -- a lane's partial sum then
    // runs over the same mics in the same order whatever the chunk size or pixels per wave, so that results do not
    // depend on the kernel shape (a device group's slabs may run other shapes than the whole grid would); a round
    // covers 32 mics in NU entries per lane; the first round's entries are requested before the chunk is staged
    constexpr int LPP = 8, NU = 32 / LPP;
    const int tail_pp = lane / LPP;
    const bool tail_lane = tail_pp < PPW && pix0 + tail_pp < a.pixel_count;
    const FastEntry *tail_row = a.lut + (size_t) (pix0 + (tail_lane ? tail_pp : 0)) * a.usable_pad;
    struct AddrG {
        unsigned addr;
        float g;
    };

    for (int m0 = 0; m0 < a.usable; m0 += a.chunk) {
        const int mc = min(a.chunk, a.usable - m0);
        const int mc4 = (mc + 3) & ~3;  // the table pads every pixel row to a multiple of 4
        AddrG te[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int j = LPP * u + ((lane - m0) & (LPP - 1));
            te[u] = *(const AddrG *) ((const char *) (tail_row + m0 + min(j, mc4 - 1)) + 4);  // fields addr, g
            if (!tail_lane || j >= mc4) te[u].g = 0.0f;
        }
        __syncthreads();                // the previous chunk is fully consumed

        // ---- stage [frame][mic][copy][wr] floats: copy q = the window shifted by q samples.
        const int rows = AWPU_DBG(a, 1) && m0 > 0 ? 0 : FPI * mc * 2;  // debug bit 0: stage once
        const bool inside = a.wstart + 1 + wr <= a.hist;  // every 16-byte piece of every row is readable
        if (inside && !AWPU_DBG(a, 64)) {
            // LDS-DMA, one wave per row: lane l's 16 bytes land at (row base + 1 KiB * k) + 16 l
            for (int r = wave; r < rows; r += NW) {
                const int b = r / (2 * mc);
                const int jr = r - b * 2 * mc;
                const int fb = min(frame0 + b, a.batch - 1);
                const float *src = a.frames + (size_t) fb * a.n_streams * a.hist + row_off[2 * m0 + jr];
                float *dst = lds + b * (FS / 4) + jr * wr;
                for (int t0 = 0; t0 < wr; t0 += 256) {
                    if (t0 + lane * 4 < wr) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (src + t0 + lane * 4),
                                                         (__attribute__((address_space(3))) void *) (dst + t0), 16, 0, 0);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else
        for (int r = wave; r < rows; r += 2 * NW) {
            const float *src[2];
            float *dst[2];
            int valid[2];  // floats of each row that lie inside the history (the shifted copy has one less)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int rr = min(r + u * NW, rows - 1);  // a duplicate row rewrites the same bytes
                const int b = rr / (2 * mc);
                const int jr = rr - b * 2 * mc;
                const int j = jr >> 1, q = jr & 1;
                const int fb = min(frame0 + b, a.batch - 1);
                const int first = a.wstart + q;
                src[u] = a.frames + ((size_t) fb * a.n_streams + index[m0 + j]) * a.hist + first;
                dst[u] = lds + b * (FS / 4) + (j * 2 + q) * wr;
                valid[u] = min(wr, a.hist - first);
            }
            if (valid[0] == wr && valid[1] == wr) {  // both rows lie inside the history: no per-element guards
                for (int t0 = 0; t0 < wr; t0 += 512) {
                    f4 v[2][2];
#pragma unroll
                    for (int u = 0; u < 2; u++)
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            const int t = t0 + k * 256 + lane * 4;
                            if (t < wr) v[u][k] = *(const f4u *) (src[u] + t);
                        }
#pragma unroll
                    for (int u = 0; u < 2; u++)
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            const int t = t0 + k * 256 + lane * 4;
                            if (t < wr) *(f4 *) (dst[u] + t) = v[u][k];
                        }
                }
            } else {
#pragma unroll
                for (int u = 0; u < 2; u++)
                    for (int t = lane; t < wr; t += 64) dst[u][t] = t < valid[u] ? src[u][t] : 0.0f;
            }
        }
        __syncthreads();

        if (!AWPU_DBG(a, 2)) {  // debug bit 1: no sweep (staging only)
            if constexpr (FPI == 1) {
                sweep_chunk_trips<PPW>(acc, a.lut, pix0, a.usable_pad, m0, mc4 >> 2, lane_addr);
            } else {
#pragma unroll
                for (int pp = 0; pp < PPW; pp++) {
                    const int p = pix0 + pp;
                    if (p < a.pixel_count) {
                        const AWPU_AS4 i16 *grp = (const AWPU_AS4 i16 *) (lut + (size_t) p * a.usable_pad + m0);
                        const int ng = mc4 >> 2;
                        EntryGroup e0, e1;
                        e0.v = grp[0];
                        for (int g = 0; g < ng; g += 2) {
                            e1.v = grp[g + 1];
                            sweep_group<FPI>(acc[pp], e0, lane_base);
                            e0.v = grp[g + 2];
                            if (g + 1 < ng) sweep_group<FPI>(acc[pp], e1, lane_base);
                        }
                    }
                }
            }
            // ---- the 257th sample of every window (X[off+256], weight g, goes to out[255]): lane
            // LPP*pp + k gathers it for pixel pp and the mics s = k (mod LPP) of this chunk.
#pragma unroll
            for (int u = 0; u < NU; u++)  // mics 0..31 of the chunk (requested before the staging)
#pragma unroll
                for (int b = 0; b < FPI; b++) {
                    const float x = lds[(te[u].addr + 1024u + (unsigned) (b * FS)) >> 2];
                    tail[b] = __builtin_fmaf(te[u].g, x, tail[b]);
                }
            for (int j0 = 32; j0 < mc4; j0 += 32) {  // chunks of more than 32 mics
                AddrG e[NU];
#pragma unroll
                for (int u = 0; u < NU; u++) {
                    const int j = j0 + LPP * u + ((lane - m0) & (LPP - 1));
                    e[u] = *(const AddrG *) ((const char *) (tail_row + m0 + min(j, mc4 - 1)) + 4);
                    if (!tail_lane || j >= mc4) e[u].g = 0.0f;
                }
#pragma unroll
                for (int u = 0; u < NU; u++)
#pragma unroll
                    for (int b = 0; b < FPI; b++) {
                        const float x = lds[(e[u].addr + 1024u + (unsigned) (b * FS)) >> 2];
                        tail[b] = __builtin_fmaf(e[u].g, x, tail[b]);
                    }
            }
        }
    }

    // combine the LPP partial tail sums of each pixel
#pragma unroll
    for (int b = 0; b < FPI; b++) {
        tail[b] = sum_lanes<LPP>(tail[b]);
    }
#pragma unroll
    for (int pp = 0; pp < PPW; pp++) {
        const int p = pix0 + pp;
        if (p < a.pixel_count) {
#pragma unroll
            for (int b = 0; b < FPI; b++) {
                const float t = lane_value(tail[b], sum_lane_of<LPP>(pp));
                const float sum = finish_pixel(acc[pp].A[b], acc[pp].Q[b], acc[pp].C[b], acc[pp].R[b], t, lane);
                if (lane == 0 && frame0 + b < a.batch) {
                    a.power[(size_t) (frame0 + b) * a.pixel_count + p] = sum / (float) (kSamples * a.usable);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Double-buffered shape for full grids: ONE 16-wave workgroup per CU, two LDS buffers.
// While chunk c is swept out of buffer c&1, the global loads of chunk c+1 are already in flight
// into registers; they are written to the other buffer after the sweep, then one barrier.
// Staging is by 16-byte pieces of the flat [mic][copy][wr] image: piece -> (row, column) is the
// same for every chunk, so each thread decodes its pieces once per launch.
// ---------------------------------------------------------------------------------------
// NW waves per workgroup, BUF bytes per LDS image (two images + the side table per workgroup),
// WPS waves per SIMD the register budget is sized for: <16, 8, 78 KiB, 4> = one workgroup per CU,
// <12, 4, 38 KiB, 6> = two per CU.
template <int NW, int PPW, int BUF, int WPS, bool DIAG>
__global__ __launch_bounds__(NW * 64, WPS) void das_fast_db_kernel(FastArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int kDbThreads = NW * 64;
    constexpr int kDbPieces = (BUF + kDbThreads * 16 - 1) / (kDbThreads * 16);  // 16-byte pieces per thread and chunk
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned) (unsigned long long) (const __attribute__((address_space(3))) char *) lds;
    // a workgroup is persistent over `frames_per_wg` consecutive frames of the batch: the chunk
    // pipeline runs on across frame boundaries (no refill bubble, one launch/drain per group)
    const int frame0 = blockIdx.x * a.frames_per_wg;
    const int n_frames = min(a.frames_per_wg, a.batch - frame0);
    const int pix0 = (blockIdx.y * NW + wave) * PPW;
    const int wr = a.wr;
    const int wr4 = wr >> 2;  // 16-byte pieces per row

    // float offset of every staged row inside a frame, kept in LDS behind the two buffers so that
    // the per-chunk address of a piece needs no dependent global load
    int *row_off_lds = (int *) (lds + 2 * (BUF / 4));
    for (int i = threadIdx.x; i < 2 * a.usable_pad; i += kDbThreads) row_off_lds[i] = a.row_off[i];

    // this thread's pieces: row (= 2*mic_slot + copy) and float column inside the row
    int piece_rc[kDbPieces];  // row << 16 | column
#pragma unroll
    for (int k = 0; k < kDbPieces; k++) {
        const int piece = threadIdx.x + k * kDbThreads;
        const int row = piece / wr4;
        piece_rc[k] = (row << 16) | ((piece - row * wr4) * 4);
    }
    __syncthreads();

    Acc<1> acc[PPW];
    float tail = 0.0f;
#pragma unroll
    for (int pp = 0; pp < PPW; pp++) acc[pp].A[0] = acc[pp].Q[0] = acc[pp].C[0] = acc[pp].R[0] = f2{0.0f, 0.0f};
    const int tail_pp = lane >> 3;
    const bool tail_lane = tail_pp < PPW && pix0 + tail_pp < a.pixel_count;
    const FastEntry *tail_row = a.lut + (size_t) (pix0 + (tail_lane ? tail_pp : 0)) * a.usable_pad;

    // LDS-DMA staging: lane l of a wave writes 16 bytes at (wave-uniform base) + 16*l, which is
    // exactly piece (threadIdx.x + k*1024) of the flat image; the source address is per lane.
    auto dma_chunk = [&](int frame, int m0, int mc, int buf) {
        const float *frame_base = a.frames + (size_t) frame * a.n_streams * a.hist;
        const int rows = 2 * mc;
#pragma unroll
        for (int k = 0; k < kDbPieces; k++) {
            const int row = piece_rc[k] >> 16;
            if (row < rows) {
                const int off = row_off_lds[2 * m0 + row];
                const float *src = frame_base + off + (piece_rc[k] & 0xffff);
                float *dst = lds + buf * (BUF / 4) + (wave * 64 + k * kDbThreads) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src,
                                                 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
            }
        }
    };

    unsigned t_wait = 0, t_all = 0;
    const long long t_begin = __builtin_readcyclecounter();
    const int n_chunks = (a.usable + a.chunk - 1) / a.chunk;
    const int n_steps = n_frames * n_chunks;  // one step = one staged chunk of one frame
    dma_chunk(frame0, 0, min(a.chunk, a.usable), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    unsigned t_ph[5] = {0, 0, 0, 0, 0};  // diagnostics: dma issue, sweep, tail, dma wait, barrier
    constexpr bool diag = DIAG;
    auto stamp = [&](int k, long long &t) {
        if (diag) {
            const long long n = __builtin_readcyclecounter();
            t_ph[k] += (unsigned) (n - t);
            t = n;
        }
    };
    int fi = 0, c = 0;  // frame within the group, chunk within the frame
    for (int step = 0; step < n_steps; step++) {
        const int m0 = c * a.chunk;
        const int mc = min(a.chunk, a.usable - m0);
        const int mc4 = (mc + 3) & ~3;
        const int buf = step & 1;
        const bool last_chunk = c + 1 == n_chunks;
        const int c1 = last_chunk ? 0 : c + 1;
        const int fi1 = last_chunk ? fi + 1 : fi;
        long long t = diag ? __builtin_readcyclecounter() : 0;
        if (step + 1 < n_steps && !AWPU_DBG(a, 1)) {  // lands in the other buffer during the sweep below
            dma_chunk(frame0 + fi1, c1 * a.chunk, min(a.chunk, a.usable - c1 * a.chunk), buf ^ 1);
        }
        stamp(0, t);

        const unsigned lane_addr = lds_base + buf * BUF + lane * 8;
        if constexpr (DIAG) {
            sweep_chunk_stamped<PPW, (WPS > 4)>(acc, a.lut, pix0, a.usable_pad, m0, mc4 >> 2, lane_addr, t_wait, t_all);
        } else if (!AWPU_DBG(a, 2)) {
            sweep_chunk_trips<PPW, (WPS > 4)>(acc, a.lut, pix0, a.usable_pad, m0, mc4 >> 2, lane_addr);
        }
        stamp(1, t);
        // the 257th sample of every window: see das_fast_kernel
        const float *buf_f = lds + buf * (BUF / 4);
        for (int j0 = 0; j0 < (AWPU_DBG(a, 4) ? 0 : mc4); j0 += 32) {
            FastEntry e[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + 8 * u + ((lane - m0) & 7);
                e[u] = tail_row[m0 + min(j, mc4 - 1)];
                if (!tail_lane || j >= mc4) e[u].g = 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) tail = __builtin_fmaf(e[u].g, buf_f[(e[u].addr + 1024u) >> 2], tail);
        }
        if (diag) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(2, t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next chunk are in LDS
        stamp(3, t);
        if (!AWPU_DBG(a, 8)) __syncthreads();
        stamp(4, t);

        if (last_chunk) {  // this frame's sums are complete: epilogue (mimo.cpp:131-137), then start over
            tail = sum8(tail);
#pragma unroll
            for (int pp = 0; pp < PPW; pp++) {
                const int p = pix0 + pp;
                const float tl = lane_value(tail, pp * 8);
                const float sum = finish_pixel(acc[pp].A[0], acc[pp].Q[0], acc[pp].C[0], acc[pp].R[0], tl, lane);
                if (lane == 0 && p < a.pixel_count) {
                    a.power[(size_t) (frame0 + fi) * a.pixel_count + p] = sum / (float) (kSamples * a.usable);
                }
                acc[pp].A[0] = acc[pp].Q[0] = acc[pp].C[0] = acc[pp].R[0] = f2{0.0f, 0.0f};
            }
            tail = 0.0f;
        }
        c = c1;
        fi = fi1;
    }

    if (DIAG && a.debug_out && lane == 0) {  // diagnostics: per-wave cycle sums
        unsigned long long *o = a.debug_out + 12 * ((size_t) (blockIdx.y * gridDim.x + blockIdx.x) * NW + wave);
        o[0] = t_wait;
        o[1] = t_all;
        o[2] = (unsigned long long) (__builtin_readcyclecounter() - t_begin);
        o[3] = (unsigned long long) n_steps * PPW;
        for (int k = 0; k < 5; k++) o[4 + k] = t_ph[k];
    }
}

// ---------------------------------------------------------------------------------------
// Frame-pair shape (the default for batches): two frames are packed sample-interleaved,
// (x_a[t], x_b[t]) = one 8-byte element, by pack_pairs_kernel; the sweep then puts the two FRAMES in
// the two lanes of every v_pk_fma_f32 instead of two neighbouring samples:
//   * any integer delay is 8-byte aligned -- no parity copies; the LDS image of a chunk,
//     [mic][W] elements, has the footprint the two copies had and holds two frames;
//   * lane l owns samples l, l+64, l+128, l+192; an item = 1 address add + 4 ds_read_b64 +
//     8 v_pk_fma_f32 (A_k += f x_k, Q_k += g x_k), so 8 of 9 VALU instructions are FMAs (4 of 5
//     before) and a table entry serves two frames;
//   * the packed rows of a chunk are contiguous in HBM: staging is one linear LDS-DMA stream.
// One 16-wave workgroup per CU, two LDS images (chunk c+1 lands while chunk c is swept), 4 pixels
// per wave; grid = (frame pairs, 64-pixel tiles).
// ---------------------------------------------------------------------------------------
// FILTER: the rows carry Y[t] = X[t]/2 - (X[t+1] + X[t-1])/4 instead of X[t] -- the reference's moving-average
// stencil (mimo.cpp:131-134) applied to the SAMPLES.  The stencil and the delay-and-sum are both linear, so
//     MA[i] = out[i]/2 - (out[i+1] + out[i-1])/4 = sum_m f Y_m[off+i] + (1-f) Y_m[off+i+1]     (i = 1..254)
// comes straight out of the sweep: the frame-pair kernels need no stencil in their epilogue and no 257th sample
// (MA[254] needs Y[off+255], whose X[off+256] the pack pass reads like any other sample).  Y[off] and Y[off+256]
// -- the two that would need samples outside [off, off+256] -- are never used (MA[0] and MA[255] do not exist).
// In fp32 the filtered samples are a third of the size of the raw ones at the carrier, and so are all rounding
// errors downstream: against exact sums this order is closer than the reference's own (docs/HISTORY.md 4.2g).
template <bool FILTER>
__device__ __forceinline__ void pack_one_row(const float *frames, int n_streams, int hist, int wstart, const int32_t *index,
                                             int usable, const float *gain, int wp, int batch, float *packed, int pair, int s,
                                             int rows_out) {
    f2 *dst = (f2 *) packed + ((size_t) pair * rows_out + s) * wp;
    if (s >= usable) {  // padding rows (the quad shape sweeps whole groups of four mics): silence
        for (int t = threadIdx.x; t < wp; t += blockDim.x) dst[t] = f2{0.0f, 0.0f};
        return;
    }
    const int fa = min(2 * pair, batch - 1), fb = min(2 * pair + 1, batch - 1);
    const float *xa = frames + ((size_t) fa * n_streams + index[s]) * hist + wstart;
    const float *xb = frames + ((size_t) fb * n_streams + index[s]) * hist + wstart;
    const float gm = gain ? gain[s] : 1.0f;  // optional per-mic gain (awpu_hip_set_mic_gains); x * 1.0f is x
    const int valid = min(wp, hist - wstart);
    for (int t = threadIdx.x; t < wp; t += blockDim.x) {
        f2 v = f2{0.0f, 0.0f};
        if (t < valid) {
            v = f2{xa[t] * gm, xb[t] * gm};
            if constexpr (FILTER) {
                // neighbours outside the row count as 0 (the values they produce are never used, see above)
                const bool lo = wstart + t > 0, hi = wstart + t + 1 < hist;
                const f2 nb = f2{(lo ? xa[t - 1] * gm : 0.0f) + (hi ? xa[t + 1] * gm : 0.0f),
                                 (lo ? xb[t - 1] * gm : 0.0f) + (hi ? xb[t + 1] * gm : 0.0f)};
                v = __builtin_elementwise_fma(f2{-0.25f, -0.25f}, nb, 0.5f * v);
            }
        }
        dst[t] = v;
    }
}

template <bool FILTER>
__global__ void pack_pairs_kernel(const float *frames, int n_streams, int hist, int wstart, const int32_t *index,
                                  int usable, const float *gain, int wp, int batch, float *packed) {
    pack_one_row<FILTER>(frames, n_streams, hist, wstart, index, usable, gain, wp, batch, packed, blockIdx.y, blockIdx.x, gridDim.x);
}

// out[] of one pixel (both frames at once) from the skewed accumulators, then mimo.cpp:131-137.
// P[0..3] = A_k: f-terms of samples l+64k; P[4..7] = Q_k: g-terms, belonging to samples l+64k-1.
// The neighbour exchanges are whole-wave rotations by one lane (wave_rotate1): lane 63's (lane 0's) wrap-around value
// is exactly the element the neighbouring register needs at its seam, so one rotation per register serves both.
template <int CTRL>
__device__ __forceinline__ f2 wave_rotate(f2 v) {
    return f2{wave_rotate1<CTRL>(v.x), wave_rotate1<CTRL>(v.y)};
}

__device__ __forceinline__ f2 finish_pixel_pair(const f2 (&P)[8], f2 tail, int lane) {
    f2 o[4];
    f2 rq = wave_rotate<kDppWaveRol1>(P[4]);  // Q_k one lane down; lane 63 holds Q_k[0]
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // lane 63 of register k takes sample 64(k+1)'s term = Q_{k+1}[0]; sample 255 takes X[off+256]'s term
        const f2 rq_next = k < 3 ? wave_rotate<kDppWaveRol1>(P[k < 3 ? 5 + k : 7]) : tail;
        o[k] = P[k] + (lane == 63 ? rq_next : rq);  // out[l + 64k]
        rq = rq_next;
    }
    f2 sum = f2{0.0f, 0.0f};
    f2 dn = wave_rotate<kDppWaveRol1>(o[0]);  // out[l+1 + 64k]; lane 63: out[64k]
    f2 up_before = wave_rotate<kDppWaveRor1>(o[0]);  // (k = 0, lane 0: sample 0 is not summed)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const f2 dn_after = k < 3 ? wave_rotate<kDppWaveRol1>(o[k < 3 ? k + 1 : 3]) : dn;  // (k = 3, lane 63: sample 255 is not summed)
        const f2 up = wave_rotate<kDppWaveRor1>(o[k]);  // out[l-1 + 64k]; lane 0: out[63 + 64k]
        const f2 next = lane == 63 ? dn_after : dn;
        const f2 prev = lane == 0 ? up_before : up;
        const int i = lane + 64 * k;
        const f2 ma = o[k] * 0.5f - 0.25f * (next + prev);
        if (i >= 1 && i <= kSamples - 2) sum += ma * ma;
        dn = dn_after;
        up_before = up;
    }
    sum.x = wave_sum(sum.x);
    sum.y = wave_sum(sum.y);
    return sum;
}

// The same for sweeps of PRE-FILTERED samples (pack_one_row<true>): the un-skewed sums ARE the moving average,
// MA[i] = A[i] + Q[i+1], i = 1..254 -- no stencil, no 257th sample.  P as above.
__device__ __forceinline__ f2 pixel_pair_partial_filtered(const f2 (&P)[8], int lane) {  // a lane's share of sum MA^2, both frames
    f2 sum = f2{0.0f, 0.0f};
    f2 rq = wave_rotate<kDppWaveRol1>(P[4]);  // Q_k one lane down; lane 63 holds Q_k[0]
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // lane 63 of register k takes sample 64(k+1)'s term = Q_{k+1}[0] (k = 3: MA[255] is not summed)
        const f2 rq_next = k < 3 ? wave_rotate<kDppWaveRol1>(P[k < 3 ? 5 + k : 7]) : rq;
        const f2 ma = P[k] + (lane == 63 ? rq_next : rq);  // MA[l + 64k]
        const int i = lane + 64 * k;
        if (i >= 1 && i <= kSamples - 2) sum = __builtin_elementwise_fma(ma, ma, sum);
        rq = rq_next;
    }
    return sum;
}
__device__ __forceinline__ f2 finish_pixel_pair_filtered(const f2 (&P)[8], int lane) {
    f2 sum = pixel_pair_partial_filtered(P, lane);
    sum.x = wave_sum(sum.x);
    sum.y = wave_sum(sum.y);
    return sum;
}

// tell the compiler a pointer / int is wave-uniform (it is: built from block and wave ids)
__device__ __forceinline__ const void *uniform_ptr(const void *p) {
    const unsigned long long v = (unsigned long long) p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned) (v >> 32));
    return (const void *) (((unsigned long long) hi << 32) | lo);
}

// SHARE: the two pixels of a block are swept in mic-major order and share a mic's sample reads whenever
// their table entries carry the same LDS address (sweep_duo_shared); same sums bit for bit.
template <int PPW, bool DIAG, bool SHARE>
__global__ __launch_bounds__(1024, 4) void das_pair_kernel(PairArgs a) {
    static_assert(PPW % 2 == 0 && PPW <= 8, "pixels go through the asm blocks two at a time");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = 16, kThreads = NW * 64, BUF = kFastLdsBytes;
    constexpr int kPieces = (BUF + kThreads * 16 - 1) / (kThreads * 16);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned) (unsigned long long) (const __attribute__((address_space(3))) char *) lds;
    // dispatch order (round 4: das_quad_kernel's): the items (frame pair, tile), ordered (pair group, tile, pair), are cut into 8
    // contiguous runs, one per XCD (blockIdx & 7: round-robin placement, assumed for speed only): an XCD's workgroups sweep
    // `pair_group` frame pairs x a few tiles at a time, the pairs' samples stay in its L2 while it walks the table
    const int total = a.n_pairs * a.tiles;
    const int per_xcd = (total + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= min(total, ((int) (blockIdx.x & 7) + 1) * per_xcd)) return;  // (uniform for the workgroup)
    int pair, tile;
    {
        const int full_items = (a.n_pairs / a.pair_group) * a.pair_group * a.tiles;
        const int ga = item < full_items ? a.pair_group : a.n_pairs % a.pair_group;
        const int rem = item < full_items ? item : item - full_items;
        const int grp = rem / (a.tiles * ga), in = rem - grp * a.tiles * ga;
        tile = in / ga;
        pair = (item < full_items ? grp * a.pair_group : a.n_pairs - ga) + (in - tile * ga);
    }
    // Which pixels this wave sweeps.  Neighbouring pixels in a grid column usually differ less in delay than
    // neighbours in a row (arrays are wider than tall), and the shared-read block profits from pixels whose
    // integer delays coincide, so when the grid's row length is known (a.cols > 0) a workgroup takes 32
    // columns x 2 rows and a wave two vertical pairs; otherwise 4 consecutive pixels (two horizontal pairs).
    int pix[PPW];   // pixel of slot q (slots 2j, 2j+1 form a pair: the two pixels of one asm block)
    bool live[PPW];  // false: outside the grid, swept on another pixel's (or a null) row and not stored
    int pair_rows;  // table rows between the two pixels of a pair
    if (a.cols > 0) {
        static_assert(PPW == 4, "vertical pairing: 2 columns x 2 rows per wave");
        const int tiles_per_rowpair = (a.cols + 2 * NW - 1) / (2 * NW);
        const int row2 = tile / tiles_per_rowpair, col0 = (tile - row2 * tiles_per_rowpair) * 2 * NW + 2 * wave;
        const int rows = a.pixel_count / a.cols;
#pragma unroll
        for (int q = 0; q < PPW; q++) {
            const int row = 2 * row2 + (q & 1), col = col0 + (q >> 1);
            live[q] = row < rows && col < a.cols;
            pix[q] = min(2 * row2 * a.cols + col, a.pixel_count - 1) + (q & 1) * a.cols;  // B's row = A's + cols rows
        }
        pair_rows = a.cols;
    } else {
#pragma unroll
        for (int q = 0; q < PPW; q++) {
            pix[q] = (tile * NW + wave) * PPW + q;
            live[q] = pix[q] < a.pixel_count;
        }
        pair_rows = 1;
    }
    const size_t row_floats = (size_t) a.wp * 2;
    const float *pair_base = a.packed + (size_t) pair * a.usable * row_floats;

    f2 acc[PPW][8];
#pragma unroll
    for (int pp = 0; pp < PPW; pp++)
#pragma unroll
        for (int k = 0; k < 8; k++) acc[pp][k] = f2{0.0f, 0.0f};

    // one chunk = rows m0 .. m0+mc of this pair, contiguous in HBM and in the LDS image
    auto dma_chunk = [&](int m0, int mc, int buf) {
        const float *src = pair_base + (size_t) m0 * row_floats;
        const int n_pieces = (int) ((size_t) mc * row_floats / 4);
#pragma unroll
        for (int k = 0; k < kPieces; k++) {
            const int piece = threadIdx.x + k * kThreads;
            if (piece < n_pieces) {
                float *dst = lds + buf * (BUF / 4) + (wave * 64 + k * kThreads) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (src + (size_t) piece * 4),
                                                 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
            }
        }
    };

    unsigned t_wait = 0, t_all = 0;
    const long long t_begin = __builtin_readcyclecounter();
    const int n_chunks = (a.usable + a.chunk - 1) / a.chunk;
    dma_chunk(0, min(a.chunk, a.usable), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    unsigned t_ph[5] = {0, 0, 0, 0, 0};
    auto stamp = [&](int k, long long &t) {
        if (DIAG) {
            const long long n = __builtin_readcyclecounter();
            t_ph[k] += (unsigned) (n - t);
            t = n;
        }
    };
    const int stride = pair_rows * a.usable_pad * (int) sizeof(FastEntry);  // from pixel A's row to pixel B's
    const int rank = wave >> 2;  // age order of this wave among the four that share its SIMD
    for (int c = 0; c < n_chunks; c++) {
        const int m0 = c * a.chunk;
        const int mc = min(a.chunk, a.usable - m0);
        const int mc4 = (mc + 3) & ~3;
        const int buf = c & 1;
        long long t = DIAG ? __builtin_readcyclecounter() : 0;
        if (c + 1 < n_chunks && !AWPU_DBG(a, 1)) dma_chunk(m0 + a.chunk, min(a.chunk, a.usable - m0 - a.chunk), buf ^ 1);
        stamp(0, t);
        const unsigned lane_addr = lds_base + buf * BUF + lane * 8;
#pragma unroll
        for (int q = 0; q < PPW; q += 2) {
            const void *row = uniform_ptr(a.lut + (size_t) pix[q] * a.usable_pad + m0);
            const int ng = __builtin_amdgcn_readfirstlane(mc4 >> 2);
            if constexpr (DIAG) {
                unsigned dw = 0, da = 0;
                if constexpr (SHARE) sweep_duo_shared_stamped(acc[q], acc[q + 1], row, stride, ng, lane_addr, rank, dw, da);
                else sweep_duo_pairs_stamped(acc[q], acc[q + 1], row, stride, ng, lane_addr, rank, dw, da);
                t_wait += dw;
                t_all += da;
            } else {
                if constexpr (SHARE) sweep_duo_shared(acc[q], acc[q + 1], row, stride, ng, lane_addr, rank);
                else sweep_duo_pairs(acc[q], acc[q + 1], row, stride, ng, lane_addr, rank);
            }
        }
        stamp(1, t);
        // (no 257th-sample pass: the rows carry pre-filtered samples, pack_one_row<true>)
        if (DIAG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(2, t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(3, t);
        if (!AWPU_DBG(a, 8)) __syncthreads();
        stamp(4, t);
    }

    if (DIAG && a.debug_out && lane == 0) {
        unsigned long long *o = a.debug_out + 12 * ((size_t) item * NW + wave);
        o[0] = t_wait;
        o[1] = t_all;
        o[2] = (unsigned long long) (__builtin_readcyclecounter() - t_begin);
        o[3] = (unsigned long long) n_chunks * PPW;
        for (int k = 0; k < 5; k++) o[4 + k] = t_ph[k];
    }
    const float norm = (float) (kSamples * a.usable);
    static_assert(PPW == 4, "store_powers8: four pixels x two frames per wave");
    f2 part[PPW];
#pragma unroll
    for (int pp = 0; pp < PPW; pp++) {
        part[pp] = pixel_pair_partial_filtered(acc[pp], lane);
    }
    store_powers8(part, pix, live, pair, a.batch, a.pixel_count, norm, a.power, lane);
}


// ---------------------------------------------------------------------------------------
// Stationary shape: when the touched window of EVERY active mic of a frame pair fits the CU's LDS at once -- the
// reference's own configuration does: one 8x8 array, 64 mics x 285 samples x 8 bytes = 146 KB of the 156 KB the two
// images span -- a workgroup stages the pair once and then sweeps `tiles_per_wg` tiles of the grid from it, every
// wave on its own: no refill, no chunk loop, no barrier after the first, and the staging is amortised over
// several tiles (das_pair_kernel re-stages the pair for every tile and meets a barrier per chunk: on such small
// problems that machinery, not the arithmetic, was where the time went).  Same blocks, same per-pixel order of the
// items as das_pair_kernel: the sums are the same bits.  grid = (frame pairs, groups of tiles).
// ---------------------------------------------------------------------------------------
template <bool SHARE>
__global__ __launch_bounds__(1024, 4) void das_pair_stationary_kernel(PairArgs a, int n_tiles, int tiles_per_wg) {
    constexpr int PPW = 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = 16, kThreads = NW * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned) (unsigned long long) (const __attribute__((address_space(3))) char *) lds;
    const int pair = blockIdx.x;
    const size_t row_floats = (size_t) a.wp * 2;
    const float *pair_base = a.packed + (size_t) pair * a.usable * row_floats;

    if (a.frames) {
        // ---- stage the pair from the caller's frames (round 4): a wave takes every 16th active stream, a lane every 64th
        // element of its row, and writes pack_one_row<true>'s value -- the stencil of the two frames' samples, neighbours
        // outside the history counting as 0 -- straight into the image: the bits pack_pairs_kernel<true> + the LDS-DMA stream
        // below would have put there, without the pre-pass (a launch and 28 MB of HBM traffic per 128 frames of one 8x8 array)
        const int fa = min(2 * pair, a.batch - 1), fb = min(2 * pair + 1, a.batch - 1);
        const int valid = min(a.wp, a.hist - a.wstart);
        f2 *image = (f2 *) lds;
        for (int s = wave; s < a.usable; s += NW) {
            const int stream = __builtin_amdgcn_readfirstlane(a.index[s]);
            const float *xa = a.frames + ((size_t) fa * a.n_streams + stream) * a.hist + a.wstart;
            const float *xb = a.frames + ((size_t) fb * a.n_streams + stream) * a.hist + a.wstart;
            f2 *row = image + (size_t) s * a.wp;
            // five elements per lane at a time, all their loads in flight together (a row of the reference's array is 286
            // elements: one round trip per row instead of five)
            for (int t0 = lane; t0 < a.wp; t0 += 5 * 64) {
                float c[5][2], l[5][2], h[5][2];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const int t = t0 + 64 * k;
                    const bool in = t < valid, lo = in && a.wstart + t > 0, hi = in && a.wstart + t + 1 < a.hist;
                    c[k][0] = in ? xa[t] : 0.0f;
                    c[k][1] = in ? xb[t] : 0.0f;
                    l[k][0] = lo ? xa[t - 1] : 0.0f;
                    l[k][1] = lo ? xb[t - 1] : 0.0f;
                    h[k][0] = hi ? xa[t + 1] : 0.0f;
                    h[k][1] = hi ? xb[t + 1] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const int t = t0 + 64 * k;
                    // (an element past the history is +0 in pack_one_row; here fma(-0.25, 0 + 0, 0.5 * 0) = +0 as well)
                    const f2 v = __builtin_elementwise_fma(f2{-0.25f, -0.25f}, f2{l[k][0] + h[k][0], l[k][1] + h[k][1]}, 0.5f * f2{c[k][0], c[k][1]});
                    if (t < a.wp) row[t] = v;
                }
            }
        }
        __syncthreads();
    } else {
    // ---- stage all rows of the (pre-packed) pair: one linear LDS-DMA stream (16 bytes per lane, 16 KiB per workgroup pass)
    const unsigned lane_bytes = threadIdx.x * 16;
    const unsigned n_bytes = (unsigned) ((size_t) a.usable * row_floats * 4);
    for (unsigned base = 0; base < n_bytes; base += kThreads * 16) {
        if (base + lane_bytes < n_bytes) {
            const char *src = (const char *) uniform_ptr((const char *) pair_base + base);
            float *dst = lds + (base >> 2) + wave * 256;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (src + lane_bytes),
                                             (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    }

    const int rank = wave >> 2;
    const int ng = __builtin_amdgcn_readfirstlane(a.usable_pad >> 2);
    const unsigned lane_addr = lds_base + lane * 8;
    const float norm = (float) (kSamples * a.usable);
    const int tile_end = min(n_tiles, (int) (blockIdx.y + 1) * tiles_per_wg);
    for (int tile = blockIdx.y * tiles_per_wg; tile < tile_end; tile++) {
        int pix[PPW];
        bool live[PPW];
        int pair_rows;
        if (a.cols > 0) {  // 2 rows x 32 columns per workgroup, a wave two vertical pixel pairs (see das_pair_kernel)
            const int tiles_per_rowpair = (a.cols + 2 * NW - 1) / (2 * NW);
            const int row2 = tile / tiles_per_rowpair, col0 = (tile - row2 * tiles_per_rowpair) * 2 * NW + 2 * wave;
            const int rows = a.pixel_count / a.cols;
#pragma unroll
            for (int q = 0; q < PPW; q++) {
                const int row = 2 * row2 + (q & 1), col = col0 + (q >> 1);
                live[q] = row < rows && col < a.cols;
                pix[q] = min(2 * row2 * a.cols + col, a.pixel_count - 1) + (q & 1) * a.cols;
            }
            pair_rows = a.cols;
        } else {
#pragma unroll
            for (int q = 0; q < PPW; q++) {
                pix[q] = (tile * NW + wave) * PPW + q;
                live[q] = pix[q] < a.pixel_count;
            }
            pair_rows = 1;
        }
        f2 acc[PPW][8];
#pragma unroll
        for (int pp = 0; pp < PPW; pp++)
#pragma unroll
            for (int k = 0; k < 8; k++) acc[pp][k] = f2{0.0f, 0.0f};
        const int stride = pair_rows * a.usable_pad * (int) sizeof(FastEntry);
#pragma unroll
        for (int q = 0; q < PPW; q += 2) {
            const void *row = uniform_ptr(a.lut + (size_t) pix[q] * a.usable_pad);
            if constexpr (SHARE) sweep_duo_shared(acc[q], acc[q + 1], row, stride, ng, lane_addr, rank);
            else sweep_duo_pairs(acc[q], acc[q + 1], row, stride, ng, lane_addr, rank);
        }
        static_assert(PPW == 4, "store_powers8: four pixels x two frames per wave");
        f2 part[PPW];
#pragma unroll
        for (int pp = 0; pp < PPW; pp++) {
            part[pp] = pixel_pair_partial_filtered(acc[pp], lane);
        }
        store_powers8(part, pix, live, pair, a.batch, a.pixel_count, norm, a.power, lane);
    }
}

// ---------------------------------------------------------------------------------------
// Reference-order sweep on the frame-pair layout (AWPU_MATH_F32_EXACT, round 4): das_pair_kernel's staging -- rows of
// RAW samples (pack_pairs_kernel<false>: the gain, if any, applied to the samples as das_exact_kernel applies it), one
// linear LDS-DMA stream per chunk, two images, one barrier per chunk -- around the block sweep_duo_exact, which performs
// delay.cpp:19-25's three operations per sample in the reference's order with the mics in antenna.index[] order
// (mimo.cpp:124-130).  out[i] of a pixel lives in acc[k] of lane l (i = l + 64 k), both frames of the pair side by side,
// and never leaves its register between the first mic and the epilogue: the pre-epilogue sums are bit-identical to the
// reference's (and to das_exact_kernel's; a.sums exports them for the tests).  The epilogue is mimo.cpp:131-137 with
// whole-wave rotations for the neighbours.  Padding mics (usable rounded up to 4) read rows of zeros with fraction 0:
// t = fma(0, 0 - 0, 0) = +0 and out + 0 = out bit for bit (out starts at +0 and can never become -0).
// Workgroup = 16 waves, a wave 4 pixels (two vertical pairs when the row length is known); 1-D grid of items, XCD-aware order.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ f2 pixel_pair_partial_exact(const f2 (&o)[4], int lane) {  // a lane's share of sum MA^2, both frames
    f2 sum = f2{0.0f, 0.0f};
    f2 dn = wave_rotate<kDppWaveRol1>(o[0]);         // out[l+1 + 64k]; lane 63: out[64k]
    f2 up_before = wave_rotate<kDppWaveRor1>(o[0]);  // (k = 0, lane 0: sample 0 is not summed)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const f2 dn_after = k < 3 ? wave_rotate<kDppWaveRol1>(o[k < 3 ? k + 1 : 3]) : dn;  // (k = 3, lane 63: sample 255 is not summed)
        const f2 up = wave_rotate<kDppWaveRor1>(o[k]);  // out[l-1 + 64k]; lane 0: out[63 + 64k]
        const f2 next = lane == 63 ? dn_after : dn;
        const f2 prev = lane == 0 ? up_before : up;
        const int i = lane + 64 * k;
        const f2 ma = o[k] * 0.5f - 0.25f * (next + prev);  // mimo.cpp:132-134
        if (i >= 1 && i <= kSamples - 2) sum += ma * ma;
        dn = dn_after;
        up_before = up;
    }
    return sum;
}
// (das_exact_pair_kernel and the FIR8 plane kernel keep one wave sum per pixel and frame: through store_powers8 -- per-lane
// selects of the wave's four pixel indices -- the joint reduction measured 0.2-0.5 % SLOWER there, gpurun_out/epi3ab, round 4;
// das_exact_quad_kernel computes the lane's pixel from row and column like das_quad_kernel and gains 0.65 %, gpurun_out/exq1)
__device__ __forceinline__ f2 finish_pixel_pair_exact(const f2 (&o)[4], int lane) {
    f2 sum = pixel_pair_partial_exact(o, lane);
    sum.x = wave_sum(sum.x);
    sum.y = wave_sum(sum.y);
    return sum;
}

__global__ __launch_bounds__(1024, 4) void das_exact_pair_kernel(ExactPairArgs a) {
    constexpr int PPW = 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = 16, kThreads = NW * 64, BUF = kFastLdsBytes;
    constexpr int kPieces = (BUF + kThreads * 16 - 1) / (kThreads * 16);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned) (unsigned long long) (const __attribute__((address_space(3))) char *) lds;
    // items (frame pair, tile) in das_quad_kernel's order -- (pair group, tile, pair), one contiguous run per XCD -- so that an
    // XCD's workgroups share a few pairs' samples in its L2 while they walk the table (see das_exact_quad_kernel)
    const int total = a.n_pairs * a.tiles;
    const int per_xcd = (total + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= min(total, ((int) (blockIdx.x & 7) + 1) * per_xcd)) return;  // (uniform for the workgroup)
    int pair, tile;
    {
        const int full_items = (a.n_pairs / a.pair_group) * a.pair_group * a.tiles;
        const int ga = item < full_items ? a.pair_group : a.n_pairs % a.pair_group;
        const int rem = item < full_items ? item : item - full_items;
        const int grp = rem / (a.tiles * ga), in = rem - grp * a.tiles * ga;
        tile = in / ga;
        pair = (item < full_items ? grp * a.pair_group : a.n_pairs - ga) + (in - tile * ga);
    }
    int pix[PPW];    // pixel of slot q (slots 2j, 2j+1 are the two pixels of one block)
    bool live[PPW];  // false: outside the grid, swept on another pixel's (or a null) row and not stored
    int pair_rows;   // table rows between the two pixels of a block
    if (a.cols > 0) {  // 2 rows x 32 columns per workgroup, a wave two vertical pixel pairs (see das_pair_kernel)
        const int tiles_per_rowpair = (a.cols + 2 * NW - 1) / (2 * NW);
        const int row2 = tile / tiles_per_rowpair, col0 = (tile - row2 * tiles_per_rowpair) * 2 * NW + 2 * wave;
        const int rows = a.pixel_count / a.cols;
#pragma unroll
        for (int q = 0; q < PPW; q++) {
            const int row = 2 * row2 + (q & 1), col = col0 + (q >> 1);
            live[q] = row < rows && col < a.cols;
            pix[q] = min(2 * row2 * a.cols + col, a.pixel_count - 1) + (q & 1) * a.cols;
        }
        pair_rows = a.cols;
    } else {
#pragma unroll
        for (int q = 0; q < PPW; q++) {
            pix[q] = (tile * NW + wave) * PPW + q;
            live[q] = pix[q] < a.pixel_count;
        }
        pair_rows = 1;
    }
    const size_t row_floats = (size_t) a.wp * 2;
    const float *pair_base = a.packed + (size_t) pair * a.usable_pad * row_floats;  // usable_pad rows per pair, the extra ones zero

    f2 acc[PPW][4];
#pragma unroll
    for (int pp = 0; pp < PPW; pp++)
#pragma unroll
        for (int k = 0; k < 4; k++) acc[pp][k] = f2{0.0f, 0.0f};  // float out[N_SAMPLES] = {0.0}, mimo.cpp:122

    auto dma_chunk = [&](int m0, int mc, int buf) {  // rows m0 .. m0+mc of this pair: contiguous in HBM and in the image
        const float *src = pair_base + (size_t) m0 * row_floats;
        const int n_pieces = (int) ((size_t) mc * row_floats / 4);
#pragma unroll
        for (int k = 0; k < kPieces; k++) {
            const int piece = threadIdx.x + k * kThreads;
            if (piece < n_pieces) {
                float *dst = lds + buf * (BUF / 4) + (wave * 64 + k * kThreads) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (src + (size_t) piece * 4),
                                                 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
            }
        }
    };

    const int n_chunks = (a.usable_pad + a.chunk - 1) / a.chunk;
    dma_chunk(0, min(a.chunk, a.usable_pad), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int stride = pair_rows * a.usable_pad * (int) sizeof(FastEntry);  // from pixel A's row to pixel B's
    const int rank = wave >> 2;  // age order of this wave among the four that share its SIMD
    for (int c = 0; c < n_chunks; c++) {
        const int m0 = c * a.chunk;
        const int mc = min(a.chunk, a.usable_pad - m0);  // a multiple of 4: chunk and usable_pad both are
        const int buf = c & 1;
        if (c + 1 < n_chunks) dma_chunk(m0 + a.chunk, min(a.chunk, a.usable_pad - m0 - a.chunk), buf ^ 1);
        const unsigned lane_addr = lds_base + buf * BUF + lane * 8;
#pragma unroll
        for (int q = 0; q < PPW; q += 2) {
            const void *row = uniform_ptr(a.lut + (size_t) pix[q] * a.usable_pad + m0);
            const int ng = __builtin_amdgcn_readfirstlane(mc >> 2);
            sweep_duo_exact(acc[q], acc[q + 1], row, stride, ng, lane_addr, rank);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const float norm = (float) (kSamples * a.usable);
#pragma unroll
    for (int pp = 0; pp < PPW; pp++) {
        const int p = pix[pp];
        if (a.sums && live[pp]) {  // the pre-epilogue sums, for the tests: [batch][pixel_count][256]
#pragma unroll
            for (int k = 0; k < 4; k++) {
                a.sums[((size_t) (2 * pair) * a.pixel_count + p) * kSamples + lane + 64 * k] = acc[pp][k].x;
                if (2 * pair + 1 < a.batch) a.sums[((size_t) (2 * pair + 1) * a.pixel_count + p) * kSamples + lane + 64 * k] = acc[pp][k].y;
            }
        }
        const f2 sum = finish_pixel_pair_exact(acc[pp], lane);
        if (lane == 0 && live[pp]) {
            a.power[(size_t) (2 * pair) * a.pixel_count + p] = sum.x / norm;
            if (2 * pair + 1 < a.batch) a.power[(size_t) (2 * pair + 1) * a.pixel_count + p] = sum.y / norm;
        }
    }
}

// ---------------------------------------------------------------------------------------
// The same reference-order sweep with FOUR vertically adjacent pixels per wave (round 4; taken where the grid's row length is
// known and vertical neighbours share their integer delays more often than horizontal ones): das_exact_pair_kernel's staging
// around the block sweep_quad_exact, in which the reads of a mic's samples and the difference cur - next are shared by
// every pixel of the column that carries the reference pixel's LDS address.  Per pixel the three operations and the mic
// order are unchanged: the same bits as das_exact_pair_kernel and das_exact_kernel (a.sums exports them).
// Tile = 4 rows x 16 columns (a wave one column), quad-major table as das_quad_kernel's with the RAW fraction; 1-D grid of items.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024, 4) void das_exact_quad_kernel(ExactQuadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = 16, kThreads = NW * 64, BUF = kFastLdsBytes;
    constexpr int kPieces = (BUF + kThreads * 16 - 1) / (kThreads * 16);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned) (unsigned long long) (const __attribute__((address_space(3))) char *) lds;
    // Dispatch as das_quad_kernel's: the items (frame pair, tile), ordered (pair group, tile, pair), are cut into 8 contiguous
    // runs, one per XCD (blockIdx & 7: round-robin placement, assumed for speed only), so that at any moment an XCD's
    // workgroups sweep `pair_group` frame pairs x a few tiles: the pairs' samples stay in that XCD's 4 MiB L2 while it walks the
    // table once per pair group (a (pairs, tiles) grid streamed 3.2 GB per headline launch from beyond the L2: PMC).
    const int total = a.n_pairs * a.tiles;
    const int per_xcd = (total + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= min(total, ((int) (blockIdx.x & 7) + 1) * per_xcd)) return;  // (uniform for the workgroup)
    int pair, tile;
    {
        const int full_items = (a.n_pairs / a.pair_group) * a.pair_group * a.tiles;  // items of whole pair groups
        const int ga = item < full_items ? a.pair_group : a.n_pairs % a.pair_group;   // the last group may be smaller
        const int rem = item < full_items ? item : item - full_items;
        const int grp = rem / (a.tiles * ga), in = rem - grp * a.tiles * ga;
        tile = in / ga;
        pair = (item < full_items ? grp * a.pair_group : a.n_pairs - ga) + (in - tile * ga);
    }
    const int tiles_per_row4 = (a.cols + NW - 1) / NW;
    const int row4 = tile / tiles_per_row4;
    const int col = (tile - row4 * tiles_per_row4) * NW + wave;
    const int quad = row4 * tiles_per_row4 * NW + col;  // the table's quads: columns padded to whole tiles
    const int groups_total = a.usable_pad >> 2;
    const QuadEntry *quad_lut = a.lut + (size_t) quad * groups_total * 16;
    const size_t row_floats = (size_t) a.wp * 2;
    const float *pair_base = a.packed + (size_t) pair * a.usable_pad * row_floats;

    f8 O0 = {0, 0, 0, 0, 0, 0, 0, 0}, O1 = O0, O2 = O0, O3 = O0;  // float out[N_SAMPLES] = {0.0}, mimo.cpp:122

    auto dma_chunk = [&](int m0, int mc, int buf) {
        const float *src = pair_base + (size_t) m0 * row_floats;
        const int n_pieces = (int) ((size_t) mc * row_floats / 4);
#pragma unroll
        for (int k = 0; k < kPieces; k++) {
            const int piece = threadIdx.x + k * kThreads;
            if (piece < n_pieces) {
                float *dst = lds + buf * (BUF / 4) + (wave * 64 + k * kThreads) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (src + (size_t) piece * 4),
                                                 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
            }
        }
    };

    const int n_chunks = (a.usable_pad + a.chunk - 1) / a.chunk;
    dma_chunk(0, min(a.chunk, a.usable_pad), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int rank = wave >> 2;
    for (int c = 0; c < n_chunks; c++) {
        const int m0 = c * a.chunk;
        const int mc = min(a.chunk, a.usable_pad - m0);
        const int buf = c & 1;
        if (c + 1 < n_chunks) dma_chunk(m0 + a.chunk, min(a.chunk, a.usable_pad - m0 - a.chunk), buf ^ 1);
        const unsigned lane_addr = lds_base + buf * BUF + lane * 8;
        const void *row = uniform_ptr(quad_lut + (size_t) (m0 >> 2) * 16);
        sweep_quad_exact(O0, O1, O2, O3, row, __builtin_amdgcn_readfirstlane(mc >> 2), lane_addr, rank);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const float norm = (float) (kSamples * a.usable);
    auto finish = [&](const f8 &O, int q) {
        const int row = 4 * row4 + q;
        const bool live = row < a.rows && col < a.cols;
        const int p = min(row, a.rows - 1) * a.cols + min(col, a.cols - 1);
        f2 o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = f2{O[2 * k], O[2 * k + 1]};
        if (a.sums && live) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                a.sums[((size_t) (2 * pair) * a.pixel_count + p) * kSamples + lane + 64 * k] = o[k].x;
                if (2 * pair + 1 < a.batch) a.sums[((size_t) (2 * pair + 1) * a.pixel_count + p) * kSamples + lane + 64 * k] = o[k].y;
            }
        }
        return pixel_pair_partial_exact(o, lane);
    };
    const f2 s0 = finish(O0, 0), s1 = finish(O1, 1), s2 = finish(O2, 2), s3 = finish(O3, 3);
    {   // the eight wave sums together, one division and one store per quad and frame pair (as das_quad_kernel's epilogue)
        const float total = wave_sum8(s0.x, s0.y, s1.x, s1.y, s2.x, s2.y, s3.x, s3.y, lane);
        const int value = kWaveSum8Value(lane >> 3), row = 4 * row4 + (value >> 1), frame = 2 * pair + (value & 1);
        if ((lane & 7) == 0 && row < a.rows && col < a.cols && frame < a.batch)
            a.power[(size_t) frame * a.pixel_count + (size_t) row * a.cols + col] = total / norm;
    }
}

// ---------------------------------------------------------------------------------------
// The reference's order on the {next, d} layout (round 5; the default of AWPU_MATH_F32_EXACT wherever the grid's row length is known).
// delay.cpp:19-25 performs, per pixel, mic and sample:  d = cur - next;  t = fma(frac, d, next);  out += t.  The first operation
// depends on (mic, sample, frame) only -- never on the pixel or the fraction -- and an fp32 subtraction of the same two operands gives
// the same bits wherever it is done: pack_nd_kernel forms it ONCE per sample and writes, for the frame pair (a, b) and sample t of a
// mic's window, the 16-byte element { next_a, next_b, d_a, d_b } (next = X[t+1], d = X[t] - X[t+1]; the per-mic gain, if any, multiplies
// the samples first, as in das_exact_kernel).  The sweep is then 4 ds_read_b128 per distinct address + v_pk_fma_f32 t, frac, d, next +
// v_pk_add_f32 out, out, t per register: 8 packed VALU instructions per (pixel, mic, frame pair) flat, where das_exact_quad_kernel paid
// 8 + 4 per distinct address -- and the pre-epilogue sums are still the reference's bits (a.sums exports them).
// Workgroup = 16 waves, a wave NQ quads of four vertically adjacent pixels (tile = 4 NQ rows x 16 columns: twice the pixels per
// barrier, the 16-byte elements halve the mics per chunk); the whole item -- chunk loop, in-block refill, barrier -- runs inside
// sweep_exact_nd_item<NQ> (tools/gen_trip_asm.py, block_exact_nd).  Epilogue, item order: das_exact_quad_kernel's.
// ---------------------------------------------------------------------------------------
// (next, cur - next) of one sample with the per-mic gain on both samples first, every operation rounded on its own: hipcc's
// __fmul_rn / __fsub_rn are plain `*` and `-`, which it contracts into an FMA where it can (caught by the gains case of
// test_exact_mode_single_frames_are_the_reference_bits); with contraction off in this scope the three roundings survive inlining
__device__ __forceinline__ f2 next_and_difference(float cur, float next, float gm) {
#pragma clang fp contract(off)
    const float c = cur * gm, n = next * gm;  // x * 1.0f is x
    return f2{n, c - n};                      // delay.cpp:21: cur - next
}

__global__ void pack_nd_kernel(const float *frames, int n_streams, int hist, int wstart, const int32_t *index, int usable,
                               const float *gain, int wq, int batch, float *packed) {
    const int pair = blockIdx.y, s = blockIdx.x, rows_out = gridDim.x;
    f4 *dst = (f4 *) packed + ((size_t) pair * rows_out + s) * wq;
    if (s >= usable) {  // padding rows (whole groups of four mics are swept): next = 0, d = 0 -> t = fma(0, 0, 0) = +0
        for (int t = threadIdx.x; t < wq; t += blockDim.x) dst[t] = f4{0.0f, 0.0f, 0.0f, 0.0f};
        return;
    }
    const int fa = min(2 * pair, batch - 1), fb = min(2 * pair + 1, batch - 1);
    const float *xa = frames + ((size_t) fa * n_streams + index[s]) * hist + wstart;
    const float *xb = frames + ((size_t) fb * n_streams + index[s]) * hist + wstart;
    const float gm = gain ? gain[s] : 1.0f;  // x * 1.0f is x
    const int valid = min(wq, hist - wstart - 1);  // elements whose two samples lie inside the stream's history
    for (int t = threadIdx.x; t < wq; t += blockDim.x) {
        f4 v = f4{0.0f, 0.0f, 0.0f, 0.0f};
        if (t < valid) {
            const f2 ea = next_and_difference(xa[t], xa[t + 1], gm), eb = next_and_difference(xb[t], xb[t + 1], gm);
            v = f4{ea.x, eb.x, ea.y, eb.y};
        }
        dst[t] = v;
    }
}

// Work distribution (round 5): PERSISTENT workgroups -- one per CU: the two LDS images leave room for no second -- that take items
// (frame pair, tile) from queues (NdQueues below).  The items are ordered (pair group, tile, pair) and cut into eight contiguous runs,
// one per XCD (HW_REG_XCC_ID), as before: an XCD's workgroups sweep a few frame pairs x a few tiles at a time, the pairs' samples stay
// in its 4 MiB L2 while it walks the table.  Measured on that static split, the XCDs of one chip finished 6 % apart (their clocks
// differ) and every CU idled 3.3 % of the launch waiting for the slowest; with ONE queue for the whole chip the tail shrank to 1.9 %
// (-1.6 ... -2.6 %) but every XCD then loaded every pair's rows (L2 hit rate 96 -> 83 %, 2.8 GB instead of 0.7 GB from beyond the L2
// per launch); a run's last eighth in a common queue keeps both: another -2.3 % (profiles/r05_ablation_nd_kernel.txt).  A workgroup
// knows its next item one item ahead (thread 0 takes it while the current one is swept; the mailbox is two ints behind the images),
// so the block refills the next item's first chunk beside the current item's last: no staging gap between items (2.2 % before).
// The item list of das_exact_nd_kernel: items[i] = (frame pair, first table quad of the tile) in the order (pair group, tile, pair) --
// `pair_group` frame pairs x all tiles, then the next group (the last one may be smaller): consecutive items share a few pairs' samples.
__global__ void nd_items_kernel(int2 *items, int n_pairs, int tiles, int pair_group, int tiles_per_row, int nq) {
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= n_pairs * tiles) return;
    const int full_items = (n_pairs / pair_group) * pair_group * tiles;
    const int ga = item < full_items ? pair_group : n_pairs % pair_group;
    const int rem = item < full_items ? item : item - full_items;
    const int grp = rem / (tiles * ga), in = rem - grp * tiles * ga;
    const int tile = in / ga;
    const int pair = (item < full_items ? grp * pair_group : n_pairs - ga) + (in - tile * ga);
    const int rowq = tile / tiles_per_row, colt = tile - rowq * tiles_per_row;
    items[item] = int2{pair, (nq * rowq * tiles_per_row + colt) * 16};
}

// The queues.  The item order is cut into eight runs of `per` items, run x = XCD x's: its frame pairs' rows stay in that XCD's L2.  Each
// run's last `tail` items belong to a NINTH, common queue (dealt run by run, round-robin); queue x < 8 holds the rest of run x.  A
// workgroup of XCD x drains queue x, then the common queue, then -- last resort -- the other XCDs' queues: the XCDs of a chip finish their
// equal shares 6 % apart (their clocks differ), and the common tail is what the fast ones take from the slow ones.  tail = the whole
// run (a.tail >= per) makes it ONE queue for the chip.
struct NdQueues {
    unsigned *q;   // [9] counters, zeroed by the launcher
    int per, total, tail;
    __device__ __forceinline__ int len(int x) const { return max(0, min(per, total - x * per)); }
    __device__ __forceinline__ int head(int x) const { return len(x) - min(len(x), tail); }
    __device__ __forceinline__ int take_head(int x) const {
        if (head(x) <= 0) return -1;
        const unsigned i = atomicAdd(&q[x], 1u);
        return i < (unsigned) head(x) ? x * per + (int) i : -1;
    }
    __device__ __forceinline__ int take_common() const {
        // run after run, not a ticket to every run in turn: the workgroups that share the tail then work on ONE frame pair group's rows
        // at a time.  (Dealt round-robin, every XCD had eight pair groups' rows in flight during the tail -- more than its L2 holds --
        // and an eighth of the launch's items re-read their rows from beyond it: 2.81 -> 2.04 GB per launch, FAST 1.82 -> 1.00; 0.1-0.3 % of
        // the time: profiles/r05_ablation_nd_kernel.txt.)
        if (tail <= 0) return -1;
        for (;;) {
            const unsigned j = atomicAdd(&q[8], 1u);
            const int run = (int) (j / (unsigned) tail), k = (int) (j - (unsigned) run * (unsigned) tail);
            if (run >= 8) return -1;
            if (k < len(run) - head(run)) return run * per + head(run) + k;
        }
    }
    __device__ __forceinline__ int take(int xcd) const {  // own queue, the common tail, then anybody's
        int item = take_head(xcd);
        if (item < 0) item = take_common();
        for (int y = 1; item < 0 && y < 8; y++) item = take_head((xcd + y) & 7);
        return item;
    }
};

template <int NQ, bool SUMS>
__global__ __launch_bounds__(1024, 4) void das_exact_nd_kernel(ExactNdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = 16, kThreads = NW * 64, BUF = kFastLdsBytes;
    constexpr int kPieces = (BUF + kThreads * 16 - 1) / (kThreads * 16);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned) (unsigned long long) (const __attribute__((address_space(3))) char *) lds;
    int *mail = (int *) (lds + 2 * (BUF / 4));  // two ints behind the images: the items thread 0 took for the workgroup
    const int total = a.n_pairs * a.tiles;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const NdQueues Q{a.queue, (total + 7) >> 3, total, a.tail};
    const int xcd = (int) (xcc & 7u);
    if (threadIdx.x == 0) {
        const int first = Q.take(xcd);
        mail[0] = first;
        mail[1] = first >= 0 ? Q.take(xcd) : -1;
    }
    __syncthreads();
    int cur = __builtin_amdgcn_readfirstlane(mail[0]), nxt = __builtin_amdgcn_readfirstlane(mail[1]);
    if (cur < 0) return;  // (uniform for the workgroup)
    __syncthreads();      // (the mailbox is written again below)

    // where an item lies: a.items[item] = (frame pair, first quad of the tile in the table), written by nd_items_kernel in the order
    // (pair group, tile, pair) -- one scalar load per item here instead of five integer divisions and their constants
    auto decode = [&](int item, int &pair, int &tile_quad) {
        const int2 d = a.items[item];
        pair = __builtin_amdgcn_readfirstlane(d.x);
        tile_quad = __builtin_amdgcn_readfirstlane(d.y);
    };
    const int tiles_per_row = (a.cols + NW - 1) / NW;
    const int groups_total = a.usable_pad >> 2;
    const size_t row_floats = (size_t) a.wq * 4;
    const int n_chunks = (a.usable_pad + a.chunk - 1) / a.chunk;
    const int first_mics = min(a.chunk, a.usable_pad), last_mics = a.usable_pad - (n_chunks - 1) * a.chunk;
    const int ngf = __builtin_amdgcn_readfirstlane(first_mics >> 2), ngl = __builtin_amdgcn_readfirstlane(last_mics >> 2);
    const unsigned dbf = __builtin_amdgcn_readfirstlane((unsigned) ((size_t) a.chunk * row_floats * 4));
    const unsigned dbl = __builtin_amdgcn_readfirstlane((unsigned) ((size_t) last_mics * row_floats * 4));
    const unsigned db0 = __builtin_amdgcn_readfirstlane((unsigned) ((size_t) first_mics * row_floats * 4));
    const int rank = wave >> 2;  // age order of this wave among the four that share its SIMD
    const int own_head = Q.head(xcd);  // items of this XCD's own queue

    int pair, tile;
    decode(cur, pair, tile);
#ifdef AWPU_TUNING_BUILD
    const unsigned long long rt_begin = a.debug_out ? __builtin_amdgcn_s_memrealtime() : 0ull;  // 100 MHz
    long long t_sweep = 0, t_other = 0, t_mark = __builtin_readcyclecounter();
    int n_items = 0;
#endif
    {   // the first item's chunk 0 into image 0 (the block refills everything after it, the next item's first chunk included)
        const float *pair_base = a.packed + (size_t) pair * a.usable_pad * row_floats;
        const int n_pieces = (int) ((size_t) first_mics * row_floats / 4);
#pragma unroll
        for (int k = 0; k < kPieces; k++) {
            const int piece = threadIdx.x + k * kThreads;
            if (piece < n_pieces) {
                float *dst = lds + (wave * 64 + k * kThreads) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (pair_base + (size_t) piece * 4),
                                                 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned step = 0;  // chunks swept so far: the chunk about to be swept lives in image step & 1

    for (;;) {
        // tile = the table's first quad of the tile ([quad row][columns padded to whole tiles]): quad row NQ * rowq, column 16 x ...
        const int cols_pad = tiles_per_row * NW;
        const int rowq = tile / (NQ * cols_pad);  // (one division per item, for the epilogue's rows)
        const int col = tile - NQ * rowq * cols_pad + wave;
        const int quad0 = tile + wave;
        const QuadEntry *quad_lut = a.lut + (size_t) quad0 * groups_total * 16;
        const float *pair_base = a.packed + (size_t) pair * a.usable_pad * row_floats;
        int pair_next = pair, tile_next = tile;
        if (nxt >= 0) decode(nxt, pair_next, tile_next);
        const float *next_base = a.packed + (size_t) pair_next * a.usable_pad * row_floats;
        // Wave 0 asks its XCD's queue for the item after the next one INSIDE the sweep block (qptr: lane 0 adds one to that counter as
        // the block begins) and thread 0 looks at the answer after the sweep: the atomic's round trip -- microseconds, with every
        // workgroup of the XCD on the same counter -- runs beside the item, not in front of it.  (From C++ the compiler's atomic
        // optimizer reads the answer back at once, and any scratch reload in front of the block waits for it with vmcnt(0).)
        unsigned ticket = 0;
        const bool asked = wave == 0 && nxt >= 0 && own_head > 0;
        const unsigned *qptr = (const unsigned *) uniform_ptr(asked ? a.queue + xcd : nullptr);

        f8 O[NQ][4];  // (float out[N_SAMPLES] = {0.0}, mimo.cpp:122: the block zeroes them itself)
#ifdef AWPU_TUNING_BUILD
        { const long long n = __builtin_readcyclecounter(); t_other += n - t_mark; t_mark = n; }
#endif
        {
            // (per-lane values are derived from the thread index HERE, through a volatile asm that is not hoisted out of the item loop:
            // the block's 116 pinned and clobbered registers leave the compiler 12, and every per-lane value it keeps across the items
            // -- it kept five -- is one it spills to scratch)
            unsigned tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const unsigned lane_bytes = tid * 16;
            const int buf = (int) (step & 1);
            unsigned lane_addr = lds_base + buf * BUF + (tid & 63) * 16;
            const unsigned ddst = __builtin_amdgcn_readfirstlane(lds_base + (buf ^ 1) * BUF + wave * 1024);
            const int delta = __builtin_amdgcn_readfirstlane(buf ? -BUF : BUF);
            const unsigned dbn = __builtin_amdgcn_readfirstlane(nxt >= 0 ? db0 : 0u);
            if constexpr (NQ == 1) {
                sweep_exact_nd_item1(O[0][0], O[0][1], O[0][2], O[0][3], uniform_ptr(quad_lut), ngf, ngl, __builtin_amdgcn_readfirstlane(n_chunks),
                                     lane_addr, rank, uniform_ptr(pair_base), dbf, dbl, uniform_ptr(next_base), dbn, ddst, delta, lane_bytes, qptr, ticket);
            } else {
                static_assert(NQ == 2, "blocks are generated for one and two quads per wave");
                const int qstride = __builtin_amdgcn_readfirstlane(tiles_per_row * NW * groups_total * 16 * (int) sizeof(QuadEntry));
                sweep_exact_nd_item2(O[0][0], O[0][1], O[0][2], O[0][3], O[1][0], O[1][1], O[1][2], O[1][3], uniform_ptr(quad_lut), qstride, ngf, ngl,
                                     __builtin_amdgcn_readfirstlane(n_chunks), lane_addr, rank, uniform_ptr(pair_base), dbf, dbl, uniform_ptr(next_base),
                                     dbn, ddst, delta, lane_bytes, qptr, ticket);
            }
            step += n_chunks;
        }
        if (threadIdx.x == 0 && nxt >= 0) {  // the answer, into the mailbox at once (not carried through the epilogue)
            int got = own_head > 0 && ticket < (unsigned) own_head ? xcd * Q.per + (int) ticket : -1;
            if (got < 0) {  // own queue empty: the common tail, then anybody's (only near the launch's end)
                got = Q.take_common();
                for (int y = 1; got < 0 && y < 8; y++) got = Q.take_head((xcd + y) & 7);
            }
            mail[0] = got;
        }
#ifdef AWPU_TUNING_BUILD
        { const long long n = __builtin_readcyclecounter(); t_sweep += n - t_mark; t_mark = n; n_items++; }
#endif

        unsigned tid_e = threadIdx.x;  // (the same for the epilogue's per-lane values)
        asm volatile("" : "+v"(tid_e));
        const int lane = (int) (tid_e & 63);
        const float norm = (float) (kSamples * a.usable);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int row4 = NQ * rowq + q;
            auto finish = [&](const f8 &Op, int pp) {
                const int row = 4 * row4 + pp;
                const bool live = row < a.rows && col < a.cols;
                const int p = min(row, a.rows - 1) * a.cols + min(col, a.cols - 1);
                f2 o[4];
#pragma unroll
                for (int k = 0; k < 4; k++) o[k] = f2{Op[2 * k], Op[2 * k + 1]};
                if (SUMS && live) {  // the pre-epilogue sums, for the tests: [batch][pixel_count][256] (an instance of its own)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        a.sums[((size_t) (2 * pair) * a.pixel_count + p) * kSamples + lane + 64 * k] = o[k].x;
                        if (2 * pair + 1 < a.batch) a.sums[((size_t) (2 * pair + 1) * a.pixel_count + p) * kSamples + lane + 64 * k] = o[k].y;
                    }
                }
                return pixel_pair_partial_exact(o, lane);
            };
            const f2 s0 = finish(O[q][0], 0), s1 = finish(O[q][1], 1), s2 = finish(O[q][2], 2), s3 = finish(O[q][3], 3);
            // the eight wave sums together, one division and one store per quad and frame pair (das_quad_kernel's epilogue)
            const float sum = wave_sum8(s0.x, s0.y, s1.x, s1.y, s2.x, s2.y, s3.x, s3.y, lane);
            const int value = kWaveSum8Value(lane >> 3), row = 4 * row4 + (value >> 1), frame = 2 * pair + (value & 1);
            if ((lane & 7) == 0 && row < a.rows && col < a.cols && frame < a.batch)
                a.power[(size_t) frame * a.pixel_count + (size_t) row * a.cols + col] = sum / norm;
        }

        if (nxt < 0) break;  // (uniform)
        __syncthreads();     // (the mailbox was written before the epilogue)
        cur = nxt;
        pair = pair_next;
        tile = tile_next;
        nxt = __builtin_amdgcn_readfirstlane(mail[0]);
        __syncthreads();  // (everybody has read the mailbox before thread 0 writes it again)
    }
#ifdef AWPU_TUNING_BUILD
    if (a.debug_out && threadIdx.x == 0) {  // the workgroup's timeline: which CU it ran on, when, how many items, and the share of the sweep block
        unsigned hw_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        unsigned long long *o = a.debug_out + 8 * (size_t) blockIdx.x;
        o[0] = rt_begin;
        o[1] = __builtin_amdgcn_s_memrealtime();
        o[2] = (unsigned long long) t_other;
        o[3] = (unsigned long long) t_sweep;
        o[4] = (unsigned long long) (__builtin_readcyclecounter() - t_mark);
        o[5] = ((unsigned long long) xcc << 32) | hw_id;
        o[6] = (unsigned long long) n_items;
        o[7] = 1;
    }
#endif
}

// The synchronous one-frame host call learns that a sweep's powers are in its pinned buffer from a flag instead of from the stream's
// completion signal (the end-of-kernel release and the signal cost ~7 us more in the call; tools/microbench/done_flag.hip).  The powers
// must then be written THROUGH as they are stored: as 4-byte stores that is 10 000 acknowledged PCIe writes (+60 us, measured); here the
// 4-row x 16-column tile of a 16-wave workgroup is gathered in the LDS (its rows are dead once every wave has arrived) and leaves as four
// 64-byte stores of wave 0.  No __threadfence_system(): on this chip it writes back and invalidates the whole L2 (+35 us in the
// microbenchmark).  `holder`: this lane holds the power of row `tile_row` of the wave's column.
__device__ __forceinline__ void store_tile_and_signal(float *lds, bool holder, int tile_row, int wave, int lane, float value, float *power, int row0,
                                                      int col0, int rows, int cols, const DoneFlag &done) {
    __syncthreads();
    if (holder) lds[tile_row * 16 + wave] = value;
    __syncthreads();
    if (wave != 0) return;
    const int r = row0 + (lane >> 4), c = col0 + (lane & 15);
    if (r < rows && c < cols) __hip_atomic_store(&power[(size_t) r * cols + c], lds[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // acknowledged
    if (lane == 0 && __hip_atomic_fetch_add(done.counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull == done.target)
        __hip_atomic_store(done.flag, done.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------
// Single frames in the reference's order (round 5): the halves form of the {next, d} layout.  MIMOWorker::update sweeps ONE
// 256-sample block per call (worker.h:212-224, mimo.cpp:97-151); das_exact_nd_kernel would sweep it as a pair with itself, half its
// packed lanes idle.  Here the two packed lanes are the two HALVES of the block (samples i and i + 128), as in das_quadh_kernel:
// element t of a mic's row = { X[t+1], X[t+129], X[t] - X[t+1], X[t+128] - X[t+129] }, lane l owns samples l and l + 64 of either half
// (register pair k of a pixel = out[l + 64 k], out[128 + l + 64 k]), an item = 2 ds_read_b128 + 2 x (v_pk_fma_f32, v_pk_add_f32):
// the three operations of delay.cpp:19-25 per sample in its order, mics in table order -- the pre-epilogue sums are the
// reference's bits, and the powers equal das_exact_nd_kernel's bit for bit (the same per-lane sums, the same reduction tree).
// STATIONARY (one array: every active mic's row fits the LDS): the workgroup forms the elements itself from the caller's frame
// (global loads -> registers -> LDS, no pre-pass, no chunks, nothing staged twice); otherwise rows packed by pack_ndh_kernel, the
// item block refilling the other image chunk by chunk.  Tile = 4 rows x 16 NQ columns; grid = frames x tiles.
// ---------------------------------------------------------------------------------------
__global__ void pack_ndh_kernel(const float *frames, int n_streams, int pitch, int wstart, const int32_t *index, int usable,
                                const float *gain, int wh, float *packed) {
    const int frame = blockIdx.y, s = blockIdx.x, rows_out = gridDim.x;
    f4 *dst = (f4 *) packed + ((size_t) frame * rows_out + s) * wh;
    if (s >= usable) {
        for (int t = threadIdx.x; t < wh; t += blockDim.x) dst[t] = f4{0.0f, 0.0f, 0.0f, 0.0f};
        return;
    }
    // (index == null: the identity list -- the common case -- spares the look-up, a dependent round trip in a pass that is all latency)
    const float *x = frames + ((size_t) frame * n_streams + (index ? index[s] : s)) * pitch + wstart;
    const float gm = gain ? gain[s] : 1.0f;
    for (int t = threadIdx.x; t < wh; t += blockDim.x) {  // (t + 129 <= window - 1: inside the stream's history)
        const f2 lo = next_and_difference(x[t], x[t + 1], gm), hi = next_and_difference(x[t + 128], x[t + 129], gm);
        dst[t] = f4{lo.x, hi.x, lo.y, hi.y};
    }
}

// a lane's share of sum MA^2 of one pixel (mimo.cpp:131-137) from out[] in sample order o[k] = out[l + 64 k]: the scalar twin of
// pixel_pair_partial_exact -- the same expressions in the same order, so a frame swept alone gives the bits it gives in a pair
__device__ __forceinline__ float pixel_partial_exact(const float (&o)[4], int lane) {
    float sum = 0.0f;
    float dn = wave_rotate1<kDppWaveRol1>(o[0]);
    float up_before = wave_rotate1<kDppWaveRor1>(o[0]);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float dn_after = k < 3 ? wave_rotate1<kDppWaveRol1>(o[k < 3 ? k + 1 : 3]) : dn;
        const float up = wave_rotate1<kDppWaveRor1>(o[k]);
        const float next = lane == 63 ? dn_after : dn;
        const float prev = lane == 0 ? up_before : up;
        const int i = lane + 64 * k;
        const float ma = o[k] * 0.5f - 0.25f * (next + prev);  // mimo.cpp:132-134
        if (i >= 1 && i <= kSamples - 2) sum += ma * ma;
        dn = dn_after;
        up_before = up;
    }
    return sum;
}

template <int NQ, bool STATIONARY>
__global__ __launch_bounds__(1024) void das_exact_ndh_kernel(ExactNdhArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = 16, kThreads = NW * 64, BUF = kFastLdsBytes;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned) (unsigned long long) (const __attribute__((address_space(3))) char *) lds;
    const int frame = blockIdx.x / a.tiles, tile = blockIdx.x - frame * a.tiles;
    constexpr int tile_cols = NW * NQ;
    const int tiles_per_row4 = (a.cols + tile_cols - 1) / tile_cols;
    const int row4 = tile / tiles_per_row4;
    const int col0 = (tile - row4 * tiles_per_row4) * tile_cols + wave * NQ;  // the wave's NQ quads: columns col0 .. col0 + NQ - 1
    const int groups_total = a.usable_pad >> 2;
    const QuadEntry *quad_lut = a.lut + ((size_t) row4 * a.lut_cols + col0) * groups_total * 16;  // (the table's columns: whole tiles of 32)
    const size_t row_floats = (size_t) a.wh * 4;

    int n_chunks = 1, first_mics = a.usable_pad, last_mics = a.usable_pad;
    const float *frame_rows = nullptr;
    if constexpr (STATIONARY) {
        // every active mic's row, formed here: waves take rows s = wave, wave + 16, ...; per pass kUnits (row, 64-element stretch) units =
        // 48 global loads in flight per lane -- one array's rows (4 rows x 3 stretches per wave) in ONE pass: a frame that has just
        // arrived by DMA is cold in every cache, and three passes of 16 loads were three memory latencies -- then the differences and
        // the ds_write_b128s
        const float *frame_base = a.frames + (size_t) frame * a.n_streams * a.pitch + a.wstart;
        const int stretches = (a.wh + 63) >> 6;
        const int units = ((a.usable_pad - wave + NW - 1) / NW) * stretches;
        int r = 0, st = 0;  // the unit about to be taken: row wave + 16 r, elements 64 st .. 64 st + 63 (wave-uniform counters)
        constexpr int kUnits = 12;
        for (int u0 = 0; u0 < units; u0 += kUnits) {
            float c0[kUnits], n0[kUnits], c1[kUnits], n1[kUnits], gm[kUnits];
            int slot[kUnits];
#pragma unroll
            for (int j = 0; j < kUnits; j++) {
                const int s = wave + NW * r, t = 64 * st + lane;
                const bool on = u0 + j < units && t < a.wh;
                slot[j] = on ? s * a.wh + t : -1;
                c0[j] = n0[j] = c1[j] = n1[j] = 0.0f;  // padding mics: silence
                gm[j] = 1.0f;
                if (on && s < a.usable) {
                    const float *x = frame_base + (size_t) (a.identity ? s : a.index[s]) * a.pitch + t;
                    c0[j] = x[0], n0[j] = x[1], c1[j] = x[128], n1[j] = x[129];
                    if (a.gain) gm[j] = a.gain[s];
                }
                if (++st == stretches) st = 0, r++;
            }
#pragma unroll
            for (int j = 0; j < kUnits; j++) {
                if (slot[j] < 0) continue;
                const f2 lo = next_and_difference(c0[j], n0[j], gm[j]), hi = next_and_difference(c1[j], n1[j], gm[j]);
                ((f4 *) lds)[slot[j]] = f4{lo.x, hi.x, lo.y, hi.y};
            }
        }
        __syncthreads();
#ifdef AWPU_NDH_STAGE_ONLY  // timing experiment (tools/build_variant.sh): what the launch and the staging cost without the sweep
        if (lds[threadIdx.x] != 12345.678f) return;
#endif
    } else {
        constexpr int kPieces = (BUF + kThreads * 16 - 1) / (kThreads * 16);
        frame_rows = a.packed + (size_t) frame * a.usable_pad * row_floats;
        n_chunks = (a.usable_pad + a.chunk - 1) / a.chunk;
        first_mics = min(a.chunk, a.usable_pad);
        last_mics = a.usable_pad - (n_chunks - 1) * a.chunk;
        const int n_pieces = (int) ((size_t) first_mics * row_floats / 4);
#pragma unroll
        for (int k = 0; k < kPieces; k++) {  // chunk 0 into image 0 (the block refills from chunk 1 on)
            const int piece = threadIdx.x + k * kThreads;
            if (piece < n_pieces) {
                float *dst = lds + (wave * 64 + k * kThreads) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (frame_rows + (size_t) piece * 4),
                                                 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    f4 O[NQ][4];  // (float out[N_SAMPLES] = {0.0}, mimo.cpp:122: the block zeroes them itself)
    {
        unsigned lane_addr = lds_base + lane * 16;
        const int ngf = __builtin_amdgcn_readfirstlane(first_mics >> 2), ngl = __builtin_amdgcn_readfirstlane(last_mics >> 2);
        const unsigned dbf = __builtin_amdgcn_readfirstlane((unsigned) ((size_t) a.chunk * row_floats * 4));
        const unsigned dbl = __builtin_amdgcn_readfirstlane((unsigned) ((size_t) last_mics * row_floats * 4));
        const unsigned ddst = __builtin_amdgcn_readfirstlane(lds_base + BUF + wave * 1024);
        const int rank = wave >> 2;  // (age order among the waves that share a SIMD)
        const unsigned lane_bytes = threadIdx.x * 16;
        const void *isrc = uniform_ptr(STATIONARY ? (const void *) a.lut : (const void *) frame_rows);  // (one chunk: nothing is refilled)
        const unsigned *no_queue = (const unsigned *) uniform_ptr(nullptr);  // (one item per workgroup: no queue)
        unsigned no_ticket;
        if constexpr (NQ == 1 && STATIONARY) {  // (the block without refill code: one chunk)
            sweep_exact_ndh_resident1(O[0][0], O[0][1], O[0][2], O[0][3], uniform_ptr(quad_lut), ngf, ngl, __builtin_amdgcn_readfirstlane(n_chunks),
                                      lane_addr, rank, isrc, dbf, dbl, isrc, 0u, ddst, BUF, lane_bytes, no_queue, no_ticket);
        } else if constexpr (NQ == 1) {
            sweep_exact_ndh_item1(O[0][0], O[0][1], O[0][2], O[0][3], uniform_ptr(quad_lut), ngf, ngl, __builtin_amdgcn_readfirstlane(n_chunks),
                                  lane_addr, rank, isrc, dbf, dbl, isrc, 0u, ddst, BUF, lane_bytes, no_queue, no_ticket);
        } else if constexpr (STATIONARY) {
            static_assert(NQ == 2, "blocks are generated for one and two quads per wave");
            const int qstride = __builtin_amdgcn_readfirstlane(groups_total * 16 * (int) sizeof(QuadEntry));  // the next column's quad
            sweep_exact_ndh_resident2(O[0][0], O[0][1], O[0][2], O[0][3], O[1][0], O[1][1], O[1][2], O[1][3], uniform_ptr(quad_lut), qstride, ngf, ngl,
                                      __builtin_amdgcn_readfirstlane(n_chunks), lane_addr, rank, isrc, dbf, dbl, isrc, 0u, ddst, BUF, lane_bytes, no_queue,
                                      no_ticket);
        } else {
            static_assert(NQ == 2, "blocks are generated for one and two quads per wave");
            const int qstride = __builtin_amdgcn_readfirstlane(groups_total * 16 * (int) sizeof(QuadEntry));  // the next column's quad
            sweep_exact_ndh_item2(O[0][0], O[0][1], O[0][2], O[0][3], O[1][0], O[1][1], O[1][2], O[1][3], uniform_ptr(quad_lut), qstride, ngf, ngl,
                                  __builtin_amdgcn_readfirstlane(n_chunks), lane_addr, rank, isrc, dbf, dbl, isrc, 0u, ddst, BUF, lane_bytes, no_queue,
                                  no_ticket);
        }
    }

    const float norm = (float) (kSamples * a.usable);
    float part[8];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int pp = 0; pp < 4; pp++) {
            part[4 * q + pp] = 0.0f;
            if (q >= NQ) continue;
            const int row = 4 * row4 + pp, col = col0 + q;
            const bool live = row < a.rows && col < a.cols;
            const f4 &Op = O[q < NQ ? q : 0][pp];
            const float o[4] = {Op[0], Op[2], Op[1], Op[3]};  // sample order: l, 64 + l, 128 + l, 192 + l
            if (a.sums && live) {
                const size_t p = (size_t) row * a.cols + col;
#pragma unroll
                for (int k = 0; k < 4; k++) a.sums[((size_t) frame * a.pixel_count + p) * kSamples + lane + 64 * k] = o[k];
            }
            part[4 * q + pp] = pixel_partial_exact(o, lane);
        }
    // das_exact_nd_kernel's reduction tree (wave_sum8 of eight per-lane sums) whatever NQ: the same bits as a frame swept in a pair
    const float total = wave_sum8(part[0], part[1], part[2], part[3], part[4], part[5], part[6], part[7], lane);
    const int value = kWaveSum8Value(lane >> 3), row = 4 * row4 + (value & 3), col = col0 + (value >> 2);
    if constexpr (STATIONARY && NQ == 1) {
        if (a.done.flag) {  // (uniform) the one-frame host call: powers into pinned memory, completion by flag
            store_tile_and_signal(lds, (lane & 7) == 0 && (value >> 2) < NQ, value & 3, wave, lane, total / norm, a.power + (size_t) frame * a.pixel_count,
                                  4 * row4, (tile - row4 * tiles_per_row4) * tile_cols, a.rows, a.cols, a.done);
            return;
        }
    }
    if ((lane & 7) == 0 && (value >> 2) < NQ && row < a.rows && col < a.cols)
        a.power[(size_t) frame * a.pixel_count + (size_t) row * a.cols + col] = total / norm;
}

// ---------------------------------------------------------------------------------------
// Single frames in the reference's order, ONE PIXEL PER WAVE (round 5): das_exact_ndh_kernel's layout, table, arithmetic and
// epilogue for grids that have too few quads to give every SIMD more than one wave (c2: 64 x 64 pixels = 1024 quads on 1024 SIMDs).
// A quad block's trip is ~150 instructions issued by one wave, 64 of them arithmetic; with nothing else to run beside it the
// frame takes as long as that wave's instruction issue (profiles/r05_single_frame_ablation.txt: 58 of c2's 69 us remain with NO
// arithmetic at all).  Here a wave owns one pixel: no sharing of reads, hence no compare tree -- sweep_exact_ndp_item
// (tools/gen_trip_asm.py: block_exact_solo), ~45 instructions per trip of which 16 are arithmetic -- and four times the waves: a
// workgroup is 4 rows x 4 columns (16 waves; wave = 4 x column + row), 16 pixels per CU when the grid fits one round of workgroups.
// The same per-lane sums and the same reduction tree as every other reference-order kernel: the same bits.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void das_exact_ndp_kernel(ExactNdhArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int kThreads = 1024, BUF = kFastLdsBytes;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned) (unsigned long long) (const __attribute__((address_space(3))) char *) lds;
    const int frame = blockIdx.x / a.tiles, tile = blockIdx.x - frame * a.tiles;
    const int tiles_per_row4 = (a.cols + 3) >> 2;
    const int row4 = tile / tiles_per_row4;
    const int col = (tile - row4 * tiles_per_row4) * 4 + (wave >> 2), pp = wave & 3;  // (col < the table's columns: padded to 32)
    const int groups_total = a.usable_pad >> 2;
    const QuadEntry *pixel_lut = a.lut + ((size_t) row4 * a.lut_cols + col) * groups_total * 16 + pp * 4;  // 32 bytes of every 128-byte group
    const size_t row_floats = (size_t) a.wh * 4;

    constexpr int kPieces = (BUF + kThreads * 16 - 1) / (kThreads * 16);
    const float *frame_rows = a.packed + (size_t) frame * a.usable_pad * row_floats;
    const int n_chunks = (a.usable_pad + a.chunk - 1) / a.chunk;
    const int first_mics = min(a.chunk, a.usable_pad);
    const int last_mics = a.usable_pad - (n_chunks - 1) * a.chunk;
    const int n_pieces = (int) ((size_t) first_mics * row_floats / 4);
#pragma unroll
    for (int k = 0; k < kPieces; k++) {  // chunk 0 into image 0 (the block refills from chunk 1 on)
        const int piece = threadIdx.x + k * kThreads;
        if (piece < n_pieces) {
            float *dst = lds + (wave * 64 + k * kThreads) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (frame_rows + (size_t) piece * 4),
                                             (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
        }
    }
    // (no wait, no barrier here: the block has both, behind its first table loads)

    f4 O;  // (float out[N_SAMPLES] = {0.0}, mimo.cpp:122: the block zeroes it itself)
    {
        unsigned lane_addr = lds_base + lane * 16;
        const int ngf = __builtin_amdgcn_readfirstlane(first_mics >> 2), ngl = __builtin_amdgcn_readfirstlane(last_mics >> 2);
        const unsigned dbf = __builtin_amdgcn_readfirstlane((unsigned) ((size_t) a.chunk * row_floats * 4));
        const unsigned dbl = __builtin_amdgcn_readfirstlane((unsigned) ((size_t) last_mics * row_floats * 4));
        const unsigned ddst = __builtin_amdgcn_readfirstlane(lds_base + BUF + wave * 1024);
        const unsigned lane_bytes = threadIdx.x * 16;
        sweep_exact_ndp_item(O, uniform_ptr(pixel_lut), ngf, ngl, __builtin_amdgcn_readfirstlane(n_chunks), lane_addr, uniform_ptr(frame_rows), dbf,
                             dbl, ddst, BUF, lane_bytes);
    }

    const int row = 4 * row4 + pp;
    const bool live = row < a.rows && col < a.cols;
    const float o[4] = {O[0], O[2], O[1], O[3]};  // sample order: l, 64 + l, 128 + l, 192 + l
    if (a.sums && live) {
        const size_t p = (size_t) row * a.cols + col;
#pragma unroll
        for (int k = 0; k < 4; k++) a.sums[((size_t) frame * a.pixel_count + p) * kSamples + lane + 64 * k] = o[k];
    }
    // the other kernels' reduction tree (wave_sum8; a value's tree does not depend on which of the eight places it takes)
    const float total = wave_sum8(pixel_partial_exact(o, lane), 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, lane);
    if (lane == 0 && live) a.power[(size_t) frame * a.pixel_count + (size_t) row * a.cols + col] = total / (float) (kSamples * a.usable);
}

// ---------------------------------------------------------------------------------------
// FIR8, four-plane frame-pair layout (the default for FIR8 batches): das_fir8_pair_kernel reads 32 LDS elements for
// 32 FMAs, and the LDS array -- one per CU, 2 cycles per ds_read_b64, shared by four SIMDs that each want a
// v_pk_fma_f32 every 4 cycles -- then holds the sweep at half the VALU rate.  Here a lane owns four CONSECUTIVE
// outputs n = 4l..4l+3, so that the eleven samples X[e+4l .. e+4l+10] feed its 32 FMAs: 11 reads per item.  For
// consecutive lanes to read consecutive 8-byte elements (conflict-free) whatever the delay, pack_planes_kernel
// stores a row as four planes (sample i in plane i % 4, index i / 4); the 64-byte table entry carries the byte
// address of each rotated plane and the item's eight coefficients.  Inner loop: sweep_fir8_planes (generated,
// tools/gen_trip_asm.py: block_fir8).  Chunk pipeline, grid and epilogue arithmetic as in das_fir8_pair_kernel.
// ---------------------------------------------------------------------------------------
__global__ void pack_planes_kernel(const float *frames, int n_streams, int hist, int wstart, const int32_t *index,
                                   const float *gain, int wp, int batch, float *packed) {
    const int pair = blockIdx.y, s = blockIdx.x, usable = gridDim.x, plane = wp >> 2;
    f2 *dst = (f2 *) packed + ((size_t) pair * usable + s) * wp;
    const int fa = min(2 * pair, batch - 1), fb = min(2 * pair + 1, batch - 1);
    const float *xa = frames + ((size_t) fa * n_streams + index[s]) * hist + wstart;
    const float *xb = frames + ((size_t) fb * n_streams + index[s]) * hist + wstart;
    const float gm = gain ? gain[s] : 1.0f;
    const int valid = min(wp, hist - wstart);
    for (int t = threadIdx.x; t < wp; t += blockDim.x)
        dst[(t & 3) * plane + (t >> 2)] = t < valid ? f2{xa[t] * gm, xb[t] * gm} : f2{0.0f, 0.0f};
}

// One dword per (pixel, mic): bits 0..17 the LDS byte offset, in the chunk's image, of X[off] (its plane, its index),
// bits 18..19 that plane, bits 20..26 the coefficient row (delay.cpp:32-33; row 101 = zeros, for padding entries);
// the block derives the other three plane addresses and fetches the coefficients from `coeffs` (block_fir8).
typedef uint32_t FirPlaneEntry;
__host__ __device__ constexpr uint32_t fir_plane_entry(uint32_t addr, uint32_t plane, uint32_t k) {
    return (addr & 0x3ffffu) | ((plane & 3u) << 18) | ((k & 0x7fu) << 20);
}

// lut: [pixel][usable_pad] entries (+ 4 spare), usable_pad and the chunk multiples of 4 (null entries: coefficient
// row 101 on row 0); coeffs [128][8]: the caller's 101 rows, then zeros.
template <int VAR>
__global__ __launch_bounds__(1024, 4) void das_fir8_plane_kernel(PairArgs a, const FirPlaneEntry *lut, const float *coeffs, int n_tiles,
                                                                 int pair_group) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = 16, PPW = 4, kThreads = NW * 64, BUF = kFastLdsBytes;
    constexpr int kPieces = (BUF + kThreads * 16 - 1) / (kThreads * 16);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned) (unsigned long long) (const __attribute__((address_space(3))) char *) lds;
    // Item order as in das_quad_kernel: the (frame pair, tile) items, ordered (pair group, tile, pair), are cut into 8
    // contiguous runs, one per XCD (blockIdx & 7: round-robin placement, assumed for speed only), so that the workgroups an
    // XCD runs side by side sweep `pair_group` frame pairs x consecutive tiles: the pairs' packed samples stay in that
    // XCD's L2 instead of 32 different pairs streaming through it.
    const int n_pairs = (a.batch + 1) / 2;
    const int total = n_pairs * n_tiles;
    const int per_xcd = (total + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int) (blockIdx.x >> 3) >= per_xcd || item >= total) return;  // (uniform for the workgroup)
    int pair, tile;
    {
        const int full_items = (n_pairs / pair_group) * pair_group * n_tiles;  // items of whole pair groups
        const int ga = item < full_items ? pair_group : n_pairs % pair_group;  // the last group may be smaller
        const int rem = item < full_items ? item : item - full_items;
        const int grp = rem / (n_tiles * ga), in = rem - grp * n_tiles * ga;
        tile = in / ga;
        pair = (item < full_items ? grp * pair_group : n_pairs - ga) + (in - tile * ga);
    }
    // Which pixels this wave sweeps.  With the grid's row length known (a.cols > 0: set by the host only together with the
    // static plane pitch) a workgroup takes 4 rows x 16 columns and a wave the four VERTICALLY adjacent pixels of one
    // column, whose integer delays mostly coincide (sweep_fir8_planes_shared reuses the samples); otherwise four
    // consecutive pixels.  Slots past the grid repeat a pixel inside it and are not stored.
    int pix[PPW];
    bool live[PPW];
    if (a.cols > 0) {
        const int tiles_per_row4 = (a.cols + NW - 1) / NW, rows = a.pixel_count / a.cols;
        const int row4 = tile / tiles_per_row4, col = (tile - row4 * tiles_per_row4) * NW + wave;
#pragma unroll
        for (int q = 0; q < PPW; q++) {
            live[q] = 4 * row4 + q < rows && col < a.cols;
            pix[q] = min(4 * row4 + q, rows - 1) * a.cols + min(col, a.cols - 1);
        }
    } else {
#pragma unroll
        for (int q = 0; q < PPW; q++) {
            live[q] = (tile * NW + wave) * PPW + q < a.pixel_count;
            pix[q] = min((tile * NW + wave) * PPW + q, a.pixel_count - 1);
        }
    }
    const size_t row_floats = (size_t) a.wp * 2;
    const float *pair_base = a.packed + (size_t) pair * a.usable * row_floats;
    const unsigned lane_bytes = threadIdx.x * 16;
    auto dma_chunk = [&](int m0, int mc, int buf) {
        const unsigned n_bytes = (unsigned) ((size_t) mc * row_floats * 4);
        const char *src = (const char *) (pair_base + (size_t) m0 * row_floats);
#pragma unroll
        for (int k = 0; k < kPieces; k++) {
            if (lane_bytes + k * kThreads * 16 < n_bytes) {
                const char *base = (const char *) uniform_ptr(src + k * kThreads * 16);
                float *dst = lds + buf * (BUF / 4) + (wave * 64 + k * kThreads) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (base + lane_bytes),
                                                 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
            }
        }
    };
    f8 acc[PPW];  // acc[pp][2o], acc[pp][2o+1] = out[4l + o] of the pair's two frames
#pragma unroll
    for (int pp = 0; pp < PPW; pp++) acc[pp] = f8{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};

    const int n_chunks = (a.usable + a.chunk - 1) / a.chunk;
    dma_chunk(0, min(a.chunk, a.usable), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = 0; c < n_chunks; c++) {
        const int m0 = c * a.chunk;
        const int mc = __builtin_amdgcn_readfirstlane(min(a.chunk, a.usable - m0));
        const int buf = c & 1;
        if (c + 1 < n_chunks) dma_chunk(m0 + a.chunk, min(a.chunk, a.usable - m0 - a.chunk), buf ^ 1);
        const unsigned lane_addr = lds_base + buf * BUF + lane * 8;
        const FirPlaneEntry *row[PPW];
#pragma unroll
        for (int pp = 0; pp < PPW; pp++) row[pp] = (const FirPlaneEntry *) uniform_ptr(lut + (size_t) pix[pp] * a.usable_pad + m0);
        // the block also touches the next chunk's entries of its pixels (256 bytes from the chunk's end on: 64 entries)
        const unsigned pfoff = (unsigned) a.chunk * 4u + lane * 4u;
        const int pfn = __builtin_amdgcn_readfirstlane(c + 1 < n_chunks ? 1 : 0);
        const unsigned plane_bytes = (unsigned) a.wp * 2u;  // a staged row is wp 8-byte elements in four planes
        const void *coef = uniform_ptr(coeffs);
#ifdef AWPU_QUAD_VARIANTS
        if constexpr (VAR == 1) sweep_fir8_planes_v1(acc[0], acc[1], acc[2], acc[3], row[0], row[1], row[2], row[3], (mc + 3) / 4, lane_addr, coef, plane_bytes, pfoff, pfn);
        else if constexpr (VAR == 2) sweep_fir8_planes_v2(acc[0], acc[1], acc[2], acc[3], row[0], row[1], row[2], row[3], (mc + 3) / 4, lane_addr, coef, plane_bytes, pfoff, pfn);
        else if constexpr (VAR == 3) sweep_fir8_planes(acc[0], acc[1], acc[2], acc[3], row[0], row[1], row[2], row[3], (mc + 3) / 4, lane_addr, coef, plane_bytes, pfoff, 0);
        else
#endif
        if (a.cols > 0)  // (uniform) vertical pixel quads: samples shared between pixels with the same integer delay
            sweep_fir8_planes_shared(acc[0], acc[1], acc[2], acc[3], row[0], row[1], row[2], row[3], mc, lane_addr, coef, pfoff, pfn, wave >> 2);
        else if (plane_bytes == kFirStaticPlaneBytes)  // (uniform: rows staged at the pitch the one-address block is generated for, fir8_plane_plan)
            sweep_fir8_planes_static(acc[0], acc[1], acc[2], acc[3], row[0], row[1], row[2], row[3], (mc + 3) / 4, lane_addr, coef, plane_bytes, pfoff, pfn, wave >> 2);
        else
        sweep_fir8_planes(acc[0], acc[1], acc[2], acc[3], row[0], row[1], row[2], row[3], (mc + 3) / 4, lane_addr, coef, plane_bytes, pfoff, pfn, wave >> 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // epilogue, mimo.cpp:131-137, on consecutive outputs: the neighbours of out[4l+o] are in this lane but for the
    // first's left one (lane l-1) and the last's right one (lane l+1)
    const float norm = (float) (kSamples * a.usable);
#pragma unroll
    for (int pp = 0; pp < PPW; pp++) {
        f2 o[6];
#pragma unroll
        for (int k = 0; k < 4; k++) o[k + 1] = f2{acc[pp][2 * k], acc[pp][2 * k + 1]};
        o[0] = wave_rotate<kDppWaveRor1>(o[4]);  // (lane 0's wrap-around value belongs to sample 0, which is not summed)
        o[5] = wave_rotate<kDppWaveRol1>(o[1]);  // (lane 63's to sample 255)
        f2 sum = f2{0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = 4 * lane + k;
            const f2 ma = o[k + 1] * 0.5f - 0.25f * (o[k + 2] + o[k]);
            if (i >= 1 && i <= kSamples - 2) sum += ma * ma;
        }
        sum.x = wave_sum(sum.x);
        sum.y = wave_sum(sum.y);
        const int p = pix[pp];
        if (lane == 0 && live[pp]) {
            a.power[(size_t) (2 * pair) * a.pixel_count + p] = sum.x / norm;
            if (2 * pair + 1 < a.batch) a.power[(size_t) (2 * pair + 1) * a.pixel_count + p] = sum.y / norm;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Quad shape (the default for batches on grids whose row length is known and whose neighbouring rows mostly
// share their integer delays): the frame-pair layout and chunk pipeline of das_pair_kernel, with the
// arithmetic rearranged so that pixels share work and not only sample reads:
//     out_p[i] = sum_m f X[o+i] + (1-f) X[o+i+1] = A_p[i] - A_p[i+1] + (S_p[i] + S_p[i+1]) / 2,
//     A_p[j] = sum_m (f_pm - 1/2) X_m[o_pm + j],   S_p[j] = sum_m X_m[o_pm + j]   (j = 0..256)
// (the weights centred on zero keep A_p an incoherent sum even where the beam adds up coherently, so its rounding
// errors stay small beside S_p, which is a plain sum).
// S_p does not depend on the fractions, so pixels whose INTEGER delays coincide for a mic share that mic's term
// of it.  A wave sweeps four vertically adjacent pixels (rows 4r..4r+3 of one grid column; arrays are wider
// than tall, so the delay changes least between vertical neighbours); the second is the reference: its samples
// feed T = S_ref (one packed add per register) and the A of every pixel whose entry carries the same LDS
// address (one packed FMA per register).  A pixel that differs for this mic reads its own samples and pays
// three instructions per register (A_p += f x_p; V_p += x_p; V_p -= x_ref; S_p = T + V_p).  Per mic and quad
// that is 20 packed VALU instructions when all four coincide, +8 per pixel that does not, against 32 (+2
// address adds) in das_pair_kernel; the epilogue turns (A, S) into the (S/2 + A, S/2 - A) pair finish_pixel_pair
// takes.  Differs from the other fast kernels by fp32 rounding only (measured 2e-6 of the reference).
// Table: 8-byte entries (f, address), quad-major -- [quad][group of 4 mics][pixel][mic] -- so that a trip's
// entries are one 128-byte line; pixels past the grid carry weight 0 and their neighbour's address, padding
// mics weight 0 and the address of a zero row (pack_pairs_kernel writes those).
// Dispatch: a 1-D grid whose blocks are dealt to the 8 XCDs in contiguous runs of work items ordered
// (pair group, tile, pair): at any moment an XCD's 32 workgroups sweep `pair_group` frame pairs x 32/pair_group
// tiles, so the pairs' samples stay in that XCD's 4 MiB L2 while it walks the table once per pair group.
// (Which XCD a block lands on is the hardware's business: the mapping only assumes round-robin for speed.)
// ---------------------------------------------------------------------------------------
template <bool DIAG, int VAR>
__global__ __launch_bounds__(1024, 4) void das_quad_kernel(QuadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = 16, kThreads = NW * 64, BUF = kFastLdsBytes;
    constexpr int kPieces = (BUF + kThreads * 16 - 1) / (kThreads * 16);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned) (unsigned long long) (const __attribute__((address_space(3))) char *) lds;

    // ---- which work items (frame pair, tile) this workgroup sweeps: the item space, ordered (pair group, tile,
    // pair), is cut into 8 contiguous runs, one per XCD (blockIdx & 7: round-robin placement, assumed for speed
    // only); the workgroups of an XCD walk their run side by side, so that at any moment they work on consecutive
    // items.  A workgroup is persistent: the chunk pipeline runs on across its items (the first chunk of the next
    // item lands while the last chunk of this one is swept), so only its very first chunk is waited for.
    const int total = a.n_pairs * a.tiles;
    const int per_xcd = (total + 7) >> 3;
    const int wgs_per_xcd = gridDim.x >> 3;
    const int run_begin = (blockIdx.x & 7) * per_xcd, run_end = min(total, run_begin + per_xcd);
    const int full_items = (a.n_pairs / a.pair_group) * a.pair_group * a.tiles;  // items of whole pair groups
    auto decode = [&](int item, int &pair, int &tile) {
        const int ga = item < full_items ? a.pair_group : a.n_pairs % a.pair_group;  // the last group may be smaller
        const int rem = item < full_items ? item : item - full_items;
        const int grp = rem / (a.tiles * ga), in = rem - grp * a.tiles * ga;
        tile = in / ga;
        pair = (item < full_items ? grp * a.pair_group : a.n_pairs - ga) + (in - tile * ga);
    };
    // a.queue (the production launch): persistent workgroups that take their items from the queues of das_exact_nd_kernel (NdQueues:
    // one per XCD -- the run above -- whose last eighth is common to the chip) instead of every (gridDim / 8)-th item of their XCD's
    // run: the XCDs of a chip finish equal shares 6 % apart.  The item after the next is taken inside the sweep block (qptr).
    const bool queued = !DIAG && VAR == 0 && a.queue != nullptr;
    const NdQueues Q{a.queue, per_xcd, total, a.tail};
    int *mail = (int *) (lds + 2 * (BUF / 4));  // two ints behind the images
    int xcd = (int) (blockIdx.x & 7);
    int item, nxt = -1;
    if (queued) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcd = (int) (xcc & 7u);
        if (threadIdx.x == 0) {
            const int first = Q.take(xcd);
            mail[0] = first;
            mail[1] = first >= 0 ? Q.take(xcd) : -1;
        }
        __syncthreads();
        item = __builtin_amdgcn_readfirstlane(mail[0]);
        nxt = __builtin_amdgcn_readfirstlane(mail[1]);
        if (item < 0) return;  // (uniform for the workgroup)
        __syncthreads();
    } else {
        item = run_begin + (blockIdx.x >> 3);
        if (item >= run_end) return;  // (uniform for the workgroup)
    }

    const int tiles_per_row4 = (a.cols + NW - 1) / NW;
    const int groups_total = a.usable_pad >> 2;
    const size_t row_floats = (size_t) a.wp * 2;
    const int n_chunks = (a.usable + a.chunk - 1) / a.chunk;
    const int rank = wave >> 2;  // age order of this wave among the four that share its SIMD
    const float norm = (float) (kSamples * a.usable);
    // which (pixel of the quad, frame of the pair) this lane's 8-lane group holds after wave_sum8 of (s0.x, s0.y, s1.x, ... s3.y)
    const int out_value = kWaveSum8Value(lane >> 3), out_pp = out_value >> 1, out_frame = out_value & 1;
    auto chunk_mics = [&](int m0) { return (min(a.chunk, a.usable - m0) + 3) & ~3; };
    // one chunk = mc4 rows of a pair (whole groups: the padding rows are zero), contiguous in HBM from `src` (a
    // wave-uniform pointer: the transfers take it as a scalar base plus one per-lane byte offset)
    const unsigned lane_bytes = threadIdx.x * 16;
    auto dma_chunk = [&](const float *src, int mc4, int buf) {
        const unsigned n_bytes = (unsigned) ((size_t) mc4 * row_floats * 4);
#pragma unroll
        for (int k = 0; k < kPieces; k++) {
            if (lane_bytes + k * kThreads * 16 < n_bytes) {
                const char *base = (const char *) uniform_ptr((const char *) src + k * kThreads * 16);
                float *dst = lds + buf * (BUF / 4) + (wave * 64 + k * kThreads) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (base + lane_bytes),
                                                 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
            }
        }
    };

    unsigned t_wait = 0, t_all = 0, n_blocks = 0;
    const long long t_begin = __builtin_readcyclecounter();
    const unsigned long long rt_begin = DIAG ? __builtin_amdgcn_s_memrealtime() : 0ull;  // 100 MHz: the in-kernel clock's yardstick
    unsigned t_ph[5] = {0, 0, 0, 0, 0};
    auto stamp = [&](int k, long long &t) {
        if (DIAG) {
            const long long n = __builtin_readcyclecounter();
            t_ph[k] += (unsigned) (n - t);
            t = n;
        }
    };

    int pair, tile;
    decode(item, pair, tile);
    dma_chunk(a.packed + (size_t) pair * a.usable_pad * row_floats, chunk_mics(0), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned step = 0;  // chunks swept so far: chunk `step` lives in image step & 1

    for (;;) {
        // ---- this item's pixels
        const int row4 = tile / tiles_per_row4;
        const int col = (tile - row4 * tiles_per_row4) * NW + wave;
        const int quad = row4 * tiles_per_row4 * NW + col;
        // (pixels of the quad: rows 4 row4 .. 4 row4 + 3 of column `col`; slots outside the grid sweep a clamped table row --
        // build_quad_lut -- and are not stored: the epilogue below)
        const QuadEntry *quad_lut = a.lut + (size_t) quad * groups_total * 16;
        const float *pair_base = a.packed + (size_t) pair * a.usable_pad * row_floats;
        // ---- the item after it
        const int item_next = queued ? nxt : item + wgs_per_xcd;
        const bool more = queued ? nxt >= 0 : item_next < run_end;
        int pair_next = 0, tile_next = 0;
        if (more) decode(item_next, pair_next, tile_next);

        f8 A0 = {0, 0, 0, 0, 0, 0, 0, 0}, A1 = A0, A2 = A0, A3 = A0, T = A0, V0 = A0, V2 = A0, V3 = A0;
        if constexpr (!DIAG && VAR == 0) {
            // The production build sweeps the whole item in ONE block: chunk loop, refill, vmcnt wait and workgroup barrier
            // inside the asm (sweep_quad_item), so that a chunk begins on table entries that are in SGPRs already.
            const int buf = step & 1;
            unsigned lane_addr = lds_base + buf * BUF + lane * 8;
            const int ngf = __builtin_amdgcn_readfirstlane(chunk_mics(0) >> 2);
            const int ngl = __builtin_amdgcn_readfirstlane(chunk_mics((n_chunks - 1) * a.chunk) >> 2);
            const unsigned dbf = __builtin_amdgcn_readfirstlane((unsigned) ((size_t) a.chunk * row_floats * 4));
            const unsigned dbl = __builtin_amdgcn_readfirstlane((unsigned) ((size_t) ngl * 4 * row_floats * 4));
            const float *nsrc = more ? a.packed + (size_t) pair_next * a.usable_pad * row_floats : pair_base;
            const unsigned dbn = __builtin_amdgcn_readfirstlane(more ? (unsigned) ((size_t) chunk_mics(0) * row_floats * 4) : 0u);
            const unsigned ddst = __builtin_amdgcn_readfirstlane(lds_base + (buf ^ 1) * BUF + wave * 1024);
            const int delta = __builtin_amdgcn_readfirstlane(buf ? -BUF : BUF);
            unsigned ticket = 0;
            const bool asked = queued && wave == 0 && more && Q.head(xcd) > 0;
            const unsigned *qptr = (const unsigned *) uniform_ptr(asked ? a.queue + xcd : nullptr);
            sweep_quad_item(A0, A1, A2, A3, T, V0, V2, V3, uniform_ptr(quad_lut), ngf, ngl, __builtin_amdgcn_readfirstlane(n_chunks),
                            lane_addr, rank, uniform_ptr(pair_base), dbf, dbl, uniform_ptr(nsrc), dbn, ddst, delta, lane_bytes, qptr, ticket);
            step += n_chunks;
            if (queued && threadIdx.x == 0 && more) {  // the answer, into the mailbox at once (not carried through the epilogue)
                int got = Q.head(xcd) > 0 && ticket < (unsigned) Q.head(xcd) ? xcd * Q.per + (int) ticket : -1;
                if (got < 0) {  // own queue empty: the common tail, then anybody's (only near the launch's end)
                    got = Q.take_common();
                    for (int y = 1; got < 0 && y < 8; y++) got = Q.take_head((xcd + y) & 7);
                }
                mail[0] = got;
            }
        } else
        for (int c = 0; c < n_chunks; c++, step++) {
            const int m0 = c * a.chunk;
            const int mc4 = chunk_mics(m0);
            const int buf = step & 1;
            long long t = DIAG ? __builtin_readcyclecounter() : 0;
            // The stamped and the tuning builds issue the refill themselves (the production build lets the sweep block
            // do it piece by piece, below): all pieces at the head of the chunk, from every wave at once (the order
            // of round 1); or, with a.debug & 512, the waves of a SIMD taking turns, rank r sweeping r quarters of
            // the chunk first (measured: 1.8 % slower: the later refills land later and the barrier waits for them).
            const int ng = __builtin_amdgcn_readfirstlane(mc4 >> 2);
            const int g_head = (a.debug & 512) ? __builtin_amdgcn_readfirstlane((ng * rank) >> 2) : 0;
            const unsigned lane_addr = lds_base + buf * BUF + lane * 8;
            auto sweep = [&](int g0, int n) {
                const void *row = uniform_ptr(quad_lut + (size_t) ((m0 >> 2) + g0) * 16);
                if constexpr (DIAG) {
                    unsigned dw = 0, da = 0;
                    sweep_quad_sum_stamped(A0, A1, A2, A3, T, V0, V2, V3, row, n, lane_addr, rank, dw, da);
                    t_wait += dw;
                    t_all += da;
                    n_blocks++;
                } else {
#ifdef AWPU_QUAD_VARIANTS  // tuning builds: the wave-priority schemes of tools/gen_trip_asm.py side by side
                    if constexpr (VAR == 1) sweep_quad_sum_v0(A0, A1, A2, A3, T, V0, V2, V3, row, n, lane_addr, rank);
                    else if constexpr (VAR == 2) sweep_quad_sum_v3(A0, A1, A2, A3, T, V0, V2, V3, row, n, lane_addr, rank);
                    else if constexpr (VAR == 3) sweep_quad_sum_v4(A0, A1, A2, A3, T, V0, V2, V3, row, n, lane_addr, rank);
                    else
#endif
                    sweep_quad_sum(A0, A1, A2, A3, T, V0, V2, V3, row, n, lane_addr, rank);
                }
            };
            // the refill of the other image: this item's next chunk, or the next item's first
            const float *next_src = nullptr;
            int next_mc4 = 0;
            if (c + 1 < n_chunks) {
                next_src = pair_base + (size_t) (m0 + a.chunk) * row_floats;
                next_mc4 = chunk_mics(m0 + a.chunk);
            } else if (item_next < run_end) {
                next_src = a.packed + (size_t) pair_next * a.usable_pad * row_floats;
                next_mc4 = chunk_mics(0);
            }
            if (AWPU_DBG(a, 1)) next_mc4 = 0;
            // (the stamped and the tuning builds: one block per chunk, the refill issued here, outside the block)
            if (g_head > 0) sweep(0, g_head);
            long long t_dma = DIAG ? __builtin_readcyclecounter() : 0;
            if (next_mc4) dma_chunk(next_src, next_mc4, buf ^ 1);
            if (DIAG) {
                const long long n = __builtin_readcyclecounter();
                t_ph[0] += (unsigned) (n - t_dma);
                t += n - t_dma;  // (the sweep's share below excludes it)
            }
            if (ng - g_head > 0) sweep(g_head, ng - g_head);
            stamp(1, t);
            // (no 257th-sample pass: the rows carry pre-filtered samples, pack_one_row<true>)
            if (DIAG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stamp(2, t);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(3, t);
            if (!AWPU_DBG(a, 8)) __syncthreads();
            stamp(4, t);
        }

        // ---- this item's powers (the next item's first chunk is in its image already)
        auto partial = [&](const f8 &A, const f8 &S) {  // a lane's share of sum MA^2 of one pixel, both frames
            f2 P[8];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const f2 Ak = f2{A[2 * k], A[2 * k + 1]}, Sk = f2{S[2 * k], S[2 * k + 1]};
                P[k] = __builtin_elementwise_fma(f2{0.5f, 0.5f}, Sk, Ak);       // sum f Y      (A was accumulated with f - 1/2: see QuadEntry)
                P[4 + k] = __builtin_elementwise_fma(f2{0.5f, 0.5f}, Sk, -Ak);  // sum (1 - f) Y
            }
            return pixel_pair_partial_filtered(P, lane);
        };
        const f2 s0 = partial(A0, T + V0), s1 = partial(A1, T);
        const f8 S2 = T + V2;
        const f2 s2 = partial(A2, S2);
        const f2 s3 = partial(A3, (kQuadChain ? S2 : T) + V3);  // (the blocks keep V3 = S3 - S2: tools/gen_trip_asm.py, pixels_23_chain)
        // the eight wave sums together (wave_sum8); the 8-lane group of a lane then holds ONE (pixel, frame) of the quad, whose
        // first lane divides and stores: one vector division and one store per quad and frame pair instead of eight of each
        {
            const float total = wave_sum8(s0.x, s0.y, s1.x, s1.y, s2.x, s2.y, s3.x, s3.y, lane);
            const int row = 4 * row4 + out_pp, frame = 2 * pair + out_frame;
            if ((lane & 7) == 0 && row < a.rows && col < a.cols && frame < a.batch)
                a.power[(size_t) frame * a.pixel_count + (size_t) row * a.cols + col] = total / norm;
        }
        if (!more) break;
        item = item_next;
        pair = pair_next;
        tile = tile_next;
        if (queued) {
            __syncthreads();  // (the mailbox was written before the epilogue)
            nxt = __builtin_amdgcn_readfirstlane(mail[0]);
            __syncthreads();  // (everybody has read it before thread 0 writes it again)
        }
    }

    if (DIAG && a.debug_out && lane == 0) {
        unsigned long long *o = a.debug_out + 12 * ((size_t) blockIdx.x * NW + wave);
        o[0] = t_wait;
        o[1] = t_all;
        o[2] = (unsigned long long) (__builtin_readcyclecounter() - t_begin);
        o[3] = n_blocks;
        for (int k = 0; k < 5; k++) o[4 + k] = t_ph[k];
        o[9] = __builtin_amdgcn_s_memrealtime() - rt_begin;
    }
}

// ---------------------------------------------------------------------------------------
// Quad shape for single frames on the HALVES layout (round 3; the default for calls of one frame where the quad table
// pays): a pre-pass (pack_halves_kernel) writes, per active mic, the touched window as 8-byte elements
//     element t = (Y[wstart + t], Y[wstart + t + 128]),      Y = the pre-filtered samples of pack_one_row<true>,
// i.e. the two HALVES of the 256-sample block ride in the two lanes of every packed instruction, as the two frames of a
// pair do in das_quad_kernel.  Lane l owns samples l and l + 64 of either half: register pair 0 = (sample l, 128 + l),
// pair 1 = (64 + l, 192 + l).  Against das_quad1_kernel's layout (adjacent samples in the packed lanes, the window staged
// twice so that odd delays stay 8-byte aligned):
//   * any integer delay is 8-byte aligned: no parity copies -- a mic's row is (W - 128) x 8 bytes instead of 2 x W x 4
//     (headline: 1760 against 2784), 44 mics per chunk instead of 28, 6 chunks instead of 10;
//   * a chunk is contiguous in HBM: the refill is one linear stream that the sweep block issues itself, one 16 KiB
//     piece per trip (sweep_quad1_sum_a_dma) -- no row table, no queue of 80 LDS-DMA instructions at the CU's address
//     unit right after every barrier;
//   * pre-filtered samples: no 257th-sample pass, and the epilogue is rotation, squares, wave sum.
// The block (always-read schedule), the quad-major table and the tile geometry are das_quad1_kernel's; the table
// carries this layout's LDS addresses (slot x row bytes + (off - wstart) x 8).  grid = (frames, tiles).
// ---------------------------------------------------------------------------------------
// Epilogue of the halves-layout kernels: the un-skewed sums ARE the moving average (pre-filtered samples): MA[i] = P[i] +
// G[i+1], i = 1..254, with P = sum f Y = S/2 + A, G = sum (1 - f) Y = S/2 - A (A was accumulated with f - 1/2).  Sample
// order of a lane's four values: pair 0 low (l), pair 1 low (64 + l), pair 0 high (128 + l), pair 1 high (192 + l).
// Returns sum MA^2 over the wave (every lane).
__device__ __forceinline__ float quadh_pixel_partial(const f4 &A, const f4 &S, int lane) {
    const float Av[4] = {A[0], A[2], A[1], A[3]}, Sv[4] = {S[0], S[2], S[1], S[3]};  // in sample order
    float P[4], G[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        P[r] = __builtin_fmaf(0.5f, Sv[r], Av[r]);
        G[r] = __builtin_fmaf(0.5f, Sv[r], -Av[r]);
    }
    float sum = 0.0f;
    float rq = wave_rotate1<kDppWaveRol1>(G[0]);  // G_r one lane down; lane 63 holds G_r[0]
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float rq_next = r < 3 ? wave_rotate1<kDppWaveRol1>(G[r < 3 ? r + 1 : 3]) : rq;
        const float ma = P[r] + (lane == 63 ? rq_next : rq);  // MA[l + 64 r]
        const int i = lane + 64 * r;
        if (i >= 1 && i <= kSamples - 2) sum = __builtin_fmaf(ma, ma, sum);
        rq = rq_next;
    }
    return sum;
}

// The filtered sample pack_halves_kernel and das_quadh_stationary_kernel stage: Y[i] = X[i]/2 - (X[i+1] + X[i-1])/4 with the
// gain on every sample first; neighbours outside the history count as 0 (values that would need them are never used:
// pack_one_row).  One expression for both, so that the two paths stage the same bits.
__device__ __forceinline__ float filtered_sample(const float *x, int i, int hist, float gm) {
    if (i < 0 || i >= hist) return 0.0f;
    const float lo = i > 0 ? x[i - 1] * gm : 0.0f, hi = i + 1 < hist ? x[i + 1] * gm : 0.0f;
    return __builtin_fmaf(-0.25f, lo + hi, 0.5f * (x[i] * gm));
}

__global__ void pack_halves_kernel(const float *frames, int n_streams, int pitch, int hist, int wstart, const int32_t *index,
                                   int usable, const float *gain, int wp, float *packed) {
    const int frame = blockIdx.y, s = blockIdx.x, rows_out = gridDim.x;
    f2 *dst = (f2 *) packed + ((size_t) frame * rows_out + s) * wp;
    if (s >= usable) {  // padding rows (whole groups of four mics are swept): silence
        for (int t = threadIdx.x; t < wp; t += blockDim.x) dst[t] = f2{0.0f, 0.0f};
        return;
    }
    // (index == null: the identity list -- the common case -- spares the look-up, a dependent round trip in a pass that is
    // nothing but two of them)
    const float *x = frames + ((size_t) frame * n_streams + (index ? index[s] : s)) * pitch;
    const float gm = gain ? gain[s] : 1.0f;
    for (int t = threadIdx.x; t < wp; t += blockDim.x)
        dst[t] = f2{filtered_sample(x, wstart + t, hist, gm), filtered_sample(x, wstart + t + 128, hist, gm)};
}

template <int QPW, bool DIAG>
__global__ __launch_bounds__(1024, 4) void das_quadh_kernel(QuadhArgs a) {
    static_assert(QPW == 1 || QPW == 2, "one or two quads per wave");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = 16, kThreads = NW * 64, BUF = kFastLdsBytes;
    constexpr int kPieces = (BUF + kThreads * 16 - 1) / (kThreads * 16);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned) (unsigned long long) (const __attribute__((address_space(3))) char *) lds;
    const int frame = blockIdx.x;
    const int tile = blockIdx.y;

    const int tile_cols = NW * QPW;
    const int tiles_per_row4 = (a.cols + tile_cols - 1) / tile_cols;
    const int cols_pad = (a.cols + 15) / 16 * 16;  // the table's quads: columns padded to 16 (quad_count)
    const int row4 = tile / tiles_per_row4;
    const int col0 = (tile - row4 * tiles_per_row4) * tile_cols + wave * QPW;
    const int groups_total = a.usable_pad >> 2;
    const size_t row_floats = (size_t) a.wp * 2;
    const float *frame_base = a.packed + (size_t) frame * a.usable_pad * row_floats;
    const unsigned lane_bytes = threadIdx.x * 16;
    auto chunk_mics = [&](int m0) { return (min(a.chunk, a.usable - m0) + 3) & ~3; };
    auto dma_chunk = [&](const float *src, int mc4, int buf) {  // (first chunk, and the stamped build's refills)
        const unsigned n_bytes = (unsigned) ((size_t) mc4 * row_floats * 4);
#pragma unroll
        for (int k = 0; k < kPieces; k++) {
            if (lane_bytes + k * kThreads * 16 < n_bytes) {
                const char *base = (const char *) uniform_ptr((const char *) src + k * kThreads * 16);
                float *dst = lds + buf * (BUF / 4) + (wave * 64 + k * kThreads) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (base + lane_bytes),
                                                 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
            }
        }
    };

    f4 A0a = {0, 0, 0, 0}, A1a = A0a, A2a = A0a, A3a = A0a, Ta = A0a, V0a = A0a, V2a = A0a, V3a = A0a;
    f4 A0b = A0a, A1b = A0a, A2b = A0a, A3b = A0a, Tb = A0a, V0b = A0a, V2b = A0a, V3b = A0a;

    unsigned t_wait = 0, t_all = 0;
    const long long t_begin = __builtin_readcyclecounter();
    const int n_chunks = (a.usable + a.chunk - 1) / a.chunk;
    dma_chunk(frame_base, chunk_mics(0), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    unsigned t_ph[5] = {0, 0, 0, 0, 0};
    auto stamp = [&](int k, long long &t) {
        if (DIAG) {
            const long long n = __builtin_readcyclecounter();
            t_ph[k] += (unsigned) (n - t);
            t = n;
        }
    };
    const int rank = wave >> 2;
    for (int c = 0; c < n_chunks; c++) {
        const int m0 = c * a.chunk;
        const int mc4 = chunk_mics(m0);
        const int buf = c & 1;
        long long t = DIAG ? __builtin_readcyclecounter() : 0;
        const float *next_src = frame_base + (size_t) (m0 + a.chunk) * row_floats;
        const int next_mc4 = c + 1 < n_chunks ? chunk_mics(m0 + a.chunk) : 0;
        const unsigned lane_addr = lds_base + buf * BUF + lane * 8;
        const int ng = __builtin_amdgcn_readfirstlane(mc4 >> 2);
        {
            const void *row = uniform_ptr(a.lut + (((size_t) row4 * cols_pad + min(col0, cols_pad - 1)) * groups_total + (m0 >> 2)) * 16);
            if constexpr (DIAG) {
                if (next_mc4) dma_chunk(next_src, next_mc4, buf ^ 1);
                stamp(0, t);
                unsigned dw = 0, da = 0;
                sweep_quad1_sum_a_stamped(A0a, A1a, A2a, A3a, Ta, V0a, V2a, V3a, row, ng, lane_addr, rank, dw, da);
                t_wait += dw;
                t_all += da;
            } else {  // the block issues the refill of the other image itself, one 16 KiB piece per trip
                const unsigned dst0 = __builtin_amdgcn_readfirstlane(lds_base + (buf ^ 1) * BUF + wave * 1024);
                const unsigned n_bytes = __builtin_amdgcn_readfirstlane((unsigned) ((size_t) next_mc4 * row_floats * 4));
                const unsigned dnp = wave < kQuadDmaWaves ? (n_bytes + kQuadDmaWaves * 1024 - 1) / (kQuadDmaWaves * 1024) : 0;
                sweep_quad1_sum_a_dma(A0a, A1a, A2a, A3a, Ta, V0a, V2a, V3a, row, ng, lane_addr, rank, uniform_ptr(next_src), dst0,
                                      n_bytes, lane_bytes, dnp);
            }
        }
        if constexpr (QPW == 2) {
            const void *row = uniform_ptr(a.lut + (((size_t) row4 * cols_pad + min(col0 + 1, cols_pad - 1)) * groups_total + (m0 >> 2)) * 16);
            sweep_quad1_sum_b(A0b, A1b, A2b, A3b, Tb, V0b, V2b, V3b, row, ng, lane_addr, rank);
        }
        stamp(1, t);
        stamp(2, t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(3, t);
        __syncthreads();
        stamp(4, t);
    }

    if (DIAG && a.debug_out && lane == 0) {
        unsigned long long *o = a.debug_out + 12 * ((size_t) (blockIdx.y * gridDim.x + blockIdx.x) * NW + wave);
        o[0] = t_wait;
        o[1] = t_all;
        o[2] = (unsigned long long) (__builtin_readcyclecounter() - t_begin);
        o[3] = (unsigned long long) n_chunks;
        for (int k = 0; k < 5; k++) o[4 + k] = t_ph[k];
    }
    const float norm = (float) (kSamples * a.usable);
    // the wave sums of the quad's pixels together (wave_sum4 / wave_sum8); the first lane of the lane group that holds pixel
    // `slot` (a-quad: 0..3, b-quad: 4..7; row = slot & 3 of its quad) divides and stores
    const float p0 = quadh_pixel_partial(A0a, Ta + V0a, lane), p1 = quadh_pixel_partial(A1a, Ta, lane);
    const float p2 = quadh_pixel_partial(A2a, Ta + V2a, lane), p3 = quadh_pixel_partial(A3a, Ta + V3a, lane);
    float total;
    int slot;
    bool first;
    if constexpr (QPW == 2) {
        const float p4 = quadh_pixel_partial(A0b, Tb + V0b, lane), p5 = quadh_pixel_partial(A1b, Tb, lane);
        const float p6 = quadh_pixel_partial(A2b, Tb + V2b, lane), p7 = quadh_pixel_partial(A3b, Tb + V3b, lane);
        total = wave_sum8(p0, p1, p2, p3, p4, p5, p6, p7, lane);
        slot = kWaveSum8Value(lane >> 3);
        first = (lane & 7) == 0;
    } else {
        total = wave_sum4(p0, p1, p2, p3);
        slot = kWaveSum4Value(lane >> 4);
        first = (lane & 15) == 0;
    }
    const int col = col0 + (slot >> 2), row = 4 * row4 + (slot & 3);
    if (first && col < a.cols && row < a.rows) a.power[(size_t) frame * a.pixel_count + (size_t) row * a.cols + col] = total / norm;
}

// ---------------------------------------------------------------------------------------
// Single frames of SMALL arrays (round 4): when the halves rows of EVERY active mic fit the CU's LDS at once -- the
// reference's own configuration does: one 8x8 array, 64 mics x 158 elements x 8 bytes = 79 KiB of the 156 KiB the two
// images span -- there is nothing to chunk, and a call of one frame (MIMOWorker::update, once per 5.24 ms block,
// src/dsp/worker.h:212-224) is all latency: das_quadh_kernel behind pack_halves_kernel is two launches, a DMA wait and two
// chunks with a barrier each for 6 us of arithmetic.  Here a workgroup stages the window ITSELF -- the raw samples of every mic
// by LDS-DMA straight from the caller's frame (or the ingest ring: `pitch`), then filtered from LDS into the halves image with
// pack_halves_kernel's expression, so the image holds the same bits -- and sweeps all mics in one block per quad: one launch,
// no refill.  LDS: [usable][raw_wr] floats raw + [usable_pad][wp] elements + a 4 KiB row table <= 156 KiB.
// Table: the quad-major table with slot = mic (build_quad_lut, kQuadHalvesStationary).  grid = (frames, tiles).
// ---------------------------------------------------------------------------------------
constexpr int kQuadhsRowTableOffset = (2 * kFastLdsBytes - 4096) / 4;  // floats: the last 4 KiB hold the streams' row offsets
template <int QPW>
__global__ __launch_bounds__(1024, 4) void das_quadh_stationary_kernel(QuadhStationaryArgs a) {
    static_assert(QPW == 1 || QPW == 2, "one or two quads per wave");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // waves per workgroup = columns per tile: the host launches 16.  (Measured on the reference's 100 x 100 grid, us per frame:
    // 16 waves / 175 workgroups 19.7, 14 / 200 20.4, 12 / 225 20.1, 10 / 250 -- every CU busy, 2.5 waves per SIMD -- 21.1,
    // 8 / 325 -- two rounds -- 39.0: a workgroup's time hardly depends on its wave count, the call is a chain of latencies --
    // launch, DMA, three barriers, 64 dependent mic stages -- so what counts is the staging work per wave.)
    const int NW = (int) (blockDim.x >> 6), kThreads = (int) blockDim.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned) (unsigned long long) (const __attribute__((address_space(3))) char *) lds;
    const int frame = blockIdx.x, tile = blockIdx.y;
    const int tile_cols = NW * QPW;
    const int tiles_per_row4 = (a.cols + tile_cols - 1) / tile_cols;
    const int cols_pad = (a.cols + 15) / 16 * 16;
    const int row4 = tile / tiles_per_row4;
    const int col0 = (tile - row4 * tiles_per_row4) * tile_cols + wave * QPW;
    const int groups_total = a.usable_pad >> 2;

    // ---- stage, phase A: the raw window of every active mic, [usable][raw_wr] floats from history sample raw_begin on, by
    // LDS-DMA: one wave instruction moves 64 pieces of 16 bytes from 64 addresses into 1 KiB of LDS, nothing passes through
    // registers and all of a wave's instructions are in flight together (a register-staged loop was latency, row after row)
    const float *frame_base = a.frames + (size_t) frame * a.n_streams * a.pitch;
    float *raw = lds;
    f2 *image = (f2 *) (lds + a.image_offset);              // [usable_pad][wp] elements (Y[wstart + t], Y[wstart + t + 128])
    int32_t *row_of = (int32_t *) (lds + kQuadhsRowTableOffset);  // float offset of stream index[s] in the frame
    if (!a.identity) {  // (the default mic list 0..usable-1 needs no table: one global round trip and one barrier less)
        for (int s = threadIdx.x; s < a.usable; s += kThreads) row_of[s] = a.index[s] * a.pitch;
        __syncthreads();
    }
    const int ppr = a.raw_wr >> 2;  // 16-byte pieces per row
    const int n_pieces = a.usable * ppr;
    for (int p0 = __builtin_amdgcn_readfirstlane(wave * 64); p0 < n_pieces; p0 += kThreads) {  // (p0 stays wave-uniform)
        const int p = p0 + lane;
        if (p < n_pieces) {
            const int s = p / ppr, col = a.raw_begin + 4 * (p - s * ppr);
            if (col + 4 <= a.row_limit)  // (never past the stream's row: the last stream of the last frame ends the allocation)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (frame_base + (a.identity ? s * a.pitch : row_of[s]) + col),
                                                 (__attribute__((address_space(3))) void *) (raw + (size_t) p0 * 4), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- phase B: filter on the way from the raw rows to the halves image: pack_halves_kernel's expression (the same bits);
    // samples outside the history count as 0 (the values that would need them are never used)
    for (int s = wave; s < a.usable_pad; s += NW) {
        f2 *dst = image + (size_t) s * a.wp;
        if (s < a.usable) {
            const float *x = raw + (size_t) s * a.raw_wr - a.raw_begin;  // x[i] = history sample i of this mic
            const float gm = a.gain ? a.gain[s] : 1.0f;
            for (int t = lane; t < a.wp; t += 64)
                dst[t] = f2{filtered_sample(x, a.wstart + t, a.hist, gm), filtered_sample(x, a.wstart + t + 128, a.hist, gm)};
        } else {
            for (int t = lane; t < a.wp; t += 64) dst[t] = f2{0.0f, 0.0f};  // padding mics: silence
        }
    }
    __syncthreads();

    f4 A0a = {0, 0, 0, 0}, A1a = A0a, A2a = A0a, A3a = A0a, Ta = A0a, V0a = A0a, V2a = A0a, V3a = A0a;
    f4 A0b = A0a, A1b = A0a, A2b = A0a, A3b = A0a, Tb = A0a, V0b = A0a, V2b = A0a, V3b = A0a;
    const int rank = wave >> 2;
    const unsigned lane_addr = lds_base + a.image_offset * 4 + lane * 8;
    const int ng = __builtin_amdgcn_readfirstlane(groups_total);
    {
        const void *row = uniform_ptr(a.lut + ((size_t) row4 * cols_pad + min(col0, cols_pad - 1)) * groups_total * 16);
        sweep_quad1_sum_a(A0a, A1a, A2a, A3a, Ta, V0a, V2a, V3a, row, ng, lane_addr, rank);
    }
    if constexpr (QPW == 2) {
        const void *row = uniform_ptr(a.lut + ((size_t) row4 * cols_pad + min(col0 + 1, cols_pad - 1)) * groups_total * 16);
        sweep_quad1_sum_b(A0b, A1b, A2b, A3b, Tb, V0b, V2b, V3b, row, ng, lane_addr, rank);
    }

    const float norm = (float) (kSamples * a.usable);
    // the wave sums of the quad's pixels together (wave_sum4 / wave_sum8); the first lane of the lane group that holds pixel
    // `slot` (a-quad: 0..3, b-quad: 4..7; row = slot & 3 of its quad) divides and stores
    const float p0 = quadh_pixel_partial(A0a, Ta + V0a, lane), p1 = quadh_pixel_partial(A1a, Ta, lane);
    const float p2 = quadh_pixel_partial(A2a, Ta + V2a, lane), p3 = quadh_pixel_partial(A3a, Ta + V3a, lane);
    float total;
    int slot;
    bool first;
    if constexpr (QPW == 2) {
        const float p4 = quadh_pixel_partial(A0b, Tb + V0b, lane), p5 = quadh_pixel_partial(A1b, Tb, lane);
        const float p6 = quadh_pixel_partial(A2b, Tb + V2b, lane), p7 = quadh_pixel_partial(A3b, Tb + V3b, lane);
        total = wave_sum8(p0, p1, p2, p3, p4, p5, p6, p7, lane);
        slot = kWaveSum8Value(lane >> 3);
        first = (lane & 7) == 0;
    } else {
        total = wave_sum4(p0, p1, p2, p3);
        slot = kWaveSum4Value(lane >> 4);
        first = (lane & 15) == 0;
    }
    if constexpr (QPW == 1) {
        if (a.done.flag && NW == 16) {  // (uniform) the one-frame host call: completion by flag (store_tile_and_signal)
            store_tile_and_signal(lds, first, slot & 3, wave, lane, total / norm, a.power + (size_t) frame * a.pixel_count, 4 * row4,
                                  (tile - row4 * tiles_per_row4) * tile_cols, a.rows, a.cols, a.done);
            return;
        }
    }
    const int col = col0 + (slot >> 2), row = 4 * row4 + (slot & 3);
    if (first && col < a.cols && row < a.rows) a.power[(size_t) frame * a.pixel_count + (size_t) row * a.cols + col] = total / norm;
}

// ---------------------------------------------------------------------------------------
// host side: geometry of the LDS image and the launch
// ---------------------------------------------------------------------------------------
bool fast_plan(int window, int usable, int fpi, int image_bytes, FastPlan *plan) {
    if (fpi != 1 && fpi != 2) return false;
    const int wr = (window + 3) & ~3;  // rows are whole 16-byte pieces (and start 16-byte aligned)
    const size_t row_bytes = (size_t) wr * sizeof(float);
    const size_t frame_bytes = (size_t) image_bytes / fpi;
    int chunk = (int) (frame_bytes / (2 * row_bytes));
    chunk &= ~3;  // whole entry groups per chunk
    if (chunk > 64) chunk = 64;
    if (chunk < 4) return false;
    const int usable_pad = (usable + 3) & ~3;
    if (chunk > usable_pad) chunk = usable_pad;
    plan->fpi = fpi;
    plan->wr = wr;
    plan->chunk = chunk;
    plan->usable_pad = usable_pad;
    plan->row_bytes = (int) row_bytes;
    plan->image_bytes = image_bytes;
    return true;
}

// Kernels that ask for more than 64 KiB of dynamic LDS need the limit raised once per function AND
// per device (a process may hold handles on several GPUs); `done` is the caller's per-function flags.
// The flags are atomics: every worker thread launches through its own handle, and two first launches may
// meet here (setting the attribute twice is harmless; a torn flag would not be).
typedef std::atomic<bool> LdsFlags[64];
static hipError_t allow_lds(const void *kernel, int bytes, LdsFlags &done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !done[dev].load(std::memory_order_acquire)) {
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) done[dev].store(true, std::memory_order_release);
    }
    return hipSuccess;
}

// reach of the kernels (das_kernels.h, Extents): the furthest table entry / packed float a launch can load
static bool within(const Extents &reach, const Extents &have) {
    return reach.lut_entries <= have.lut_entries && reach.sample_floats <= have.sample_floats;
}
// pixel-major tables ([pixel][usable_pad] FastEntry / dword): rows a frame-pair launch can address -- with the grid's row
// length known, the clamped pixel of slot A plus one grid row for its vertical partner; else whole 64-pixel tiles
static size_t pair_table_rows(int pixel_count, int cols) {
    return cols > 0 ? (size_t) pixel_count + cols : ((size_t) pixel_count + 63) / 64 * 64;
}
// quad-major tables ([quad][group][4 pixels][4 mics] QuadEntry): every quad of the padded grid, one group of prefetch
static size_t quad_table_reach(int rows, int cols, int usable_pad) {
    return (size_t) quad_count(rows, cols) * (usable_pad / 4) * 16 + kQuadTablePrefetch;
}

template <int NW, int PPW, int FPI, int WPS>
static hipError_t launch_variant(const FastArgs &a, hipStream_t stream) {
    static LdsFlags attr_set = {};
    if (hipError_t e = allow_lds((const void *) das_fast_kernel<NW, PPW, FPI, WPS>, kFastLdsBytes, attr_set); e != hipSuccess)
        return e;
    const int pix_per_block = NW * PPW;
    dim3 grid((a.batch + FPI - 1) / FPI, (a.pixel_count + pix_per_block - 1) / pix_per_block);
    if (grid.y > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL((das_fast_kernel<NW, PPW, FPI, WPS>), grid, dim3(NW * 64), kFastLdsBytes, stream, a);
    return hipGetLastError();
}

template <int NW, int PPW, int BUF, int WPS, bool DIAG>
static hipError_t launch_db(const FastArgs &a, hipStream_t stream) {
    static LdsFlags attr_set = {};
    constexpr int lds_bytes = 2 * BUF + kFastSideBytes;
    if (hipError_t e = allow_lds((const void *) das_fast_db_kernel<NW, PPW, BUF, WPS, DIAG>, lds_bytes, attr_set); e != hipSuccess)
        return e;
    const int pix_per_block = NW * PPW;
    dim3 grid((a.batch + a.frames_per_wg - 1) / a.frames_per_wg, (a.pixel_count + pix_per_block - 1) / pix_per_block);
    if (grid.y > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL((das_fast_db_kernel<NW, PPW, BUF, WPS, DIAG>), grid, dim3(NW * 64), lds_bytes, stream, a);
    return hipGetLastError();
}

bool pair_plan(int window, int usable, FastPlan *plan) {
    const int wp = (window + 1) & ~1;  // rows of 8-byte elements, whole 16-byte pieces
    const size_t row_bytes = (size_t) wp * 8;
    int chunk = (int) ((size_t) kFastLdsBytes / row_bytes);
    chunk &= ~3;
    if (chunk > 64) chunk = 64;
    if (chunk < 4) return false;
    const int usable_pad = (usable + 3) & ~3;
    if (chunk > usable_pad) chunk = usable_pad;
    plan->fpi = 2;
    plan->wr = wp;
    plan->chunk = chunk;
    plan->usable_pad = usable_pad;
    plan->row_bytes = (int) row_bytes;
    plan->image_bytes = -1;  // marks the frame-pair layout
    return true;
}

hipError_t launch_pack_pairs(const float *d_frames, int n_streams, int hist, int wstart, const int32_t *d_index,
                             int usable, int rows_out, const float *d_gain, int wp, int batch, float *d_packed,
                             bool filter, hipStream_t stream) {
    dim3 grid(rows_out, (batch + 1) / 2);
    if (filter) {
        hipLaunchKernelGGL(pack_pairs_kernel<true>, grid, dim3(128), 0, stream, d_frames, n_streams, hist, wstart, d_index,
                           usable, d_gain, wp, batch, d_packed);
    } else {
        hipLaunchKernelGGL(pack_pairs_kernel<false>, grid, dim3(128), 0, stream, d_frames, n_streams, hist, wstart, d_index,
                           usable, d_gain, wp, batch, d_packed);
    }
    return hipGetLastError();
}

template <int PPW, bool DIAG, bool SHARE>
static hipError_t launch_pair_variant(const PairArgs &a, const Extents &have, hipStream_t stream) {
    static LdsFlags attr_set = {};
    constexpr int lds_bytes = 2 * kFastLdsBytes;
    if (hipError_t e = allow_lds((const void *) das_pair_kernel<PPW, DIAG, SHARE>, lds_bytes, attr_set); e != hipSuccess) return e;
    static_assert(16 * PPW == 64, "pair_tiles() counts 64-pixel tiles");
    if (a.n_pairs != (a.batch + 1) / 2 || a.tiles != pair_tiles(a.pixel_count, a.cols) || a.pair_group < 1) return hipErrorInvalidValue;
    // reach: the last pixel's row + one group of prefetch; `usable` rows of wp elements per frame pair, in whole 16-byte pieces
    if ((a.wp & 1) || !within({pair_table_rows(a.pixel_count, a.cols) * a.usable_pad + kPairTablePrefetch,
                               (size_t) a.n_pairs * a.usable * a.wp * 2}, have))
        return hipErrorInvalidValue;
    const long total = (long) a.n_pairs * a.tiles;
    dim3 grid((unsigned) (8 * ((total + 7) / 8)));
    hipLaunchKernelGGL((das_pair_kernel<PPW, DIAG, SHARE>), grid, dim3(1024), lds_bytes, stream, a);
    return hipGetLastError();
}

bool pair_plan_stationary(int window, int usable, FastPlan *plan) {
    const int wp = (window + 1) & ~1;
    const size_t row_bytes = (size_t) wp * 8;
    const int usable_pad = (usable + 3) & ~3;
    if ((size_t) usable * row_bytes > (size_t) 2 * kFastLdsBytes) return false;  // (null entries of padding mics point at row 0)
    plan->fpi = 2;
    plan->wr = wp;
    plan->chunk = usable_pad;  // every mic has its own slot: addresses are not folded into chunks
    plan->usable_pad = usable_pad;
    plan->row_bytes = (int) row_bytes;
    plan->image_bytes = -2;  // marks the stationary layout
    return true;
}

hipError_t launch_das_pairs_stationary(const PairArgs &a, int tiles_per_wg, const Extents &have, hipStream_t stream) {
    static LdsFlags attr_set[2] = {};
    constexpr int lds_bytes = 2 * kFastLdsBytes;
    const int n_tiles = pair_tiles(a.pixel_count, a.cols);
    // (self-staged pairs read the caller's frames inside [0, hist) of a stream: the kernel clips to the history itself)
    if (tiles_per_wg < 1 || (a.wp & 1) || (size_t) a.usable * a.wp * 8 > (size_t) lds_bytes ||
        (a.frames && (!a.index || a.n_streams < 1 || a.wstart < 0 || a.wstart >= a.hist)) ||
        !within({pair_table_rows(a.pixel_count, a.cols) * a.usable_pad + kPairTablePrefetch,
                 a.frames ? 0 : (size_t) ((a.batch + 1) / 2) * a.usable * a.wp * 2}, have))
        return hipErrorInvalidValue;
    dim3 grid((a.batch + 1) / 2, (n_tiles + tiles_per_wg - 1) / tiles_per_wg);
    if (grid.y > 65535) return hipErrorInvalidValue;
#ifdef AWPU_TUNING_BUILD  // debug bit 4096: the pixel-major block without read sharing (the same bits)
    if (a.debug & 4096) {
        if (hipError_t e = allow_lds((const void *) das_pair_stationary_kernel<false>, lds_bytes, attr_set[0]); e != hipSuccess) return e;
        hipLaunchKernelGGL(das_pair_stationary_kernel<false>, grid, dim3(1024), lds_bytes, stream, a, n_tiles, tiles_per_wg);
        return hipGetLastError();
    }
#endif
    if (hipError_t e = allow_lds((const void *) das_pair_stationary_kernel<true>, lds_bytes, attr_set[1]); e != hipSuccess) return e;
    hipLaunchKernelGGL(das_pair_stationary_kernel<true>, grid, dim3(1024), lds_bytes, stream, a, n_tiles, tiles_per_wg);
    return hipGetLastError();
}

bool fir8_plane_plan(int window, int usable, FastPlan *plan) {
    int wp = (window + 3) & ~3;  // four planes of wp / 4 elements
    // windows of 321..384 samples (every BASELINE shape but the single 8x8 array) are staged at the plane pitch the
    // one-address block is generated for (sweep_fir8_planes_static: 33 instead of 36 VALU instructions per item);
    // AWPU_FIR8_STATIC=0 keeps the natural pitch for A/B runs
#ifdef AWPU_TUNING_BUILD
    static const bool allow_static = !(std::getenv("AWPU_FIR8_STATIC") && std::atoi(std::getenv("AWPU_FIR8_STATIC")) == 0);
#else
    constexpr bool allow_static = true;
#endif
    const int wp_static = (int) (kFirStaticPlaneBytes / 2);
    if (allow_static && wp > wp_static - 64 && wp <= wp_static) wp = wp_static;
    const size_t row_bytes = (size_t) wp * 8;
    int chunk = (int) ((size_t) kFastLdsBytes / row_bytes);
    chunk &= ~3;
    if (chunk > 64) chunk = 64;
    if (chunk < 4) return false;
    const int usable_pad = (usable + 3) & ~3;
    if (chunk > usable_pad) chunk = usable_pad;
    plan->fpi = 2;
    plan->wr = wp;
    plan->chunk = chunk;
    plan->usable_pad = usable_pad;
    plan->row_bytes = (int) row_bytes;
    plan->image_bytes = -3;  // marks the four-plane frame-pair layout
    return true;
}

hipError_t launch_pack_planes(const float *d_frames, int n_streams, int hist, int wstart, const int32_t *d_index, int usable,
                              const float *d_gain, int wp, int batch, float *d_packed, hipStream_t stream) {
    dim3 grid(usable, (batch + 1) / 2);
    hipLaunchKernelGGL(pack_planes_kernel, grid, dim3(128), 0, stream, d_frames, n_streams, hist, wstart, d_index, d_gain,
                       wp, batch, d_packed);
    return hipGetLastError();
}

template <int VAR>
static hipError_t launch_fir8_plane_variant(const PairArgs &a, const void *d_entries, const float *d_coeffs, const Extents &have,
                                            hipStream_t stream) {
    static LdsFlags attr_set = {};
    constexpr int lds_bytes = 2 * kFastLdsBytes;
    if (hipError_t e = allow_lds((const void *) das_fir8_plane_kernel<VAR>, lds_bytes, attr_set); e != hipSuccess) return e;
    const int n_pairs = (a.batch + 1) / 2;
    const int n_tiles = a.cols > 0 ? ((a.pixel_count / a.cols + 3) / 4) * ((a.cols + 15) / 16) : (a.pixel_count + 63) / 64;
    if (a.cols > 0 && (unsigned) a.wp * 2u != kFirStaticPlaneBytes) return hipErrorInvalidValue;  // (the shared block is generated for that pitch)
    // reach: slots past the grid repeat a pixel inside it -- pixel_count rows of one dword per mic, and the block's prefetch
    if ((a.wp & 3) || !within({(size_t) a.pixel_count * a.usable_pad + kFir8PlaneTablePrefetch, (size_t) n_pairs * a.usable * a.wp * 2}, have))
        return hipErrorInvalidValue;
    // frame pairs an XCD works on at a time: as many as keep their packed samples in its 4 MiB L2 beside the table slices
    const size_t pair_bytes = (size_t) a.usable * a.wp * 8;
    int g = (int) std::max<size_t>(1, (3u << 20) / pair_bytes);
    g = g >= 8 ? 8 : g >= 4 ? 4 : g >= 2 ? 2 : 1;
    while (g > 1 && g > n_pairs) g >>= 1;
    const long total = (long) n_pairs * n_tiles;
    const long per_xcd = (total + 7) / 8;
    hipLaunchKernelGGL(das_fir8_plane_kernel<VAR>, dim3((unsigned) (8 * per_xcd)), dim3(1024), lds_bytes, stream, a,
                       (const FirPlaneEntry *) d_entries, d_coeffs, n_tiles, g);
    return hipGetLastError();
}

hipError_t launch_das_fir8_planes(const PairArgs &a, const void *d_entries, const float *d_coeffs, int variant, const Extents &have,
                                  hipStream_t stream) {
#ifdef AWPU_QUAD_VARIANTS  // tuning builds (tools/gen_trip_asm.py with QUAD_VARIANTS=1): timing-only and alternative blocks
    if (variant == 1) return launch_fir8_plane_variant<1>(a, d_entries, d_coeffs, have, stream);
    if (variant == 2) return launch_fir8_plane_variant<2>(a, d_entries, d_coeffs, have, stream);
    if (variant == 3) return launch_fir8_plane_variant<3>(a, d_entries, d_coeffs, have, stream);  // no table prefetch
#endif
    (void) variant;
    return launch_fir8_plane_variant<0>(a, d_entries, d_coeffs, have, stream);
}

bool exact_nd_plan(int window, int usable, FastPlan *plan) {
    const int wq = window - 1;  // element t holds X[t+1] and X[t] - X[t+1]: one element less than samples
    const size_t row_bytes = (size_t) wq * 16;
    int chunk = (int) ((size_t) kFastLdsBytes / row_bytes);
    chunk &= ~3;
    if (chunk > 64) chunk = 64;
    if (chunk < 4 || wq < kSamples) return false;
    const int usable_pad = (usable + 3) & ~3;
    if (chunk > usable_pad) chunk = usable_pad;
    plan->fpi = 2;
    plan->wr = wq;
    plan->chunk = chunk;
    plan->usable_pad = usable_pad;
    plan->row_bytes = (int) row_bytes;
    plan->image_bytes = -4;  // marks the {next, d} layout
    return true;
}

hipError_t launch_pack_nd(const float *d_frames, int n_streams, int hist, int wstart, const int32_t *d_index, int usable, int rows_out,
                          const float *d_gain, int wq, int batch, float *d_packed, hipStream_t stream) {
    hipLaunchKernelGGL(pack_nd_kernel, dim3(rows_out, (batch + 1) / 2), dim3(128), 0, stream, d_frames, n_streams, hist, wstart, d_index,
                       usable, d_gain, wq, batch, d_packed);
    return hipGetLastError();
}

template <int NQ, bool SUMS>
static hipError_t launch_exact_nd_variant(const ExactNdArgs &a, hipStream_t stream) {
    static LdsFlags attr_set = {};
    constexpr int lds_bytes = 2 * kFastLdsBytes + 64;  // two images + the item mailbox
    if (hipError_t e = allow_lds((const void *) das_exact_nd_kernel<NQ, SUMS>, lds_bytes, attr_set); e != hipSuccess) return e;
    // one persistent workgroup per CU (the LDS holds no second one), never more than there are items; the queues start at zero
    const long total = (long) a.n_pairs * a.tiles;
    if (hipError_t e = hipMemsetAsync(a.queue, 0, 9 * sizeof(unsigned), stream); e != hipSuccess) return e;
    if (a.build_items)
        hipLaunchKernelGGL(nd_items_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, stream, const_cast<int2 *>(a.items), a.n_pairs,
                           a.tiles, a.pair_group, (a.cols + 15) / 16, NQ);
    hipLaunchKernelGGL((das_exact_nd_kernel<NQ, SUMS>), dim3((unsigned) std::min<long>(total, std::max(1, a.wgs))), dim3(1024), lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_das_exact_nd(const ExactNdArgs &a, const Extents &have, hipStream_t stream) {
    if ((a.nq != 1 && a.nq != 2) || !a.queue || !a.items || a.wgs < 1 || a.tail < 1) return hipErrorInvalidValue;
    if (a.chunk < 4 || (a.chunk & 3) || (a.usable_pad & 3) || a.usable < 1 || a.usable > a.usable_pad || a.wq < kSamples ||
        (size_t) a.chunk * a.wq * 16 > (size_t) kFastLdsBytes || a.cols < 1 || a.rows * a.cols != a.pixel_count)
        return hipErrorInvalidValue;
    if (a.n_pairs != (a.batch + 1) / 2 || a.tiles != nd_tiles(a.rows, a.cols, a.nq) || a.pair_group < 1) return hipErrorInvalidValue;
    // reach: every quad of the grid padded to whole tiles (quad rows to a multiple of nq) + one group of prefetch; usable_pad rows of wq
    // 16-byte elements per frame pair
    if (!within({(size_t) nd_quad_count(a.rows, a.cols, a.nq) * (a.usable_pad / 4) * 16 + kQuadTablePrefetch,
                 (size_t) a.n_pairs * a.usable_pad * a.wq * 4}, have))
        return hipErrorInvalidValue;
    if (a.sums) return a.nq == 2 ? launch_exact_nd_variant<2, true>(a, stream) : launch_exact_nd_variant<1, true>(a, stream);
    return a.nq == 2 ? launch_exact_nd_variant<2, false>(a, stream) : launch_exact_nd_variant<1, false>(a, stream);
}

bool exact_ndh_plan(int window, int usable, bool stationary, FastPlan *plan) {
    const int wh = window - 129;  // element t holds samples t, t+1, t+128, t+129 of the window
    if (wh < kSamples / 2) return false;
    const size_t row_bytes = (size_t) wh * 16;
    const int usable_pad = (usable + 3) & ~3;
    int chunk;
    if (stationary) {
        if ((size_t) usable_pad * row_bytes > (size_t) 2 * kFastLdsBytes) return false;
        chunk = usable_pad;  // every mic has its own slot
    } else {
        chunk = (int) ((size_t) kFastLdsBytes / row_bytes) & ~3;
        if (chunk > 64) chunk = 64;
        if (chunk < 4) return false;
        if (chunk > usable_pad) chunk = usable_pad;
    }
    plan->fpi = 1;
    plan->wr = wh;
    plan->chunk = chunk;
    plan->usable_pad = usable_pad;
    plan->row_bytes = (int) row_bytes;
    plan->image_bytes = stationary ? -6 : -5;  // marks the halves form of the {next, d} layout
    return true;
}

hipError_t launch_pack_ndh(const float *d_frames, int n_streams, int pitch, int wstart, const int32_t *d_index, int usable, int rows_out,
                           const float *d_gain, int wh, int batch, float *d_packed, hipStream_t stream) {
    hipLaunchKernelGGL(pack_ndh_kernel, dim3(rows_out, batch), dim3(256), 0, stream, d_frames, n_streams, pitch, wstart, d_index, usable, d_gain,
                       wh, d_packed);
    return hipGetLastError();
}

template <int NQ, bool STATIONARY>
static hipError_t launch_exact_ndh_variant(const ExactNdhArgs &a, hipStream_t stream) {
    static LdsFlags attr_set = {};
    constexpr int lds_bytes = 2 * kFastLdsBytes;
    if (hipError_t e = allow_lds((const void *) das_exact_ndh_kernel<NQ, STATIONARY>, lds_bytes, attr_set); e != hipSuccess) return e;
    hipLaunchKernelGGL((das_exact_ndh_kernel<NQ, STATIONARY>), dim3((unsigned) ((long) a.batch * a.tiles)), dim3(1024), lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_das_exact_ndh(const ExactNdhArgs &a, bool stationary, const Extents &have, hipStream_t stream) {
    if (a.nq != 1 && a.nq != 2) return hipErrorInvalidValue;
    if (a.chunk < 4 || (a.chunk & 3) || (a.usable_pad & 3) || a.usable < 1 || a.usable > a.usable_pad || a.wh < kSamples / 2 || a.cols < 1 ||
        a.rows * a.cols != a.pixel_count || a.batch < 1 || a.tiles != ndh_tiles(a.rows, a.cols, a.nq) || (long) a.batch * a.tiles > 0x7fffffffL)
        return hipErrorInvalidValue;
    if (stationary ? (a.chunk != a.usable_pad || (size_t) a.usable_pad * a.wh * 16 > (size_t) 2 * kFastLdsBytes || !a.frames || !a.index || a.pitch < 1)
                   : ((size_t) a.chunk * a.wh * 16 > (size_t) kFastLdsBytes || !a.packed))
        return hipErrorInvalidValue;
    // reach: every quad of the grid with its columns padded to whole tiles (16 nq) + one group of prefetch; chunked: usable_pad rows of wh
    // 16-byte elements per frame (stationary: the caller's frames, read inside [wstart, wstart + wh + 129) of a stream)
    if (a.lut_cols < (a.cols + 16 * a.nq - 1) / (16 * a.nq) * 16 * a.nq) return hipErrorInvalidValue;
    const size_t quads = (size_t) ((a.rows + 3) / 4) * a.lut_cols;
    if (!within({quads * (a.usable_pad / 4) * 16 + kQuadTablePrefetch, stationary ? 0 : (size_t) a.batch * a.usable_pad * a.wh * 4}, have))
        return hipErrorInvalidValue;
    if (stationary) return a.nq == 2 ? launch_exact_ndh_variant<2, true>(a, stream) : launch_exact_ndh_variant<1, true>(a, stream);
    return a.nq == 2 ? launch_exact_ndh_variant<2, false>(a, stream) : launch_exact_ndh_variant<1, false>(a, stream);
}

hipError_t launch_das_exact_ndp(const ExactNdhArgs &a, const Extents &have, hipStream_t stream) {
    static LdsFlags attr_set = {};
    constexpr int lds_bytes = 2 * kFastLdsBytes;
    if (hipError_t e = allow_lds((const void *) das_exact_ndp_kernel, lds_bytes, attr_set); e != hipSuccess) return e;
    if (a.chunk < 4 || (a.chunk & 3) || (a.usable_pad & 3) || a.usable < 1 || a.usable > a.usable_pad || a.wh < kSamples / 2 || a.cols < 1 ||
        a.rows * a.cols != a.pixel_count || a.batch < 1 || a.tiles != ndp_tiles(a.rows, a.cols) || (long) a.batch * a.tiles > 0x7fffffffL ||
        (size_t) a.chunk * a.wh * 16 > (size_t) kFastLdsBytes || !a.packed || a.lut_cols < (a.cols + 3) / 4 * 4)
        return hipErrorInvalidValue;
    // reach: every quad of the table + TWO groups of prefetch (the block requests entries two trips ahead); usable_pad rows of wh elements per frame
    const size_t quads = (size_t) ((a.rows + 3) / 4) * a.lut_cols;
    if (!within({quads * (a.usable_pad / 4) * 16 + 2 * kQuadTablePrefetch, (size_t) a.batch * a.usable_pad * a.wh * 4}, have)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(das_exact_ndp_kernel, dim3((unsigned) ((long) a.batch * a.tiles)), dim3(1024), lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_das_exact_quads(const ExactQuadArgs &a, const Extents &have, hipStream_t stream) {
    static LdsFlags attr_set = {};
    constexpr int lds_bytes = 2 * kFastLdsBytes;
    if (hipError_t e = allow_lds((const void *) das_exact_quad_kernel, lds_bytes, attr_set); e != hipSuccess) return e;
    if (a.chunk < 4 || (a.chunk & 3) || (a.usable_pad & 3) || (size_t) a.chunk * a.wp * 8 > (size_t) kFastLdsBytes || a.cols < 1 ||
        a.rows * a.cols != a.pixel_count)
        return hipErrorInvalidValue;
    if (a.n_pairs != (a.batch + 1) / 2 || a.tiles != quad_tiles(a.rows, a.cols) || a.pair_group < 1) return hipErrorInvalidValue;
    if ((a.wp & 1) || !within({quad_table_reach(a.rows, a.cols, a.usable_pad), (size_t) a.n_pairs * a.usable_pad * a.wp * 2}, have))
        return hipErrorInvalidValue;
    const long total = (long) a.n_pairs * a.tiles;
    dim3 grid((unsigned) (8 * ((total + 7) / 8)));
    hipLaunchKernelGGL(das_exact_quad_kernel, grid, dim3(1024), lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_das_exact_pairs(const ExactPairArgs &a, const Extents &have, hipStream_t stream) {
    static LdsFlags attr_set = {};
    constexpr int lds_bytes = 2 * kFastLdsBytes;
    if (hipError_t e = allow_lds((const void *) das_exact_pair_kernel, lds_bytes, attr_set); e != hipSuccess) return e;
    if (a.chunk < 4 || (a.chunk & 3) || (a.usable_pad & 3) || (size_t) a.chunk * a.wp * 8 > (size_t) kFastLdsBytes) return hipErrorInvalidValue;
    if (a.n_pairs != (a.batch + 1) / 2 || a.tiles != pair_tiles(a.pixel_count, a.cols) || a.pair_group < 1) return hipErrorInvalidValue;
    if ((a.wp & 1) || !within({pair_table_rows(a.pixel_count, a.cols) * a.usable_pad + kPairTablePrefetch,
                               (size_t) a.n_pairs * a.usable_pad * a.wp * 2}, have))
        return hipErrorInvalidValue;
    const long total = (long) a.n_pairs * a.tiles;
    dim3 grid((unsigned) (8 * ((total + 7) / 8)));
    hipLaunchKernelGGL(das_exact_pair_kernel, grid, dim3(1024), lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_das_pairs(const PairArgs &a, const Extents &have, hipStream_t stream) {
#ifdef AWPU_TUNING_BUILD  // stamped builds (bit 16) and the pixel-major block without read sharing (bit 4096: the same bits)
    const bool share = (a.debug & 4096) == 0;
    if (a.debug & 16) return share ? launch_pair_variant<4, true, true>(a, have, stream) : launch_pair_variant<4, true, false>(a, have, stream);
    if (!share) return launch_pair_variant<4, false, false>(a, have, stream);
#endif
    return launch_pair_variant<4, false, true>(a, have, stream);
}

template <bool DIAG, int VAR>
static hipError_t launch_quad_variant(const QuadArgs &a, const Extents &have, hipStream_t stream) {
    static LdsFlags attr_set = {};
    constexpr int lds_bytes = 2 * kFastLdsBytes + 64;  // two images + the item mailbox of the queued launch
    if (hipError_t e = allow_lds((const void *) das_quad_kernel<DIAG, VAR>, lds_bytes, attr_set); e != hipSuccess) return e;
    // One workgroup per item by default: the hardware hands items to CUs as they free up, which stays balanced when
    // something else (an RCCL broadcast of the next batch) holds a few CUs.  a.wgs > 0 (AWPU_FAST_WGS) launches that
    // many persistent workgroups instead, each walking several items with the next item's first chunk prefetched:
    // measured equal at the headline shape on an otherwise idle chip (5.41 vs 5.41 ms), and fragile when CUs are
    // shared (a static share of the items per workgroup).
    if (a.rows * a.cols != a.pixel_count || a.n_pairs != (a.batch + 1) / 2 || a.tiles != quad_tiles(a.rows, a.cols) || a.pair_group < 1 ||
        (a.wp & 1) || (a.usable_pad & 3) || (a.chunk & 3) || (size_t) a.chunk * a.wp * 8 > (size_t) kFastLdsBytes)
        return hipErrorInvalidValue;
    if (!within({quad_table_reach(a.rows, a.cols, a.usable_pad), (size_t) a.n_pairs * a.usable_pad * a.wp * 2}, have)) return hipErrorInvalidValue;
    const long items = (long) a.n_pairs * a.tiles;
    const long per_xcd = (items + 7) / 8;
    if (!DIAG && VAR == 0 && a.queue && a.wgs > 0) {  // persistent workgroups on the item queues (das_quad_kernel: `queued`)
        if (a.tail < 1) return hipErrorInvalidValue;
        if (hipError_t e = hipMemsetAsync(a.queue, 0, 9 * sizeof(unsigned), stream); e != hipSuccess) return e;
        hipLaunchKernelGGL((das_quad_kernel<DIAG, VAR>), dim3((unsigned) std::min<long>(items, a.wgs)), dim3(1024), lds_bytes, stream, a);
        return hipGetLastError();
    }
    QuadArgs b = a;
    b.queue = nullptr;  // (the stamped and the tuning instances keep the static shares)
    const long wgs_per_xcd = a.wgs > 0 ? std::min<long>(per_xcd, std::max(1, a.wgs / 8)) : per_xcd;
    if (8 * wgs_per_xcd > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((das_quad_kernel<DIAG, VAR>), dim3((unsigned) (8 * wgs_per_xcd)), dim3(1024), lds_bytes, stream, b);
    return hipGetLastError();
}

hipError_t launch_das_quads(const QuadArgs &a, const Extents &have, hipStream_t stream) {
#ifdef AWPU_TUNING_BUILD
    if (a.debug & 16) return launch_quad_variant<true, 0>(a, have, stream);
#endif
#ifdef AWPU_QUAD_VARIANTS
    if (a.variant == 1) return launch_quad_variant<false, 1>(a, have, stream);
    if (a.variant == 2) return launch_quad_variant<false, 2>(a, have, stream);
    if (a.variant == 3) return launch_quad_variant<false, 3>(a, have, stream);
#endif
    return launch_quad_variant<false, 0>(a, have, stream);
}

int fast_image_bytes(int nw) { return nw == 24 ? kFastLdsBytesSmall : kFastLdsBytes; }

bool fast_db_fits(const FastPlan &plan) {
    return (size_t) 2 * plan.usable_pad * sizeof(int) <= (size_t) kFastSideBytes;
}

template <int QPW, bool DIAG>
static hipError_t launch_quadh_variant(const QuadhArgs &a, const Extents &have, hipStream_t stream) {
    static LdsFlags attr_set = {};
    constexpr int lds_bytes = 2 * kFastLdsBytes;
    if (hipError_t e = allow_lds((const void *) das_quadh_kernel<QPW, DIAG>, lds_bytes, attr_set); e != hipSuccess) return e;
    if (a.rows * a.cols != a.pixel_count || (a.wp & 1) || (a.usable_pad & 3) || (a.chunk & 3) || a.chunk < 4 ||
        (size_t) a.chunk * a.wp * 8 > (size_t) kFastLdsBytes ||
        !within({quad_table_reach(a.rows, a.cols, a.usable_pad), (size_t) a.batch * a.usable_pad * a.wp * 2}, have))
        return hipErrorInvalidValue;
    dim3 grid(a.batch, quad1_tiles(a.rows, a.cols, QPW));
    if (grid.y > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL((das_quadh_kernel<QPW, DIAG>), grid, dim3(1024), lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_das_quadh(const QuadhArgs &a, int qpw, const Extents &have, hipStream_t stream) {
#ifdef AWPU_TUNING_BUILD
    if (a.debug & 16) return qpw == 2 ? launch_quadh_variant<2, true>(a, have, stream) : launch_quadh_variant<1, true>(a, have, stream);
#endif
    return qpw == 2 ? launch_quadh_variant<2, false>(a, have, stream) : launch_quadh_variant<1, false>(a, have, stream);
}

bool quadh_stationary_plan(int window, int usable, FastPlan *plan) {
    const int wp = (window - 128 + 1) & ~1;  // elements (sample t, sample t + 128) per row
    if (wp < 130) return false;
    const size_t row_bytes = (size_t) wp * 8;
    const int usable_pad = (usable + 3) & ~3;
    // the halves image + the raw rows it is filtered from (at most wp + 136 floats each) + the row table
    if ((size_t) usable_pad * row_bytes + (size_t) usable * (wp + 130) * 4 > (size_t) kQuadhsRowTableOffset * 4) return false;
    plan->fpi = 1;
    plan->wr = wp;
    plan->chunk = usable_pad;  // every mic has its own slot
    plan->usable_pad = usable_pad;
    plan->row_bytes = (int) row_bytes;
    plan->image_bytes = -3;
    return true;
}

// the raw rows a launch stages: history samples [raw_begin, raw_begin + raw_wr) of every active stream, whole 16-byte pieces;
// false if they do not fit beside the image (the caller then takes das_quadh_kernel)
bool quadh_stationary_raw(const FastPlan &plan, int usable, int wstart, int row_limit, int *raw_begin, int *raw_wr, int *image_offset) {
    const int begin = std::max(0, wstart - 1) & ~3;
    const int end = std::min(row_limit, (wstart + plan.wr + 128 + 1 + 3) & ~3);
    if (end <= begin || ((end - begin) & 3)) return false;
    const size_t raw_floats = (size_t) usable * (end - begin);
    const size_t image_off = (raw_floats + 3) & ~(size_t) 3;
    if (image_off * 4 + (size_t) plan.usable_pad * plan.row_bytes > (size_t) kQuadhsRowTableOffset * 4) return false;
    if (usable > 1024) return false;  // (the row table)
    *raw_begin = begin;
    *raw_wr = end - begin;
    *image_offset = (int) image_off;
    return true;
}

template <int QPW>
static hipError_t launch_quadh_stationary_variant(const QuadhStationaryArgs &a, const Extents &have, hipStream_t stream) {
    static LdsFlags attr_set = {};
    constexpr int lds_bytes = 2 * kFastLdsBytes;
    if (hipError_t e = allow_lds((const void *) das_quadh_stationary_kernel<QPW>, lds_bytes, attr_set); e != hipSuccess) return e;
    if ((size_t) a.image_offset * 4 + (size_t) a.usable_pad * a.wp * 8 > (size_t) kQuadhsRowTableOffset * 4 || (a.usable_pad & 3) ||
        (a.raw_wr & 3) || (a.image_offset & 3) || (size_t) a.usable * a.raw_wr > (size_t) a.image_offset || a.usable > 1024 ||
        a.raw_begin + a.raw_wr > a.row_limit)
        return hipErrorInvalidValue;
    if (a.waves < 4 || a.waves > 16) return hipErrorInvalidValue;
    // (the samples are the caller's frames: rows [raw_begin, raw_begin + raw_wr) of a stream, inside row_limit -- checked above)
    if (a.rows * a.cols != a.pixel_count || !within({quad_table_reach(a.rows, a.cols, a.usable_pad), 0}, have)) return hipErrorInvalidValue;
    const int tile_cols = a.waves * QPW;
    dim3 grid(a.batch, ((a.rows + 3) / 4) * ((a.cols + tile_cols - 1) / tile_cols));
    if (grid.y > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL((das_quadh_stationary_kernel<QPW>), grid, dim3(a.waves * 64), lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_das_quadh_stationary(const QuadhStationaryArgs &a, int qpw, const Extents &have, hipStream_t stream) {
    return qpw == 2 ? launch_quadh_stationary_variant<2>(a, have, stream) : launch_quadh_stationary_variant<1>(a, have, stream);
}

hipError_t launch_pack_halves(const float *d_frames, int n_streams, int pitch, int hist, int wstart, const int32_t *d_index, int usable,
                              int rows_out, const float *d_gain, int wp, int batch, float *d_packed, hipStream_t stream) {
    // one thread per element where a row allows (a row is 130 .. 400 elements): the pass is a few microseconds of latency
    // in front of a 55 us sweep, not bandwidth
    const int threads = wp <= 256 ? 256 : 512;
    hipLaunchKernelGGL(pack_halves_kernel, dim3(rows_out, batch), dim3(threads), 0, stream, d_frames, n_streams, pitch, hist, wstart,
                       d_index, usable, d_gain, wp, d_packed);
    return hipGetLastError();
}

hipError_t launch_das_fast(const FastArgs &a, int fpi, int ppw, int nw, const Extents &have, hipStream_t stream) {
    {   // reach: whole workgroup tiles of pixels (rows past the grid are null rows of the table) + one group of prefetch; the
        // samples are the caller's frames, staged row by row through row_off[] with the valid length clamped to hist
        const size_t tile = (size_t) (nw == 32 ? 16 : nw == 24 ? 12 : 8) * ppw;
        if (ppw < 1 || !within({((size_t) a.pixel_count + tile - 1) / tile * tile * a.usable_pad + kPairTablePrefetch, 0}, have))
            return hipErrorInvalidValue;
    }
    if (nw == 32) {  // double-buffered, one 16-wave workgroup per CU
#ifdef AWPU_TUNING_BUILD
        if (a.debug & 16) return ppw == 4 ? launch_db<16, 4, kFastLdsBytes, 4, true>(a, stream)
                                          : launch_db<16, 8, kFastLdsBytes, 4, true>(a, stream);
#endif
        if (ppw == 4) return launch_db<16, 4, kFastLdsBytes, 4, false>(a, stream);
        return launch_db<16, 8, kFastLdsBytes, 4, false>(a, stream);
    }
#ifdef AWPU_TUNING_BUILD  // shapes only AWPU_FAST_VARIANT reaches (measured and not taken: docs/HISTORY.md 8)
    if (nw == 24) {  // double-buffered, two 12-wave workgroups per CU, 6 waves per SIMD
        if (a.debug & 16) return launch_db<12, 4, kFastLdsBytesSmall, 6, true>(a, stream);
        return launch_db<12, 4, kFastLdsBytesSmall, 6, false>(a, stream);
    }
    if (fpi == 2) {
        if (ppw == 2) return launch_variant<8, 2, 2, 4>(a, stream);
        return launch_variant<8, 4, 2, 4>(a, stream);
    }
    if (ppw == 8) return launch_variant<8, 8, 1, 4>(a, stream);
#endif
    if (fpi != 1 || nw != 8 || (ppw != 2 && ppw != 4)) return hipErrorInvalidValue;
    if (ppw == 2) return launch_variant<8, 2, 1, 4>(a, stream);
    return launch_variant<8, 4, 1, 4>(a, stream);
}

}  // namespace awpu
