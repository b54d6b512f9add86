#!/usr/bin/env python3
"""Generate beamforming-lk_amd/csrc/das_fast_trip.inc: the hand-scheduled inner loop of the
fast sweep kernel (das_fast.hip) -- one asm block = all items of ONE pixel in the staged chunk.

An item (pixel x mic, one frame) costs
    v_add_u32 (LDS address) ; 2 x ds_read_b64 ; 4 x v_pk_fma_f32
and a "trip" is 8 consecutive items whose table entries (f, addr, g, pad: 4 dwords each) sit in
32 SGPRs.  Two SGPR sets ping-pong: while a trip runs out of one set, two s_load_dwordx16 fill
the other with the next trip's entries.

Schedule inside a trip: LDS reads run DEPTH items ahead of the FMAs that consume them; every
wait is a counted s_waitcnt lgkmcnt(n) with n = the number of YOUNGER LDS reads of this wave.
Scalar loads share that counter and may return out of order, but since n never budgets for them
a pending scalar load can only make a wait longer, never let it pass early.  Each trip ends on
lgkmcnt(0), which is also what proves the other SGPR set has landed.

Fixed registers (all in the clobber list, so hipcc keeps nothing of its own there):
    SGPR  s[36:67] set X, s[68:99] set Y, s34 trip counter, s35 byte offset into the table row
    VGPR  five read slots of 4 registers + 1 address temp, from `vbase`
The block is self-contained: it begins with none of its own loads pending and ends drained, so
the compiler never sees a register with a load in flight.
"""
from pathlib import Path

DEPTH = 4  # items of LDS read-ahead
SET = {"X": 36, "Y": 68}
CNT, OFF = 34, 35


def trip(n_items, sbase, vbase, prefetch_into=None):
    """Lines of one trip over `n_items` entries held at s[sbase...]."""
    slots = DEPTH + 1
    addr_t = vbase + 4 * slots

    def fpair(i):
        return f"s[{sbase + 4 * i}:{sbase + 4 * i + 1}]"

    def gpair(i):
        return f"s[{sbase + 4 * i + 2}:{sbase + 4 * i + 3}]"

    def issue(i):
        s = vbase + 4 * (i % slots)
        return [f"v_add_u32 v{addr_t}, s{sbase + 4 * i + 1}, %[lane]",
                f"ds_read_b64 v[{s}:{s + 1}], v{addr_t}",
                f"ds_read_b64 v[{s + 2}:{s + 3}], v{addr_t} offset:512"]

    def fma(i):
        s = vbase + 4 * (i % slots)
        x, y = f"v[{s}:{s + 1}]", f"v[{s + 2}:{s + 3}]"
        return [f"v_pk_fma_f32 %[A], {fpair(i)}, {x}, %[A] op_sel_hi:[0,1,1]",
                f"v_pk_fma_f32 %[Q], {gpair(i)}, {x}, %[Q] op_sel_hi:[0,1,1]",
                f"v_pk_fma_f32 %[C], {fpair(i)}, {y}, %[C] op_sel_hi:[0,1,1]",
                f"v_pk_fma_f32 %[R], {gpair(i)}, {y}, %[R] op_sel_hi:[0,1,1]"]

    lines = []
    for i in range(min(DEPTH, n_items)):
        lines += issue(i)
    if prefetch_into is not None:  # next trip's entries; s35 already points at them
        lines += [f"s_load_dwordx16 s[{prefetch_into}:{prefetch_into + 15}], %[ptr], s{OFF}",
                  f"s_add_u32 s{OFF}, s{OFF}, 64",
                  f"s_load_dwordx16 s[{prefetch_into + 16}:{prefetch_into + 31}], %[ptr], s{OFF}",
                  f"s_add_u32 s{OFF}, s{OFF}, 64"]
    for k in range(n_items):
        if k + DEPTH < n_items:
            lines += issue(k + DEPTH)
        younger = min(DEPTH, n_items - 1 - k)
        lines.append(f"s_waitcnt lgkmcnt({2 * younger})")
        lines += fma(k)
    return lines


def pixel_block(name, vbase):
    """asm block: ng groups of four items starting at table address %[ptr]."""
    x, y = SET["X"], SET["Y"]
    L = []
    L += [f"s_load_dwordx16 s[{x}:{x + 15}], %[ptr], 0x0",
          f"s_load_dwordx16 s[{x + 16}:{x + 31}], %[ptr], 0x40",
          f"s_lshr_b32 s{CNT}, %[ng], 1",          # full trips
          f"s_movk_i32 s{OFF}, 0x80",               # where the next trip's entries start
          "s_waitcnt lgkmcnt(0)",
          f"s_cmp_eq_u32 s{CNT}, 0",
          "s_cbranch_scc1 .Lhalf_x_%="]
    L += [".Ltrip_x_%=:"]
    L += trip(8, x, vbase, prefetch_into=y)
    L += [f"s_sub_u32 s{CNT}, s{CNT}, 1",
          f"s_cmp_eq_u32 s{CNT}, 0",
          "s_cbranch_scc1 .Lhalf_y_%="]
    L += trip(8, y, vbase, prefetch_into=x)
    L += [f"s_sub_u32 s{CNT}, s{CNT}, 1",
          f"s_cmp_lg_u32 s{CNT}, 0",
          "s_cbranch_scc1 .Ltrip_x_%="]
    # odd number of groups: four more items, from the set that was filled last
    L += [".Lhalf_x_%=:",
          "s_bitcmp0_b32 %[ng], 0",
          "s_cbranch_scc1 .Ldone_%="]
    L += trip(4, x, vbase)
    L += ["s_branch .Ldone_%=",
          ".Lhalf_y_%=:",
          "s_bitcmp0_b32 %[ng], 0",
          "s_cbranch_scc1 .Ldone_%="]
    L += trip(4, y, vbase)
    L += [".Ldone_%=:"]

    body = "\n".join(f'        "{l}\\n\\t"' for l in L)
    vregs = list(range(vbase, vbase + 4 * (DEPTH + 1) + 1))
    sregs = [CNT, OFF] + list(range(SET["X"], SET["Y"] + 32))
    clobbers = ", ".join([f'"v{r}"' for r in vregs] + [f'"s{r}"' for r in sregs] + ['"scc"'])
    return f'''// all items of one pixel in the staged chunk: ng groups of four (ng >= 1), entries at `row`.
// Reads the table up to two groups past the last one it uses (the table carries spare groups).
// temps v{vregs[0]}..v{vregs[-1]}, s{CNT}, s{OFF}, s[{SET["X"]}:{SET["Y"] + 31}]
__device__ __forceinline__ void {name}(f2 &A, f2 &Q, f2 &C, f2 &R, const void *row, int ng, unsigned lane_addr) {{
    asm volatile(
{body}
        : [A] "+v"(A), [Q] "+v"(Q), [C] "+v"(C), [R] "+v"(R)
        : [ptr] "s"(row), [ng] "s"(ng), [lane] "v"(lane_addr)
        : {clobbers});
}}
'''


def main():
    out = ["// GENERATED by tools/gen_trip_asm.py -- do not edit.  See that script for the schedule.", ""]
    out.append(pixel_block("sweep_pixel_hi", 104))  # kernels with a 128-VGPR budget
    out.append(pixel_block("sweep_pixel_lo", 40))   # low temps, for shapes with few accumulators
    path = Path(__file__).resolve().parent.parent / "beamforming-lk_amd" / "csrc" / "das_fast_trip.inc"
    path.write_text("\n".join(out))
    print("wrote", path, sum(1 for _ in path.read_text().splitlines()), "lines")


if __name__ == "__main__":
    main()
