#!/usr/bin/env python3
"""Generate beamforming-lk_amd/csrc/das_fast_trip.inc: the hand-scheduled inner loop of the
fast sweep kernel (das_fast.hip).

An item (pixel x mic, one frame) costs
    v_add_u32 (LDS address) ; 2 x ds_read_b64 ; 4 x v_pk_fma_f32
and a "trip" is 8 consecutive items of one pixel (or 4: a pixel's last, odd group) whose table
entries (f, addr, g, pad: 4 dwords each) sit in 32 SGPRs.  Two SGPR sets ping-pong: while a trip
runs out of one set, two s_load_dwordx16 fill the other with the entries of the NEXT trip -- of
the same pixel, or the first trip of the next pixel, so that a wave meets the scalar-load latency
once per asm block, not once per pixel.

Schedule inside a trip: LDS reads run DEPTH items ahead of the FMAs that consume them; every wait
is a counted s_waitcnt lgkmcnt(n) with n = the number of YOUNGER LDS reads of this wave.  Scalar
loads share that counter and may return out of order, but since n never budgets for them a pending
scalar load can only make a wait longer, never let it pass early.  Each trip ends on lgkmcnt(0),
which is also what proves the other SGPR set has landed.

Block flavours:
  sweep_pixel_hi      one pixel per block
  sweep_quad_hi       four pixels per block (rows `stride` bytes apart in the table)
  sweep_quad_stamped  the same with s_memtime stamps (diagnostic builds only)

Fixed registers (all in the clobber list, so hipcc keeps nothing of its own there):
    SGPR  s[36:67] set X, s[68:99] set Y, s23..s31 and s35 loop state (s32..s34 are left alone: SP/FP/BP)
    VGPR  DEPTH+1 read slots of 4 registers + 1 address temp, from `vbase`
A block is self-contained: it begins with none of its own loads pending and ends drained, so the
compiler never sees a register with a load in flight.
"""
import os
from pathlib import Path

PRIO = int(os.environ.get("TRIP_PRIO", "5"))  # pair blocks: 3 rotation, 4 youngest-first, 5 alternate the two; others: 2 = four levels
PRIO_BASE = [0]  # +2 on odd pixels of a block: four levels, so that ties (decided by age) are rarer
DEPTH = int(os.environ.get("TRIP_DEPTH", "4"))  # items of LDS read-ahead (2*DEPTH <= 15: lgkmcnt is 4 bits)
SET = {"X": 36, "Y": 68}
OTHER = {"X": "Y", "Y": "X"}
# loop state: s[28:29] stamp t0, s[30:31] temp pair, s23 row offset, s24 groups left,
# s25 refill offset, s35 offset of the entries after the loaded ones
S_T0, S_T1, S_ROW, S_LEFT, S_PF, S_OFF = 28, 30, 23, 24, 25, 35  # not s32..s34: stack/frame/base pointer registers
S_PRIO = 27  # PRIO 3: rotating priority counter (starts at the wave's age rank on its SIMD)


S_RANK = 26  # the wave's age rank on its SIMD (0 = oldest), constant


def select_prio(sel, step, pmap=(0, 1, 2, 3)):
    """s_setprio takes an immediate, so the priority is chosen by a 4-way branch on s{sel} & 3 after
    adding `step` to it; pmap[k] = the priority a wave whose s{sel} is k takes."""
    L = []
    if step:
        L += [f"s_add_u32 s{sel}, s{sel}, {step}", f"s_and_b32 s{sel}, s{sel}, 3"]
    for k in range(3):
        L += [f"s_cmp_eq_u32 s{sel}, {k}", f"s_cbranch_scc1 .Lprio{k}_%=_{COUNTER[0]}"]
    L += [f"s_setprio {pmap[3]}", f"s_branch .Lprio_done_%=_{COUNTER[0]}"]
    for k in range(3):
        L += [f".Lprio{k}_%=_{COUNTER[0]}:", f"s_setprio {pmap[k]}"]
        if k < 2:
            L += [f"s_branch .Lprio_done_%=_{COUNTER[0]}"]
    L += [f".Lprio_done_%=_{COUNTER[0]}:"]
    COUNTER[0] += 1
    return L


def trip_prio(cur):
    """Waves of a SIMD are served by priority, then oldest first.  Left alone (measured, 4 waves per
    SIMD, one barrier per chunk) the oldest wave sweeps a chunk in 123k cycles and the youngest in
    217k, so every chunk ends with one starved wave running alone.  Equal-time rotation of the top
    priority (each trip the next wave) still leaves the oldest ahead (157k vs 206k), a static
    youngest-first order overshoots (228k vs 136k); alternating the two -- X trips rotate, Y trips
    youngest-first -- lands near balance."""
    if PRIO == 4:
        return select_prio(S_RANK, 0)
    if PRIO == 3 or cur == "X":
        return select_prio(S_PRIO, 1)
    return select_prio(S_RANK, 0)


COUNTER = [0]


def trip(n_items, cur, vbase, acc):
    """One trip over `n_items` entries held in set `cur`; the other set is refilled from table
    byte offset s{S_PF} (two s_load_dwordx16), after which s{S_OFF} = s{S_PF} + 128."""
    sbase = SET[cur]
    pbase = SET[OTHER[cur]]
    slots = DEPTH + 1
    addr_t = vbase + 4 * slots
    A, Q, C, R = acc

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
        return [f"v_pk_fma_f32 {A}, {fpair(i)}, {x}, {A} op_sel_hi:[0,1,1]",
                f"v_pk_fma_f32 {Q}, {gpair(i)}, {x}, {Q} op_sel_hi:[0,1,1]",
                f"v_pk_fma_f32 {C}, {fpair(i)}, {y}, {C} op_sel_hi:[0,1,1]",
                f"v_pk_fma_f32 {R}, {gpair(i)}, {y}, {R} op_sel_hi:[0,1,1]"]

    lines = []
    if PRIO:
        # Waves of a SIMD are served oldest first; left alone, the youngest is starved and every
        # chunk ends with that wave running alone while the others sit at the barrier (measured:
        # sweep 196k vs 247k cycles, 63k cycles of barrier wait for the oldest).  Alternating the
        # priority trip by trip (X trips high, Y trips low) time-slices the SIMD between waves at
        # different phases and roughly halves that skew.
        lines.append(f"s_setprio {(1 if cur == 'X' else 0) + PRIO_BASE[0]}")
    for i in range(min(DEPTH, n_items)):
        lines += issue(i)
    lines += [f"s_load_dwordx16 s[{pbase}:{pbase + 15}], %[ptr], s{S_PF}",
              f"s_add_u32 s{S_OFF}, s{S_PF}, 64",
              f"s_load_dwordx16 s[{pbase + 16}:{pbase + 31}], %[ptr], s{S_OFF}",
              f"s_add_u32 s{S_OFF}, s{S_PF}, 128"]
    for k in range(n_items):
        if k + DEPTH < n_items:
            lines += issue(k + DEPTH)
        younger = min(DEPTH, n_items - 1 - k)
        lines.append(f"s_waitcnt lgkmcnt({2 * younger})")
        lines += fma(k)
    return lines


def trip2(n_items, cur, vbase, acc, depth):
    """Frame-pair flavour of trip(): the LDS image holds (frame a, frame b) sample-interleaved, so one
    ds_read_b64 returns one sample of both frames and the two lanes of a packed FMA are the two frames.
    Lane l owns samples l, l+64, l+128, l+192: per item one address add, FOUR ds_read_b64 (512 bytes
    apart) and EIGHT v_pk_fma_f32 into acc = (A0..A3, Q0..Q3):  A_k += f * x_k  (-> out[l+64k]),
    Q_k += g * x_k  (-> out[l+64k-1]; the one-sample skew is undone once per pixel)."""
    sbase = SET[cur]
    pbase = SET[OTHER[cur]]
    slots = depth + 1
    addr_t = vbase + 8 * slots

    def fpair(i):
        return f"s[{sbase + 4 * i}:{sbase + 4 * i + 1}]"

    def gpair(i):
        return f"s[{sbase + 4 * i + 2}:{sbase + 4 * i + 3}]"

    def issue(i):
        s = vbase + 8 * (i % slots)
        L = [f"v_add_u32 v{addr_t}, s{sbase + 4 * i + 1}, %[lane]"]
        for k in range(4):
            off = f" offset:{512 * k}" if k else ""
            L.append(f"ds_read_b64 v[{s + 2 * k}:{s + 2 * k + 1}], v{addr_t}{off}")
        return L

    def fma(i):
        s = vbase + 8 * (i % slots)
        L = []
        for k in range(4):
            x = f"v[{s + 2 * k}:{s + 2 * k + 1}]"
            L.append(f"v_pk_fma_f32 {acc[k]}, {fpair(i)}, {x}, {acc[k]} op_sel_hi:[0,1,1]")
            L.append(f"v_pk_fma_f32 {acc[4 + k]}, {gpair(i)}, {x}, {acc[4 + k]} op_sel_hi:[0,1,1]")
        return L

    lines = []
    if PRIO >= 3:
        lines += trip_prio(cur)
    elif PRIO:
        lines.append(f"s_setprio {(1 if cur == 'X' else 0) + PRIO_BASE[0]}")
    for i in range(min(depth, n_items)):
        lines += issue(i)
    lines += [f"s_load_dwordx16 s[{pbase}:{pbase + 15}], %[ptr], s{S_PF}",
              f"s_add_u32 s{S_OFF}, s{S_PF}, 64",
              f"s_load_dwordx16 s[{pbase + 16}:{pbase + 31}], %[ptr], s{S_OFF}",
              f"s_add_u32 s{S_OFF}, s{S_PF}, 128"]
    for k in range(n_items):
        if k + depth < n_items:
            lines += issue(k + depth)
        younger = min(depth, n_items - 1 - k)
        lines.append(f"s_waitcnt lgkmcnt({4 * younger})")
        lines += fma(k)
    return lines


def next_pixel_setup(last_pixel):
    if last_pixel:
        return []
    return [f"s_add_u32 s{S_ROW}, s{S_ROW}, %[stride]",
            f"s_mov_b32 s{S_LEFT}, %[ng]"]  # s{S_OFF} was set by the refill: new row + 128


def pixel_code(j, n_pix, vbase, acc, pair_depth=0):
    """State machine of pixel j.  On entry at LPj_X / LPj_Y the named set holds the pixel's next
    entries, s{S_LEFT} = groups of four left (>= 1), s{S_OFF} = table offset of the entries after
    those, s{S_ROW} = table offset of this pixel's row."""
    last_pixel = j == n_pix - 1
    PRIO_BASE[0] = 2 * (j & 1) if PRIO >= 2 else 0
    L = []
    for cur in ("X", "Y"):
        oth = OTHER[cur]
        nxt_entry = ".Ldone_%=" if last_pixel else f".LP{j + 1}_{oth}_%="
        L += [f".LP{j}_{cur}_%=:",
              f"s_cmp_lt_u32 s{S_LEFT}, 2",
              f"s_cbranch_scc1 .LP{j}_{cur}_half_%=",
              # full trip; if it is this pixel's last, the refill comes from the next pixel's row
              f"s_sub_u32 s{S_LEFT}, s{S_LEFT}, 2",
              f"s_add_u32 s{S_T1}, s{S_ROW}, %[stride]",
              f"s_cmp_eq_u32 s{S_LEFT}, 0",
              f"s_cselect_b32 s{S_PF}, s{S_T1}, s{S_OFF}"]
        L += trip2(8, cur, vbase, acc, pair_depth) if pair_depth else trip(8, cur, vbase, acc)
        L += [f"s_cmp_lg_u32 s{S_LEFT}, 0",
              f"s_cbranch_scc1 .LP{j}_{oth}_%="]
        # pixel finished on a full trip: the other set holds the next pixel's first entries
        L += next_pixel_setup(last_pixel) + [f"s_branch {nxt_entry}"]
        L += [f".LP{j}_{cur}_half_%=:",
              f"s_add_u32 s{S_PF}, s{S_ROW}, %[stride]"]
        L += trip2(4, cur, vbase, acc, pair_depth) if pair_depth else trip(4, cur, vbase, acc)
        L += next_pixel_setup(last_pixel) + [f"s_branch {nxt_entry}"]
    return L


def block(name, n_pix, vbase, stamp=False, pair_depth=0):
    x = SET["X"]
    if pair_depth:  # frame-pair flavour: 8 accumulators per pixel, named A0..A3, Q0..Q3 per pixel
        accs = [tuple(f"%[P{j}{n}{k}]" for n in "AQ" for k in range(4)) for j in range(n_pix)]
    else:
        accs = [(f"%[A{j}]", f"%[Q{j}]", f"%[C{j}]", f"%[R{j}]") for j in range(n_pix)]
    L = []
    if stamp:
        L += [f"s_memtime s[{S_T0}:{S_T0 + 1}]", "s_waitcnt lgkmcnt(0)"]
    if PRIO >= 3 and pair_depth:
        L += [f"s_mov_b32 s{S_PRIO}, %[rank]", f"s_mov_b32 s{S_RANK}, %[rank]"]
    L += [f"s_load_dwordx16 s[{x}:{x + 15}], %[ptr], 0x0",
          f"s_load_dwordx16 s[{x + 16}:{x + 31}], %[ptr], 0x40",
          f"s_mov_b32 s{S_LEFT}, %[ng]",
          f"s_mov_b32 s{S_ROW}, 0",
          f"s_movk_i32 s{S_OFF}, 0x80",
          "s_waitcnt lgkmcnt(0)"]
    if stamp:
        L += [f"s_memtime s[{S_T1}:{S_T1 + 1}]", "s_waitcnt lgkmcnt(0)",
              f"s_sub_u32 %[t_wait], s{S_T1}, s{S_T0}"]
    for j in range(n_pix):
        L += pixel_code(j, n_pix, vbase, accs[j], pair_depth)
    L += [".Ldone_%=:"]
    if PRIO:
        L += [f"s_setprio {BLOCK_END_PRIO}"]
    if stamp:
        L += [f"s_memtime s[{S_T1}:{S_T1 + 1}]", "s_waitcnt lgkmcnt(0)",
              f"s_sub_u32 %[t_all], s{S_T1}, s{S_T0}"]

    body = "\n".join(f'        "{l}\\n\\t"' for l in L)
    vregs = list(range(vbase, vbase + (8 * (pair_depth + 1) + 1 if pair_depth else 4 * (DEPTH + 1) + 1)))
    sregs = sorted({S_RANK, S_PRIO, S_ROW, S_LEFT, S_PF, S_OFF, S_T0, S_T0 + 1, S_T1, S_T1 + 1}) + list(range(SET["X"], SET["Y"] + 32))
    clobbers = ", ".join([f'"v{r}"' for r in vregs] + [f'"s{r}"' for r in sregs] + ['"scc"'])
    if pair_depth:
        acc_params = ", ".join(f"f2 (&P{j})[8]" for j in range(n_pix))
        acc_ops = ", ".join(f'[P{j}{n}{k}] "+v"(P{j}[{4 * "AQ".index(n) + k}])' for j in range(n_pix) for n in "AQ" for k in range(4))
    else:
        acc_params = ", ".join(f"f2 &A{j}, f2 &Q{j}, f2 &C{j}, f2 &R{j}" for j in range(n_pix))
        acc_ops = ", ".join(f'[A{j}] "+v"(A{j}), [Q{j}] "+v"(Q{j}), [C{j}] "+v"(C{j}), [R{j}] "+v"(R{j})' for j in range(n_pix))
    stamp_params = ", unsigned &t_wait, unsigned &t_all" if stamp else ""
    stamp_ops = ', [t_wait] "=&s"(t_wait), [t_all] "=&s"(t_all)' if stamp else ""  # this call's cycles
    return f'''// {n_pix} pixel(s) of the staged chunk, ng groups of four items each (ng >= 1); pixel j's entries
// start at `row` + j * stride bytes.  Reads the table up to one row start + two groups past the
// last row it sweeps (the table carries spare groups).
// temps v{vregs[0]}..v{vregs[-1]}, s{sregs[0]}..s{sregs[-1]}
__device__ __forceinline__ void {name}({acc_params}, const void *row, int stride, int ng, unsigned lane_addr{", int rank" if pair_depth else ""}{stamp_params}) {{
    asm volatile(
{body}
        : {acc_ops}{stamp_ops}
        : [ptr] "s"(row), [stride] "s"(stride), [ng] "s"(ng), [lane] "v"(lane_addr){', [rank] "s"(rank)' if pair_depth else ""}
        : {clobbers});
}}
'''


def block_shared(name, vbase, stamp=False):
    """Frame-pair sweep of TWO pixels in mic-major order with the sample reads shared at run time: for
    every mic the block handles pixel A's item, then pixel B's; when B's entry carries the same LDS
    address as A's (the two pixels' integer delays coincide: most mics for neighbouring pixels) B's
    eight packed FMAs take their samples from the registers A's item loaded, and B issues neither an
    address add nor LDS reads.  Same items, same per-pixel order as sweep_duo_pairs, so the sums are
    bit-identical; the table format is unchanged.

    Trips of 4 mics.  SGPR sets: A0/B0 (entries of pixel A / B for this trip) and A1/B1 (the next trip's,
    being fetched), 16 SGPRs each.  VGPR slots: two for A's samples (this mic / next mic), ONE for B's.
    Stage s of a trip:  issue A[s+1];  wait until at most A[s+1]'s four reads are in flight (LDS returns
    in order: A[s] and B[s] are then in);  8 FMAs of A[s];  8 FMAs of B[s] from A's slot or B's;  issue
    B[s+1] unless shared.  At most 4 + 4 + 4 LDS reads and 2 scalar loads are in flight (lgkmcnt <= 15)."""
    setA, setB = (36, 68), (52, 84)
    accA = tuple(f"%[P0{n}{k}]" for n in "AQ" for k in range(4))
    accB = tuple(f"%[P1{n}{k}]" for n in "AQ" for k in range(4))
    slotA = (vbase, vbase + 8)
    slotB = vbase + 16
    addr_t = vbase + 24

    def reads(slot, addr_sgpr):
        L = [f"v_add_u32 v{addr_t}, s{addr_sgpr}, %[lane]"]
        for k in range(4):
            off = f" offset:{512 * k}" if k else ""
            L.append(f"ds_read_b64 v[{slot + 2 * k}:{slot + 2 * k + 1}], v{addr_t}{off}")
        return L

    def fmas(acc, base, i, slot):
        L = []
        for k in range(4):
            x = f"v[{slot + 2 * k}:{slot + 2 * k + 1}]"
            L.append(f"v_pk_fma_f32 {acc[k]}, s[{base + 4 * i}:{base + 4 * i + 1}], {x}, {acc[k]} op_sel_hi:[0,1,1]")
            L.append(f"v_pk_fma_f32 {acc[4 + k]}, s[{base + 4 * i + 2}:{base + 4 * i + 3}], {x}, {acc[4 + k]} op_sel_hi:[0,1,1]")
        return L

    def uid():
        COUNTER[0] += 1
        return f"%=_{COUNTER[0]}"

    cold = []  # the not-shared paths, kept out of line so that the common path takes no branch

    def issue_b(sA, sB, i):
        u = uid()
        cold.extend([f".Lread{u}:"] + reads(slotB, sB + 4 * i + 1) + [f"s_branch .Lreadback{u}"])
        return [f"s_cmp_lg_u32 s{sA + 4 * i + 1}, s{sB + 4 * i + 1}", f"s_cbranch_scc1 .Lread{u}", f".Lreadback{u}:"]

    def trip_s(par):
        sA, sB = setA[par], setB[par]
        nA, nB = setA[1 - par], setB[1 - par]
        L = trip_prio("X" if par == 0 else "Y") if PRIO >= 3 else []
        L += reads(slotA[0], sA + 1)
        L += issue_b(sA, sB, 0)
        L += [f"s_load_dwordx16 s[{nA}:{nA + 15}], %[ptr], s{S_PF}",
              f"s_add_u32 s{S_OFF}, s{S_PF}, %[stride]",
              f"s_load_dwordx16 s[{nB}:{nB + 15}], %[ptr], s{S_OFF}",
              f"s_add_u32 s{S_PF}, s{S_PF}, 64"]
        for st in range(4):
            if st < 3:
                L += reads(slotA[(st + 1) % 2], sA + 4 * (st + 1) + 1)
            L.append(f"s_waitcnt lgkmcnt({4 if st < 3 else 0})")
            L += fmas(accA, sA, st, slotA[st % 2])
            u = uid()
            cold.extend([f".Lown{u}:"] + fmas(accB, sB, st, slotB) + [f"s_branch .Lfmadone{u}"])
            L += [f"s_cmp_lg_u32 s{sA + 4 * st + 1}, s{sB + 4 * st + 1}", f"s_cbranch_scc1 .Lown{u}"]
            L += fmas(accB, sB, st, slotA[st % 2]) + [f".Lfmadone{u}:"]
            if st < 3:
                L += issue_b(sA, sB, st + 1)
        return L

    L = []
    if stamp:
        L += [f"s_memtime s[{S_T0}:{S_T0 + 1}]", "s_waitcnt lgkmcnt(0)"]
    if PRIO >= 3:
        L += [f"s_mov_b32 s{S_PRIO}, %[rank]", f"s_mov_b32 s{S_RANK}, %[rank]"]
    L += [f"s_load_dwordx16 s[{setA[0]}:{setA[0] + 15}], %[ptr], 0x0",
          f"s_load_dwordx16 s[{setB[0]}:{setB[0] + 15}], %[ptr], %[stride]",
          f"s_mov_b32 s{S_LEFT}, %[ng]",
          f"s_movk_i32 s{S_PF}, 0x40",
          "s_waitcnt lgkmcnt(0)"]
    if stamp:
        L += [f"s_memtime s[{S_T1}:{S_T1 + 1}]", "s_waitcnt lgkmcnt(0)",
              f"s_sub_u32 %[t_wait], s{S_T1}, s{S_T0}"]
    L += [".LT0_%=:"] + trip_s(0)
    L += [f"s_sub_u32 s{S_LEFT}, s{S_LEFT}, 1", f"s_cmp_eq_u32 s{S_LEFT}, 0", "s_cbranch_scc1 .Ldone_%="]
    L += trip_s(1)
    L += [f"s_sub_u32 s{S_LEFT}, s{S_LEFT}, 1", f"s_cmp_lg_u32 s{S_LEFT}, 0", "s_cbranch_scc1 .LT0_%="]
    L += ["s_branch .Ldone_%="] + cold + [".Ldone_%=:"]
    if PRIO:
        L += [f"s_setprio {BLOCK_END_PRIO}"]
    if stamp:
        L += [f"s_memtime s[{S_T1}:{S_T1 + 1}]", "s_waitcnt lgkmcnt(0)",
              f"s_sub_u32 %[t_all], s{S_T1}, s{S_T0}"]
    body = "\n".join(f'        "{l}\\n\\t"' for l in L)
    vregs = list(range(vbase, vbase + 25))
    sregs = sorted({S_RANK, S_PRIO, S_LEFT, S_PF, S_OFF, S_T0, S_T0 + 1, S_T1, S_T1 + 1}) + list(range(36, 100))
    clobbers = ", ".join([f'"v{r}"' for r in vregs] + [f'"s{r}"' for r in sregs] + ['"scc"'])
    acc_ops = ", ".join(f'[P{j}{n}{k}] "+v"(P{j}[{4 * "AQ".index(n) + k}])' for j in range(2) for n in "AQ" for k in range(4))
    stamp_params = ", unsigned &t_wait, unsigned &t_all" if stamp else ""
    stamp_ops = ', [t_wait] "=&s"(t_wait), [t_all] "=&s"(t_all)' if stamp else ""
    return f'''// Two pixels of the staged chunk in mic-major order, ng groups of four mics each (ng >= 1); pixel A's
// entries start at `row`, pixel B's at `row` + stride bytes.  Sample reads are shared whenever the two
// entries of a mic carry the same LDS address.  Reads the table up to one group past each row's end.
// temps v{vregs[0]}..v{vregs[-1]}, s{sregs[0]}..s{sregs[-1]}
__device__ __forceinline__ void {name}(f2 (&P0)[8], f2 (&P1)[8], const void *row, int stride, int ng, unsigned lane_addr, int rank{stamp_params}) {{
    asm volatile(
{body}
        : {acc_ops}{stamp_ops}
        : [ptr] "s"(row), [stride] "s"(stride), [ng] "s"(ng), [lane] "v"(lane_addr), [rank] "s"(rank)
        : {clobbers});
}}
'''


def block_exact_shared(name, vbase):
    """Reference-order sweep (AWPU_MATH_F32_EXACT) of TWO pixels on the frame-pair layout, RAW samples.

    Per pixel, mic and sample the operations are delay.cpp:19-25's, in its order, on both frames of the pair at once:
        d = cur - next            v_pk_add_f32 with the second operand negated (IEEE: a - b == a + (-b), bit for bit)
        t = fma(frac, d, next)    v_pk_fma_f32, frac a scalar operand (the reference's `fraction`)
        out = out + t             v_pk_add_f32
    and mics are visited in table order (= antenna.index[] order, mimo.cpp:124-130): the pre-epilogue sums equal the
    reference's bit for bit.  The block is block_shared's schedule: mic-major over pixel A then pixel B, trips of 4 mics,
    entry sets of 16 SGPRs per pixel ping-ponging, B's entry compared with A's -- when the two carry the same LDS address
    (same integer delay: most mics of vertical neighbours) B takes A's `next` and `d = cur - next` registers (d does not
    depend on the fraction) and issues no address add, no LDS read and no subtraction: 8 packed VALU instructions instead
    of 13.  Lane l owns samples l + 64 k; an item reads cur_k (offset 512 k) and next_k (512 k + 8): 8 ds_read_b64.
    VGPR slots of 16 registers (cur0..3 then next0..3): two for A (this mic / next mic), one for B; 4 register pairs for t.
    In flight at most 8 (A, next mic) + 8 (B, next mic) LDS reads + 2 scalar loads: beyond lgkmcnt's 4 bits the issue
    simply stalls; every wait names only the reads YOUNGER than the data it proves (8: A's next mic), so it cannot pass early."""
    setA, setB = (36, 68), (52, 84)
    accA = tuple(f"%[P0{k}]" for k in range(4))
    accB = tuple(f"%[P1{k}]" for k in range(4))
    slotA = (vbase, vbase + 16)
    slotB = vbase + 32
    tmp = vbase + 48
    addr_t = vbase + 56

    def reads(slot, addr_sgpr):
        L = [f"v_add_u32 v{addr_t}, s{addr_sgpr}, %[lane]"]
        for k in range(4):
            off = f" offset:{512 * k}" if k else ""
            L.append(f"ds_read_b64 v[{slot + 2 * k}:{slot + 2 * k + 1}], v{addr_t}{off}")
            L.append(f"ds_read_b64 v[{slot + 8 + 2 * k}:{slot + 9 + 2 * k}], v{addr_t} offset:{512 * k + 8}")
        return L

    def diffs(slot):  # d_k = cur_k - next_k, in place of cur_k
        return [f"v_pk_add_f32 v[{slot + 2 * k}:{slot + 2 * k + 1}], v[{slot + 2 * k}:{slot + 2 * k + 1}], "
                f"v[{slot + 8 + 2 * k}:{slot + 9 + 2 * k}] neg_lo:[0,1] neg_hi:[0,1]" for k in range(4)]

    def terms(acc, base, i, slot):  # t_k = fma(frac, d_k, next_k); out_k += t_k
        L = []
        for k in range(4):
            L.append(f"v_pk_fma_f32 v[{tmp + 2 * k}:{tmp + 2 * k + 1}], s[{base + 4 * i}:{base + 4 * i + 1}], "
                     f"v[{slot + 2 * k}:{slot + 2 * k + 1}], v[{slot + 8 + 2 * k}:{slot + 9 + 2 * k}] op_sel_hi:[0,1,1]")
        for k in range(4):
            L.append(f"v_pk_add_f32 {acc[k]}, {acc[k]}, v[{tmp + 2 * k}:{tmp + 2 * k + 1}]")
        return L

    def uid():
        COUNTER[0] += 1
        return f"%=_{COUNTER[0]}"

    cold = []

    def issue_b(sA, sB, i):
        u = uid()
        cold.extend([f".Lread{u}:"] + reads(slotB, sB + 4 * i + 1) + [f"s_branch .Lreadback{u}"])
        return [f"s_cmp_lg_u32 s{sA + 4 * i + 1}, s{sB + 4 * i + 1}", f"s_cbranch_scc1 .Lread{u}", f".Lreadback{u}:"]

    def trip_s(par):
        sA, sB = setA[par], setB[par]
        nA, nB = setA[1 - par], setB[1 - par]
        L = trip_prio("X" if par == 0 else "Y") if PRIO >= 3 else []
        L += reads(slotA[0], sA + 1)
        L += issue_b(sA, sB, 0)
        L += [f"s_load_dwordx16 s[{nA}:{nA + 15}], %[ptr], s{S_PF}",
              f"s_add_u32 s{S_OFF}, s{S_PF}, %[stride]",
              f"s_load_dwordx16 s[{nB}:{nB + 15}], %[ptr], s{S_OFF}",
              f"s_add_u32 s{S_PF}, s{S_PF}, 64"]
        for st in range(4):
            if st < 3:
                L += reads(slotA[(st + 1) % 2], sA + 4 * (st + 1) + 1)
            L.append(f"s_waitcnt lgkmcnt({8 if st < 3 else 0})")
            L += diffs(slotA[st % 2])
            L += terms(accA, sA, st, slotA[st % 2])
            u = uid()
            cold.extend([f".Lown{u}:"] + diffs(slotB) + terms(accB, sB, st, slotB) + [f"s_branch .Ltermsdone{u}"])
            L += [f"s_cmp_lg_u32 s{sA + 4 * st + 1}, s{sB + 4 * st + 1}", f"s_cbranch_scc1 .Lown{u}"]
            L += terms(accB, sB, st, slotA[st % 2]) + [f".Ltermsdone{u}:"]
            if st < 3:
                L += issue_b(sA, sB, st + 1)
        return L

    L = []
    if PRIO >= 3:
        L += [f"s_mov_b32 s{S_PRIO}, %[rank]", f"s_mov_b32 s{S_RANK}, %[rank]"]
    L += [f"s_load_dwordx16 s[{setA[0]}:{setA[0] + 15}], %[ptr], 0x0",
          f"s_load_dwordx16 s[{setB[0]}:{setB[0] + 15}], %[ptr], %[stride]",
          f"s_mov_b32 s{S_LEFT}, %[ng]",
          f"s_movk_i32 s{S_PF}, 0x40",
          "s_waitcnt lgkmcnt(0)"]
    L += [".LT0_%=:"] + trip_s(0)
    L += [f"s_sub_u32 s{S_LEFT}, s{S_LEFT}, 1", f"s_cmp_eq_u32 s{S_LEFT}, 0", "s_cbranch_scc1 .Ldone_%="]
    L += trip_s(1)
    L += [f"s_sub_u32 s{S_LEFT}, s{S_LEFT}, 1", f"s_cmp_lg_u32 s{S_LEFT}, 0", "s_cbranch_scc1 .LT0_%="]
    L += ["s_branch .Ldone_%="] + cold + [".Ldone_%=:"]
    if PRIO:
        L += [f"s_setprio {BLOCK_END_PRIO}"]
    body = "\n".join(f'        "{l}\\n\\t"' for l in L)
    vregs = list(range(vbase, vbase + 57))
    sregs = sorted({S_RANK, S_PRIO, S_LEFT, S_PF, S_OFF}) + list(range(36, 100))
    clobbers = ", ".join([f'"v{r}"' for r in vregs] + [f'"s{r}"' for r in sregs] + ['"scc"'])
    acc_ops = ", ".join(f'[P{j}{k}] "+v"(P{j}[{k}])' for j in range(2) for k in range(4))
    return f'''// Reference-order sweep of two pixels of the staged chunk (raw frame-pair samples) in mic-major order, ng groups of four
// mics each (ng >= 1); pixel A's entries start at `row`, pixel B's at `row` + stride bytes; P[k] = out[l + 64 k] of both
// frames.  delay.cpp:19-25's operations in its order; the reads and cur - next are shared whenever the two entries of a
// mic carry the same LDS address.  Reads the table up to one group past each row's end.
// temps v{vregs[0]}..v{vregs[-1]}, s{sregs[0]}..s{sregs[-1]}
__device__ __forceinline__ void {name}(f2 (&P0)[4], f2 (&P1)[4], const void *row, int stride, int ng, unsigned lane_addr, int rank) {{
    asm volatile(
{body}
        : {acc_ops}
        : [ptr] "s"(row), [stride] "s"(stride), [ng] "s"(ng), [lane] "v"(lane_addr), [rank] "s"(rank)
        : {clobbers});
}}
'''


EXQ_ACC = 26   # block_exact_quad: out[] of the four pixels, 8 registers each, pinned at v26..v57
EXQ_TMP = 58   # its temps: two slots for the reference pixel's samples, one for pixel 0's own, one that pixels 2 and 3 share, t, address


def block_exact_quad(name):
    """Reference-order sweep (AWPU_MATH_F32_EXACT) of FOUR vertically adjacent pixels on the frame-pair layout, raw samples.

    Arithmetic per pixel, mic and sample exactly block_exact_shared's (delay.cpp:19-25 in its order, mics in table order):
    d = cur - next; t = fma(frac, d, next); out += t -- so the pre-epilogue sums stay bit-identical to the reference's.  What the
    four pixels of a column share is the SAMPLES and the difference d (neither depends on the fraction): the second pixel is
    the reference, its reads (8 ds_read_b64: cur and next of the lane's four samples) and its d serve every pixel whose entry
    carries the same LDS address -- 8 packed VALU instructions per pixel and mic, +4 (its own d) and 8 reads for a pixel whose
    integer delay differs.  Schedule and table are block_quad's: quad-major entries [group][pixel][mic] x (fraction, address),
    trips of 4 mics, two SGPR sets of 32 ping-ponging; per mic the reference's reads for the NEXT mic go out first, one
    counted wait (8 younger reads) proves this mic's samples, then pixels 3, 2, 0 -- each followed by the conditional read
    of its own samples for the next mic -- and the reference.  Pixels 2 and 3 share one slot (a delay is monotone down a
    column: when both differ from the reference they almost always carry the same address; pixel 2 reads on the spot in the
    rare other cases).  Slots are 16 registers: cur0..3 (d in their place once formed), next0..3."""
    O = [EXQ_ACC + 8 * p for p in range(4)]
    R = [EXQ_TMP, EXQ_TMP + 16]
    X0, X23 = EXQ_TMP + 32, EXQ_TMP + 48
    TT = EXQ_TMP + 64
    addr_t = EXQ_TMP + 68
    E = (36, 68)
    S_TMP, S_PF_, S_LEFT_ = 22, 23, 24

    def f_of(base, p, i):
        return base + 8 * p + 2 * i

    def a_of(base, p, i):
        return base + 8 * p + 2 * i + 1

    def pair(r, k):
        return f"v[{r + 2 * k}:{r + 2 * k + 1}]"

    def reads(slot, addr_sgpr):
        L = [f"v_add_u32 v{addr_t}, s{addr_sgpr}, %[lane]"]
        for k in range(4):
            off = f" offset:{512 * k}" if k else ""
            L.append(f"ds_read_b64 {pair(slot, k)}, v{addr_t}{off}")
            L.append(f"ds_read_b64 {pair(slot + 8, k)}, v{addr_t} offset:{512 * k + 8}")
        return L

    def diffs(slot):  # d_k = cur_k - next_k, in place of cur_k
        return [f"v_pk_add_f32 {pair(slot, k)}, {pair(slot, k)}, {pair(slot + 8, k)} neg_lo:[0,1] neg_hi:[0,1]" for k in range(4)]

    def terms(p, base, i, slot):  # t_k = fma(frac, d_k, next_k); out_k += t_k   (two t registers, used alternately)
        fs = f"s[{f_of(base, p, i)}:{f_of(base, p, i) + 1}]"
        L = []
        for k0 in (0, 2):
            for k in (k0, k0 + 1):
                L.append(f"v_pk_fma_f32 {pair(TT, k - k0)}, {fs}, {pair(slot, k)}, {pair(slot + 8, k)} op_sel_hi:[0,1,1]")
            for k in (k0, k0 + 1):
                L.append(f"v_pk_add_f32 {pair(O[p], k)}, {pair(O[p], k)}, {pair(TT, k - k0)}")
        return L

    def uid():
        COUNTER[0] += 1
        return f"%=_{COUNTER[0]}"

    cold = []

    def maybe_read(p, slot, base, i):
        """pixel p's own reads for the mic whose entries sit at (base, i), unless it shares the reference's samples"""
        u = uid()
        cold.extend([f".Leread{u}:"] + reads(slot, a_of(base, p, i)) + [f"s_branch .Lereadback{u}"])
        return [f"s_cmp_lg_u32 s{a_of(base, p, i)}, s{a_of(base, REF, i)}", f"s_cbranch_scc1 .Leread{u}", f".Lereadback{u}:"]

    def pixel0(base, i, rslot):
        u = uid()
        cold.extend([f".Leown{u}:"] + diffs(X0) + terms(0, base, i, X0) + [f"s_branch .Ledone{u}"])
        return ([f"s_cmp_lg_u32 s{a_of(base, 0, i)}, s{a_of(base, REF, i)}", f"s_cbranch_scc1 .Leown{u}"] + terms(0, base, i, rslot) +
                [f".Ledone{u}:"])

    def pixels_23(base, i, rslot):
        """X23 holds pixel 3's samples whenever its address differs from the reference's (requested a mic ahead); pixel 2 takes
        the reference's, or pixel 3's when it carries pixel 3's address, or reads into X23 on the spot once pixel 3 is done."""
        u = uid()
        a1, a2, a3 = a_of(base, REF, i), a_of(base, 2, i), a_of(base, 3, i)
        spot = reads(X23, a2) + ["s_waitcnt lgkmcnt(0)"] + diffs(X23) + terms(2, base, i, X23)
        cold.extend(
            [f".Le3diff{u}:"] + diffs(X23) + terms(3, base, i, X23) +
            [f"s_cmp_eq_u32 s{a2}, s{a3}", f"s_cbranch_scc1 .Letog{u}", f"s_cmp_lg_u32 s{a2}, s{a1}", f"s_cbranch_scc1 .Letwo{u}"] +
            terms(2, base, i, rslot) + [f"s_branch .Le23done{u}"] +          # a-a-a-b
            [f".Letog{u}:"] + terms(2, base, i, X23) + [f"s_branch .Le23done{u}"] +  # a-a-b-b: pixel 3's samples and d serve pixel 2
            [f".Letwo{u}:"] + spot + [f"s_branch .Le23done{u}"] +                # a-a-b-c
            [f".Le2odd{u}:"] + spot + [f"s_branch .Le23done{u}"])               # a-a-b-a (not monotone: rare)
        return ([f"s_cmp_lg_u32 s{a3}, s{a1}", f"s_cbranch_scc1 .Le3diff{u}"] + terms(3, base, i, rslot) +
                [f"s_cmp_lg_u32 s{a2}, s{a1}", f"s_cbranch_scc1 .Le2odd{u}"] + terms(2, base, i, rslot) + [f".Le23done{u}:"])

    def load_set(base, off, literal=False):
        if literal:
            return [f"s_load_dwordx16 s[{base}:{base + 15}], %[ptr], {hex(off)}",
                    f"s_load_dwordx16 s[{base + 16}:{base + 31}], %[ptr], {hex(off + 64)}"]
        return [f"s_load_dwordx16 s[{base}:{base + 15}], %[ptr], s{off}",
                f"s_add_u32 s{S_TMP}, s{off}, 64",
                f"s_load_dwordx16 s[{base + 16}:{base + 31}], %[ptr], s{S_TMP}"]

    def trip_q(par):
        cur, nxt = E[par], E[1 - par]
        L = select_prio(S_PRIO, 1) if par == 0 else select_prio(S_RANK, 0)  # rotation / youngest first, trip by trip
        L += load_set(nxt, S_PF_) + [f"s_add_u32 s{S_PF_}, s{S_PF_}, 128"]
        for st in range(4):
            rslot = R[st & 1]
            if st < 3:
                L += reads(R[(st + 1) & 1], a_of(cur, REF, st + 1))
                L.append("s_waitcnt lgkmcnt(8)")  # all but the eight reads just issued: this mic's samples are in
                nbase, ni = cur, st + 1
            else:
                L.append("s_waitcnt lgkmcnt(0)")  # this mic's samples, and the next trip's entries
                L += reads(R[0], a_of(nxt, REF, 0))
                nbase, ni = nxt, 0
            L += diffs(rslot)
            L += pixels_23(cur, st, rslot) + maybe_read(3, X23, nbase, ni)
            L += pixel0(cur, st, rslot) + maybe_read(0, X0, nbase, ni)
            L += terms(REF, cur, st, rslot)
        return L

    L = [f"s_mov_b32 s{S_PRIO}, %[rank]", f"s_mov_b32 s{S_RANK}, %[rank]"]
    L += load_set(E[0], 0, literal=True)
    L += [f"s_mov_b32 s{S_LEFT_}, %[ng]", f"s_movk_i32 s{S_PF_}, 0x80", "s_waitcnt lgkmcnt(0)"]
    L += reads(R[0], a_of(E[0], REF, 0))
    L += maybe_read(3, X23, E[0], 0) + maybe_read(0, X0, E[0], 0)
    L += [".LE0_%=:"] + trip_q(0)
    L += [f"s_sub_u32 s{S_LEFT_}, s{S_LEFT_}, 1", f"s_cmp_eq_u32 s{S_LEFT_}, 0", "s_cbranch_scc1 .LEdone_%="]
    L += trip_q(1)
    L += [f"s_sub_u32 s{S_LEFT_}, s{S_LEFT_}, 1", f"s_cmp_lg_u32 s{S_LEFT_}, 0", "s_cbranch_scc1 .LE0_%="]
    L += ["s_branch .LEdone_%="] + cold + [".LEdone_%=:", "s_waitcnt lgkmcnt(0)"]  # (the reads issued for a trip that does not come)
    L += [f"s_setprio {BLOCK_END_PRIO}"]
    body = "\n".join(f'        "{l}\\n\\t"' for l in L)
    vregs = list(range(EXQ_TMP, addr_t + 1))
    sregs = sorted({S_TMP, S_PF_, S_LEFT_, S_RANK, S_PRIO}) + list(range(36, 100))
    clobbers = ", ".join([f'"v{r}"' for r in vregs] + [f'"s{r}"' for r in sregs] + ['"scc"'])
    acc_params = ", ".join(f"f8 &O{p}" for p in range(4))
    acc_ops = ", ".join(f'"+{{v[{O[p]}:{O[p] + 7}]}}"(O{p})' for p in range(4))
    return f"""// Reference-order sweep of four vertically adjacent pixels of the staged chunk (raw frame-pair samples), ng groups of four
// mics each (ng >= 1): see tools/gen_trip_asm.py, block_exact_quad.  `row` = the quad's entries of the chunk's first group in
// the quad-major table ([group][pixel][mic] x (fraction, address)); reads one group past the last.  O_p = out[l + 64 k] of
// pixel p, both frames, pinned at v[{O[0]}+8p ..]; temps v{vregs[0]}..v{vregs[-1]}, s{sregs[0]}..s{sregs[-1]}.
__device__ __forceinline__ void {name}({acc_params}, const void *row, int ng, unsigned lane_addr, int rank) {{
    asm volatile(
{body}
        : {acc_ops}
        : [ptr] "s"(row), [ng] "s"(ng), [lane] "v"(lane_addr), [rank] "s"(rank)
        : {clobbers});
}}
"""


ND_ACC = 10    # block_exact_nd: out[] of quad q, pixel p pinned at v[ND_ACC + 32 q + 8 p ..+7]
ND_TMP = 74    # its 52 temps (two 16-register slots for the reference pixel's elements, one for whichever pixel leaves it, t, address)
ND_PPT_H = int(os.environ.get("ND_PPT_H", "1"))  # refill pieces per trip of the single-frame blocks, in units of 16 KiB
ND_TIMING = os.environ.get("ND_TIMING", "")  # tuning builds only (see the end of block_exact_nd)
ND_PRIO = int(os.environ.get("ND_PRIO", "5"))  # tuning builds only: the wave-priority scheme of block_exact_nd's trips


def block_exact_nd(name, nq, nk=4, nw=16, refill=True):
    """Reference-order sweep (AWPU_MATH_F32_EXACT, round 5) of a WHOLE item -- frame pair x tile, `nq` quads of four vertically
    adjacent pixels per wave -- on the {next, d} layout: pack_nd_kernel stores, per mic and sample t of the window, the 16-byte element
        { next_a, next_b, d_a, d_b },   next = X[t + 1],  d = X[t] - X[t + 1]      (a, b = the two frames of the pair)
    i.e. delay.cpp:19-25's first operation done ONCE per sample instead of once per (pixel, mic, sample): the fp32 subtraction of the
    same two operands gives the same bits wherever it is done.  What is left per pixel, mic and register is
        t = fma(frac, d, next)    v_pk_fma_f32, frac a scalar operand
        out = out + t             v_pk_add_f32
    in the reference's order, mics in table order: the pre-epilogue sums stay the reference's bits.  8 packed VALU instructions per
    (pixel, mic, frame pair) flat; an item's reads are 4 ds_read_b128 (lane l owns samples l + 64 k: elements 1 KiB apart), shared
    by every pixel of the column whose entry carries the reference pixel's LDS address.

    One own-sample slot X serves the whole quad: a delay is monotone down a column and steps at most once inside four rows at the
    resolutions this shape is chosen for, so the pixels that leave the reference (pixel 1) are {0}, {3} or {2, 3} with one address --
    X is requested a mic ahead for pixel 0 if it differs, else for pixel 3 if it differs; any other pattern reads on the spot.  Per
    mic: the reference's reads for the NEXT mic go out first, one counted wait (4 younger reads) proves this mic's elements (the
    reference's and X), then the pixels that sweep X, then the request of X for the next mic, then the pixels on the reference's
    elements -- so X has three pixels' worth of FMAs to land.

    Item structure (the production quad block's, _block_quad_item): chunk loop, in-block refill of the other LDS image (one 16 KiB
    piece at the head of each trip), vmcnt wait and workgroup barrier inside the block, the next chunk's first entries already in
    SGPRs when a chunk begins (a quad's table is contiguous across chunks; the running prefetch switches between the quads' tables
    one trip before a quad's chunk ends).  nq = 2: per chunk quad A then quad B (`qstride` table bytes apart), each with its own
    pinned accumulators and its own copy of the trip code.

    nk = 2: the same block for SINGLE frames on the halves form of the layout (das_exact_ndh_kernel): element t of a mic's row =
    { X[t+1], X[t+129], X[t] - X[t+1], X[t+128] - X[t+129] }, the two packed lanes are the two halves of the 256-sample block, lane l
    owns samples l and l + 64 of either half -- two register pairs per pixel, two ds_read_b128 per distinct address, four packed
    VALU instructions per (pixel, mic).

    refill = False: the block of the RESIDENT single-frame kernel (every mic's row in the LDS, one chunk): no refill code at all -- the
    pieces' per-trip test alone is two instructions of a trip that the kernel's one wave per SIMD issues back to back.

    nw = waves per workgroup (a refill piece is nw x 1 KiB): 16.  (8 and 4 were built for single frames on small grids -- c2 76.6 ->
    71.7 / 68.9 us -- until one pixel per wave, block_exact_solo, replaced them.)"""
    DMA_PIECE = nw * 1024
    w = 2 * nk  # registers per pixel's out[]
    O = [[ND_ACC + 4 * w * q + w * p for p in range(4)] for q in range(nq)]
    R = [ND_TMP, ND_TMP + 4 * nk]
    X = ND_TMP + 8 * nk
    TT = ND_TMP + 12 * nk
    # the address temp is t's last register: a reads() sequence (address add, then its ds_reads, which take the address as they
    # issue) never runs between an FMA into t and the add that consumes it
    addr_t = TT + 3
    # nk = 2 (single frames: registers to spare): the refill's running byte offset of this lane inside the chunk being fetched, so that a
    # piece is seven instructions -- M0 the running LDS destination -- instead of thirteen (block_exact_solo's; the frame-pair blocks
    # have no register left for it)
    LB = addr_t + 1 if nk == 2 else None
    E = (36, 68)
    S_NG, S_PFO, S_CH, S_SB, S_TMP, S_PF_, S_LEFT_, S_DST, S_REM, S_K, S_NP, S_M0, S_DELTA = 17, 18, 19, 20, 22, 23, 24, 25, 28, 29, 30, 31, 35

    def f_of(base, p, i):
        return base + 8 * p + 2 * i

    def a_of(base, p, i):
        return base + 8 * p + 2 * i + 1

    def quadreg(r, k):
        return f"v[{r + 4 * k}:{r + 4 * k + 3}]"

    def nxt_of(slot, k):
        return f"v[{slot + 4 * k}:{slot + 4 * k + 1}]"

    def d_of(slot, k):
        return f"v[{slot + 4 * k + 2}:{slot + 4 * k + 3}]"

    def opair(r, k):
        return f"v[{r + 2 * k}:{r + 2 * k + 1}]"

    def reads(slot, addr_sgpr):
        L = [f"v_add_u32 v{addr_t}, s{addr_sgpr}, %[lane]"]
        for k in range(nk):
            off = f" offset:{1024 * k}" if k else ""
            L.append(f"ds_read_b128 {quadreg(slot, k)}, v{addr_t}{off}")
        return L

    def uid():
        COUNTER[0] += 1
        return f"%=_{COUNTER[0]}"

    cold = []
    qz = [0]  # the quad whose code is being generated

    def terms(p, base, i, slot):  # t_k = fma(frac, d_k, next_k); out_k += t_k   (two t registers, used alternately)
        fs = f"s[{f_of(base, p, i)}:{f_of(base, p, i) + 1}]"
        Op = O[qz[0]][p]
        L = []
        for k0 in range(0, nk, 2):
            for k in (k0, k0 + 1):
                L.append(f"v_pk_fma_f32 {opair(TT, k - k0)}, {fs}, {d_of(slot, k)}, {nxt_of(slot, k)} op_sel_hi:[0,1,1]")
            for k in (k0, k0 + 1):
                L.append(f"v_pk_add_f32 {opair(Op, k)}, {opair(Op, k)}, {opair(TT, k - k0)}")
        return L

    def spot(addr_sgpr):  # a pixel's own elements read where they are needed (rare patterns)
        return reads(X, addr_sgpr) + ["s_waitcnt lgkmcnt(0)"]

    def request_x(base, i):
        """X for the mic whose entries sit at (base, i): pixel 0's elements if its address differs from the reference's, else pixel 3's"""
        u = uid()
        cold.extend([f".Lnx0{u}:"] + reads(X, a_of(base, 0, i)) + [f"s_branch .Lnxb{u}",
                     f".Lnx3{u}:"] + reads(X, a_of(base, 3, i)) + [f"s_branch .Lnxb{u}"])
        return [f"s_cmp_lg_u32 s{a_of(base, 0, i)}, s{a_of(base, REF, i)}", f"s_cbranch_scc1 .Lnx0{u}",
                f"s_cmp_lg_u32 s{a_of(base, 3, i)}, s{a_of(base, REF, i)}", f"s_cbranch_scc1 .Lnx3{u}", f".Lnxb{u}:"]

    def mic_step(base, i, rslot, nbase, ni):
        u = uid()
        a0, a1, a2, a3 = (a_of(base, p, i) for p in range(4))
        t = lambda p, slot: terms(p, base, i, slot)
        # a-a-a-a
        hot = ([f"s_cmp_lg_u32 s{a0}, s{a1}", f"s_cbranch_scc1 .Ln0{u}", f"s_cmp_lg_u32 s{a3}, s{a1}", f"s_cbranch_scc1 .Ln3{u}",
                f"s_cmp_lg_u32 s{a2}, s{a1}", f"s_cbranch_scc1 .Lnslow{u}"] +
               request_x(nbase, ni) + t(3, rslot) + t(2, rslot) + t(0, rslot) + t(REF, rslot) + [f".Lnjoin{u}:"])
        # b-a-a-a
        cold.extend([f".Ln0{u}:", f"s_cmp_lg_u32 s{a3}, s{a1}", f"s_cbranch_scc1 .Lnslow{u}", f"s_cmp_lg_u32 s{a2}, s{a1}", f"s_cbranch_scc1 .Lnslow{u}"] +
                    t(0, X) + request_x(nbase, ni) + t(3, rslot) + t(2, rslot) + t(REF, rslot) + [f"s_branch .Lnjoin{u}"])
        # a-a-b-b and a-a-a-b
        cold.extend([f".Ln3{u}:", f"s_cmp_eq_u32 s{a2}, s{a1}", f"s_cbranch_scc1 .Ln3only{u}", f"s_cmp_lg_u32 s{a2}, s{a3}", f"s_cbranch_scc1 .Lnslow{u}"] +
                    t(3, X) + t(2, X) + request_x(nbase, ni) + t(0, rslot) + t(REF, rslot) + [f"s_branch .Lnjoin{u}"])
        cold.extend([f".Ln3only{u}:"] + t(3, X) + request_x(nbase, ni) + t(2, rslot) + t(0, rslot) + t(REF, rslot) + [f"s_branch .Lnjoin{u}"])
        # anything else (two steps inside the quad, or not monotone): every differing pixel after the slot's owner reads on the spot
        cold.extend(
            [f".Lnslow{u}:", f"s_cmp_eq_u32 s{a0}, s{a1}", f"s_cbranch_scc1 .Lns1{u}"] + t(0, X) +
            [f"s_cmp_eq_u32 s{a3}, s{a1}", f"s_cbranch_scc1 .Lns2{u}"] + spot(a3) + t(3, X) + [f"s_branch .Lns3{u}",
             f".Lns1{u}:", f"s_cmp_eq_u32 s{a3}, s{a1}", f"s_cbranch_scc1 .Lns2{u}"] + t(3, X) +
            [f".Lns3{u}:", f"s_cmp_eq_u32 s{a2}, s{a1}", f"s_cbranch_scc1 .Lns4{u}", f"s_cmp_eq_u32 s{a2}, s{a3}", f"s_cbranch_scc1 .Lns3b{u}"] + spot(a2) +
            [f".Lns3b{u}:"] + t(2, X) + [f"s_branch .Lns4{u}",
             f".Lns2{u}:", f"s_cmp_eq_u32 s{a2}, s{a1}", f"s_cbranch_scc1 .Lns4{u}"] + spot(a2) + t(2, X) +
            [f".Lns4{u}:"] + request_x(nbase, ni) +
            [f"s_cmp_lg_u32 s{a0}, s{a1}", f"s_cbranch_scc1 .Lns5{u}"] + t(0, rslot) +
            [f".Lns5{u}:", f"s_cmp_lg_u32 s{a3}, s{a1}", f"s_cbranch_scc1 .Lns6{u}"] + t(3, rslot) +
            [f".Lns6{u}:", f"s_cmp_lg_u32 s{a2}, s{a1}", f"s_cbranch_scc1 .Lns7{u}"] + t(2, rslot) +
            [f".Lns7{u}:"] + t(REF, rslot) + [f"s_branch .Lnjoin{u}"])
        return hot

    def load_set(base, off, literal=False):
        if literal:
            return [f"s_load_dwordx16 s[{base}:{base + 15}], %[ptr], {hex(off)}",
                    f"s_load_dwordx16 s[{base + 16}:{base + 31}], %[ptr], {hex(off + 64)}"]
        return [f"s_load_dwordx16 s[{base}:{base + 15}], %[ptr], s{off}",
                f"s_add_u32 s{S_TMP}, s{off}, 64",
                f"s_load_dwordx16 s[{base + 16}:{base + 31}], %[ptr], s{S_TMP}"]

    def dma_piece():
        """one 16 KiB piece of the refill, if any is left (block_quad's)"""
        if not refill:
            return []
        u = uid()
        if LB is not None:
            return [f"v_cmp_gt_u32 vcc, s{S_REM}, v{LB}",  # lanes whose 16 bytes lie inside the chunk
                    f"s_cbranch_vccz .Lndskip{u}",
                    "s_mov_b64 exec, vcc",
                    f"global_load_lds_dwordx4 v{LB}, s[{S_SB}:{S_SB + 1}]",
                    "s_mov_b64 exec, -1",
                    f"v_add_u32 v{LB}, {hex(DMA_PIECE)}, v{LB}",
                    f"s_add_u32 m0, m0, {hex(DMA_PIECE)}", f".Lndskip{u}:"]
        return [f"s_cmp_ge_u32 s{S_K}, s{S_NP}", f"s_cbranch_scc1 .Lndskip{u}",
                f"v_cmp_gt_u32 vcc, s{S_REM}, %[lbytes]",  # lanes whose 16 bytes lie inside the chunk
                "s_mov_b64 exec, vcc",
                f"s_mov_b32 m0, s{S_DST}", "s_nop 0",
                f"global_load_lds_dwordx4 %[lbytes], s[{S_SB}:{S_SB + 1}]",
                "s_mov_b64 exec, -1",
                f"s_add_u32 s{S_SB}, s{S_SB}, {hex(DMA_PIECE)}", f"s_addc_u32 s{S_SB + 1}, s{S_SB + 1}, 0",
                f"s_add_u32 s{S_DST}, s{S_DST}, {hex(DMA_PIECE)}", f"s_sub_u32 s{S_REM}, s{S_REM}, {hex(DMA_PIECE)}",
                f"s_add_u32 s{S_K}, s{S_K}, 1", f".Lndskip{u}:"]

    def trip_n(par):
        cur, nxt = E[par], E[1 - par]
        L = []
        # 16 KiB of the refill per trip, whatever the piece (tuning knob ND_PPT_H: x that in the single-frame blocks -- measured SLOWER:
        # c2 69 -> 71 / 81 us at 2 / 5 pieces per trip; profiles/r05_single_frame_ablation.txt)
        for _ in range((16 // nw) * (ND_PPT_H if nk == 2 else 1)):
            L += dma_piece()
        # rotation / youngest first, trip by trip (tuning knob ND_PRIO: 3 = rotation in every trip, 4 = youngest first in every trip)
        if ND_PRIO == 3 or (ND_PRIO == 5 and par == 0):
            L += select_prio(S_PRIO, 1, QUAD_XMAP)
        else:
            L += select_prio(S_RANK, 0, QUAD_YMAP)
        if nq > 1:  # a quad's last trip of the chunk fetches the OTHER quad's next entries (its own continue where they stopped)
            u = uid()
            L += [f"s_cmp_lg_u32 s{S_LEFT_}, 1", f"s_cbranch_scc1 .Lnsw{u}",
                  f"s_mov_b32 s{S_TMP}, s{S_PF_}", f"s_mov_b32 s{S_PF_}, s{S_PFO}", f"s_mov_b32 s{S_PFO}, s{S_TMP}", f".Lnsw{u}:"]
        L += load_set(nxt, S_PF_) + [f"s_add_u32 s{S_PF_}, s{S_PF_}, 128"]
        for st in range(4):
            rslot = R[st & 1]
            if st < 3:
                L += reads(R[(st + 1) & 1], a_of(cur, REF, st + 1))
                L.append(f"s_waitcnt lgkmcnt({nk})")  # all but the reads just issued: this mic's elements are in
                nbase, ni = cur, st + 1
            else:
                L.append("s_waitcnt lgkmcnt(0)")  # this mic's elements, and the next trip's entries
                L += reads(R[0], a_of(nxt, REF, 0))
                nbase, ni = nxt, 0
            L += mic_step(cur, st, rslot, nbase, ni)
        return L

    def first_reads(base):
        return reads(R[0], a_of(base, REF, 0)) + request_x(base, 0)

    def refill_params(first):
        """S_SB / S_REM / S_NP / S_K for the refill that runs beside the chunk about to be swept (S_CH = chunks left, that one
        included): the item's next chunk, dbf bytes on, or -- beside the item's last chunk -- the NEXT item's first chunk (nsrc, dbn
        bytes; dbn = 0: none), so that a persistent workgroup begins its next item on rows that are in the LDS already."""
        if not refill:
            return []
        u = uid()
        L = []
        if LB is not None:  # (nothing to undo: neither the source nor the destination base was advanced)
            if not first:
                L += [f"s_sub_u32 s{S_DST}, s{S_DST}, s{S_DELTA}", f"s_sub_u32 s{S_DELTA}, 0, s{S_DELTA}"]
            return L + [f"s_cmp_eq_u32 s{S_CH}, 1", f"s_cbranch_scc1 .Lnrnext{u}",
                        f"s_add_u32 s{S_SB}, s{S_SB}, %[dbf]", f"s_addc_u32 s{S_SB + 1}, s{S_SB + 1}, 0",
                        f"s_mov_b32 s{S_REM}, %[dbf]", f"s_cmp_eq_u32 s{S_CH}, 2", f"s_cselect_b32 s{S_REM}, %[dbl], s{S_REM}",
                        f"s_branch .Lnrset{u}", f".Lnrnext{u}:",
                        f"s_mov_b64 s[{S_SB}:{S_SB + 1}], %[nsrc]", f"s_mov_b32 s{S_REM}, %[dbn]",
                        f".Lnrset{u}:",
                        f"v_mov_b32 v{LB}, %[lbytes]", f"s_mov_b32 m0, s{S_DST}"]
        if not first:
            L += [f"s_lshl_b32 s{S_TMP}, s{S_K}, {DMA_PIECE.bit_length() - 1}",   # undo the pieces' advance
                  f"s_sub_u32 s{S_SB}, s{S_SB}, s{S_TMP}", f"s_subb_u32 s{S_SB + 1}, s{S_SB + 1}, 0",
                  f"s_sub_u32 s{S_DST}, s{S_DST}, s{S_TMP}",
                  f"s_sub_u32 s{S_DST}, s{S_DST}, s{S_DELTA}",            # the refill alternates images like the sweep, one ahead
                  f"s_sub_u32 s{S_DELTA}, 0, s{S_DELTA}"]
        L += [f"s_cmp_eq_u32 s{S_CH}, 1", f"s_cbranch_scc1 .Lnrnext{u}",
              f"s_add_u32 s{S_SB}, s{S_SB}, %[dbf]", f"s_addc_u32 s{S_SB + 1}, s{S_SB + 1}, 0",
              f"s_mov_b32 s{S_REM}, %[dbf]", f"s_cmp_eq_u32 s{S_CH}, 2", f"s_cselect_b32 s{S_REM}, %[dbl], s{S_REM}",
              f"s_branch .Lnrset{u}", f".Lnrnext{u}:",
              f"s_mov_b64 s[{S_SB}:{S_SB + 1}], %[nsrc]", f"s_mov_b32 s{S_REM}, %[dbn]",
              f".Lnrset{u}:",
              f"s_mov_b32 s{S_K}, 0", f"s_add_u32 s{S_NP}, s{S_REM}, {DMA_PIECE - 1}",
              f"s_lshr_b32 s{S_NP}, s{S_NP}, {DMA_PIECE.bit_length() - 1}"]
        return L

    def chunk_groups():  # S_NG = groups of four mics in the chunk about to be swept
        return [f"s_mov_b32 s{S_NG}, %[ngf]", f"s_cmp_eq_u32 s{S_CH}, 1", f"s_cselect_b32 s{S_NG}, %[ngl], s{S_NG}", f"s_mov_b32 s{S_LEFT_}, s{S_NG}"]

    def boundary(next_set, resume):
        u = uid()
        if not refill:
            L = ["s_waitcnt lgkmcnt(0)"]
        elif LB is not None:
            piece = dma_piece()
            L = ["s_waitcnt lgkmcnt(0)", f".Lnmore{u}:"] + piece[:-1] + [f"s_branch .Lnmore{u}"] + piece[-1:]
        else:
            L = ["s_waitcnt lgkmcnt(0)",  # this chunk's last elements, and the reads issued for a trip that does not come
                 f".Lnmore{u}:", f"s_cmp_ge_u32 s{S_K}, s{S_NP}", f"s_cbranch_scc1 .Lnnomore{u}"] + dma_piece() + [f"s_branch .Lnmore{u}", f".Lnnomore{u}:"]
        L += ["s_waitcnt vmcnt(0)", "s_barrier",  # my pieces of the next chunk have landed; so has everybody's, and all are done with this image
              f"s_sub_u32 s{S_CH}, s{S_CH}, 1", f"s_cmp_eq_u32 s{S_CH}, 0", "s_cbranch_scc1 .LNexit_%=",
              f"v_add_u32 %[lane], s{S_DELTA}, %[lane]"]  # the sweep moves to the image just filled
        L += refill_params(first=False) + chunk_groups() + first_reads(next_set) + [f"s_branch {resume}"]
        return L

    L = [f"s_mov_b32 s{S_PRIO}, %[rank]", f"s_mov_b32 s{S_RANK}, %[rank]"]
    # float out[N_SAMPLES] = {0.0} (mimo.cpp:122), here and not in front of the block: the accumulators are outputs only, so the compiler
    # holds none of their registers while it computes the block's inputs
    L += [f"v_mov_b32 v{r}, 0" for q in range(nq) for p in range(4) for r in range(O[q][p], O[q][p] + w)]
    # the persistent kernel's ticket for its next-but-one item: lane 0 of the wave that is handed the queue's address (qptr != 0) adds
    # one to that counter HERE, where no compiler-made wait can follow it (a scratch reload's vmcnt(0) in front of the block would have
    # waited for the answer: microseconds); the first chunk boundary's vmcnt(0) proves the answer long before the block ends
    L += ["s_cmp_eq_u64 %[qptr], 0", "s_cbranch_scc1 .LNnoq_%=", "s_mov_b64 exec, 1",
          f"v_mov_b32 v{TT}, 0", f"v_mov_b32 v{TT + 1}, 1", f"global_atomic_add %[ticket], v{TT}, v{TT + 1}, %[qptr] sc0",
          "s_mov_b64 exec, -1", ".LNnoq_%=:"]
    L += [f"s_mov_b32 s{S_M0}, m0", f"s_mov_b32 s{S_CH}, %[nch]", f"s_mov_b32 s{S_DELTA}, %[delta]",
          f"s_mov_b64 s[{S_SB}:{S_SB + 1}], %[isrc]", f"s_mov_b32 s{S_DST}, %[ddst]"]
    L += refill_params(first=True)
    L += load_set(E[0], 0, literal=True)
    L += chunk_groups() + [f"s_movk_i32 s{S_PF_}, 0x80"]
    if nq > 1:
        L += [f"s_mov_b32 s{S_PFO}, %[qstride]"]
    L += ["s_waitcnt lgkmcnt(0)"] + first_reads(E[0])
    # quad q's trips: .LNq_0 runs out of set 0, .LNq_1 out of set 1; a quad's chunk that ends on a trip out of set s hands set 1 - s
    # (already loaded: the next quad's, or -- after the last quad -- the first quad's entries of the next chunk) to what follows
    for q in range(nq):
        qz[0] = q
        last = q == nq - 1
        after0 = f".LNhand{q}_%=" if not last else ".LNbndA_%="
        L += [f".LN{q}_0_%=:"] + trip_n(0)
        L += [f"s_sub_u32 s{S_LEFT_}, s{S_LEFT_}, 1", f"s_cmp_eq_u32 s{S_LEFT_}, 0", f"s_cbranch_scc1 {after0}"]
        L += [f".LN{q}_1_%=:"] + trip_n(1)
        L += [f"s_sub_u32 s{S_LEFT_}, s{S_LEFT_}, 1", f"s_cmp_lg_u32 s{S_LEFT_}, 0", f"s_cbranch_scc1 .LN{q}_0_%="]
        if not last:  # ended on a trip out of set 1: the next quad begins on set 0 (falls through); out of set 0: on set 1
            L += [f"s_mov_b32 s{S_LEFT_}, s{S_NG}", f"s_branch .LN{q + 1}_0_%=",
                  f".LNhand{q}_%=:", f"s_mov_b32 s{S_LEFT_}, s{S_NG}", f"s_branch .LN{q + 1}_1_%="]
        else:
            qz[0] = 0  # (the boundary's first reads belong to no quad's accumulators; request_x / reads use none)
            L += boundary(E[0], ".LN0_0_%=")
            L += [".LNbndA_%=:"] + boundary(E[1], ".LN0_1_%=")
    L += cold
    L += [".LNexit_%=:", f"s_mov_b32 m0, s{S_M0}", f"s_setprio {QUAD_END_PRIO}"]
    # timing experiments (tools/build_variant.sh with ND_TIMING=...; WRONG results): what the block costs without one of its parts
    if "nobarrier" in ND_TIMING:
        L = ["s_nop 0" if l == "s_barrier" else l for l in L]
    if "nolds" in ND_TIMING:
        L = [l for l in L if not l.startswith("ds_read")]
    if "novalu" in ND_TIMING:
        L = [l for l in L if not l.startswith(("v_pk_fma_f32", "v_pk_add_f32"))]
    if "nodma" in ND_TIMING:
        L = [l for l in L if not l.startswith("global_load_lds")]
    if "noprio" in ND_TIMING:
        L = [l for l in L if not l.startswith("s_setprio")]
    body = "\n".join(f'        "{l}\\n\\t"' for l in L)
    vregs = list(range(ND_TMP, (LB if LB is not None else addr_t) + 1))
    sregs = sorted({S_NG, S_PFO, S_CH, S_SB, S_SB + 1, S_TMP, S_PF_, S_LEFT_, S_DST, S_RANK, S_PRIO, S_REM, S_K, S_NP, S_M0, S_DELTA}) + list(range(36, 100))
    clobbers = ", ".join([f'"v{r}"' for r in vregs] + [f'"s{r}"' for r in sregs] + ['"scc"', '"vcc"', '"memory"'])
    acc_params = ", ".join(f"{'f8' if nk == 4 else 'f4'} &O{q}{p}" for q in range(nq) for p in range(4))
    acc_ops = ", ".join(f'"=&{{v[{O[q][p]}:{O[q][p] + w - 1}]}}"(O{q}{p})' for q in range(nq) for p in range(4))
    qs_param = ", int qstride" if nq > 1 else ""
    qs_op = ', [qstride] "s"(qstride)' if nq > 1 else ""
    return f"""// Reference-order sweep of a whole item (frame pair x tile) on the {{next, d}} layout, {nq} quad(s) of four vertically adjacent
// pixels per wave: tools/gen_trip_asm.py, block_exact_nd.  `row` = the first quad's entries of the item's first group in the quad-major
// table ([group][pixel][mic] x (fraction, address), contiguous across chunks){"; the second quad's lie qstride bytes on" if nq > 1 else ""}; reads one group
// past a quad's last.  ngf / ngl = groups of four mics in a full / in the last chunk, nch = chunks, isrc = chunk 0's rows in HBM (chunk c's
// follow dbf bytes apart; dbl = bytes of the last chunk), ddst = this wave's first LDS-DMA destination in the image chunk 0 does NOT occupy,
// delta = (that image) - (chunk 0's image) in bytes, lbytes = 16 x thread index; lane_addr = the sweep's LDS address in chunk 0's image on
// entry, in the last chunk's on exit; nsrc / dbn = the NEXT item's first chunk, refilled beside this item's last (dbn = 0: none);
// qptr != null (one wave of the workgroup): lane 0 adds 1 to that counter (device scope) and `ticket` returns what it held before.
// O_qp (outputs: the block zeroes them itself) = out[l + 64 k] of pixel p of quad q, both {"frames" if nk == 4 else "halves of the block"}, pinned at v[{ND_ACC} + {4 * w} q + {w} p ..]; temps v{vregs[0]}..v{vregs[-1]}, s{sregs[0]}..s{sregs[-1]}.
// Executes nch s_barrier instructions.
__device__ __forceinline__ void {name}({acc_params}, const void *row{qs_param}, int ngf, int ngl, int nch, unsigned &lane_addr, int rank,
                                       const void *isrc, unsigned dbf, unsigned dbl, const void *nsrc, unsigned dbn, unsigned ddst, int delta,
                                       unsigned lbytes, const unsigned *qptr, unsigned &ticket) {{
    asm volatile(
{body}
        : {acc_ops}, [lane] "+v"(lane_addr), [ticket] "=&v"(ticket)
        : [ptr] "s"(row){qs_op}, [ngf] "s"(ngf), [ngl] "s"(ngl), [nch] "s"(nch), [rank] "s"(rank), [isrc] "s"(isrc), [dbf] "s"(dbf), [dbl] "s"(dbl),
          [nsrc] "s"(nsrc), [dbn] "s"(dbn), [ddst] "s"(ddst), [delta] "s"(delta), [lbytes] "v"(lbytes), [qptr] "s"(qptr)
        : {clobbers});
}}
"""


SOLO_ACC = 10   # block_exact_solo: out[] of the wave's pixel pinned at v[10:13]
SOLO_TMP = 14   # two sets of four 8-register slots (a trip's elements: the set being swept, the set in flight), t, two address temps


def block_exact_solo(name, nw=16):
    """Reference-order sweep of ONE pixel per wave, single frames on the halves form of the {next, d} layout (das_exact_ndp_kernel):
    block_exact_nd(nk = 2)'s arithmetic and item structure without the quad.  A quad block's trip of four mics is ~150 instructions
    of which 64 are arithmetic (the rest picks, per mic, which pixels share the reference pixel's reads), issued by ONE wave: on a
    grid of a few thousand pixels there are too few quads to give every SIMD more than one wave, and the frame waits for that one
    wave's instruction issue (c2, 64x64 x 256 mics: 69 us, 58 of them with no arithmetic at all; profiles/r05_single_frame_ablation.txt).
    A pixel per wave has four times the waves, no sharing and therefore no compare tree: per trip one s_load_dwordx8 (the pixel's
    32 bytes of the quad's 128-byte line), 4 address adds + 8 ds_read_b128 for the NEXT trip, 16 packed VALU for this one, and ONE
    wait -- lgkmcnt(0) at the head of a trip, for reads and entries that were requested a whole trip earlier.

    Registers: three sets of 8 SGPRs and three sets of four 8-register slots rotate trip by trip (the trip code exists three times):
    at the head of trip n the set n % 3 holds entries(n) -- its fractions feed this trip's arithmetic -- and set (n+1) % 3 entries(n+1),
    whose addresses issue the reads of trip n+1 into slot set (n+1) % 3; set (n+2) % 3 (entries(n-1): dead) is reloaded with
    entries(n+2).  A chunk's last trip issues no reads (the next trip's rows are in the image still being refilled): it runs a copy
    of the trip's tail that ends in the chunk boundary, which issues them after the barrier.  Reads two groups past the pixel's last
    (never used)."""
    DMA_PIECE = nw * 1024
    O = SOLO_ACC
    SL = [[SOLO_TMP + 32 * s + 8 * i for i in range(4)] for s in range(3)]
    TT = SOLO_TMP + 96
    AT = (TT + 4, TT + 5)
    LB = TT + 6  # the refill's running byte offset of this lane inside the chunk being fetched (lbytes + pieces issued x the piece)
    E = (36, 44, 52)
    S_NG, S_CH, S_SB, S_PF_, S_LEFT_, S_DST, S_REM, S_M0, S_DELTA = 17, 19, 20, 23, 24, 25, 28, 31, 35

    def uid():
        COUNTER[0] += 1
        return f"%=_{COUNTER[0]}"

    def reads(slots, base):
        L = []
        for i in range(4):
            L.append(f"v_add_u32 v{AT[i & 1]}, s{base + 2 * i + 1}, %[lane]")
            L.append(f"ds_read_b128 v[{slots[i]}:{slots[i] + 3}], v{AT[i & 1]}")
            L.append(f"ds_read_b128 v[{slots[i] + 4}:{slots[i] + 7}], v{AT[i & 1]} offset:1024")
        return L

    def terms(base, i, slot):  # t_k = fma(frac, d_k, next_k); out_k += t_k  (delay.cpp:21-25 on {next, d}), k = the lane's two sample pairs
        fs = f"s[{base + 2 * i}:{base + 2 * i + 1}]"
        L = [f"v_pk_fma_f32 v[{TT + 2 * k}:{TT + 2 * k + 1}], {fs}, v[{slot + 4 * k + 2}:{slot + 4 * k + 3}], v[{slot + 4 * k}:{slot + 4 * k + 1}] op_sel_hi:[0,1,1]"
             for k in range(2)]
        L += [f"v_pk_add_f32 v[{O + 2 * k}:{O + 2 * k + 1}], v[{O + 2 * k}:{O + 2 * k + 1}], v[{TT + 2 * k}:{TT + 2 * k + 1}]" for k in range(2)]
        return L

    def dma_piece():
        """one piece (a 16-byte element per lane) of the refill, if a lane still has one inside the chunk: seven instructions -- M0 is the
        running LDS destination, v{LB} the running offset (no piece counter, no scalar copy of either)"""
        u = uid()
        return [f"v_cmp_gt_u32 vcc, s{S_REM}, v{LB}",  # lanes whose 16 bytes lie inside the chunk
                f"s_cbranch_vccz .Lpdskip{u}",
                "s_mov_b64 exec, vcc",
                f"global_load_lds_dwordx4 v{LB}, s[{S_SB}:{S_SB + 1}]",
                "s_mov_b64 exec, -1",
                f"v_add_u32 v{LB}, {hex(DMA_PIECE)}, v{LB}",
                f"s_add_u32 m0, m0, {hex(DMA_PIECE)}", f".Lpdskip{u}:"]

    def refill_params(first):  # block_exact_nd's, without a next item: beside the last chunk nothing is refilled
        u = uid()
        L = []
        if not first:
            L += [f"s_sub_u32 s{S_DST}, s{S_DST}, s{S_DELTA}",   # the refill alternates images like the sweep, one ahead
                  f"s_sub_u32 s{S_DELTA}, 0, s{S_DELTA}"]
        L += [f"s_mov_b32 s{S_REM}, 0", f"s_cmp_eq_u32 s{S_CH}, 1", f"s_cbranch_scc1 .Lprset{u}",
              f"s_add_u32 s{S_SB}, s{S_SB}, %[dbf]", f"s_addc_u32 s{S_SB + 1}, s{S_SB + 1}, 0",
              f"s_mov_b32 s{S_REM}, %[dbf]", f"s_cmp_eq_u32 s{S_CH}, 2", f"s_cselect_b32 s{S_REM}, %[dbl], s{S_REM}",
              f".Lprset{u}:",
              f"v_mov_b32 v{LB}, %[lbytes]", f"s_mov_b32 m0, s{S_DST}"]
        return L

    def chunk_groups():
        # (S_LEFT = trips of the chunk after its first: the borrow of a trip's decrement is what marks the chunk's last trip)
        return [f"s_mov_b32 s{S_NG}, %[ngf]", f"s_cmp_eq_u32 s{S_CH}, 1", f"s_cselect_b32 s{S_NG}, %[ngl], s{S_NG}", f"s_sub_u32 s{S_LEFT_}, s{S_NG}, 1"]

    cold = []

    def boundary(j_next):
        u = uid()
        piece = dma_piece()
        L = ["s_waitcnt lgkmcnt(0)", f".Lpmore{u}:"] + piece[:-1] + [f"s_branch .Lpmore{u}"] + piece[-1:]  # the pieces the trips did not get to
        L += ["s_waitcnt vmcnt(0)", "s_barrier",
              f"s_sub_u32 s{S_CH}, s{S_CH}, 1", f"s_cmp_eq_u32 s{S_CH}, 0", "s_cbranch_scc1 .LPexit_%=",
              f"v_add_u32 %[lane], s{S_DELTA}, %[lane]"]
        L += refill_params(first=False) + chunk_groups() + reads(SL[j_next], E[j_next]) + [f"s_branch .LP{j_next}_%="]
        return L

    def trip_p(j):
        cur, nxt, nn = E[j], E[(j + 1) % 3], E[(j + 2) % 3]
        u = uid()
        L = [f".LP{j}_%=:", "s_waitcnt lgkmcnt(0)",  # this trip's elements (requested a trip ago) and entries(n+1)
             f"s_load_dwordx8 s[{nn}:{nn + 7}], %[ptr], s{S_PF_}", f"s_add_u32 s{S_PF_}, s{S_PF_}, 128",
             f"s_sub_u32 s{S_LEFT_}, s{S_LEFT_}, 1", f"s_cbranch_scc1 .LPlast{u}"]
        L += reads(SL[(j + 1) % 3], nxt) + dma_piece()
        tail = []
        for i in range(4):
            tail += terms(cur, i, SL[j][i])
        L += tail
        if j == 2:
            L.append("s_branch .LP0_%=")
        cold.extend([f".LPlast{u}:"] + dma_piece() + tail + boundary((j + 1) % 3))  # the chunk's last trip
        return L

    L = [f"v_mov_b32 v{r}, 0" for r in range(O, O + 4)]  # float out[N_SAMPLES] = {0.0} (mimo.cpp:122)
    L += [f"s_mov_b32 s{S_M0}, m0", f"s_mov_b32 s{S_CH}, %[nch]", f"s_mov_b32 s{S_DELTA}, %[delta]",
          f"s_mov_b64 s[{S_SB}:{S_SB + 1}], %[isrc]", f"s_mov_b32 s{S_DST}, %[ddst]"]
    L += refill_params(first=True)
    L += [f"s_load_dwordx8 s[{E[0]}:{E[0] + 7}], %[ptr], 0x0", f"s_load_dwordx8 s[{E[1]}:{E[1] + 7}], %[ptr], 0x80"]
    # chunk 0 was requested by the kernel in front of the block (LDS-DMA); its wait and the workgroup's barrier stand HERE, behind the
    # first table loads, so that the entries travel while the rows land
    L += chunk_groups() + [f"s_movk_i32 s{S_PF_}, 0x100", "s_waitcnt vmcnt(0)", "s_barrier", "s_waitcnt lgkmcnt(0)"] + reads(SL[0], E[0])
    L += trip_p(0) + trip_p(1) + trip_p(2) + cold
    L += [".LPexit_%=:", "s_waitcnt lgkmcnt(0)", f"s_mov_b32 m0, s{S_M0}"]
    if "nobarrier" in ND_TIMING:
        L = ["s_nop 0" if l == "s_barrier" else l for l in L]
    if "nolds" in ND_TIMING:
        L = [l for l in L if not l.startswith("ds_read")]
    if "novalu" in ND_TIMING:
        L = [l for l in L if not l.startswith(("v_pk_fma_f32", "v_pk_add_f32"))]
    if "nodma" in ND_TIMING:
        L = [l for l in L if not l.startswith("global_load_lds")]
    body = "\n".join(f'        "{l}\\n\\t"' for l in L)
    vregs = list(range(SOLO_TMP, LB + 1))
    sregs = sorted({S_NG, S_CH, S_SB, S_SB + 1, S_PF_, S_LEFT_, S_DST, S_REM, S_M0, S_DELTA}) + list(range(36, 60))
    clobbers = ", ".join([f'"v{r}"' for r in vregs] + [f'"s{r}"' for r in sregs] + ['"scc"', '"vcc"', '"memory"'])
    return f"""// Reference-order sweep of one pixel's whole item (one frame x tile) on the halves form of the {{next, d}} layout: tools/gen_trip_asm.py,
// block_exact_solo.  `row` = the PIXEL's entries of the item's first group in the quad-major table (32 bytes of each group's 128-byte line:
// [mic] x (fraction, address); groups 128 bytes apart, contiguous across chunks); reads two groups past the last.  ngf / ngl / nch / isrc /
// dbf / dbl / ddst / delta / lbytes / lane_addr as sweep_exact_ndh_item1.  O (output: zeroed here) = out[l + 64 k] of either half, pinned
// at v[{O}:{O + 3}]; temps v{vregs[0]}..v{vregs[-1]}, s{sregs[0]}..s{sregs[-1]}.  Executes nch + 1 s_barrier instructions: the first one -- behind
// s_waitcnt vmcnt(0) -- is the barrier of chunk 0's staging, which the caller has requested and NOT waited for.
__device__ __forceinline__ void {name}(f4 &O, const void *row, int ngf, int ngl, int nch, unsigned &lane_addr, const void *isrc, unsigned dbf,
                                       unsigned dbl, unsigned ddst, int delta, unsigned lbytes) {{
    asm volatile(
{body}
        : "=&{{v[{O}:{O + 3}]}}"(O), [lane] "+v"(lane_addr)
        : [ptr] "s"(row), [ngf] "s"(ngf), [ngl] "s"(ngl), [nch] "s"(nch), [isrc] "s"(isrc), [dbf] "s"(dbf), [dbl] "s"(dbl), [ddst] "s"(ddst),
          [delta] "s"(delta), [lbytes] "v"(lbytes)
        : {clobbers});
}}
"""


# ---------------------------------------------------------------------------------------------------
# Quad block with a shared integer-delay sum (das_quad_kernel).
#
# out_p[i] = sum_m f*X[o+i] + (1-f)*X[o+i+1] = A_p[i] - A_p[i+1] + S_p[i+1],  A_p[j] = sum_m f_pm X_m[o_pm+j],
# S_p[j] = sum_m X_m[o_pm+j].  S_p does not depend on the fractions: pixels whose INTEGER delays coincide for a
# mic share that mic's term of it.  A block sweeps four vertically adjacent pixels (rows r..r+3 of one grid
# column); the second one is the reference: its sample reads feed T = S_ref (one packed add per register) and
# every pixel whose entry carries the same LDS address (one packed FMA per register into A_p).  A pixel whose
# integer delay differs for this mic reads its own samples and pays A_p += f x_p, V_p += x_p, V_p -= x_ref
# (S_p = T + V_p).  Per mic and quad: 20 packed VALU instructions when the four delays coincide instead of 32,
# +8 per pixel that differs.  Entries are 8 bytes (f, LDS address); g = 1 - f is never needed in the sweep.
#
# Fixed registers: accumulators pinned by the caller's operand constraints
#     A_p (p = 0..3) v[QA+8p : +7],  T v[QT:QT+7],  V_p (p = 0, 2, 3) v[QV.. : +7]  (register 2k,2k+1 = samples l+64k)
# temps (clobbers): two slots for the reference's samples (this mic / next mic), one slot for pixel 0's own samples,
# one that pixels 2 and 3 share (when both differ from the reference they almost always carry the same address:
# the delay is monotone down a column; the slot is prefetched for pixel 3, pixel 2 uses it when its address is
# the same and otherwise -- rare -- reads into it on the spot, after pixel 3 is done with it), one address
# register; SGPR sets E0 = s[36:67], E1 = s[68:99]: pixel p, mic i of the trip: f at +8p+2i, address +1.
# the refill pieces of the quad block's in-block DMA: 1 KiB per issuing wave, QUAD_DMA_WAVES waves take part (waves
# 0 .. n-1 of the workgroup; the kernel gives the others no pieces)
DMA_WAVES = int(os.environ.get("QUAD_DMA_WAVES", "16"))
DMA_STRIDE = DMA_WAVES * 1024
QUAD_END_PRIO = int(os.environ.get("QUAD_END_PRIO", "3"))
PAIR23 = int(os.environ.get("QUAD_PAIR23", "1"))  # quad blocks: pixels 2 and 3 share one difference where they leave the reference together
BLOCK_END_PRIO = int(os.environ.get("BLOCK_END_PRIO", "0"))  # the same for the pair-, single-frame and single-frame quad blocks  # priority a wave keeps after the quad block (tail pass, barrier, next block's head)
QUAD_ACC = 30          # first pinned accumulator register; 64 of them
QUAD_TMP = QUAD_ACC + 64  # 33 temps
REF = 1                # which pixel of the quad is the reference (a middle one: fewest differing neighbours)


def quad_regs(nk=4, acc=QUAD_ACC, tmp=QUAD_TMP):
    """nk = register pairs per accumulator: 4 in the frame-pair layout (lane l owns samples l+64k, the two packed
    lanes are two frames), 2 in the single-frame layout (lane l owns samples {2l,2l+1} and {128+2l,129+2l})"""
    w = 2 * nk
    A = [acc + w * p for p in range(4)]
    T = acc + 4 * w
    V = {0: acc + 5 * w, 2: acc + 6 * w, 3: acc + 7 * w}
    R = (tmp, tmp + w)
    X = {0: tmp + 2 * w, 2: tmp + 3 * w, 3: tmp + 3 * w}  # pixels 2 and 3 share a slot (see block_quad)
    addr_t = tmp + 4 * w
    return A, T, V, R, X, addr_t


# single-frame layout: two quads per wave, each with its own pinned accumulators (32 registers), shared temps
QUAD1_ACC = (16, 48)
QUAD1_TMP = 80


QUAD_YMAP = tuple(int(x) for x in os.environ.get("QUAD_YMAP", "0,1,2,3").split(","))  # frame-pair quad blocks: priority by age rank in the youngest-first trips
QUAD_XMAP = tuple(int(x) for x in os.environ.get("QUAD_XMAP", "0,1,2,3").split(","))  # ... and by rotating role in the other trips
# frame-pair quad blocks, tuning (see trip_q): the priority scheme alternating per pair of trips (1) or per two pairs (2) instead of
# trip by trip (0).  Measured, headline, alternating runs on one box: 4.470 / 4.497 / 4.520 ms per launch -- coarser is worse here
# (the FIR8 blocks, whose items are eight times longer, want it the other way round: FIR_PRIO).
QUAD_PRIO_COARSE = int(os.environ.get("QUAD_PRIO_COARSE", "0"))
# the same knob for the single-frame quad blocks (block_quad_ar).  Measured, headline, one frame per call, alternating runs on one
# box, frames/s: trip by trip 14.1 k | per pair of trips 13.9 k | per two pairs 13.8 k | rotation only (TRIP_PRIO=3) 13.5 k | static
# youngest-first (4) 13.6 k | no priorities (0) 13.3 k
QUAD1_PRIO_COARSE = int(os.environ.get("QUAD1_PRIO_COARSE", "0"))
CHAIN = int(os.environ.get("QUAD_CHAIN", "1"))  # frame-pair quad blocks: V3 = S3 - S2 (a step in the column costs one difference, wherever it lies)


def block_quad(name, stamp=False, prio=None, nk=4, acc=QUAD_ACC, tmp=QUAD_TMP, dma=False, early_x=False, chain=False, item=False):
    """dma=True: the block also issues the refill of the other LDS image (the next chunk: `dbytes` bytes from `dsrc`,
    16 KiB pieces of 64 lanes x 16 bytes per wave, this wave's first piece landing at LDS address `ddst`), one
    piece at the head of each trip instead of all of them before the sweep: the 16 waves of the workgroup then do not
    queue at the CU's one address unit right after the barrier, and every wave's pieces go out beside the other
    waves' FMAs.  Pieces a short chunk has no trip for follow the last trip.  The caller's s_waitcnt vmcnt(0)
    before the barrier covers them (hipcc does not count loads issued inside an asm)."""
    prio = PRIO if prio is None else prio
    # early_x (the single-frame layout, where a stage is too short to hide a read issued at its end): the own-sample
    # slots are double-buffered like the reference's, and the conditional reads of the NEXT mic go out at the head of
    # the stage, right behind the wait that proves this mic's samples -- a whole stage of flight, no more LDS reads
    # than are needed.  MEASURED SLOWER than the always-read schedule of block_quad_ar (headline, one frame per call:
    # 79-80 us against 70-73 us in the same kernel): five compare-and-branch pairs per stage instead of three weigh
    # more than the LDS reads saved.  Kept as a generator option (QUAD1_EARLY_X=1), not built by default.
    assert not (dma and stamp), "the stamped builds keep the refill outside the block (s28..s31 hold the stamps)"
    S_SB, S_DST, S_REM, S_K, S_NP, S_M0 = 20, 25, 28, 29, 30, 31  # s[20:21] running source address, ...
    A, T, V, R, X, addr_t = quad_regs(nk, acc, tmp)
    XS = [X, X]
    if early_x:
        w_ = 2 * nk
        XS = [{0: tmp + 2 * w_, 2: tmp + 3 * w_, 3: tmp + 3 * w_}, {0: tmp + 4 * w_, 2: tmp + 5 * w_, 3: tmp + 5 * w_}]
        addr_t = tmp + 6 * w_
    xz = [0]  # which own-sample slot set the code being generated uses
    E = (36, 68)
    S_TMP, S_PF_, S_LEFT_ = 22, 23, 24

    def f_of(base, p, i):
        return base + 8 * p + 2 * i

    def a_of(base, p, i):
        return base + 8 * p + 2 * i + 1

    def pair(r, k):
        return f"v[{r + 2 * k}:{r + 2 * k + 1}]"

    def reads(slot, addr_sgpr):
        L = [f"v_add_u32 v{addr_t}, s{addr_sgpr}, %[lane]"]
        for k in range(nk):
            off = f" offset:{512 * k}" if k else ""
            L.append(f"ds_read_b64 {pair(slot, k)}, v{addr_t}{off}")
        return L

    def uid():
        COUNTER[0] += 1
        return f"%=_{COUNTER[0]}"

    cold = []
    skip23 = [None]

    def maybe_read_x(p, base, i):
        """issue pixel p's own reads for the mic whose entries sit at (base, i) unless it shares the reference's"""
        u = uid()
        cold.extend([f".Lqread{u}:"] + reads(XS[xz[0]][p], a_of(base, p, i)) + [f"s_branch .Lqreadback{u}"])
        return [f"s_cmp_lg_u32 s{a_of(base, p, i)}, s{a_of(base, REF, i)}", f"s_cbranch_scc1 .Lqread{u}", f".Lqreadback{u}:"]

    def own_ops(p, fs, rslot):
        X = XS[xz[0]]
        own = []
        for k in range(nk):
            own.append(f"v_pk_fma_f32 {pair(A[p], k)}, {fs}, {pair(X[p], k)}, {pair(A[p], k)} op_sel_hi:[0,1,1]")
        for k in range(nk):
            own.append(f"v_pk_add_f32 {pair(V[p], k)}, {pair(V[p], k)}, {pair(X[p], k)}")
        for k in range(nk):
            own.append(f"v_pk_add_f32 {pair(V[p], k)}, {pair(V[p], k)}, {pair(rslot, k)} neg_lo:[0,1] neg_hi:[0,1]")
        return own

    def pixel_ops(p, base, i, rslot):
        """pixel p's share of mic (base, i): one FMA per register from the reference's samples, or the
        three-instruction form from its own"""
        u = uid()
        fs = f"s[{f_of(base, p, i)}:{f_of(base, p, i) + 1}]"
        if p == 2:  # shares pixel 3's slot: usable as it is only if pixel 3 differs too and carries the same address
            cold.extend([f".Lqown{u}:",
                         f"s_cmp_eq_u32 s{a_of(base, 2, i)}, s{a_of(base, 3, i)}",
                         f"s_cbranch_scc1 .Lqhave{u}"] +
                        reads(XS[xz[0]][2], a_of(base, 2, i)) + ["s_waitcnt lgkmcnt(0)", f".Lqhave{u}:"] +
                        own_ops(p, fs, rslot) + [f"s_branch .Lqdone{u}"])
        elif p == 3 and PAIR23:
            # Pixels 2 and 3 both away from the reference but together (the pattern a-a-b-b of a delay that steps once
            # inside the quad): one difference x_b - x_ref serves both shared-sum corrections -- 5 instructions per
            # register for the two pixels instead of 6.  (The slot is pixel 2's and 3's alone: changed in place.)
            X3 = XS[xz[0]][3]
            fs2 = f"s[{f_of(base, 2, i)}:{f_of(base, 2, i) + 1}]"
            both = []
            for q, fq in ((3, fs), (2, fs2)):
                both += [f"v_pk_fma_f32 {pair(A[q], k)}, {fq}, {pair(X3, k)}, {pair(A[q], k)} op_sel_hi:[0,1,1]" for k in range(nk)]
            both += [f"v_pk_add_f32 {pair(X3, k)}, {pair(X3, k)}, {pair(rslot, k)} neg_lo:[0,1] neg_hi:[0,1]" for k in range(nk)]
            for q in (3, 2):
                both += [f"v_pk_add_f32 {pair(V[q], k)}, {pair(V[q], k)}, {pair(X3, k)}" for k in range(nk)]
            skip23[0] = u
            cold.extend([f".Lqown{u}:", f"s_cmp_eq_u32 s{a_of(base, 2, i)}, s{a_of(base, 3, i)}", f"s_cbranch_scc1 .Lqboth{u}"] +
                        own_ops(p, fs, rslot) + [f"s_branch .Lqdone{u}", f".Lqboth{u}:"] + both + [f"s_branch .Lqskip{u}"])
        else:
            cold.extend([f".Lqown{u}:"] + own_ops(p, fs, rslot) + [f"s_branch .Lqdone{u}"])
        L = [f"s_cmp_lg_u32 s{a_of(base, p, i)}, s{a_of(base, REF, i)}", f"s_cbranch_scc1 .Lqown{u}"]
        for k in range(nk):
            L.append(f"v_pk_fma_f32 {pair(A[p], k)}, {fs}, {pair(rslot, k)}, {pair(A[p], k)} op_sel_hi:[0,1,1]")
        L.append(f".Lqdone{u}:")
        if p == 2 and skip23[0] is not None:  # where pixel 3's combined path rejoins
            L.append(f".Lqskip{skip23[0]}:")
            skip23[0] = None
        return L

    def pixels_23_chain(base, i, rslot):
        """Pixels 3 and 2 of mic (base, i) with the shared sums kept as a CHAIN of differences down the column:
        V2 = S2 - S1 (S1 = T, the reference's), V3 = S3 - S2.  The integer delay is monotone down a column, so the
        quad's addresses look like a-a-a-a, a-a-a-b, a-a-b-b (or, rarely, a-a-b-c): a step costs one difference
        x_b - x_a and one add wherever it lies -- the a-a-b-b pattern 4 instructions per register for the two pixels
        (two FMAs, the difference, one add) where V3 = S3 - S1 needed 5.  The slot X23 holds pixel 3's samples whenever
        its address differs from the reference's (requested a mic ahead by maybe_read_x)."""
        X3 = XS[xz[0]][3]
        u = uid()
        a1, a2, a3 = a_of(base, REF, i), a_of(base, 2, i), a_of(base, 3, i)
        f2s = f"s[{f_of(base, 2, i)}:{f_of(base, 2, i) + 1}]"
        f3s = f"s[{f_of(base, 3, i)}:{f_of(base, 3, i) + 1}]"
        fma = lambda p, fs, src: [f"v_pk_fma_f32 {pair(A[p], k)}, {fs}, {pair(src, k)}, {pair(A[p], k)} op_sel_hi:[0,1,1]" for k in range(nk)]
        sub_ref = [f"v_pk_add_f32 {pair(X3, k)}, {pair(X3, k)}, {pair(rslot, k)} neg_lo:[0,1] neg_hi:[0,1]" for k in range(nk)]
        add_v = lambda p, src: [f"v_pk_add_f32 {pair(V[p], k)}, {pair(V[p], k)}, {pair(src, k)}" for k in range(nk)]
        sub_v = lambda p, src: [f"v_pk_add_f32 {pair(V[p], k)}, {pair(V[p], k)}, {pair(src, k)} neg_lo:[0,1] neg_hi:[0,1]" for k in range(nk)]
        spot = reads(X3, a2) + ["s_waitcnt lgkmcnt(0)"]  # pixel 2's own samples, read on the spot (rare)
        cold.extend(
            [f".Lq3diff{u}:"] + fma(3, f3s, X3) +
            [f"s_cmp_eq_u32 s{a2}, s{a3}", f"s_cbranch_scc1 .Lqtog{u}", f"s_cmp_lg_u32 s{a2}, s{a1}", f"s_cbranch_scc1 .Lqtwo{u}"] +
            # a-a-a-b: the step lies between pixels 2 and 3
            fma(2, f2s, rslot) + sub_ref + add_v(3, X3) + [f"s_branch .Lq23done{u}"] +
            # a-a-b-b: between the reference and pixel 2; pixel 3 follows pixel 2 (V3 += 0)
            [f".Lqtog{u}:"] + fma(2, f2s, X3) + sub_ref + add_v(2, X3) + [f"s_branch .Lq23done{u}"] +
            # a-a-b-c: two steps
            [f".Lqtwo{u}:"] + add_v(3, X3) + spot + fma(2, f2s, X3) + sub_v(3, X3) + sub_ref + add_v(2, X3) + [f"s_branch .Lq23done{u}"] +
            # a-a-b-a (not monotone: rare): pixel 2 alone leaves and pixel 3 comes back
            [f".Lq2odd{u}:"] + spot + fma(2, f2s, X3) + sub_ref + add_v(2, X3) + sub_v(3, X3) + [f"s_branch .Lq23done{u}"])
        return ([f"s_cmp_lg_u32 s{a3}, s{a1}", f"s_cbranch_scc1 .Lq3diff{u}"] + fma(3, f3s, rslot) +
                [f"s_cmp_lg_u32 s{a2}, s{a1}", f"s_cbranch_scc1 .Lq2odd{u}"] + fma(2, f2s, rslot) + [f".Lq23done{u}:"])

    def ref_ops(base, i, rslot):
        fs = f"s[{f_of(base, REF, i)}:{f_of(base, REF, i) + 1}]"
        L = []
        for k in range(nk):
            L.append(f"v_pk_add_f32 {pair(T, k)}, {pair(T, k)}, {pair(rslot, k)}")
        for k in range(nk):
            L.append(f"v_pk_fma_f32 {pair(A[REF], k)}, {fs}, {pair(rslot, k)}, {pair(A[REF], k)} op_sel_hi:[0,1,1]")
        return L

    def load_set(base, off, literal=False):
        """one trip's entries -- [pixel][mic] x (f, address), 128 contiguous bytes in the quad-major table -- into
        the set at `base`; `off` = a literal byte offset or the SGPR that holds it"""
        if literal:
            return [f"s_load_dwordx16 s[{base}:{base + 15}], %[ptr], {hex(off)}",
                    f"s_load_dwordx16 s[{base + 16}:{base + 31}], %[ptr], {hex(off + 64)}"]
        return [f"s_load_dwordx16 s[{base}:{base + 15}], %[ptr], s{off}",
                f"s_add_u32 s{S_TMP}, s{off}, 64",
                f"s_load_dwordx16 s[{base + 16}:{base + 31}], %[ptr], s{S_TMP}"]

    others = [p for p in (3, 2, 0)]

    def dma_piece():
        """one 16 KiB piece of the refill, if any is left"""
        u = uid()
        return [f"s_cmp_ge_u32 s{S_K}, s{S_NP}", f"s_cbranch_scc1 .Ldskip{u}",
                f"v_cmp_gt_u32 vcc, s{S_REM}, %[lbytes]",  # lanes whose 16 bytes lie inside the chunk
                "s_mov_b64 exec, vcc",                      # (the block runs with all 64 lanes on: restored to -1 below)
                f"s_mov_b32 m0, s{S_DST}", "s_nop 0",
                f"global_load_lds_dwordx4 %[lbytes], s[{S_SB}:{S_SB + 1}]",
                "s_mov_b64 exec, -1",
                f"s_add_u32 s{S_SB}, s{S_SB}, {hex(DMA_STRIDE)}", f"s_addc_u32 s{S_SB + 1}, s{S_SB + 1}, 0",
                f"s_add_u32 s{S_DST}, s{S_DST}, {hex(DMA_STRIDE)}", f"s_sub_u32 s{S_REM}, s{S_REM}, {hex(DMA_STRIDE)}",
                f"s_add_u32 s{S_K}, s{S_K}, 1", f".Ldskip{u}:"]

    def trip_q(par):
        cur, nxt = E[par], E[1 - par]
        L = dma_piece() if dma else []
        if QUAD_PRIO_COARSE and prio == 5 and nk == 4:
            # coarser alternation (tuning): the scheme changes every QUAD_PRIO_COARSE-th pair of trips -- rotation while bit
            # `QUAD_PRIO_COARSE` of the groups-left counter is clear, youngest-first while it is set -- and is applied once per pair
            if par == 0:
                COUNTER[0] += 1
                uc = f"%=_{COUNTER[0]}"
                COUNTER[0] += 1
                L += ([f"s_bitcmp1_b32 s{S_LEFT_}, {QUAD_PRIO_COARSE}", f"s_cbranch_scc1 .LQpy{uc}"] + select_prio(S_PRIO, 1, QUAD_XMAP) +
                      [f"s_branch .LQpz{uc}", f".LQpy{uc}:"] + select_prio(S_RANK, 0, QUAD_YMAP) + [f".LQpz{uc}:"])
        elif prio == 3 or (prio == 5 and par == 0):
            L += select_prio(S_PRIO, 1, QUAD_XMAP if nk == 4 else (0, 1, 2, 3))   # the top priority moves on to the next wave of the SIMD
        elif prio == 5:
            L += select_prio(S_RANK, 0, QUAD_YMAP if nk == 4 else (0, 1, 2, 3))   # youngest first
        L += load_set(nxt, S_PF_) + [f"s_add_u32 s{S_PF_}, s{S_PF_}, 128"]
        for st in range(4):
            rslot = R[st & 1]
            if st < 3:
                L += reads(R[(st + 1) & 1], a_of(cur, REF, st + 1))
                L.append(f"s_waitcnt lgkmcnt({nk})")  # all but the reads just issued: this mic's samples are in
                nbase, ni = cur, st + 1
            else:
                L.append("s_waitcnt lgkmcnt(0)")  # this mic's samples, and the next trip's entries
                L += reads(R[0], a_of(nxt, REF, 0))
                nbase, ni = nxt, 0
            if early_x:
                xz[0] = (st + 1) & 1  # the next mic's own samples, into the other slot set, a whole stage ahead
                L += maybe_read_x(3, nbase, ni) + maybe_read_x(0, nbase, ni)
                xz[0] = st & 1
                L += pixel_ops(3, cur, st, rslot) + pixel_ops(2, cur, st, rslot) + pixel_ops(0, cur, st, rslot)
            elif chain:
                L += pixels_23_chain(cur, st, rslot) + maybe_read_x(3, nbase, ni)
                L += pixel_ops(0, cur, st, rslot) + maybe_read_x(0, nbase, ni)
            else:
                L += pixel_ops(3, cur, st, rslot) + pixel_ops(2, cur, st, rslot) + maybe_read_x(3, nbase, ni)
                L += pixel_ops(0, cur, st, rslot) + maybe_read_x(0, nbase, ni)
            L += ref_ops(cur, st, rslot)
        return L

    if item:
        return _block_quad_item(name, locals())
    L = []
    if stamp:
        L += [f"s_memtime s[{S_T0}:{S_T0 + 1}]", "s_waitcnt lgkmcnt(0)"]
    L += [f"s_mov_b32 s{S_PRIO}, %[rank]", f"s_mov_b32 s{S_RANK}, %[rank]"]
    if prio == 4:
        L += select_prio(S_RANK, 0)  # static: youngest first for the whole block
    if dma:
        L += [f"s_mov_b32 s{S_M0}, m0",
              f"s_mov_b64 s[{S_SB}:{S_SB + 1}], %[dsrc]", f"s_mov_b32 s{S_DST}, %[ddst]", f"s_mov_b32 s{S_REM}, %[dbytes]",
              f"s_mov_b32 s{S_K}, 0", f"s_mov_b32 s{S_NP}, %[dnp]"]
    L += load_set(E[0], 0, literal=True)
    L += [f"s_mov_b32 s{S_LEFT_}, %[ng]", f"s_movk_i32 s{S_PF_}, 0x80", "s_waitcnt lgkmcnt(0)"]
    if stamp:
        L += [f"s_memtime s[{S_T1}:{S_T1 + 1}]", "s_waitcnt lgkmcnt(0)", f"s_sub_u32 %[t_wait], s{S_T1}, s{S_T0}"]
    L += reads(R[0], a_of(E[0], REF, 0))
    xz[0] = 0
    L += maybe_read_x(3, E[0], 0) + maybe_read_x(0, E[0], 0)
    L += [".LQ0_%=:"] + trip_q(0)
    L += [f"s_sub_u32 s{S_LEFT_}, s{S_LEFT_}, 1", f"s_cmp_eq_u32 s{S_LEFT_}, 0", "s_cbranch_scc1 .LQdone_%="]
    L += trip_q(1)
    L += [f"s_sub_u32 s{S_LEFT_}, s{S_LEFT_}, 1", f"s_cmp_lg_u32 s{S_LEFT_}, 0", "s_cbranch_scc1 .LQ0_%="]
    L += ["s_branch .LQdone_%="] + cold + [".LQdone_%=:", "s_waitcnt lgkmcnt(0)"]  # the reads issued for a trip that does not come
    if dma:  # pieces a short chunk had no trip for
        L += [".LQmore_%=:", f"s_cmp_ge_u32 s{S_K}, s{S_NP}", "s_cbranch_scc1 .LQnomore_%="] + dma_piece() + ["s_branch .LQmore_%=", ".LQnomore_%=:",
              f"s_mov_b32 m0, s{S_M0}"]
    if prio:
        L += [f"s_setprio {QUAD_END_PRIO}"]
    if stamp:
        L += [f"s_memtime s[{S_T1}:{S_T1 + 1}]", "s_waitcnt lgkmcnt(0)", f"s_sub_u32 %[t_all], s{S_T1}, s{S_T0}"]
    body = "\n".join(f'        "{l}\\n\\t"' for l in L)
    vregs = list(range(tmp, tmp + (12 if early_x else 8) * nk + 1))
    sregs = sorted({S_TMP, S_PF_, S_LEFT_, S_RANK, S_PRIO, S_T0, S_T0 + 1, S_T1, S_T1 + 1} |
                   ({S_SB, S_SB + 1, S_DST, S_REM, S_K, S_NP, S_M0} if dma else set())) + list(range(36, 100))
    clobbers = ", ".join([f'"v{r}"' for r in vregs] + [f'"s{r}"' for r in sregs] + ['"scc"'] + (['"vcc"'] if dma else []))
    names = [f"A{p}" for p in range(4)] + ["T"] + [f"V{p}" for p in (0, 2, 3)]
    bases = A + [T] + [V[0], V[2], V[3]]
    acc_params = ", ".join(f"{'f8' if nk == 4 else 'f4'} &{n}" for n in names)
    acc_ops = ", ".join(f'"+{{v[{b}:{b + 2 * nk - 1}]}}"({n})' for n, b in zip(names, bases))
    dma_params = ", const void *dsrc, unsigned ddst, unsigned dbytes, unsigned lbytes, unsigned dnp" if dma else ""
    dma_ops = ', [dsrc] "s"(dsrc), [ddst] "s"(ddst), [dbytes] "s"(dbytes), [lbytes] "v"(lbytes), [dnp] "s"(dnp)' if dma else ""
    stamp_params = ", unsigned &t_wait, unsigned &t_all" if stamp else ""
    stamp_ops = ', [t_wait] "=&s"(t_wait), [t_all] "=&s"(t_all)' if stamp else ""
    return f"""// Four vertically adjacent pixels of the staged chunk, ng groups of four mics each (ng >= 1), with the shared
// integer-delay sum: see tools/gen_trip_asm.py.  `row` = the quad's entries of the chunk's first group in the
// quad-major table ([group][pixel][mic] x 8 bytes: 128 contiguous bytes per group); reads one group past the last.
// Accumulators are pinned: A_p v[{A[0]}+8p..], T v[{T}..], V0/V2/V3 v[{V[0]}..]/v[{V[2]}..]/v[{V[3]}..]; temps v{vregs[0]}..v{vregs[-1]},
// s{sregs[0]}..s{sregs[-1]}.
__device__ __forceinline__ void {name}({acc_params}, const void *row, int ng, unsigned lane_addr, int rank{dma_params}{stamp_params}) {{
    asm volatile(
{body}
        : {acc_ops}{stamp_ops}
        : [ptr] "s"(row), [ng] "s"(ng), [lane] "v"(lane_addr), [rank] "s"(rank){dma_ops}
        : {clobbers});
}}
"""


def _block_quad_item(name, env):
    """The quad block for a WHOLE item (frame pair x tile): the chunk loop, the wait for the refill and the workgroup
    barrier live inside the block.  A chunk's trips run exactly as in block_quad(dma=True); at a chunk's end the block
    drains its reads, issues what is left of the refill, waits for its own pieces (vmcnt(0)), meets the workgroup at
    s_barrier, flips to the other LDS image and goes on -- with the NEXT chunk's first table entries already in
    SGPRs (the table of a quad is contiguous across chunks, and the running prefetch has fetched them during the last
    trip), where a block per chunk began by waiting for its first two scalar loads with every wave of the workgroup in
    the same place.  The refill runs one chunk ahead across the whole item and on into the next item's first chunk.
    Inputs (all wave-uniform): ngf / ngl = groups of four mics in a full / in the last chunk, nch = chunks,
    isrc = chunk 0's rows in HBM (chunk c's follow dbf bytes apart), dbf / dbl = bytes of a full / the last chunk,
    nsrc, dbn = the next item's first chunk (dbn = 0: none), ddst = this wave's first LDS-DMA destination in the image
    that chunk 0 does NOT occupy, delta = (that image) - (chunk 0's image) in bytes."""
    g = env
    S_SB, S_DST, S_REM, S_K, S_NP, S_M0 = g["S_SB"], g["S_DST"], g["S_REM"], g["S_K"], g["S_NP"], g["S_M0"]
    S_TMP, S_PF_, S_LEFT_ = g["S_TMP"], g["S_PF_"], g["S_LEFT_"]
    E, R, REF_ = g["E"], g["R"], REF
    reads, maybe_read_x, trip_q, load_set, dma_piece, a_of, cold, uid = (g["reads"], g["maybe_read_x"], g["trip_q"], g["load_set"],
                                                                        g["dma_piece"], g["a_of"], g["cold"], g["uid"])
    nk, acc, tmp, A, T, V, prio = g["nk"], g["acc"], g["tmp"], g["A"], g["T"], g["V"], g["prio"]
    S_CH, S_DELTA = 19, 35  # chunks left (this one included); byte distance to the OTHER image, sign flipping per chunk
    assert nk == 4 and g["dma"] and not g["stamp"]

    def first_reads(base):
        g["xz"][0] = 0
        return reads(R[0], a_of(base, REF_, 0)) + maybe_read_x(3, base, 0) + maybe_read_x(0, base, 0)

    def refill_params(first):
        """S_SB / S_REM / S_NP / S_K for the refill that runs beside the chunk about to be swept; S_CH = chunks left with
        that chunk included.  first: S_SB holds chunk 0's source; else S_SB / S_DST have advanced S_K pieces."""
        u = uid()
        L = []
        if not first:
            L += [f"s_lshl_b32 s{S_TMP}, s{S_K}, {DMA_STRIDE.bit_length() - 1}",   # undo the pieces' advance
                  f"s_sub_u32 s{S_SB}, s{S_SB}, s{S_TMP}", f"s_subb_u32 s{S_SB + 1}, s{S_SB + 1}, 0",
                  f"s_sub_u32 s{S_DST}, s{S_DST}, s{S_TMP}",
                  f"s_sub_u32 s{S_DST}, s{S_DST}, s{S_DELTA}",            # the refill alternates images like the sweep, one ahead
                  f"s_sub_u32 s{S_DELTA}, 0, s{S_DELTA}"]
        L += [f"s_cmp_eq_u32 s{S_CH}, 1", f"s_cbranch_scc1 .Lrnext{u}",
              # another chunk of this item follows the one about to be swept: its rows lie dbf bytes on
              f"s_add_u32 s{S_SB}, s{S_SB}, %[dbf]", f"s_addc_u32 s{S_SB + 1}, s{S_SB + 1}, 0",
              f"s_mov_b32 s{S_REM}, %[dbf]", f"s_cmp_eq_u32 s{S_CH}, 2", f"s_cselect_b32 s{S_REM}, %[dbl], s{S_REM}",
              f"s_branch .Lrset{u}", f".Lrnext{u}:",
              # the chunk about to be swept is the item's last: the refill is the next item's first chunk (or nothing)
              f"s_mov_b64 s[{S_SB}:{S_SB + 1}], %[nsrc]", f"s_mov_b32 s{S_REM}, %[dbn]",
              f".Lrset{u}:",
              f"s_mov_b32 s{S_K}, 0", f"s_add_u32 s{S_NP}, s{S_REM}, {DMA_STRIDE - 1}",
              f"s_lshr_b32 s{S_NP}, s{S_NP}, {DMA_STRIDE.bit_length() - 1}"]
        assert DMA_WAVES == 16, "round 5 dropped the block's wave input (the refill by fewer than 16 waves was measured slower in round 2)"
        return L

    def boundary(next_set, resume):
        u = uid()
        L = ["s_waitcnt lgkmcnt(0)",  # this chunk's last samples, and the reads issued for a trip that does not come
             f".LQmore{u}:", f"s_cmp_ge_u32 s{S_K}, s{S_NP}", f"s_cbranch_scc1 .LQnomore{u}"] + dma_piece() + [f"s_branch .LQmore{u}", f".LQnomore{u}:"]
        L += ["s_waitcnt vmcnt(0)", "s_barrier",  # my pieces of the next chunk have landed; so has everybody's, and all are done with this image
              f"s_sub_u32 s{S_CH}, s{S_CH}, 1", f"s_cmp_eq_u32 s{S_CH}, 0", "s_cbranch_scc1 .LQexit_%=",
              f"v_add_u32 %[lane], s{S_DELTA}, %[lane]"]  # the sweep moves to the image just filled
        L += refill_params(first=False)
        L += [f"s_mov_b32 s{S_LEFT_}, %[ngf]", f"s_cmp_eq_u32 s{S_CH}, 1", f"s_cselect_b32 s{S_LEFT_}, %[ngl], s{S_LEFT_}"]
        L += first_reads(next_set) + [f"s_branch {resume}"]
        return L

    L = [f"s_mov_b32 s{S_PRIO}, %[rank]", f"s_mov_b32 s{S_RANK}, %[rank]"]
    if prio == 4:
        L += select_prio(S_RANK, 0)
    # the persistent kernel's ticket for its next-but-one item (round 5, as in block_exact_nd): lane 0 of the wave that is handed a
    # queue's address adds one to that counter here, beside the sweep; the first chunk boundary's vmcnt(0) proves the answer
    L += ["s_cmp_eq_u64 %[qptr], 0", "s_cbranch_scc1 .LQnoq_%=", "s_mov_b64 exec, 1",
          f"v_mov_b32 v{R[1]}, 0", f"v_mov_b32 v{R[1] + 1}, 1", f"global_atomic_add %[ticket], v{R[1]}, v{R[1] + 1}, %[qptr] sc0",
          "s_mov_b64 exec, -1", ".LQnoq_%=:"]
    L += [f"s_mov_b32 s{S_M0}, m0", f"s_mov_b32 s{S_CH}, %[nch]", f"s_mov_b32 s{S_DELTA}, %[delta]",
          f"s_mov_b64 s[{S_SB}:{S_SB + 1}], %[isrc]", f"s_mov_b32 s{S_DST}, %[ddst]"]
    L += refill_params(first=True)
    L += load_set(E[0], 0, literal=True)
    L += [f"s_mov_b32 s{S_LEFT_}, %[ngf]", f"s_cmp_eq_u32 s{S_CH}, 1", f"s_cselect_b32 s{S_LEFT_}, %[ngl], s{S_LEFT_}",
          f"s_movk_i32 s{S_PF_}, 0x80", "s_waitcnt lgkmcnt(0)"]
    L += first_reads(E[0])
    L += [".LQ0_%=:"] + trip_q(0)
    L += [f"s_sub_u32 s{S_LEFT_}, s{S_LEFT_}, 1", f"s_cmp_eq_u32 s{S_LEFT_}, 0", "s_cbranch_scc1 .LQbndA_%="]
    L += [".LQ1_%=:"] + trip_q(1)
    L += [f"s_sub_u32 s{S_LEFT_}, s{S_LEFT_}, 1", f"s_cmp_lg_u32 s{S_LEFT_}, 0", "s_cbranch_scc1 .LQ0_%="]
    L += boundary(E[0], ".LQ0_%=")                    # the chunk ended on a trip out of set 1: the next begins on set 0
    L += [".LQbndA_%=:"] + boundary(E[1], ".LQ1_%=")  # ... on a trip out of set 0: the next begins on set 1
    L += cold
    L += [".LQexit_%=:", f"s_mov_b32 m0, s{S_M0}"]
    if prio:
        L += [f"s_setprio {QUAD_END_PRIO}"]
    body = "\n".join(f'        "{l}\\n\\t"' for l in L)
    vregs = list(range(tmp, tmp + 8 * nk + 1))
    sregs = sorted({S_TMP, S_PF_, S_LEFT_, S_RANK, S_PRIO, S_SB, S_SB + 1, S_DST, S_REM, S_K, S_NP, S_M0, S_CH, S_DELTA}) + list(range(36, 100))
    clobbers = ", ".join([f'"v{r}"' for r in vregs] + [f'"s{r}"' for r in sregs] + ['"scc"', '"vcc"', '"memory"'])
    names = [f"A{p}" for p in range(4)] + ["T"] + [f"V{p}" for p in (0, 2, 3)]
    bases = A + [T] + [V[0], V[2], V[3]]
    acc_params = ", ".join(f"f8 &{n}" for n in names)
    acc_ops = ", ".join(f'"+{{v[{b}:{b + 2 * nk - 1}]}}"({n})' for n, b in zip(names, bases))
    return f"""// A whole item (frame pair x tile) of the quad shape: chunk loop, refill, vmcnt wait and workgroup barrier inside the
// block (_block_quad_item in tools/gen_trip_asm.py).  `row` = the quad's entries of the item's first group (contiguous
// across chunks); lane_addr is the sweep's LDS address in chunk 0's image on entry and wherever the last flip left it
// on exit (the caller tracks the image by the chunk count).  Executes nch s_barrier instructions.
__device__ __forceinline__ void {name}({acc_params}, const void *row, int ngf, int ngl, int nch, unsigned &lane_addr, int rank,
                                       const void *isrc, unsigned dbf, unsigned dbl, const void *nsrc, unsigned dbn, unsigned ddst,
                                       int delta, unsigned lbytes, const unsigned *qptr, unsigned &ticket) {{
    asm volatile(
{body}
        : {acc_ops}, [lane] "+v"(lane_addr), [ticket] "=&v"(ticket)
        : [ptr] "s"(row), [ngf] "s"(ngf), [ngl] "s"(ngl), [nch] "s"(nch), [rank] "s"(rank), [isrc] "s"(isrc), [dbf] "s"(dbf), [dbl] "s"(dbl),
          [nsrc] "s"(nsrc), [dbn] "s"(dbn), [ddst] "s"(ddst), [delta] "s"(delta), [lbytes] "v"(lbytes), [qptr] "s"(qptr)
        : {clobbers});
}}
"""


def block_quad_ar(name, stamp=False, prio=None, nk=2, acc=QUAD1_ACC[0], tmp=QUAD1_TMP, dma=False):
    """The quad block for the single-frame layout, "always read" schedule.  With two register pairs per
    accumulator a mic's share of the work is only ~13 VALU instructions, too short a stage to hide an LDS read
    that is issued -- conditionally, late in the previous stage -- only when a pixel's integer delay differs from
    the reference's.  Here pixel 0 and pixel 3 ALWAYS read their own samples, one full stage ahead and together
    with the reference's (three read groups per mic into alternating slot sets, one static wait count), their
    FMAs always take their own samples, and only the correction of the shared sum (V_p += x_p - x_ref) stays
    conditional.  Pixel 2 takes the reference's samples, or pixel 3's when it carries pixel 3's address, or --
    rare -- reads on the spot.  LDS reads per mic and quad: 3 groups instead of 1.7 on average (the LDS has the
    room: the quad arithmetic needs a third of the reads of the older shapes); VALU instructions: the same.

    dma=True (the halves layout of das_quadh_kernel, whose chunks are contiguous in HBM): the block also issues the
    refill of the other LDS image, one 16 KiB piece at the head of each trip, exactly as block_quad does."""
    prio = PRIO if prio is None else prio
    assert not (dma and stamp), "the stamped builds keep the refill outside the block (s28..s31 hold the stamps)"
    S_SB, S_DST, S_REM, S_K, S_NP, S_M0 = 20, 25, 28, 29, 30, 31
    w = 2 * nk
    A = [acc + w * p for p in range(4)]
    T = acc + 4 * w
    V = {0: acc + 5 * w, 2: acc + 6 * w, 3: acc + 7 * w}
    Z = [dict(R=tmp + 3 * w * z, X0=tmp + 3 * w * z + w, X23=tmp + 3 * w * z + 2 * w) for z in range(2)]
    spot = tmp + 6 * w
    addr_t = tmp + 7 * w
    n_tmp = 7 * w + 1
    E = (36, 68)
    S_TMP, S_PF_, S_LEFT_ = 22, 23, 24

    def f_of(base, p, i):
        return base + 8 * p + 2 * i

    def a_of(base, p, i):
        return base + 8 * p + 2 * i + 1

    def pair(r, k):
        return f"v[{r + 2 * k}:{r + 2 * k + 1}]"

    def reads(slot, addr_sgpr):
        L = [f"v_add_u32 v{addr_t}, s{addr_sgpr}, %[lane]"]
        for k in range(nk):
            off = f" offset:{512 * k}" if k else ""
            L.append(f"ds_read_b64 {pair(slot, k)}, v{addr_t}{off}")
        return L

    def uid():
        COUNTER[0] += 1
        return f"%=_{COUNTER[0]}"

    cold = []

    def fmas(p, fs, src):
        return [f"v_pk_fma_f32 {pair(A[p], k)}, {fs}, {pair(src, k)}, {pair(A[p], k)} op_sel_hi:[0,1,1]" for k in range(nk)]

    def correct(p, own, ref):
        return ([f"v_pk_add_f32 {pair(V[p], k)}, {pair(V[p], k)}, {pair(own, k)}" for k in range(nk)] +
                [f"v_pk_add_f32 {pair(V[p], k)}, {pair(V[p], k)}, {pair(ref, k)} neg_lo:[0,1] neg_hi:[0,1]" for k in range(nk)])

    def issue(z, base, i):
        skip = os.environ.get("QUAD1_TIMING_SKIP", "")  # timing-only builds (wrong results): "0" = no X0 reads, "03" = neither X0 nor X23
        return (reads(Z[z]["R"], a_of(base, REF, i)) + ([] if "0" in skip else reads(Z[z]["X0"], a_of(base, 0, i))) +
                ([] if "3" in skip else reads(Z[z]["X23"], a_of(base, 3, i))))

    def stage(base, i, z):
        R, X0, X23 = Z[z]["R"], Z[z]["X0"], Z[z]["X23"]
        nobranch = os.environ.get("QUAD1_NOBRANCH") == "1"  # timing-only builds: every pixel taken as coinciding (wrong results)
        fs = lambda p: f"s[{f_of(base, p, i)}:{f_of(base, p, i) + 1}]"
        L = []
        # pixel 3: own samples always; the shared sum's correction only where its delay differs
        u = uid()
        u2 = uid()
        if PAIR23:
            # pixels 2 and 3 away from the reference together (a delay that steps once inside the quad): one difference
            # x_b - x_ref, formed in place (the slot is theirs alone), corrects both shared sums
            both = (fmas(2, fs(2), X23) +
                    [f"v_pk_add_f32 {pair(X23, k)}, {pair(X23, k)}, {pair(R, k)} neg_lo:[0,1] neg_hi:[0,1]" for k in range(nk)] +
                    [f"v_pk_add_f32 {pair(V[q], k)}, {pair(V[q], k)}, {pair(X23, k)}" for q in (3, 2) for k in range(nk)])
            cold.extend([f".Lac{u}:", f"s_cmp_eq_u32 s{a_of(base, 2, i)}, s{a_of(base, 3, i)}", f"s_cbranch_scc1 .Laf{u}"] +
                        correct(3, X23, R) + [f"s_branch .Lad{u}", f".Laf{u}:"] + both + [f"s_branch .Lad{u2}"])
        else:
            cold.extend([f".Lac{u}:"] + correct(3, X23, R) + [f"s_branch .Lad{u}"])
        L += fmas(3, fs(3), X23) + ([] if nobranch else [f"s_cmp_lg_u32 s{a_of(base, 3, i)}, s{a_of(base, REF, i)}", f"s_cbranch_scc1 .Lac{u}"]) + [f".Lad{u}:"]
        # pixel 2: the reference's samples, or pixel 3's, or (rare) its own read on the spot
        u = u2
        cold.extend([f".Lac{u}:",
                     f"s_cmp_eq_u32 s{a_of(base, 2, i)}, s{a_of(base, 3, i)}",
                     f"s_cbranch_scc0 .Lae{u}"] + fmas(2, fs(2), X23) + correct(2, X23, R) + [f"s_branch .Lad{u}",
                     f".Lae{u}:"] + reads(spot, a_of(base, 2, i)) + ["s_waitcnt lgkmcnt(0)"] + fmas(2, fs(2), spot) + correct(2, spot, R) +
                    [f"s_branch .Lad{u}"])
        L += ([] if nobranch else [f"s_cmp_lg_u32 s{a_of(base, 2, i)}, s{a_of(base, REF, i)}", f"s_cbranch_scc1 .Lac{u}"]) + fmas(2, fs(2), R) + [f".Lad{u}:"]
        # pixel 0
        u = uid()
        cold.extend([f".Lac{u}:"] + correct(0, X0, R) + [f"s_branch .Lad{u}"])
        L += fmas(0, fs(0), X0) + ([] if nobranch else [f"s_cmp_lg_u32 s{a_of(base, 0, i)}, s{a_of(base, REF, i)}", f"s_cbranch_scc1 .Lac{u}"]) + [f".Lad{u}:"]
        # the reference: the shared sum and its own A
        L += [f"v_pk_add_f32 {pair(T, k)}, {pair(T, k)}, {pair(R, k)}" for k in range(nk)] + fmas(REF, fs(REF), R)
        return L

    def load_set(base, off, literal=False):
        if literal:
            return [f"s_load_dwordx16 s[{base}:{base + 15}], %[ptr], {hex(off)}",
                    f"s_load_dwordx16 s[{base + 16}:{base + 31}], %[ptr], {hex(off + 64)}"]
        return [f"s_load_dwordx16 s[{base}:{base + 15}], %[ptr], s{off}",
                f"s_add_u32 s{S_TMP}, s{off}, 64",
                f"s_load_dwordx16 s[{base + 16}:{base + 31}], %[ptr], s{S_TMP}"]

    def dma_piece():
        """one 16 KiB piece of the refill, if any is left"""
        u = uid()
        return [f"s_cmp_ge_u32 s{S_K}, s{S_NP}", f"s_cbranch_scc1 .Ldskip{u}",
                f"v_cmp_gt_u32 vcc, s{S_REM}, %[lbytes]",  # lanes whose 16 bytes lie inside the chunk
                "s_mov_b64 exec, vcc",                      # (the block runs with all 64 lanes on: restored to -1 below)
                f"s_mov_b32 m0, s{S_DST}", "s_nop 0",
                f"global_load_lds_dwordx4 %[lbytes], s[{S_SB}:{S_SB + 1}]",
                "s_mov_b64 exec, -1",
                f"s_add_u32 s{S_SB}, s{S_SB}, {hex(DMA_STRIDE)}", f"s_addc_u32 s{S_SB + 1}, s{S_SB + 1}, 0",
                f"s_add_u32 s{S_DST}, s{S_DST}, {hex(DMA_STRIDE)}", f"s_sub_u32 s{S_REM}, s{S_REM}, {hex(DMA_STRIDE)}",
                f"s_add_u32 s{S_K}, s{S_K}, 1", f".Ldskip{u}:"]

    def trip_q(par):
        cur, nxt = E[par], E[1 - par]
        L = dma_piece() if dma else []
        if QUAD1_PRIO_COARSE and prio == 5:  # tuning: the scheme changes per pair of trips (1) / two pairs (2), applied once per pair
            if par == 0:
                COUNTER[0] += 1
                uc = f"%=_{COUNTER[0]}"
                COUNTER[0] += 1
                L += ([f"s_bitcmp1_b32 s{S_LEFT_}, {QUAD1_PRIO_COARSE}", f"s_cbranch_scc1 .LQpy{uc}"] + select_prio(S_PRIO, 1) +
                      [f"s_branch .LQpz{uc}", f".LQpy{uc}:"] + select_prio(S_RANK, 0) + [f".LQpz{uc}:"])
        elif prio == 3 or (prio == 5 and par == 0):
            L += select_prio(S_PRIO, 1)
        elif prio == 5:
            L += select_prio(S_RANK, 0)
        L += load_set(nxt, S_PF_) + [f"s_add_u32 s{S_PF_}, s{S_PF_}, 128"]
        for st in range(4):
            if st < 3:
                L += issue((st + 1) & 1, cur, st + 1)
                n_groups = 3 - len(os.environ.get("QUAD1_TIMING_SKIP", ""))
                L.append(f"s_waitcnt lgkmcnt({n_groups * nk})")  # all but the three read groups just issued
            else:
                L.append("s_waitcnt lgkmcnt(0)")  # this mic's samples, and the next trip's entries
                L += issue(0, nxt, 0)
            L += stage(cur, st, st & 1)
        return L

    L = []
    if stamp:
        L += [f"s_memtime s[{S_T0}:{S_T0 + 1}]", "s_waitcnt lgkmcnt(0)"]
    L += [f"s_mov_b32 s{S_PRIO}, %[rank]", f"s_mov_b32 s{S_RANK}, %[rank]"]
    if prio == 4:
        L += select_prio(S_RANK, 0)
    if dma:
        L += [f"s_mov_b32 s{S_M0}, m0",
              f"s_mov_b64 s[{S_SB}:{S_SB + 1}], %[dsrc]", f"s_mov_b32 s{S_DST}, %[ddst]", f"s_mov_b32 s{S_REM}, %[dbytes]",
              f"s_mov_b32 s{S_K}, 0", f"s_mov_b32 s{S_NP}, %[dnp]"]
    L += load_set(E[0], 0, literal=True)
    L += [f"s_mov_b32 s{S_LEFT_}, %[ng]", f"s_movk_i32 s{S_PF_}, 0x80", "s_waitcnt lgkmcnt(0)"]
    if stamp:
        L += [f"s_memtime s[{S_T1}:{S_T1 + 1}]", "s_waitcnt lgkmcnt(0)", f"s_sub_u32 %[t_wait], s{S_T1}, s{S_T0}"]
    L += issue(0, E[0], 0)
    L += [".LQ0_%=:"] + trip_q(0)
    L += [f"s_sub_u32 s{S_LEFT_}, s{S_LEFT_}, 1", f"s_cmp_eq_u32 s{S_LEFT_}, 0", "s_cbranch_scc1 .LQdone_%="]
    L += trip_q(1)
    L += [f"s_sub_u32 s{S_LEFT_}, s{S_LEFT_}, 1", f"s_cmp_lg_u32 s{S_LEFT_}, 0", "s_cbranch_scc1 .LQ0_%="]
    L += ["s_branch .LQdone_%="] + cold + [".LQdone_%=:", "s_waitcnt lgkmcnt(0)"]
    if dma:  # pieces a short chunk had no trip for
        L += [".LQmore_%=:", f"s_cmp_ge_u32 s{S_K}, s{S_NP}", "s_cbranch_scc1 .LQnomore_%="] + dma_piece() + ["s_branch .LQmore_%=", ".LQnomore_%=:",
              f"s_mov_b32 m0, s{S_M0}"]
    if prio:
        L += [f"s_setprio {BLOCK_END_PRIO}"]
    if stamp:
        L += [f"s_memtime s[{S_T1}:{S_T1 + 1}]", "s_waitcnt lgkmcnt(0)", f"s_sub_u32 %[t_all], s{S_T1}, s{S_T0}"]
    body = "\n".join(f'        "{l}\\n\\t"' for l in L)
    vregs = list(range(tmp, tmp + n_tmp))
    sregs = sorted({S_TMP, S_PF_, S_LEFT_, S_RANK, S_PRIO, S_T0, S_T0 + 1, S_T1, S_T1 + 1} |
                   ({S_SB, S_SB + 1, S_DST, S_REM, S_K, S_NP, S_M0} if dma else set())) + list(range(36, 100))
    clobbers = ", ".join([f'"v{r}"' for r in vregs] + [f'"s{r}"' for r in sregs] + ['"scc"'] + (['"vcc"'] if dma else []))
    dma_params = ", const void *dsrc, unsigned ddst, unsigned dbytes, unsigned lbytes, unsigned dnp" if dma else ""
    dma_ops = ', [dsrc] "s"(dsrc), [ddst] "s"(ddst), [dbytes] "s"(dbytes), [lbytes] "v"(lbytes), [dnp] "s"(dnp)' if dma else ""
    names = [f"A{p}" for p in range(4)] + ["T"] + [f"V{p}" for p in (0, 2, 3)]
    bases = A + [T] + [V[0], V[2], V[3]]
    vt = "f8" if nk == 4 else "f4"
    acc_params = ", ".join(f"{vt} &{n}" for n in names)
    acc_ops = ", ".join(f'"+{{v[{b}:{b + w - 1}]}}"({n})' for n, b in zip(names, bases))
    stamp_params = ", unsigned &t_wait, unsigned &t_all" if stamp else ""
    stamp_ops = ', [t_wait] "=&s"(t_wait), [t_all] "=&s"(t_all)' if stamp else ""
    return f"""// Four vertically adjacent pixels of the staged chunk, single-frame layout, shared integer-delay sum, reads always
// one stage ahead: see block_quad_ar in tools/gen_trip_asm.py.  `row` = the quad's entries of the chunk's first
// group in the quad-major table; reads one group past the last.  Accumulators pinned from v{acc}, temps v{vregs[0]}..v{vregs[-1]}.
__device__ __forceinline__ void {name}({acc_params}, const void *row, int ng, unsigned lane_addr, int rank{dma_params}{stamp_params}) {{
    asm volatile(
{body}
        : {acc_ops}{stamp_ops}
        : [ptr] "s"(row), [ng] "s"(ng), [lane] "v"(lane_addr), [rank] "s"(rank){dma_ops}
        : {clobbers});
}}
"""


FIR_ACC, FIR_TMP = 8, 88  # block_fir8: accumulators of the four pixels pinned at v[8:39], temps v88..v121 (static pitch: ..v124)
FIR_STATIC_PB = 768       # sweep_fir8_planes_static: rows of 384 samples (windows of 321..384: every BASELINE shape but c1)


# block_fir8 wave priorities (s_setprio, as in the quad blocks): 0 none; 1 the top priority rotates through the four waves of
# the SIMD once per group of four items; 2 rotation and youngest-first alternate within a group; 3 static youngest-first;
# 4 / 5 rotation every two items / every item; 6 rotation and youngest-first item by item; 7 rotation on even groups,
# youngest-first on odd ones (8: pairs of groups; 11: by pixel); 10 rotation every other group only.  Measured on c3 (512 mics x
# 128x128, 128 frames per launch, alternating runs on one box, ms per launch): 0: 45.79, 3: 45.60, 1: 44.05, 5: 44.03,
# 4: 43.93, 6: 43.95, 2: 43.82, 10: 43.58 | 7: 42.87, 8: 42.92, 11: 42.89 (a second box: 7: 43.24 against 2: 43.74).
FIR_PRIO = int(os.environ.get("FIR_PRIO", "7"))


def block_fir8(name, acc=FIR_ACC, tmp=FIR_TMP, timing="", static_pb=0, prio=None):
    """Four pixels of the staged chunk with the 8-tap table variant of delay() (delay.cpp:31-40) on the frame-pair
    layout stored as four sample planes (sample i of a row lives in plane i % 4 at index i / 4).

    Lane l owns the four CONSECUTIVE outputs n = 4l .. 4l+3, so an item (pixel x mic, two frames) needs the eleven
    samples X[e + 4l + w], w = 0..10, and gives 32 v_pk_fma_f32 for 11 ds_read_b64 -- against 32 reads when a lane
    owns outputs 64 apart (the LDS array, shared by the four SIMDs, then limits the sweep at half the VALU rate).
    Sample w sits in the plane (e + w) % 4: an item needs the four byte addresses of the rotated planes c = w % 4
    (4 v_add_u32 per item) and w / 4 is an immediate offset of 0, 8 or 16 bytes; consecutive lanes read consecutive
    8-byte elements, so every read is conflict-free.  The coefficient of tap t is the low or the high dword of an
    aligned SGPR pair, picked by op_sel.  Taps accumulate in the reference's order t = 0..7 for every output.

    Table (round 3: 4 bytes per item where round 2 had 64): ONE dword per (pixel, mic),
        bits 0..17  the LDS byte address of X[off] (plane r = first % 4, index first / 4 of the mic's row),
        bits 18..19 r,   bits 20..26 the coefficient row k (delay.cpp:32-33; row 101 = zeros, for padding mics);
    the other three plane addresses follow in the scalar ALU -- addr[c] = addr[0] + c PB, minus 4 PB - 8 where r + c
    wraps past plane 3 (PB = bytes of one plane) -- and the eight coefficients come by a second, dependent scalar
    load from the 3.2 KB coefficient table, which lives in the scalar cache.  The 512-mic 128x128 table shrinks from
    537 MB, re-streamed from beyond the L2 by every group of workgroups, to 33.5 MB.

    Pipeline per item i, at its head, after the one lgkmcnt(0) that proves everything requested earlier has landed:
    request the entry of item i+4; take the entry of item i+3 (requested an item ago) apart and request its
    coefficients (four sets of 8 SGPRs in rotation: they are needed from item i+2's address adds ... no, from item i+3's
    FMAs on); compute the four addresses of item i+1 (single set: the v_adds of this item consume them).  Scalar
    loads share lgkmcnt with the LDS reads and return out of order, so only an lgkmcnt(0) proves a load has landed:
    one per item -- and the LDS reads are placed so that this drain finds nothing young in flight: samples 0..3 and
    4..7 of the NEXT item are requested after this item's FMAs on its own samples 0..3 (22 FMAs before the drain;
    samples 4..7 have two register sets), samples 8..10 right after the drain (26 FMAs before their first use, waits
    counted in younger LDS reads only).

    The block sweeps its four pixels one after the other, `n4` groups of four items each (chunks and table rows are
    multiples of four mics; padding entries point at the zero coefficient row), and the stream of entry requests
    runs on from one pixel's row into the next: a wave meets the scalar-load latency once per block.

    static_pb > 0 (late round 3): the block for rows staged with a plane pitch of exactly `static_pb` bytes.  With the
    pitch a constant, every sample's offset from the row's plane 0 is an immediate once r is known -- sample w sits in
    plane (r + w) % 4 at index q + l + (r + w) / 4, i.e. at (row base + 8 q + 8 l) + [(r + w) % 4] pitch + 8 [(r + w) / 4] --
    so an item needs ONE v_add_u32 instead of four, and its eleven reads go out together, in one of four copies picked by
    two scalar compares on r (the next item's samples 8..10 get a second register set).  33 VALU instructions per item
    instead of 36; the compares cost the wave ~100 cycles of its own time per item, of the ~860 its share of the SIMD
    gives it.  Same table entries (the block subtracts r * pitch from the address field), same sums."""
    prio = FIR_PRIO if prio is None else prio
    ENT = 36                      # s[36:39]: ring of four entry dwords, entry i in s[36 + i % 4]
    CO = (40, 48, 56, 64)         # coefficient sets, item i uses CO[i % 4]
    AD = 72                       # s[72:75]: the plane addresses of the item whose reads are being issued
    S_PB, S_PB2, S_PB3, S_ADJ = 76, 77, 78, 79   # c * PB and 8 - 4 PB
    S_T, S_R, S_K = 80, 81, 82
    S_OFF_, S_LEFT_, S_PTR = 23, 24, 84  # s[84:85]: the row the entry requests currently run in
    X03 = [tmp + 2 * w for w in range(4)]
    X47 = [[tmp + 8 + 8 * z + 2 * w for w in range(4)] for z in range(2)]
    X8 = [tmp + 24 + 2 * w for w in range(3)]
    VA = [tmp + 30 + c for c in range(4)]
    n_tmp = 34
    if static_pb:
        X8 = [[tmp + 24 + 6 * z + 2 * w for w in range(3)] for z in range(2)]  # samples 8..10: two sets as well
        VA = [tmp + 36]
        n_tmp = 37
    SINK = tmp - 4  # v84: destination of the table prefetch (never read)

    def accp(pp, o):
        b = acc + 8 * pp + 2 * o
        return f"v[{b}:{b + 1}]"

    def xreg(w, z):
        if static_pb and w >= 8:
            return X8[z][w - 8]
        return X03[w] if w < 4 else (X47[z][w - 4] if w < 8 else X8[w - 8])

    def xp(w, z):
        r = xreg(w, z)
        return f"v[{r}:{r + 1}]"

    def addresses(ent):
        """the four plane addresses of the item whose entry sits in s{ent} -> s[AD:AD+3]"""
        L = [f"s_and_b32 s{AD}, s{ent}, 0x3ffff", f"s_bfe_u32 s{S_R}, s{ent}, 0x20012"]  # address; r = bits 18..19
        for c, pb in ((1, S_PB), (2, S_PB2), (3, S_PB3)):
            L += [f"s_add_u32 s{AD + c}, s{AD}, s{pb}",
                  f"s_cmp_ge_u32 s{S_R}, {4 - c}",             # r + c >= 4: the plane index wraps, the element index steps
                  f"s_cselect_b32 s{S_T}, s{S_ADJ}, 0",
                  f"s_add_u32 s{AD + c}, s{AD + c}, s{S_T}"]
        return L

    def adds():
        return [f"v_add_u32 v{VA[c]}, s{AD + c}, %[lane]" for c in range(4)]

    def read(w, z):
        off = f" offset:{8 * (w // 4)}" if w >= 4 else ""
        if timing == "nolds":  # (timing-only builds: no LDS traffic)
            return None
        return f"ds_read_b64 {xp(w, z)}, v{VA[w % 4]}{off}"

    def reads(ws, z):
        return [r for r in (read(w, z) for w in ws) if r]

    def addresses_static(ent):
        """plane-0 address (row base + 8 q) of the item whose entry sits in s{ent} -> s{AD}; r -> s{S_R}"""
        return [f"s_and_b32 s{AD}, s{ent}, 0x3ffff", f"s_bfe_u32 s{S_R}, s{ent}, 0x20012",
                f"s_mul_i32 s{S_T}, s{S_R}, {static_pb}", f"s_sub_u32 s{AD}, s{AD}, s{S_T}"]

    def reads_static(z, r):
        L = []
        for w in range(11):
            off = ((r + w) % 4) * static_pb + 8 * ((r + w) // 4)
            L.append(f"ds_read_b64 {xp(w, z)}, v{VA[0]}" + (f" offset:{off}" if off else ""))
        return L

    def dispatch_reads(z):
        """all eleven reads of the item whose r sits in s{S_R}, into register set z"""
        COUNTER[0] += 1
        u = f"%=_{COUNTER[0]}"
        return ([f"s_cmp_lt_u32 s{S_R}, 2", f"s_cbranch_scc1 .LFr01{u}", f"s_cmp_eq_u32 s{S_R}, 2", f"s_cbranch_scc1 .LFr2{u}"] +
                reads_static(z, 3) + [f"s_branch .LFrj{u}", f".LFr2{u}:"] + reads_static(z, 2) + [f"s_branch .LFrj{u}", f".LFr01{u}:",
                f"s_cmp_eq_u32 s{S_R}, 0", f"s_cbranch_scc1 .LFr0{u}"] + reads_static(z, 1) + [f"s_branch .LFrj{u}", f".LFr0{u}:"] +
                reads_static(z, 0) + [f".LFrj{u}:"])

    def load_entry(slot):
        if timing == "noload":  # (timing-only builds: the first entries over and over)
            return []
        return [f"s_load_dword s{ENT + slot}, s[{S_PTR}:{S_PTR + 1}], s{S_OFF_}", f"s_add_u32 s{S_OFF_}, s{S_OFF_}, 4"]

    def load_coeffs(ent, cset):
        if timing == "noload":
            return []
        return [f"s_bfe_u32 s{S_K}, s{ent}, 0x70014",   # k = bits 20..26
                f"s_lshl_b32 s{S_K}, s{S_K}, 5",        # 32 bytes per coefficient row
                f"s_load_dwordx8 s[{cset}:{cset + 7}], %[coef], s{S_K}"]

    def fma(pp, base, o, t, w, z):
        sp = base + (t & ~1)
        sel = t & 1
        return (f"v_pk_fma_f32 {accp(pp, o)}, s[{sp}:{sp + 1}], {xp(w, z)}, {accp(pp, o)} "
                f"op_sel:[{sel},0,0] op_sel_hi:[{sel},1,1]")

    def fmas(pp, cur, w, z):
        return [fma(pp, cur, o, w - o, w, z) for o in range(max(0, w - 7), min(3, w) + 1)]

    def item(pp, k):
        cur = CO[k % 4]
        z = k & 1
        L = ["s_waitcnt lgkmcnt(0)"]  # everything requested before this item has landed; so have its samples 0..7
        if k == 0 and pp < 3:  # a pixel's last group: the entry four items on is the next pixel's first
            L += [f"s_cmp_eq_u32 s{S_LEFT_}, 1", f"s_cselect_b64 s[{S_PTR}:{S_PTR + 1}], %[row{pp + 1}], s[{S_PTR}:{S_PTR + 1}]",
                  f"s_cselect_b32 s{S_OFF_}, 0, s{S_OFF_}"]
        L += load_entry(k % 4)                                # entry of item i + 4 (this item's slot is free)
        L += load_coeffs(ENT + (k + 3) % 4, CO[(k + 3) % 4])  # coefficients of item i + 3 (its entry landed an item ago)
        if prio == 1 and k == 0:
            L += select_prio(S_PRIO, 1)       # the top priority moves on to the next wave of the SIMD
        elif prio == 2 and k == 0:
            L += select_prio(S_PRIO, 1)
        elif prio == 2 and k == 2:
            L += select_prio(S_RANK, 0)       # youngest first
        elif prio == 4 and k in (0, 2):
            L += select_prio(S_PRIO, 1)
        elif prio == 5:
            L += select_prio(S_PRIO, 1)
        elif prio == 6:
            L += select_prio(S_PRIO, 1) if k in (0, 2) else select_prio(S_RANK, 0)
        elif prio == 10 and k == 0:            # rotation every other group only
            COUNTER[0] += 1
            u7 = f"%=_{COUNTER[0]}"
            COUNTER[0] += 1
            L += [f"s_bitcmp1_b32 s{S_LEFT_}, 0", f"s_cbranch_scc1 .LFpz{u7}"] + select_prio(S_PRIO, 1) + [f".LFpz{u7}:"]
        elif prio == 11 and k == 0:            # by pixel: even pixels rotate each group, odd pixels run youngest-first
            L += select_prio(S_PRIO, 1) if pp % 2 == 0 else (select_prio(S_RANK, 0) if True else [])
        elif prio in (7, 8) and k == 0:        # rotation every other group (8: every other pair of groups), youngest-first in between
            COUNTER[0] += 1
            u7 = f"%=_{COUNTER[0]}"
            COUNTER[0] += 1
            L += ([f"s_bitcmp1_b32 s{S_LEFT_}, {1 if prio == 8 else 0}", f"s_cbranch_scc1 .LFpy{u7}"] + select_prio(S_PRIO, 1) +
                  [f"s_branch .LFpz{u7}", f".LFpy{u7}:"] + select_prio(S_RANK, 0) + [f".LFpz{u7}:"])
        if static_pb:
            L += addresses_static(ENT + (k + 1) % 4)          # plane-0 address and r of item i + 1
            for w in range(4):
                L += fmas(pp, cur, w, z)
            L += [f"v_add_u32 v{VA[0]}, s{AD}, %[lane]"] + dispatch_reads(z ^ 1)  # all eleven samples of item i + 1
            for w in range(4, 11):
                L += fmas(pp, cur, w, z)
            return L
        L += reads(range(8, 11), z)
        L += addresses(ENT + (k + 1) % 4)                     # plane addresses of item i + 1
        for w in range(4):
            L += fmas(pp, cur, w, z)
        L += adds() + reads(range(0, 4), z ^ 1) + reads(range(4, 8), z ^ 1)
        for w in range(4, 8):
            L += fmas(pp, cur, w, z)
        younger = 10  # samples 9, 10 and the next item's 0..7
        for w in range(8, 11):
            if timing != "nolds":
                L.append(f"s_waitcnt lgkmcnt({younger})")
            younger -= 1
            L += fmas(pp, cur, w, z)
        return L

    L = [f"s_mov_b64 s[{S_PTR}:{S_PTR + 1}], %[row0]", f"s_mov_b32 s{S_OFF_}, 0"]
    if prio:
        L += [f"s_mov_b32 s{S_PRIO}, %[rank]", f"s_mov_b32 s{S_RANK}, %[rank]"]
    if prio == 3:
        L += select_prio(S_RANK, 0)
    L += [f"s_mov_b32 s{S_PB}, %[pb]", f"s_lshl_b32 s{S_PB2}, s{S_PB}, 1", f"s_add_u32 s{S_PB3}, s{S_PB2}, s{S_PB}",
         f"s_lshl_b32 s{S_ADJ}, s{S_PB}, 2", f"s_sub_u32 s{S_ADJ}, 8, s{S_ADJ}"]
    # warm the L2 with the NEXT chunk's entries of these four pixels (one dword per lane from pfoff = chunk bytes +
    # 4 * lane: a chunk is at most 64 entries; pfn = 0: last chunk): plain loads into a sink nobody reads, drained at
    # the block's end.  A scalar load that misses the L2 costs the wave more than the lead the requests have.
    L += ["s_cmp_lt_u32 %[pfn], 1", "s_cbranch_scc1 .LFpf_%="]
    L += [f"global_load_dword v{SINK}, %[pfoff], %[row{pp}]" for pp in range(4)]
    L += [".LFpf_%=:"]
    keep = timing
    timing = ""
    L += [f"s_load_dwordx4 s[{ENT}:{ENT + 3}], s[{S_PTR}:{S_PTR + 1}], 0x0", f"s_mov_b32 s{S_OFF_}, 16", "s_waitcnt lgkmcnt(0)"]
    for k in range(4 if keep == "noload" else 3):
        L += load_coeffs(ENT + k, CO[k])
    timing = keep
    if static_pb:
        L += addresses_static(ENT) + ["s_waitcnt lgkmcnt(0)", f"v_add_u32 v{VA[0]}, s{AD}, %[lane]"] + dispatch_reads(0)
    else:
        L += addresses(ENT) + ["s_waitcnt lgkmcnt(0)"] + adds() + reads(range(0, 4), 0) + reads(range(4, 8), 0)
    for pp in range(4):
        L += [f"s_mov_b32 s{S_LEFT_}, %[n4]", f".LF{pp}_%=:"]
        for k in range(4):
            L += item(pp, k)
        L += [f"s_sub_u32 s{S_LEFT_}, s{S_LEFT_}, 1", f"s_cmp_lg_u32 s{S_LEFT_}, 0", f"s_cbranch_scc1 .LF{pp}_%="]
    L += ["s_waitcnt vmcnt(0) lgkmcnt(0)"]
    if prio:
        L += ["s_setprio 0"]
    body = "\n".join(f'        "{l}\\n\\t"' for l in L)
    vregs = list(range(SINK, tmp + n_tmp))
    sregs = [S_OFF_, S_LEFT_, S_PTR, S_PTR + 1] + list(range(ENT, S_K + 1)) + ([S_RANK, S_PRIO] if prio else [])
    clobbers = ", ".join([f'"v{r}"' for r in vregs] + [f'"s{r}"' for r in sregs] + ['"scc"'])
    acc_params = ", ".join(f"f8 &A{pp}" for pp in range(4))
    acc_ops = ", ".join(f'"+{{v[{acc + 8 * pp}:{acc + 8 * pp + 7}]}}"(A{pp})' for pp in range(4))
    rows = ", ".join(f'[row{pp}] "s"(row{pp})' for pp in range(4))
    return f"""// Four pixels of the staged chunk, 8-tap variant, four-plane frame-pair layout: see block_fir8 in
// tools/gen_trip_asm.py.  row0..row3 = the pixels' 4-byte entries from the chunk's first mic; sweeps 4 * n4 entries
// of each (n4 >= 1) and reads four entries past the last swept one of row3; coef = the [102][8] coefficient table
// (row 101 zeros), pb = bytes of one sample plane of a staged row; pfn > 0: also touches the 256 bytes from byte pfoff
// of every row (the next chunk's entries: an L2 prefetch).
// Accumulators (outputs 4l..4l+3, two frames each) pinned at v[{acc}:{acc + 31}], temps v{vregs[0]}..v{vregs[-1]}.
__device__ __forceinline__ void {name}({acc_params}, const void *row0, const void *row1, const void *row2,
                                       const void *row3, int n4, unsigned lane_addr, const void *coef, unsigned pb,
                                       unsigned pfoff, int pfn, int rank = 0) {{
    asm volatile(
{body}
        : {acc_ops}
        : {rows}, [n4] "s"(n4), [lane] "v"(lane_addr), [coef] "s"(coef), [pb] "s"(pb), [pfoff] "v"(pfoff), [pfn] "s"(pfn), [rank] "s"(rank)
        : {clobbers});
}}
"""


FIR_SH_TMP = 80  # block_fir8_shared: sample sets A / B at v80..v101 / v102..v123, address v124, prefetch sink v76


def block_fir8_shared(name, acc=FIR_ACC, tmp=FIR_SH_TMP, pb=None, prio=None):
    """The FIR8 block for four VERTICALLY ADJACENT pixels swept mic by mic (late round 3): within a mic the four items
    go pixel 0, 1, 2, 3, and an item whose table entry carries the same LDS address and plane as the item before it --
    the same integer delay; only the coefficient row differs -- takes the eleven samples that are in registers already: no
    address add, no dispatch on the plane, no LDS read.  Vertical neighbours share their integer delay for 63 % (c3) to
    83 % (headline) of the mics, so an item needs 3.6 .. 5.8 LDS reads where block_fir8 issues 11.

    Two sample register sets A and B (22 registers each).  An item's FMAs read the set `sigma` its samples are in; a
    following item that does not share them has its eleven reads issued into the OTHER set beside these FMAs and flips
    sigma.  Which set an item reads is known only at run time, so every item position has its FMAs twice (from A, from B)
    behind one scalar branch on sigma.  Everything else is block_fir8(static_pb): one table dword per (pixel, mic) (entries of
    the four pixels' rows at one running offset), one v_add_u32 and eleven immediate-offset reads per NEW set of samples in
    one of four copies by the plane r, coefficients by a dependent scalar load three items ahead, one lgkmcnt(0) per item.
    Per pixel the items run in mic order and the taps in the reference's order: the sums are block_fir8's, bit for bit."""
    pb = FIR_STATIC_PB if pb is None else pb
    prio = FIR_PRIO if prio is None else prio
    ENT = 36
    CO = (40, 48, 56, 64)
    AD, S_KEYN, S_KEY, S_SH, S_SIG = 72, 73, 74, 75, 76
    S_T, S_R, S_K = 80, 81, 82
    S_OFF_, S_LEFT_ = 23, 24
    XS = (tmp, tmp + 22)
    VA = tmp + 44
    SINK = tmp - 4

    def uid():
        COUNTER[0] += 1
        return f"%=_{COUNTER[0]}"

    def accp(pp, o):
        b = acc + 8 * pp + 2 * o
        return f"v[{b}:{b + 1}]"

    def xp(w, st):
        r = XS[st] + 2 * w
        return f"v[{r}:{r + 1}]"

    def fmas(pp, ws, st):
        L = []
        for w in ws:
            for o in range(max(0, w - 7), min(3, w) + 1):
                t = w - o
                sp = CO[pp] + (t & ~1)
                sel = t & 1
                L.append(f"v_pk_fma_f32 {accp(pp, o)}, s[{sp}:{sp + 1}], {xp(w, st)}, {accp(pp, o)} op_sel:[{sel},0,0] op_sel_hi:[{sel},1,1]")
        return L

    def reads(st, r):
        L = []
        for w in range(11):
            off = ((r + w) % 4) * pb + 8 * ((r + w) // 4)
            L.append(f"ds_read_b64 {xp(w, st)}, v{VA}" + (f" offset:{off}" if off else ""))
        return L

    def issue(st):
        """the eleven samples of the item whose plane-0 address sits in s{AD} and whose plane in s{S_R}, into set st"""
        u = uid()
        return ([f"v_add_u32 v{VA}, s{AD}, %[lane]",
                 f"s_cmp_lt_u32 s{S_R}, 2", f"s_cbranch_scc1 .LSr01{u}", f"s_cmp_eq_u32 s{S_R}, 2", f"s_cbranch_scc1 .LSr2{u}"] +
                reads(st, 3) + [f"s_branch .LSrj{u}", f".LSr2{u}:"] + reads(st, 2) + [f"s_branch .LSrj{u}", f".LSr01{u}:",
                f"s_cmp_eq_u32 s{S_R}, 0", f"s_cbranch_scc1 .LSr0{u}"] + reads(st, 1) + [f"s_branch .LSrj{u}", f".LSr0{u}:"] +
                reads(st, 0) + [f".LSrj{u}:"])

    def key_of(ent):
        """key (address + plane: bits 0..19), plane-0 address and plane of the item whose entry sits in s{ent}"""
        return [f"s_and_b32 s{S_KEYN}, s{ent}, 0xfffff",
                f"s_and_b32 s{AD}, s{S_KEYN}, 0x3ffff", f"s_lshr_b32 s{S_R}, s{S_KEYN}, 18",
                f"s_mul_i32 s{S_T}, s{S_R}, {pb}", f"s_sub_u32 s{AD}, s{AD}, s{S_T}"]

    def load_coeffs(ent, cset):
        return [f"s_bfe_u32 s{S_K}, s{ent}, 0x70014", f"s_lshl_b32 s{S_K}, s{S_K}, 5",
                f"s_load_dwordx8 s[{cset}:{cset + 7}], %[coef], s{S_K}"]

    def body(pp, st):
        u = uid()
        return (fmas(pp, range(0, 4), st) +
                [f"s_cmp_eq_u32 s{S_SH}, 1", f"s_cbranch_scc1 .LSkeep{u}"] + issue(st ^ 1) + [f"s_xor_b32 s{S_SIG}, s{S_SIG}, 1", f".LSkeep{u}:"] +
                fmas(pp, range(4, 11), st))

    def item(pp):
        u = uid()
        L = ["s_waitcnt lgkmcnt(0)",
             f"s_load_dword s{ENT + pp}, %[row{pp}], s{S_OFF_}"]        # entry of item i + 4 = (next mic, this pixel)
        L += load_coeffs(ENT + (pp + 3) % 4, CO[(pp + 3) % 4])         # coefficients of item i + 3
        if prio == 7 and pp == 0:
            up = uid()
            L += ([f"s_bitcmp1_b32 s{S_LEFT_}, 0", f"s_cbranch_scc1 .LSpy{up}"] + select_prio(S_PRIO, 1) +
                  [f"s_branch .LSpz{up}", f".LSpy{up}:"] + select_prio(S_RANK, 0) + [f".LSpz{up}:"])
        elif prio in (1, 2) and pp == 0:
            L += select_prio(S_PRIO, 1)
        L += key_of(ENT + (pp + 1) % 4)                                  # item i + 1: does it share this item's samples?
        L += [f"s_cmp_eq_u32 s{S_KEYN}, s{S_KEY}", f"s_cselect_b32 s{S_SH}, 1, 0", f"s_mov_b32 s{S_KEY}, s{S_KEYN}",
              f"s_cmp_eq_u32 s{S_SIG}, 0", f"s_cbranch_scc0 .LSb{u}"]
        L += body(pp, 0) + [f"s_branch .LSj{u}", f".LSb{u}:"] + body(pp, 1) + [f".LSj{u}:"]
        return L

    L = [f"s_mov_b32 s{S_OFF_}, 0", f"s_mov_b32 s{S_SIG}, 0"]
    if prio:
        L += [f"s_mov_b32 s{S_PRIO}, %[rank]", f"s_mov_b32 s{S_RANK}, %[rank]"]
    # warm the L2 with the NEXT chunk's entries of these four pixels (as block_fir8 does)
    L += ["s_cmp_lt_u32 %[pfn], 1", f"s_cbranch_scc1 .LSpf_%="]
    L += [f"global_load_dword v{SINK}, %[pfoff], %[row{pp}]" for pp in range(4)]
    L += [".LSpf_%=:"]
    L += [f"s_load_dword s{ENT + pp}, %[row{pp}], 0x0" for pp in range(4)] + [f"s_mov_b32 s{S_OFF_}, 4", "s_waitcnt lgkmcnt(0)"]
    for k in range(3):
        L += load_coeffs(ENT + k, CO[k])
    L += key_of(ENT) + [f"s_mov_b32 s{S_KEY}, s{S_KEYN}", "s_waitcnt lgkmcnt(0)"] + issue(0)
    L += [f"s_mov_b32 s{S_LEFT_}, %[n]", ".LSloop_%=:"]
    for pp in range(4):
        L += item(pp)
    L += [f"s_add_u32 s{S_OFF_}, s{S_OFF_}, 4", f"s_sub_u32 s{S_LEFT_}, s{S_LEFT_}, 1", f"s_cmp_lg_u32 s{S_LEFT_}, 0", "s_cbranch_scc1 .LSloop_%="]
    L += ["s_waitcnt vmcnt(0) lgkmcnt(0)"]
    if prio:
        L += ["s_setprio 0"]
    body_txt = "\n".join(f'        "{l}\\n\\t"' for l in L)
    vregs = list(range(SINK, VA + 1))
    sregs = [S_OFF_, S_LEFT_] + list(range(ENT, S_K + 1)) + ([S_RANK, S_PRIO] if prio else [])
    clobbers = ", ".join([f'"v{r}"' for r in vregs] + [f'"s{r}"' for r in sregs] + ['"scc"'])
    acc_params = ", ".join(f"f8 &A{pp}" for pp in range(4))
    acc_ops = ", ".join(f'"+{{v[{acc + 8 * pp}:{acc + 8 * pp + 7}]}}"(A{pp})' for pp in range(4))
    rows = ", ".join(f'[row{pp}] "s"(row{pp})' for pp in range(4))
    return f"""// Four vertically adjacent pixels of the staged chunk, 8-tap variant, swept mic by mic with the samples shared between
// pixels whose entries carry the same address and plane: see block_fir8_shared in tools/gen_trip_asm.py.  row0..row3 = the
// pixels' 4-byte entries from the chunk's first mic; sweeps n mics (n >= 1) and reads one entry past the last of every row;
// rows staged at a plane pitch of {pb} bytes.  Accumulators pinned at v[{acc}:{acc + 31}], temps v{vregs[0]}..v{vregs[-1]}.
__device__ __forceinline__ void {name}({acc_params}, const void *row0, const void *row1, const void *row2,
                                       const void *row3, int n, unsigned lane_addr, const void *coef, unsigned pfoff, int pfn, int rank) {{
    asm volatile(
{body_txt}
        : {acc_ops}
        : {rows}, [n] "s"(n), [lane] "v"(lane_addr), [coef] "s"(coef), [pfoff] "v"(pfoff), [pfn] "s"(pfn), [rank] "s"(rank)
        : {clobbers});
}}
"""


def main():
    hi = 128 - (4 * (DEPTH + 1) + 1) - 3
    out = ["// GENERATED by tools/gen_trip_asm.py -- do not edit.  See that script for the schedule.", "",
           f"constexpr int kQuadDmaWaves = {DMA_WAVES};  // waves 0..n-1 of a workgroup issue the quad block's refill pieces", ""]
    out.append(block("sweep_pixel_hi", 1, hi))
    out.append(block("sweep_quad_hi", 4, hi))
    out.append(block("sweep_quad_stamped", 4, hi, stamp=True))
    pd = int(os.environ.get("PAIR_DEPTH", "2"))  # frame-pair items: 2 items (8 reads) of read-ahead; +4 being issued +2 scalar loads <= 15 (lgkmcnt is 4 bits)
    out.append(block("sweep_duo_pairs", 2, 128 - (8 * (pd + 1) + 1) - 3, pair_depth=pd))
    out.append(block("sweep_duo_pairs_stamped", 2, 128 - (8 * (pd + 1) + 1) - 3, stamp=True, pair_depth=pd))
    out.append(block_shared("sweep_duo_shared", 128 - 25 - 3))
    out.append(block_shared("sweep_duo_shared_stamped", 128 - 25 - 3, stamp=True))
    out.append(block_exact_shared("sweep_duo_exact", 128 - 57 - 3))  # das_exact_pair_kernel (AWPU_MATH_F32_EXACT)
    out.append(block_exact_quad("sweep_quad_exact"))  # das_exact_quad_kernel (AWPU_MATH_F32_EXACT, row length known; round 4)
    out.append(block_exact_nd("sweep_exact_nd_item1", 1))  # das_exact_nd_kernel<1>: the {next, d} layout, one quad per wave
    out.append(block_exact_nd("sweep_exact_nd_item2", 2))  # das_exact_nd_kernel<2>: two quads per wave (the default batch kernel of the reference order)
    out.append(block_exact_nd("sweep_exact_ndh_item1", 1, nk=2))  # das_exact_ndh_kernel<1, *>: single frames, the halves form of the layout
    out.append(block_exact_nd("sweep_exact_ndh_item2", 2, nk=2))  # das_exact_ndh_kernel<2, *>
    out.append(block_exact_nd("sweep_exact_ndh_resident1", 1, nk=2, refill=False))  # das_exact_ndh_kernel<1, true>: every mic resident, nothing to refill
    out.append(block_exact_nd("sweep_exact_ndh_resident2", 2, nk=2, refill=False))  # das_exact_ndh_kernel<2, true>
    out.append(block_exact_solo("sweep_exact_ndp_item"))  # das_exact_ndp_kernel: one pixel per wave (grids of at most 16 pixels per CU)
    out += [f"constexpr bool kQuadChain = {'true' if CHAIN else 'false'};  // the quad blocks keep V3 = S3 - S2 (else S3 - S1)", ""]
    out.append(block_quad("sweep_quad_sum", chain=CHAIN))
    out.append(block_quad("sweep_quad_item", dma=True, chain=CHAIN, item=True))  # the production batch kernel: one block per item
    out.append(block_quad("sweep_quad_sum_stamped", stamp=True, chain=CHAIN))
    # single frames on the halves layout (das_quadh_kernel): the first quad of a wave through the block that also issues the
    # refill (sweep_quad1_sum_a_dma, below), the second quad through this one
    out.append(block_quad_ar("sweep_quad1_sum_a", nk=2, acc=QUAD1_ACC[0], tmp=QUAD1_TMP))  # das_quadh_stationary_kernel: no refill
    out.append(block_quad_ar("sweep_quad1_sum_b", nk=2, acc=QUAD1_ACC[1], tmp=QUAD1_TMP))
    out.append(block_quad_ar("sweep_quad1_sum_a_stamped", stamp=True, nk=2, acc=QUAD1_ACC[0], tmp=QUAD1_TMP))
    out.append(block_quad_ar("sweep_quad1_sum_a_dma", nk=2, acc=QUAD1_ACC[0], tmp=QUAD1_TMP, dma=True))  # das_quadh_kernel
    if os.environ.get("QUAD1_EARLY_X"):  # tuning builds: conditional reads at the head of the stage (measured: 79 vs 71 us)
        for q, base in enumerate(QUAD1_ACC):
            out.append(block_quad(f"sweep_quad1_early_{'ab'[q]}", nk=2, acc=base, tmp=QUAD1_TMP, early_x=True))
    if os.environ.get("QUAD_VARIANTS"):  # tuning builds: the priority schemes side by side (AWPU_QUAD_VARIANT picks)
        for v in (0, 3, 4):
            out.append(block_quad(f"sweep_quad_sum_v{v}", prio=v, chain=CHAIN))
    out.append(block_fir8("sweep_fir8_planes"))
    out += [f"constexpr unsigned kFirStaticPlaneBytes = {FIR_STATIC_PB};  // sweep_fir8_planes_static: the plane pitch it is generated for", ""]
    out.append(block_fir8("sweep_fir8_planes_static", static_pb=FIR_STATIC_PB))
    out.append(block_fir8_shared("sweep_fir8_planes_shared"))
    if os.environ.get("QUAD_VARIANTS"):  # tuning builds: what the block costs without its scalar loads / its LDS reads / the read-ahead
        out.append(block_fir8("sweep_fir8_planes_v1", timing="noload", prio=0))
        out.append(block_fir8("sweep_fir8_planes_v2", timing="nolds", prio=0))
    lo = 80 - (4 * (DEPTH + 1) + 1) - 3  # shapes with an 80-VGPR budget (6 waves per SIMD)
    out.append(block("sweep_quad_lo", 4, lo))
    out.append(block("sweep_quad_lo_stamped", 4, lo, stamp=True))
    path = Path(os.environ.get("TRIP_INC_OUT") or Path(__file__).resolve().parent.parent / "beamforming-lk_amd" / "csrc" / "das_fast_trip.inc")
    tmp = path.with_suffix(f".inc.tmp{os.getpid()}")
    tmp.write_text("\n".join(out))
    os.replace(tmp, path)  # (several ranks may build at once: never a half-written include)
    print("wrote", path, sum(1 for _ in path.read_text().splitlines()), "lines")


if __name__ == "__main__":
    main()
