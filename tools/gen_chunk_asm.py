#!/usr/bin/env python3
"""Writes gpirt_amd/csrc/chunk_asm.h: the hand-scheduled halves of a 64 x 64 x 64 chunk product of the panel kernel.

Why generated assembly: (i) the compiler places every LDS operand read directly in front of the MFMA that consumes it
(s_waitcnt lgkmcnt(0) behind each ds_read, whatever the source order or sched_group_barrier says); (ii) the next chunk's
operands come in by LDS-DMA (global_load_lds_dwordx4: no VGPR destination, no ds_write pass, half the vector-memory
instructions of 8-byte loads), whose instructions have to sit BETWEEN the MFMAs to be free -- issued in front of them a
chunk's 32 loads cost 0.8 us (tools/micro/panel_bench, -DPANEL_CHUNK_PROF).

    python tools/gen_chunk_asm.py          (re-run after changing the LDS images or the schedule)

The blocks that carry LDS-DMA rewrite M0 (s_add_u32 m0, ...: that also writes SCC).  SCC is declared clobbered -- LLVM
assumes any register an asm does not name survives it, so a scalar compare kept live across the block would otherwise be
miscompiled silently (round-3 advisor finding).  M0 is a RESERVED register for this compiler (naming it in the clobber
list is diagnosed as possible undefined behaviour): it is saved and restored inside the block instead, so it does survive.
tests/test_host.py checks that the committed header is this script's output byte for byte.
"""
import os

PITCH = 136        # doubles per LDS-DMA piece slot (128 + 8 of padding): see panel.hip, "chunk operands in LDS"
GROUPS = 8         # groups of four MFMAs per half: two 16-column operand blocks I = I0, I0 + 1


def off(q, J):
    """byte offset of the A operand of group q (I = q >> 2, s = q & 3), accumulator J, from the lane's base address:
    element (row 16 J + pi, column 16 I + 4 g + s) of the pair image  (column >> 1) * PITCH + (column & 1) * 64 + row"""
    I, s = q >> 2, q & 3
    return ((8 * I + (s >> 1)) * PITCH + (s & 1) * 64 + 16 * J) * 8


def half(dma):
    """dma: None | 'sc1' | 'plain' -- eight LDS-DMA pieces of the NEXT chunk, one behind the first MFMA of each group"""
    L = ["s_nop 4"]
    if dma:
        L.append("s_mov_b32 %[keep], m0")
    rd = lambda q, J: f"ds_read_b64 %[t{(q % 3) * 4 + J}], %[addr] offset:{off(q, J)}"
    for q in (0, 1):
        for J in range(4):
            L.append(rd(q, J))
    for q in range(GROUPS):
        L.append("s_waitcnt lgkmcnt(%d)" % (4 if q + 1 < GROUPS else 0))
        for J in range(4):
            L.append(f"v_mfma_f64_16x16x4_f64 %[T{J}], %[t{(q % 3) * 4 + J}], %[x{q}], %[T{J}]")
            if q + 2 < GROUPS:
                L.append(rd(q + 2, J))
            if dma and J == 0:
                L.append(f"s_add_u32 m0, %[lb], {q * PITCH * 8 * DMA_SLOT_STEP}")
            if dma and J == 1:
                L.append("global_load_lds_dwordx4 %[va], off" + (" sc1" if dma == "sc1" else ""))
                if q + 1 < GROUPS:
                    L.append("v_lshl_add_u64 %[va], %[st], 0, %[va]")
    if dma:
        L.append("s_mov_b32 m0, %[keep]")
    # DGEMM result -> VALU / LDS-store read of the accumulators: 18 wait states
    L += ["s_nop 15", "s_nop 3"]
    return L


def dma_only(sc1):
    L = ["s_nop 4", "s_mov_b32 %[keep], m0"]
    for q in range(GROUPS):
        L.append(f"s_add_u32 m0, %[lb], {q * PITCH * 8 * DMA_SLOT_STEP}")
        L.append("s_nop 0")
        L.append("global_load_lds_dwordx4 %[va], off" + (" sc1" if sc1 else ""))
        if q + 1 < GROUPS:
            L.append("v_lshl_add_u64 %[va], %[st], 0, %[va]")
    L.append("s_mov_b32 m0, %[keep]")
    return L


def emit(lines):
    return "\n".join('        "%s\\n\\t"' % l for l in lines)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "gpirt_amd", "csrc", "chunk_asm.h")
    t_out = ", ".join(f'[T{J}] "+v"(T[{J}])' for J in range(4)) + ",\n          " + \
        ", ".join(f'[t{i}] "=&v"(t{i})' for i in range(12))
    x_in = ", ".join(f'[x{q}] "v"(x[{q}])' for q in range(GROUPS)) + ', [addr] "v"(lds_addr)'
    tdecl = "double " + ", ".join(f"t{i}" for i in range(12)) + ";"
    parts = []
    for step, name in ((4, "B"), (1, "X")):
        global DMA_SLOT_STEP
        DMA_SLOT_STEP = step
        sc1 = name == "B"
        parts.append(f'''
// half a chunk + the eight LDS-DMA pieces of the next chunk's {"B operand (another work-group's block: sc1)" if sc1 else "own-row operand (plain)"};
// piece u goes to LDS byte address lb + u * {step * PITCH * 8} from the lane's global address va + u * st
__device__ __forceinline__ void chunk_half_dma{name}(d4 (&T)[4], const double (&x)[{GROUPS}], uint32_t lds_addr,
                                                 const double* va, uint64_t st, uint32_t lb)
{{
    {tdecl}
    uint32_t keep;
    asm volatile(
{emit(half("sc1" if sc1 else "plain"))}
        : {t_out},
          [va] "+v"(va), [keep] "=&s"(keep)
        : {x_in}, [st] "s"(st), [lb] "s"(lb)
        : "memory", "scc");
}}

// the eight pieces alone (a step's first chunk)
__device__ __forceinline__ void chunk_dma{name}(const double* va, uint64_t st, uint32_t lb)
{{
    uint32_t keep;
    asm volatile(
{emit(dma_only(sc1))}
        : [va] "+v"(va), [keep] "=&s"(keep)
        : [st] "s"(st), [lb] "s"(lb)
        : "memory", "scc");
}}
''')
    text = f'''// chunk_asm.h -- GENERATED by tools/gen_chunk_asm.py, do not edit.
// Halves of a 64 x 64 x 64 chunk product of the panel kernel (panel.hip), hand-scheduled:
//     T[J] += sum_(q = 0..7) A_q,J * x[q]        (fp64 MFMA 16x16x4, J = 0..3)
// A_q,J: the lane's LDS operand, element (row 16 J + pi, column 16 I + 4 g + s) (I = q >> 2, s = q & 3) of the 64 x 64
// block in the PAIR IMAGE an LDS-DMA fill leaves: column c at (c >> 1) * {PITCH} + (c & 1) * 64 doubles (lds_addr = byte address
// of the lane's element (pi, 4 g)); x[q]: the lane's register operand, negated by the caller.  32 MFMAs; 32 ds_read_b64
// issued two groups ahead into a ring of twelve registers; per accumulator the terms come in the order q = 0, 1, ... --
// the order of strip64_update -- so the result is bit-identical to the compiler-scheduled loop it replaces.
#pragma once
#include "common.h"

namespace gpirt {{

constexpr int CHUNK_PITCH = {PITCH};           // doubles per LDS-DMA piece slot

__device__ __forceinline__ void chunk_half_plain(d4 (&T)[4], const double (&x)[{GROUPS}], uint32_t lds_addr)
{{
    {tdecl}
    asm volatile(
{emit(half(None))}
        : {t_out}
        : {x_in}
        : "memory");
}}
{"".join(parts)}
}}  // namespace gpirt
'''
    return out, text


def write():
    out, text = main()
    with open(out, "w") as f:
        f.write(text)
    print("wrote", os.path.normpath(out))


DMA_SLOT_STEP = 1
if __name__ == "__main__":
    write()
