// Shared fused epilogues of the bf16 GEMM kernels (v1 128x128, v2 / v3 256x256).
//
// Accumulator layouts.  ACC = f32x4 (MFMA 16x16x32: RBLK = 16, QN = 1) or f32x16 (MFMA 32x32x16: RBLK = 32, QN = 4);
// `fr` = lane & (RBLK-1), `hg` = lane / RBLK (0..3 resp. 0..1); register r = 4q + e:
//   row-major epilogues ("swapped" operands):  C[m = mb + RBLK i + fr][n = nb + RBLK j + 8q + 4hg + e]
//   V^T epilogue (plain operand order):        C[m = mb + RBLK i + 8q + 4hg + e][n = nb + RBLK j + fr]
// so every (i, j, q) is four consecutive n (resp. m): one 8-byte bf16x4 store.
//
// Structure matters more than arithmetic here:
//   * every optional vector (bias, LayerScale) is loaded ONCE under a single hoisted uniform branch, the activation
//     kind and "does this wave tile touch the M edge" are decided ONCE (uniform), and residual / position rows are loaded
//     with clamped (always valid) addresses in row groups before the group's first store - hipcc otherwise emits branch
//     + s_waitcnt vmcnt(0) per element, i.e. dozens of serialized L2 round trips per tile;
//   * output stores are hand-written and the hoisted vector loads are retired with a compiler-VISIBLE s_waitcnt: the
//     persistent kernels re-enter their K loop after an epilogue, and any VMEM operation hipcc still considers pending
//     there makes it put an s_waitcnt vmcnt(0) at the loop header - in front of every K-tile - which would drain the
//     LDS-DMA prefetch ring each iteration (verified in the ISA).  Stores the compiler cannot see leave nothing
//     pending; the hardware still retires them in order with the counted waits of the main loop.
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"
#include "visrep_internal.h"

#ifdef VISREP_EPI_NO_STORE   // timing-only (tools/): how much of a GEMM is its output store stream - results are not written
VR_DEV void store_b64(void* ptr, u32x2 v) { asm volatile("" ::"v"(ptr), "v"(v)); }
#else
VR_DEV void store_b64(void* ptr, u32x2 v) { asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(ptr), "v"(v) : "memory"); }
#endif
#ifdef VISREP_EPI_NO_STORE
VR_DEV void store_b128(void* ptr, f32x4 v) { asm volatile("" ::"v"(ptr), "v"(v)); }
#else
VR_DEV void store_b128(void* ptr, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory"); }
#endif
// The same stores with a compile-time BYTE offset in the instruction's 13-bit immediate: a row's stores share one 64-bit base address
// computed once per row instead of one v_mad_i64 + v_lshl_add_u64 chain per store (round 4: the address arithmetic was 20-35 % of the VALU
// work of the K = 1024 epilogues, profiles/round4_gemm.md).
#ifdef VISREP_EPI_NO_STORE
template <int OFF> VR_DEV void store_b64_at(const void* base, u32x2 v) { asm volatile("" ::"v"(base), "v"(v)); }
template <int OFF> VR_DEV void store_b128_at(const void* base, f32x4 v) { asm volatile("" ::"v"(base), "v"(v)); }
#else
template <int OFF> VR_DEV void store_b64_at(const void* base, u32x2 v) {
    static_assert(OFF >= -4096 && OFF < 4096, "global_store immediate offset range");
    asm volatile("global_store_dwordx2 %0, %1, off offset:%2" ::"v"(base), "v"(v), "n"(OFF) : "memory");
}
template <int OFF> VR_DEV void store_b128_at(const void* base, f32x4 v) {
    static_assert(OFF >= -4096 && OFF < 4096, "global_store immediate offset range");
    asm volatile("global_store_dwordx4 %0, %1, off offset:%2\n\ts_nop 1" ::"v"(base), "v"(v), "n"(OFF) : "memory");
}
#endif
// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}) - the index is usable as a template argument.
// The lambdas passed here carry __attribute__((always_inline)): a loop body the inliner leaves out of line takes the accumulators by
// reference, i.e. through scratch memory (seen: 720 bytes of scratch per lane in the EPI_ACT kernels)
template <typename F, int... Is> VR_DEV void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F> VR_DEV void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
VR_DEV void drain_visible_loads() { __builtin_amdgcn_s_waitcnt(0x0f70); }   // vmcnt(0), expcnt / lgkmcnt untouched
// An UNCONDITIONAL consumer of a compiler-visible load's result.  hipcc waits for a load where its result is first read; when every reader
// sits in a conditional region (rows past M: `if (ok) store`, with the arithmetic sunk into the region), the path around the region leaves
// the load pending as far as the wait-count pass knows, the persistent kernel's K-loop header inherits "a VMEM op may be pending" and gets
// an s_waitcnt vmcnt(0) in front of every K-tile - which drains the LDS-DMA ring each iteration.  Found in the ISA of the EPI_F32X kernels
// (all along) and of the round-4 EPI_RESID epilogue: -3 % on fc2 / out-proj until the residual registers were retired this way.
VR_DEV void retire_load(u32x2 v) { asm volatile("" ::"v"(v)); }
VR_DEV void retire_load(u32x4 v) { asm volatile("" ::"v"(v)); }
VR_DEV void retire_load(const float4& v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }

// sum of v over the lanes that differ from this one in lane bits 4 (RBLK == 16 only) and 5: the lanes that hold the other column
// groups of the same accumulator row.  gfx950 lane-swap VALU ops: no LDS traffic, nothing for the waitcnt pass to see.
template <int RBLK> VR_DEV float sum_over_hg(float v) {
    if (RBLK == 16) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

template <typename ACC> struct AccGeom { static constexpr int QN = sizeof(ACC) / 16, RBLK = QN == 1 ? 16 : 32; };

// HAS_ST (EPI_RESID): also emit, per output row and per wave tile (NJ * RBLK columns = one "slot"), the sum and the sum of squares
// of the bf16-ROUNDED outputs -> p.stat_partial[m * p.stat_slots + slot]; ln_stats_finalize turns the slots of a row into the
// (rstd, -mean * rstd) the next LayerNorm-folded GEMM wants, so that LayerNorm never reads the residual stream again.
// AL (decided once per launch by the caller: ldc % 8 == 0, 16-byte aligned C and residual): 16-byte stores and residual loads - see WIDE / RWIDE.
// Addresses: one 64-bit base per output row (this lane's first column), every store / residual load of the row at a COMPILE-TIME byte
// offset from it (the store instructions' immediate field; the column loops are static_for so that the offsets are template arguments).
// sum of v over the 16 lanes of a DPP row (lanes that differ in bits 0-3 = the 16 accumulator rows fr of one column quad): four rotate-adds
VR_DEV float sum_over_fr(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}

// (t1, t2) = this lane's sum / sum of squares over its rows of column quad c -> summed over the 16 row lanes and over the lanes a group spans,
// written to the wave's slot of GemmArgs::gn_partial by the lane that owns the group's first column
template <int NI_RBLK, int RBLK>
VR_DEV void gn_partial_store(const GemmArgs& p, float2* dst, int G, float t1, float t2, int c, int nb, int fr, int hg) {
    float s1 = sum_over_fr(t1), s2 = sum_over_fr(t2);
    if (p.gn_cpg >= 8) {                          // uniform: a group spans the column quads of the lanes hg, hg ^ 1 (and hg ^ 2 for 16)
        const auto a1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s1), false, false);
        const auto a2 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s2), __builtin_bit_cast(unsigned, s2), false, false);
        s1 = __builtin_bit_cast(float, (unsigned)a1[0]) + __builtin_bit_cast(float, (unsigned)a1[1]);
        s2 = __builtin_bit_cast(float, (unsigned)a2[0]) + __builtin_bit_cast(float, (unsigned)a2[1]);
    }
    if (p.gn_cpg >= 16) {
        const auto a1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s1), false, false);
        const auto a2 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, s2), __builtin_bit_cast(unsigned, s2), false, false);
        s1 = __builtin_bit_cast(float, (unsigned)a1[0]) + __builtin_bit_cast(float, (unsigned)a1[1]);
        s2 = __builtin_bit_cast(float, (unsigned)a2[0]) + __builtin_bit_cast(float, (unsigned)a2[1]);
    }
    const int colq = nb + c * RBLK + hg * 4;       // first column of this lane's quad c
    if (fr == 0 && (colq % p.gn_cpg) == 0) {
        const int g = colq / p.gn_cpg;
        // (asm stores: a compiler-visible VMEM operation pending at the K loop's header would cost the 256x256 kernel a vmcnt(0) per tile)
        store_b64(dst + g, u32x2{__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s2)});
        if (NI_RBLK == 128) store_b64(dst + G + g, u32x2{0u, 0u});       // this wave covers two 64-row slots: all of it in the first
    }
}

// The 256x256 ping-pong kernel's GroupNorm partial sums (EPI_BIAS convolutions): a pass over the accumulators BEFORE the plain epilogue.  Running
// sums inside the epilogue (round 4, and again in round 5 with the output values written back over their accumulators) put 170-340 registers into
// scratch, K loop included - that kernel sits at 252-256 registers and any live range added across its store stream cascades.  Here nothing is
// live across the epilogue: per column quad two packed (v_pk_*) running pairs, the bias added on the fly, reduced and stored before the next quad.
// The sums are those of the fp32 (unrounded) outputs, as the 128x128 kernel's; the order of summation differs (tests compare with a tolerance).
// RES (round 6, EPI_RESID convolutions): the sums are those of acc + bias + residual, so the pre-pass fetches the residual tile too - eight 8-byte
// loads per column quad (this lane's four columns of its eight rows), consumed at once - and the epilogue fetches it again (from L2: 128 KB per
// output tile); what that buys is the separate statistics pass over the output tensor (one more HBM read of it and a launch).
template <int NI, int NJ, bool RES = false, typename ACC>
VR_DEV void gemm_gn_partials_prepass(const GemmArgs& p, const ACC (&acc)[NI][NJ], int mb, int nb, int fr, int hg) {
    constexpr int RBLK = AccGeom<ACC>::RBLK;
    static_assert(AccGeom<ACC>::QN == 1 && NI * RBLK == 128, "16x16 accumulators, 128 rows per wave");
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    if (mb >= p.M) return;                                        // uniform per wave; gn_hw % 128 == 0: all rows valid and inside one image
    float4 bq[NJ];
#pragma unroll
    for (int c = 0; c < NJ; ++c) bq[c] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nb + hg * 4 + c * RBLK) : float4{0.f, 0.f, 0.f, 0.f};
    drain_visible_loads();
    const int g0 = mb + p.a_row0;
    const int b = g0 / p.gn_hw, slot = (g0 - b * p.gn_hw) >> 6, nblk = p.gn_hw >> 6, G = p.N / p.gn_cpg;
    float2* dst = p.gn_partial + ((size_t)b * nblk + slot) * G;
#pragma unroll
    for (int c = 0; c < NJ; ++c) {
        const f32x2 b01 = {bq[c].x, bq[c].y}, b23 = {bq[c].z, bq[c].w};
        f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
        u32x2 rq[RES ? NI : 1];
        if (RES) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
                rq[RES ? i : 0] = *reinterpret_cast<const u32x2*>(p.resid + (size_t)(mb + i * RBLK + fr) * p.ldc + nb + c * RBLK + hg * 4);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            f32x2 v01 = f32x2{acc[i][c][0], acc[i][c][1]} + b01, v23 = f32x2{acc[i][c][2], acc[i][c][3]} + b23;
            if (RES) {
                const u32x2 r = rq[RES ? i : 0];
                v01 += f32x2{bf_lo(r[0]), bf_hi(r[0])}; v23 += f32x2{bf_lo(r[1]), bf_hi(r[1])};
            }
            s01 += v01; s23 += v23;
            q01 = __builtin_elementwise_fma(v01, v01, q01); q23 = __builtin_elementwise_fma(v23, v23, q23);
        }
        const f32x2 s = s01 + s23, q = q01 + q23;
        gn_partial_store<NI * RBLK, RBLK>(p, dst, G, s[0] + s[1], q[0] + q[1], c, nb, fr, hg);
    }
}

// HAS_GN (convolutions, EPI_BIAS / EPI_RESID): also emit the GroupNorm partial sums of this wave's rows (GemmArgs::gn_partial).
template <int EPI, int NI, int NJ, int ACT, bool EDGE, bool HAS_LS, bool HAS_LN, bool HAS_ST, bool AL, bool HAS_GN, typename ACC>
VR_DEV void gemm_epilogue_rowmajor_impl(const GemmArgs& p, const ACC (&acc)[NI][NJ], int mb, int nb, int fr, int hg) {
    constexpr int QN = AccGeom<ACC>::QN, RBLK = AccGeom<ACC>::RBLK, NC = NJ * QN;   // NC column groups of 4 per lane
    static_assert(!HAS_GN || (QN == 1 && (NI * RBLK == 64 || NI * RBLK == 128)), "GroupNorm partials: 16x16 accumulators, 64 or 128 rows per wave");
    float gs1[HAS_GN ? NC : 1], gs2[HAS_GN ? NC : 1];        // per column quad: sum / sum of squares over this lane's rows
#pragma unroll
    for (int c = 0; c < (HAS_GN ? NC : 1); ++c) { gs1[c] = 0.f; gs2[c] = 0.f; }
    // Output stores.  16x16 accumulators: the four lanes hg = 0..3 of a row hold 4 consecutive columns each of every 16-column block j, i.e.
    // 8-byte stores, 32 contiguous bytes per row and instruction.  The K = 1024 GEMMs spend 15-22 % of their time in this store stream
    // (profiles/round2_store_stream.md), which is issue-bound, not bandwidth-bound.  WIDE: one v_permlane16_swap per word between the lanes
    // hg and hg ^ 1 regroups a PAIR of column blocks (j, j + 1): even lanes end up with 8 consecutive columns of block j, odd lanes with 8 of
    // block j + 1 -> one 16-byte store per pair instead of two 8-byte ones (64 contiguous bytes per row and instruction), same bytes, same
    // addresses.  RWIDE: residual rows are fetched the same way - one 16-byte load per PAIR of column blocks from the address this lane will
    // store to, then the same swap (an involution) hands each lane its own 4 + 4 columns: half the load instructions.
    constexpr bool WIDE = AL && QN == 1 && (NC % 2) == 0 && EPI != EPI_F32;
    constexpr bool RWIDE = WIDE && EPI == EPI_RESID;
    constexpr int ES = EPI == EPI_F32 ? 4 : 2;               // bytes per output element
    constexpr int CSTEP = WIDE ? 2 : 1;
    float4 bv[NC], lv[(HAS_LS || HAS_LN) ? NC : 1];           // lv: LayerScale gamma, or the LN column sums s[n]
#pragma unroll
    for (int c = 0; c < NC; ++c) bv[c] = float4{0.f, 0.f, 0.f, 0.f};
    auto coff = [](int c) { return (c / QN) * RBLK + (c % QN) * 8; };   // column of group c relative to the lane's first column
    const int lane_col = nb + hg * 4;                        // this lane's first column (group 0)
    // WIDE: first column of this lane's 8-column chunk of the pair (c0, c0 + 1) is wide_col + 16 c0: even lanes block c0's columns 8 (hg / 2) ..,
    // odd lanes block c0 + 1's (col(c0 + 1) - 4 = lane_col + 16 c0 + 12)
    const int wide_col = lane_col + ((hg & 1) ? 12 : 0);
    auto col = [&](int c) { return lane_col + coff(c); };
    if (p.bias) {
#pragma unroll
        for (int c = 0; c < NC; ++c) bv[c] = *reinterpret_cast<const float4*>(p.bias + col(c));
    }
    if (HAS_LS) {
#pragma unroll
        for (int c = 0; c < NC; ++c) lv[c] = *reinterpret_cast<const float4*>(p.ls + col(c));
    }
    float2 rt_all[HAS_LN ? NI : 1];                          // LN: (rstd, -mean * rstd) of every row of this lane, loaded up front
    if (HAS_LN) {
#pragma unroll
        for (int c = 0; c < NC; ++c) lv[c] = *reinterpret_cast<const float4*>(p.ln_s + col(c));
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int m = mb + i * RBLK + fr;
            rt_all[i] = p.ln_rt[visrep_a_row(p, (EDGE && m >= p.M) ? p.M - 1 : m)];
        }
    }
    drain_visible_loads();
    // rows are processed in groups of RB: all residual / position loads of a group are issued before its first store
    // (in-place residual: loads and stores alias as far as the compiler knows, so source order is the only batching tool)
    constexpr int INFLIGHT = (HAS_LS || QN > 1) ? 8 : 16;   // residual loads in flight per lane (register budget)
    constexpr int RB0 = (INFLIGHT / NC) < 1 ? 1 : ((INFLIGHT / NC) > NI ? NI : (INFLIGHT / NC));
    constexpr int RB = EPI == EPI_PATCH ? 1 : RB0;          // position rows are float4: one row at a time
    // (static_for, not `#pragma unroll`: every index into acc / rv / rq is a compile-time constant whatever the unroller decides - an
    // accumulator array that is indexed dynamically lives in scratch memory)
    static_assert(NI % RB == 0, "row groups must tile the accumulator rows");
    static_for<NI / RB>([&](auto I0) __attribute__((always_inline)) {
        constexpr int i0 = decltype(I0)::value * RB;
        bool ok[RB];
        size_t orow[RB];
        char* crow[RB];                                       // byte address of (output row, this lane's first column [of its wide chunk])
        u32x2 rv[RB][NC];
        float4 pv[RB][NC];
        u32x4 rq[RB][RWIDE ? NC / 2 : 1];
        static_for<RB>([&](auto II) __attribute__((always_inline)) {
            constexpr int ii = decltype(II)::value;
            const int m = mb + (i0 + ii) * RBLK + fr;
            ok[ii] = EDGE ? (m < p.M) : true;               // interior tiles: no exec-masked region at all
            const int mc = ok[ii] ? m : p.M - 1;
            orow[ii] = (size_t)mc;
            const float* posrow = nullptr;
            if (EPI == EPI_PATCH) {   // m = b*P + pidx  ->  token row b*T + cls_off + pidx ; add pos[cls_off + pidx]
                const int b = mc / p.patches, pi = mc - b * p.patches;
                orow[ii] = (size_t)b * p.tokens + p.cls_off + pi;
                posrow = p.pos + (size_t)(p.cls_off + pi) * p.N + lane_col;
            }
            crow[ii] = reinterpret_cast<char*>(p.C) + (orow[ii] * p.ldc + (WIDE ? wide_col : lane_col)) * ES;
            if (EPI == EPI_RESID) {
                const bf16_t* rrow = p.resid + orow[ii] * p.ldc + (RWIDE ? wide_col : lane_col);
                if (RWIDE) {
#pragma unroll
                    for (int c0 = 0; c0 < NC; c0 += 2) rq[ii][c0 / 2] = *reinterpret_cast<const u32x4*>(rrow + 16 * c0);
                } else {
#pragma unroll
                    for (int c = 0; c < NC; ++c) rv[ii][c] = *reinterpret_cast<const u32x2*>(rrow + coff(c));
                }
            }
            if (EPI == EPI_PATCH) {
#pragma unroll
                for (int c = 0; c < NC; ++c) pv[ii][c] = *reinterpret_cast<const float4*>(posrow + coff(c));
            }
        });
        static_for<RB>([&](auto II) __attribute__((always_inline)) {       // every row group's loads are consumed on EVERY path (see retire_load)
            constexpr int ii = decltype(II)::value;
            if (EPI == EPI_RESID) {
                if (RWIDE) {
#pragma unroll
                    for (int k = 0; k < (RWIDE ? NC / 2 : 1); ++k) retire_load(rq[ii][k]);
                } else {
#pragma unroll
                    for (int c = 0; c < NC; ++c) retire_load(rv[ii][c]);
                }
            }
            if (EPI == EPI_PATCH) {
#pragma unroll
                for (int c = 0; c < NC; ++c) retire_load(pv[ii][c]);
            }
        });
        static_for<RB>([&](auto II) __attribute__((always_inline)) {
            constexpr int ii = decltype(II)::value, i = i0 + ii;
            float st1 = 0.f, st2 = 0.f;
            static_for<NC / CSTEP>([&](auto C0) __attribute__((always_inline)) {
                constexpr int c0 = decltype(C0)::value * CSTEP;
                u32x2 o2[CSTEP];
                if (RWIDE) {                                  // even lanes loaded (own j, partner's j), odd lanes (partner's j + 1, own j + 1)
                    const u32x4 q4 = rq[ii][RWIDE ? c0 / 2 : 0];
                    const auto s0 = __builtin_amdgcn_permlane16_swap(q4[0], q4[2], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(q4[1], q4[3], false, false);
                    rv[ii][c0] = u32x2{(unsigned)s0[0], (unsigned)s1[0]};
                    rv[ii][c0 + (RWIDE ? 1 : 0)] = u32x2{(unsigned)s0[1], (unsigned)s1[1]};
                }
                static_for<CSTEP>([&](auto CC) __attribute__((always_inline)) {
                    constexpr int cc = decltype(CC)::value, c = c0 + cc, j = c / QN, q = c % QN;
                    float v0, v1, v2, v3;
                    if (HAS_LN) {       // rstd * (acc - mean * s) + bias'
                        v0 = __builtin_fmaf(acc[i][j][4 * q + 0], rt_all[HAS_LN ? i : 0].x, __builtin_fmaf(rt_all[HAS_LN ? i : 0].y, lv[HAS_LN ? c : 0].x, bv[c].x));
                        v1 = __builtin_fmaf(acc[i][j][4 * q + 1], rt_all[HAS_LN ? i : 0].x, __builtin_fmaf(rt_all[HAS_LN ? i : 0].y, lv[HAS_LN ? c : 0].y, bv[c].y));
                        v2 = __builtin_fmaf(acc[i][j][4 * q + 2], rt_all[HAS_LN ? i : 0].x, __builtin_fmaf(rt_all[HAS_LN ? i : 0].y, lv[HAS_LN ? c : 0].z, bv[c].z));
                        v3 = __builtin_fmaf(acc[i][j][4 * q + 3], rt_all[HAS_LN ? i : 0].x, __builtin_fmaf(rt_all[HAS_LN ? i : 0].y, lv[HAS_LN ? c : 0].w, bv[c].w));
                    } else {
                        v0 = acc[i][j][4 * q + 0] + bv[c].x; v1 = acc[i][j][4 * q + 1] + bv[c].y;
                        v2 = acc[i][j][4 * q + 2] + bv[c].z; v3 = acc[i][j][4 * q + 3] + bv[c].w;
                    }
                    if (EPI == EPI_ACT) {
                        v0 = apply_act(v0, ACT); v1 = apply_act(v1, ACT); v2 = apply_act(v2, ACT); v3 = apply_act(v3, ACT);
                    }
                    if (EPI == EPI_RESID) {
                        if (HAS_LS) { v0 *= lv[HAS_LS ? c : 0].x; v1 *= lv[HAS_LS ? c : 0].y; v2 *= lv[HAS_LS ? c : 0].z; v3 *= lv[HAS_LS ? c : 0].w; }
                        v0 += bf_lo(rv[ii][c][0]); v1 += bf_hi(rv[ii][c][0]); v2 += bf_lo(rv[ii][c][1]); v3 += bf_hi(rv[ii][c][1]);
                    }
                    if (EPI == EPI_PATCH) { v0 += pv[ii][c].x; v1 += pv[ii][c].y; v2 += pv[ii][c].z; v3 += pv[ii][c].w; }
                    if (HAS_GN && ok[ii]) {
                        gs1[HAS_GN ? c : 0] += (v0 + v1) + (v2 + v3);
                        gs2[HAS_GN ? c : 0] = __builtin_fmaf(v0, v0, __builtin_fmaf(v1, v1, __builtin_fmaf(v2, v2, __builtin_fmaf(v3, v3, gs2[HAS_GN ? c : 0]))));
                    }
                    if (EPI == EPI_F32) {
                        if (ok[ii]) store_b128_at<((c / QN) * RBLK + (c % QN) * 8) * 4>(crow[ii], f32x4{v0, v1, v2, v3});
                    } else {
                        o2[cc] = u32x2{pack_bf16(v0, v1), pack_bf16(v2, v3)};
                        if (HAS_ST && ok[ii]) {
                            const float r0 = bf_lo(o2[cc][0]), r1 = bf_hi(o2[cc][0]), r2 = bf_lo(o2[cc][1]), r3 = bf_hi(o2[cc][1]);
                            st1 += (r0 + r1) + (r2 + r3);
                            st2 = __builtin_fmaf(r0, r0, __builtin_fmaf(r1, r1, __builtin_fmaf(r2, r2, __builtin_fmaf(r3, r3, st2))));
                        }
                        if (!WIDE && ok[ii]) store_b64_at<((c / QN) * RBLK + (c % QN) * 8) * 2>(crow[ii], o2[cc]);
                    }
                });
                if (WIDE) {
                    // swap(vdst = block j word, src = block j + 1 word): odd 16-lane rows of vdst <-> even rows of src.  Even lanes:
                    // (own j, partner's j); odd lanes: (partner's j + 1, own j + 1) - ascending columns in both cases.
                    const auto w0 = __builtin_amdgcn_permlane16_swap(o2[0][0], o2[CSTEP - 1][0], false, false);
                    const auto w1 = __builtin_amdgcn_permlane16_swap(o2[0][1], o2[CSTEP - 1][1], false, false);
                    if (ok[ii]) {
                        const u32x4 q4 = {(unsigned)w0[0], (unsigned)w1[0], (unsigned)w0[1], (unsigned)w1[1]};
                        store_b128_at<c0 * 16 * 2>(crow[ii], __builtin_bit_cast(f32x4, q4));
                    }
                }
            });
            if (HAS_ST) {                                     // all lanes take part in the lane swaps; one lane per row stores
                st1 = sum_over_hg<RBLK>(st1);
                st2 = sum_over_hg<RBLK>(st2);
                if (hg == 0 && ok[ii]) {
                    u32x2 o = {__builtin_bit_cast(unsigned, st1), __builtin_bit_cast(unsigned, st2)};
                    store_b64(p.stat_partial + orow[ii] * p.stat_slots + nb / (NJ * RBLK), o);
                }
            }
        });
    });
    if (HAS_GN) {
        const int g0 = mb + p.a_row0;                         // first row of this wave in the whole problem (a tail launch carries its offset)
        if (mb < p.M) {                                       // uniform per wave; gn_hw % 128 == 0: all rows valid and inside one image
            const int b = g0 / p.gn_hw, slot = (g0 - b * p.gn_hw) >> 6, nblk = p.gn_hw >> 6, G = p.N / p.gn_cpg;
            float2* dst = p.gn_partial + ((size_t)b * nblk + slot) * G;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                gn_partial_store<NI * RBLK, RBLK>(p, dst, G, gs1[HAS_GN ? c : 0], gs2[HAS_GN ? c : 0], c, nb, fr, hg);
            }
        }
    }
}

template <int EPI, int NI, int NJ, int ACT, bool STATS, bool RW, typename ACC>
VR_DEV void gemm_epilogue_rowmajor_rw(const GemmArgs& p, const ACC (&acc)[NI][NJ], int mb, int nb, int fr, int hg) {
    const bool interior = mb + NI * AccGeom<ACC>::RBLK <= p.M;
    if (STATS && EPI == EPI_RESID && p.stat_partial) {       // residual GEMM that also emits the next LayerNorm's row statistics
        if (p.ls) {
            if (interior) gemm_epilogue_rowmajor_impl<EPI, NI, NJ, ACT, false, true, false, true, RW, false>(p, acc, mb, nb, fr, hg);
            else gemm_epilogue_rowmajor_impl<EPI, NI, NJ, ACT, true, true, false, true, RW, false>(p, acc, mb, nb, fr, hg);
        } else {
            if (interior) gemm_epilogue_rowmajor_impl<EPI, NI, NJ, ACT, false, false, false, true, RW, false>(p, acc, mb, nb, fr, hg);
            else gemm_epilogue_rowmajor_impl<EPI, NI, NJ, ACT, true, false, false, true, RW, false>(p, acc, mb, nb, fr, hg);
        }
    } else if (EPI == EPI_RESID && p.ls) {                   // LayerScale towers (DINOv2): its own path keeps the others lean
        if (interior) gemm_epilogue_rowmajor_impl<EPI, NI, NJ, ACT, false, true, false, false, RW, false>(p, acc, mb, nb, fr, hg);
        else gemm_epilogue_rowmajor_impl<EPI, NI, NJ, ACT, true, true, false, false, RW, false>(p, acc, mb, nb, fr, hg);
    } else if ((EPI == EPI_BIAS || EPI == EPI_ACT) && p.ln_rt) {   // LayerNorm folded into this GEMM
        if (interior) gemm_epilogue_rowmajor_impl<EPI, NI, NJ, ACT, false, false, true, false, RW, false>(p, acc, mb, nb, fr, hg);
        else gemm_epilogue_rowmajor_impl<EPI, NI, NJ, ACT, true, false, true, false, RW, false>(p, acc, mb, nb, fr, hg);
    } else {
        if (interior) gemm_epilogue_rowmajor_impl<EPI, NI, NJ, ACT, false, false, false, false, RW, false>(p, acc, mb, nb, fr, hg);
        else gemm_epilogue_rowmajor_impl<EPI, NI, NJ, ACT, true, false, false, false, RW, false>(p, acc, mb, nb, fr, hg);
    }
}

template <int EPI, int NI, int NJ, int ACT, bool STATS, typename ACC>
VR_DEV void gemm_epilogue_rowmajor_act(const GemmArgs& p, const ACC (&acc)[NI][NJ], int mb, int nb, int fr, int hg) {
    // AL: 16-byte stores (and residual loads) - rows of C (and of the residual) 16-byte aligned, 16x16 accumulators, an even number of column
    // blocks.  One uniform decision per tile; everything below it is compiled once per value (no per-store branches).
    const bool al = EPI != EPI_F32 && AccGeom<ACC>::QN == 1 && (NJ % 2) == 0 && (p.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 &&
                    (EPI != EPI_RESID || (reinterpret_cast<uintptr_t>(p.resid) & 15) == 0);
    if (al) gemm_epilogue_rowmajor_rw<EPI, NI, NJ, ACT, STATS, EPI != EPI_F32>(p, acc, mb, nb, fr, hg);
    else gemm_epilogue_rowmajor_rw<EPI, NI, NJ, ACT, STATS, false>(p, acc, mb, nb, fr, hg);
}

// the convolution kernels' epilogue when GemmArgs::gn_partial is set (plain bias / residual epilogue + GroupNorm partial sums)
template <int EPI, int NI, int NJ, typename ACC>
VR_DEV void gemm_epilogue_rowmajor_gn(const GemmArgs& p, const ACC (&acc)[NI][NJ], int mb, int nb, int fr, int hg) {
    static_assert(EPI == EPI_BIAS || EPI == EPI_RESID, "GroupNorm partials come with the bias / residual epilogues");
    const bool interior = mb + NI * AccGeom<ACC>::RBLK <= p.M;
    const bool al = (NJ % 2) == 0 && (p.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 &&
                    (EPI != EPI_RESID || (reinterpret_cast<uintptr_t>(p.resid) & 15) == 0);
    if (al) {
        if (interior) gemm_epilogue_rowmajor_impl<EPI, NI, NJ, ACT_NONE, false, false, false, false, true, true>(p, acc, mb, nb, fr, hg);
        else gemm_epilogue_rowmajor_impl<EPI, NI, NJ, ACT_NONE, true, false, false, false, true, true>(p, acc, mb, nb, fr, hg);
    } else {
        if (interior) gemm_epilogue_rowmajor_impl<EPI, NI, NJ, ACT_NONE, false, false, false, false, false, true>(p, acc, mb, nb, fr, hg);
        else gemm_epilogue_rowmajor_impl<EPI, NI, NJ, ACT_NONE, true, false, false, false, false, true>(p, acc, mb, nb, fr, hg);
    }
}

// STATS: only the kernel the dispatcher routes statistics-emitting residual GEMMs to (v2) compiles that epilogue
template <int EPI, int NI, int NJ, bool STATS = false, typename ACC>
VR_DEV void gemm_epilogue_rowmajor(const GemmArgs& p, const ACC (&acc)[NI][NJ], int mb, int nb, int fr, int hg) {
    if (EPI == EPI_ACT) {
        switch (p.act) {
            case ACT_QUICK_GELU: gemm_epilogue_rowmajor_act<EPI, NI, NJ, ACT_QUICK_GELU, false>(p, acc, mb, nb, fr, hg); break;
            case ACT_GELU_ERF: gemm_epilogue_rowmajor_act<EPI, NI, NJ, ACT_GELU_ERF, false>(p, acc, mb, nb, fr, hg); break;
            case ACT_GELU_TANH: gemm_epilogue_rowmajor_act<EPI, NI, NJ, ACT_GELU_TANH, false>(p, acc, mb, nb, fr, hg); break;
            default: gemm_epilogue_rowmajor_act<EPI, NI, NJ, ACT_NONE, false>(p, acc, mb, nb, fr, hg); break;
        }
    } else {
        gemm_epilogue_rowmajor_act<EPI, NI, NJ, ACT_NONE, STATS>(p, acc, mb, nb, fr, hg);
    }
}

// V^T scatter: vt[n * ldc + perm16(m)], four consecutive tokens per 8-byte store (layout: see attention.hip)
template <int NI, int NJ, bool HAS_LN, typename ACC>
VR_DEV void gemm_epilogue_vt_impl(const GemmArgs& p, const ACC (&acc)[NI][NJ], int mb, int nb, int fr, int hg) {
    constexpr int QN = AccGeom<ACC>::QN, RBLK = AccGeom<ACC>::RBLK;
    float bj[NJ], sj[HAS_LN ? NJ : 1];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bj[j] = 0.f;
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) bj[j] = p.bias[nb + j * RBLK + fr];
    }
    if (HAS_LN) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) sj[j] = p.ln_s[nb + j * RBLK + fr];
    }
    drain_visible_loads();
    // LN statistics of the tokens this lane stores (four consecutive m per (i, q); the buffer is padded and zero past M) are
    // loaded for half of the row blocks at a time: all at once costs 64 VGPRs on top of the live accumulators (spills)
    constexpr int IH = (HAS_LN && NI >= 4) ? NI / 2 : NI;
#pragma unroll
    for (int i0 = 0; i0 < NI; i0 += IH) {
        float4 ra[HAS_LN ? IH * QN : 1], rb[HAS_LN ? IH * QN : 1];
        if (HAS_LN) {
#pragma unroll
            for (int i = 0; i < IH; ++i)
#pragma unroll
                for (int q = 0; q < QN; ++q) {
                    const int m = mb + (i0 + i) * RBLK + q * 8 + hg * 4;
                    // four consecutive tokens: consecutive physical rows too (a row map's period is a multiple of 4).  Rows past M read the zero
                    // padding of the statistics buffer; with a row map they are clamped to the last four rows instead (never stored either way)
                    const float4* src = reinterpret_cast<const float4*>(p.ln_rt + visrep_a_row(p, (p.a_period > 0 && m + 4 > p.M) ? p.M - 4 : m));
                    ra[i * QN + q] = src[0];                // (rstd0, t0, rstd1, t1)
                    rb[i * QN + q] = src[1];                // (rstd2, t2, rstd3, t3)
                }
            drain_visible_loads();
        }
        // WIDE (16x16 accumulators): the lanes fg and fg ^ 2 of a row hold token groups that are NEIGHBOURS in the perm16 order
        // (memory order of the four 4-token groups of a 16-token block is 0, 2, 1, 3), so one v_permlane32_swap per word between the
        // row blocks (i, i + 1) gives the lower half-wave 8 consecutive stored tokens of block i and the upper half-wave 8 of block
        // i + 1: one 16-byte store per pair of blocks instead of two 8-byte ones (the store stream is issue-bound, see the row-major
        // epilogue).  A chunk's first group is the lower-numbered one (groups 0 / 1, the second is group 2 / 3 = + 8 tokens).
        constexpr bool WIDE = QN == 1 && (IH % 2) == 0;
        const bool wide = WIDE && (p.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0;      // uniform
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bf16_t* colp = p.C + (size_t)(nb + j * RBLK + fr) * p.ldc;
#pragma unroll
            for (int ii0 = 0; ii0 < IH; ii0 += (WIDE ? 2 : 1))
#pragma unroll
                for (int q = 0; q < QN; ++q) {
                    u32x2 o2[WIDE ? 2 : 1];
                    int mm[WIDE ? 2 : 1];
#pragma unroll
                    for (int t = 0; t < (WIDE ? 2 : 1); ++t) {
                        const int ii = ii0 + t, i = i0 + ii;
                        mm[t] = mb + i * RBLK + q * 8 + hg * 4;                                   // multiple of 4
                        float v0, v1, v2, v3;
                        if (HAS_LN) {
                            const float4 a4 = ra[ii * QN + q], b4 = rb[ii * QN + q];
                            v0 = __builtin_fmaf(acc[i][j][4 * q + 0], a4.x, __builtin_fmaf(a4.y, sj[j], bj[j]));
                            v1 = __builtin_fmaf(acc[i][j][4 * q + 1], a4.z, __builtin_fmaf(a4.w, sj[j], bj[j]));
                            v2 = __builtin_fmaf(acc[i][j][4 * q + 2], b4.x, __builtin_fmaf(b4.y, sj[j], bj[j]));
                            v3 = __builtin_fmaf(acc[i][j][4 * q + 3], b4.z, __builtin_fmaf(b4.w, sj[j], bj[j]));
                        } else {
                            v0 = acc[i][j][4 * q + 0] + bj[j]; v1 = acc[i][j][4 * q + 1] + bj[j];
                            v2 = acc[i][j][4 * q + 2] + bj[j]; v3 = acc[i][j][4 * q + 3] + bj[j];
                        }
                        o2[t] = u32x2{pack_bf16(v0, v1), pack_bf16(v2, v3)};
                    }
                    auto perm16 = [](int m) { return (m & ~15) | ((((m >> 2) & 1) << 1 | ((m >> 3) & 1)) << 2); };   // swap 4-token groups 1 <-> 2
                    if (WIDE && wide) {
                        // swap(vdst = block i word, src = block i + 1 word): lanes 32-63 of vdst <-> lanes 0-31 of src
                        const auto w0 = __builtin_amdgcn_permlane32_swap(o2[0][0], o2[WIDE ? 1 : 0][0], false, false);
                        const auto w1 = __builtin_amdgcn_permlane32_swap(o2[0][1], o2[WIDE ? 1 : 0][1], false, false);
                        const int m = (hg & 2) ? mm[WIDE ? 1 : 0] - 8 : mm[0];                    // first token group of this lane's 8-token chunk
                        if (m + 8 < p.M) {                                                       // both token groups of the chunk exist
                            const u32x4 q4 = {(unsigned)w0[0], (unsigned)w1[0], (unsigned)w0[1], (unsigned)w1[1]};
                            store_b128(colp + perm16(m), __builtin_bit_cast(f32x4, q4));
                        } else if (m < p.M) {                                                    // last block of the token axis: token groups >= M
                            store_b64(colp + perm16(m), u32x2{(unsigned)w0[0], (unsigned)w1[0]});   // are never written (as before)
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < (WIDE ? 2 : 1); ++t)
                            if (mm[t] < p.M) store_b64(colp + perm16(mm[t]), o2[t]);
                    }
                }
        }
    }
}

template <int NI, int NJ, typename ACC>
VR_DEV void gemm_epilogue_vt(const GemmArgs& p, const ACC (&acc)[NI][NJ], int mb, int nb, int fr, int hg) {
    if (p.ln_rt) gemm_epilogue_vt_impl<NI, NJ, true>(p, acc, mb, nb, fr, hg);
    else gemm_epilogue_vt_impl<NI, NJ, false>(p, acc, mb, nb, fr, hg);
}

// ---- EPI_F32X: fp32 epilogue of the split-bf16 GEMM (16x16 accumulators): bias, EXACT activation (libm expf / erff / tanhf like the
// fp32 tower path, f32ops.hip), LayerScale + fp32 residual, fp32 output and / or the output's own three bf16 planes (the next GEMM's operand)
VR_DEV float act_exact_f32(float x, int act) {
    switch (act) {
        case ACT_QUICK_GELU: return x * (1.0f / (1.0f + expf(-1.702f * x)));          // x * sigmoid(1.702 x) (HF QuickGELUActivation)
        case ACT_GELU_ERF: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        case ACT_GELU_TANH: return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
        default: return x;
    }
}
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): the subtractions are exact in fp32, 24 significand bits survive
VR_DEV void split_bf16x3(const float (&v)[4], u32x2& hi, u32x2& mid, u32x2& lo) {
    hi = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
    const float r0 = v[0] - bf_lo(hi[0]), r1 = v[1] - bf_hi(hi[0]), r2 = v[2] - bf_lo(hi[1]), r3 = v[3] - bf_hi(hi[1]);
    mid = u32x2{pack_bf16(r0, r1), pack_bf16(r2, r3)};
    lo = u32x2{pack_bf16(r0 - bf_lo(mid[0]), r1 - bf_hi(mid[0])), pack_bf16(r2 - bf_lo(mid[1]), r3 - bf_hi(mid[1]))};
}

template <int NI, int NJ, int ACT>
VR_DEV void gemm_epilogue_f32x_act(const GemmArgs& p, const f32x4 (&acc)[NI][NJ], int mb, int nb, int fr, int hg) {
    float4 bv[NJ], lv[NJ];
    auto col = [&](int j) { return nb + j * 16 + hg * 4; };
#pragma unroll
    for (int j = 0; j < NJ; ++j) { bv[j] = float4{0.f, 0.f, 0.f, 0.f}; lv[j] = float4{1.f, 1.f, 1.f, 1.f}; }
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) bv[j] = *reinterpret_cast<const float4*>(p.bias + col(j));
    }
    if (p.ls) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) lv[j] = *reinterpret_cast<const float4*>(p.ls + col(j));
    }
    drain_visible_loads();
    float* C32 = reinterpret_cast<float*>(p.C);
    const float* R = p.resid32;
    constexpr int RB = 1;                                     // rows whose residual loads are in flight together (2: 12 VGPR spills)
#pragma unroll
    for (int i0 = 0; i0 < NI; i0 += RB) {
        float4 rv[RB][NJ];
        bool ok[RB];
        size_t row[RB];
#pragma unroll
        for (int ii = 0; ii < RB; ++ii) {
            const int m = mb + (i0 + ii) * 16 + fr;
            ok[ii] = m < p.M;
            row[ii] = (size_t)(ok[ii] ? m : p.M - 1);
            if (R) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) rv[ii][j] = *reinterpret_cast<const float4*>(R + row[ii] * p.ldc + col(j));
#ifndef VISREP_F32X_NO_RETIRE                                    // A/B knob (tools/): leaves the K-loop header wait in place
#pragma unroll
                for (int j = 0; j < NJ; ++j) retire_load(rv[ii][j]);      // consumed on every path: rows past M skip everything below
#endif
            }
        }
#pragma unroll
        for (int ii = 0; ii < RB; ++ii)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f32x4 a = acc[i0 + ii][j];
                float v[4] = {a[0] + bv[j].x, a[1] + bv[j].y, a[2] + bv[j].z, a[3] + bv[j].w};
                if (ACT != ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_exact_f32(v[e], ACT);
                }
                if (R) {
                    v[0] = rv[ii][j].x + lv[j].x * v[0]; v[1] = rv[ii][j].y + lv[j].y * v[1];
                    v[2] = rv[ii][j].z + lv[j].z * v[2]; v[3] = rv[ii][j].w + lv[j].w * v[3];
                }
                if (!ok[ii]) continue;
                if (C32) store_b128(C32 + row[ii] * p.ldc + col(j), f32x4{v[0], v[1], v[2], v[3]});
                if (p.planes) {
                    u32x2 hi, mid, lo;
                    split_bf16x3(v, hi, mid, lo);
                    bf16_t* pr = p.planes + row[ii] * p.ldp + col(j);
                    store_b64(pr, hi);
                    store_b64(pr + p.N, mid);
                    if (p.out_planes == 3) store_b64(pr + 2 * p.N, lo);      // uniform: two-plane consumers (3 / 4 products) never read lo
                }
            }
    }
}

template <int NI, int NJ>
VR_DEV void gemm_epilogue_f32x(const GemmArgs& p, const f32x4 (&acc)[NI][NJ], int mb, int nb, int fr, int hg) {
    switch (p.epi == EPI_F32X ? p.act : ACT_NONE) {
        case ACT_QUICK_GELU: gemm_epilogue_f32x_act<NI, NJ, ACT_QUICK_GELU>(p, acc, mb, nb, fr, hg); break;
        case ACT_GELU_ERF: gemm_epilogue_f32x_act<NI, NJ, ACT_GELU_ERF>(p, acc, mb, nb, fr, hg); break;
        case ACT_GELU_TANH: gemm_epilogue_f32x_act<NI, NJ, ACT_GELU_TANH>(p, acc, mb, nb, fr, hg); break;
        default: gemm_epilogue_f32x_act<NI, NJ, ACT_NONE>(p, acc, mb, nb, fr, hg); break;
    }
}
