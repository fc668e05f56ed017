// vq_cells.h -- VectorQuantizer forward for LARGE codebooks (64 < K <= 4096, embedding_dim 16): included by vq.hip inside its
// anonymous namespace (it uses vq.hip's helpers: vq2_min / vq2_min3 / vq2_med3 / vq2_swap / vq_exact_dist / vq_better).
// Reference: HiddenStateExtractor/vq_vae.py:65-82 (distances, argmin, gather, straight-through value, counters).
//
// Same contract as vq_forward_mfma_kernel -- a bf16-split matrix FILTER decides which codes can be the reference's argmin,
// the reference's own arithmetic decides among them, so every index is the reference's -- with the inner loop rebuilt around
// what round 5 measured (2.6 vector instructions per score, 32 workgroup barriers per 256 positions):
//
//   * v_mfma_f32_32x32x16_bf16, codes on the rows, positions on the columns: a lane holds 16 scores of ONE position per
//     instruction, and the instruction keeps the SIMD's vector issue for 8 of its 32 cycles (the 16x16x32 form: 8 of 16).
//     Three products per 16 dimensions instead of four -- a_hi z_hi + a_hi z_lo + a_lo z_hi; the dropped a_lo z_lo is
//     below 2^-16 |a||z| and goes into the tolerance (ETA below): 25 % fewer matrix cycles.
//   * NO index bits in the scores and NO runner-up per score.  The 16 scores of a lane and chunk fall into 8 CELLS (pairs of
//     accumulator registers); a cell's running minimum over the 4 chunks of a GROUP of 128 codes is one v_min3_f32 per two
//     scores: 0.5 vector instructions per score.  Once per group (not per score) the 8 cell minima get their cell number
//     in their low 3 bits and the lane's two best cells are updated.  The codebook operand is permuted so that the 8 codes
//     of a cell are consecutive rows of the codebook.
//   * A position is settled by the filter when its second best CELL is farther than the proven tolerance from its best
//     cell: every code outside the best cell is then farther than the tolerance from the best score, so the reference's
//     argmin lies inside that cell, and the owning lane evaluates the cell's 8 codes in the reference's arithmetic and
//     order (first minimum).  Otherwise the position goes to the exact re-check over the groups whose minimum (kept per
//     position and group with 24 bits, rounded toward -inf) is within the tolerance: one LDS read and a ballot name them.
//   * No LDS stream and no barrier in the loop: the A operand of a chunk (32 codes: 2 x 16 bytes per lane) comes straight
//     from L2 into a register ring four chunks ahead; a wave keeps 128 positions (4 tiles) so that every operand fetched
//     is used by 12 matrix instructions.  The norms (16 KB) stay in LDS; waves never wait for each other.
//
// Layout of a pass: 128 consecutive positions of one sample; tile t, column n = position 4 n + t, so that the 8 loads of
// a lane (its 8 dimensions) are 16-byte loads of 4 consecutive positions; after the cross-half merge lane (kh, n) OWNS the
// two consecutive positions 4 n + 2 kh, + 1 (slot s = tile 2 kh + s): 16-byte index stores, 8-byte value stores.

// (constants, the row <-> code permutation and the operand preparation: vq.hip, above vq_prep_kernel)

// measurement builds only (tools/exp/vq_cells_parts.sh; results are wrong, the time is what is read): compile-time switches
// -DVQC_OFF=bits: 4 no exact re-checks, 8 no exact evaluation of the best cell, 16 no group ends, 32 no cell minima,
// 64 the operand ring is never refilled (stale codes), 128 the norms are read once per pass, 256 every other group end skipped
#ifndef VQC_OFF
#define VQC_OFF 0
#endif
#ifndef VQC_CELL_BATCH
#define VQC_CELL_BATCH 4         // rows of the best cell requested together (8: measured below)
#endif
#ifndef VQC_ZF_EARLY
#define VQC_ZF_EARLY 1
#endif
#ifndef VQC_ZPREFETCH
#define VQC_ZPREFETCH 0          // 1: the next pass's latents are requested before this pass's tail (measured: no gain, more spills)
#endif
#define VQC_DBG(bit) ((VQC_OFF & (bit)) != 0)

// diagnostic build only (-DVQ2_STAMPS, tools/exp/vq_cells_stamps.py): s_memtime at the phase boundaries of a pass, per-wave sums
// added into the workspace header -- the shipped library has no stamp
#ifdef VQ2_STAMPS
#define VQC_STAMP(i)                                                                                      \
    {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        unsigned long long t_;                                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        st_sum[i] += t_ - st_prev;                                                                        \
        st_prev = t_;                                                                                     \
    }
#else
#define VQC_STAMP(i)
#endif

template <int NPROD>
__global__ __launch_bounds__(256, 2) void vq_cells_kernel(
    const float *__restrict__ z, const float *__restrict__ cb, const vqc_u32x4 *__restrict__ cbP,
    const float *__restrict__ nrmP, const float *__restrict__ nrm, long long *__restrict__ idx, float *__restrict__ out,
    double *__restrict__ sse_slabs, int *__restrict__ hrep, int R, int *__restrict__ hdr, int K, int HW, long long P,
    int stagger)
{
    constexpr int D = 16, NT = VQC_NT;
    static_assert(NPROD == 3 || NPROD == 4, "three or four products of the bf16 split");
    constexpr float U = 5.9604645e-8f;                             // 2^-24
    // tol = 1.25 x (2 eta + 2 rho max(|z|^2 + s', 0)), eta = ETA u A, rho = (D/16 + 18) u  (derivation: vq.hip, above
    // vq_forward_mfma_kernel).  ETA here:
    //     2   the norm, rounded once
    //  + 16   the cell number in the low 3 bits of a cell minimum (7 ulp, 1 ulp <= 2 u |s'|)
    //  + 256  the split's remainders a r_z + r_a z (2 x 2^-17 |a||z|, sum |a||z| <= A)
    //  + 256  (three products only) the dropped a_lo z_lo: |a_lo| <= 2^-8 |a|, |z_lo| <= 2^-8 |z|
    //  + 50 per matrix instruction: 16 products + C added in an unspecified order, 17 additions of at most one ulp (2 u) of
    //         partial sums below 1.01 A: 35 u A, taken as 50
    constexpr float ETA = 2.f + 16.f + 256.f + (NPROD == 3 ? 256.f : 0.f) + 50.f * NPROD;
    constexpr float TOL_A = 2.5f * ETA * U, TOL_D = 2.5f * (D / 16 + 18) * U;
    __shared__ __attribute__((aligned(16))) float s_nrm[VQC_MAX_K];
    // per wave and group: the group minima of the lane's two owned positions, the top 24 bits of each (rounded toward -inf: 16
    // significand bits, 1.5e-5 relative -- a sixth of the tolerance; bf16 minima, 4e-3, sent a flagged position through 5-15
    // groups of 128 codes instead of the one or two that really hold a candidate) as 32 + 16 bits
    __shared__ unsigned s_pm[4][VQC_MAX_K / VQC_GROUP][64];
    __shared__ unsigned short s_pmh[4][VQC_MAX_K / VQC_GROUP][64];
    __shared__ float s_em[4];
    __shared__ double s_red[4];
    const int lane = threadIdx.x & 63, kh = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K128 = (K + VQC_GROUP - 1) / VQC_GROUP * VQC_GROUP, NCH = K128 >> 5, NG = NCH / VQC_GCH;
    int *__restrict__ hist = hrep + (long long)(blockIdx.x % (unsigned)R) * K;

    // ---- prologue: norms into LDS, max_k |e_k|^2 (a non-finite codebook makes it inf: every position takes the exact path)
    for (int i = threadIdx.x; i < K128; i += 256) s_nrm[i] = nrmP[i];
    float emax;
    {
        float em = 0.f;
        bool bad = false;
        for (int k = threadIdx.x; k < K; k += 256) {
            const float v = nrm[k];
            bad |= !(v < __builtin_inff());
            em = fmaxf(em, v);
        }
        em = bad ? __builtin_inff() : em;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) em = fmaxf(em, __shfl_xor(em, o, 64));
        if (lane == 0) s_em[wave] = em;
        __syncthreads();
        emax = fmaxf(fmaxf(s_em[0], s_em[1]), fmaxf(s_em[2], s_em[3]));
    }

    // Stagger (guide: two waves that run the same program on one SIMD reach their matrix work and their vector-only tails
    // together): the second workgroup of every CU -- the upper half of the grid, dispatched after the lower half has taken one
    // slot per CU -- starts `stagger` x 64 cycles late, about half a pass, so that one wave's tail (exact cell evaluation,
    // re-checks, stores: a quarter of its time, no matrix work) runs beside the other wave's code stream.  Measured: 2-4 %
    // (404 -> 387-398 us); pairing the late half by parity or by XCD neighbours instead gains nothing.
    if (stagger > 0 && blockIdx.x >= (gridDim.x + 1) / 2)
        for (int i = 0; i < stagger; i += 127) __builtin_amdgcn_s_sleep(127);
    const unsigned npass = (unsigned)(P >> 7), pps = (unsigned)HW >> 7;      // passes of 128 positions; per sample
    double sse = 0.0;
    int nflag = 0;
#ifdef VQ2_STAMPS
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
    const vqc_u32x4 *__restrict__ aP = cbP + lane;
    vqc_u32x4 ah[VQC_GCH], al[VQC_GCH];                                       // the operand ring: chunks cc .. cc + 3
#pragma unroll
    for (int c = 0; c < VQC_GCH; ++c) { ah[c] = aP[(c * 2) * 64]; al[c] = aP[(c * 2 + 1) * 64]; }

    // the lane's 8 dimensions of its 4 positions 4 n + t of a pass; the NEXT pass's are requested before this pass's tail
    f32x4 zq[8];
    auto load_z = [&](unsigned pass) {
        const unsigned pc = pass < npass ? pass : npass - 1;                 // (past the end: a harmless reload)
        const unsigned b = pc / pps, pw = pc - b * pps;
        const float *__restrict__ zb = z + (long long)b * D * HW + (long long)pw * 128;
#pragma unroll
        for (int i = 0; i < 8; ++i) zq[i] = *reinterpret_cast<const f32x4 *>(zb + (long long)(8 * kh + i) * HW + 4 * n);
    };
    if (VQC_ZPREFETCH) load_z(blockIdx.x * 4u + (unsigned)wave);

    for (unsigned pass = blockIdx.x * 4u + (unsigned)wave; pass < npass; pass += gridDim.x * 4u) {
        const unsigned b = pass / pps, pw = pass - b * pps;
        // ---- B operands: (z_hi, z_lo) of the lane's 8 dimensions for its 4 positions 4 n + t
        vqc_u32x4 Zh[NT], Zl[NT];
        {
            if (!VQC_ZPREFETCH) load_z(pass);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                unsigned hp[4], lp[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float z0 = zq[2 * q][t], z1 = zq[2 * q + 1][t];
                    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){z0, z1}, vqc_bf16x2));
                    const float r0 = z0 - __builtin_bit_cast(float, h01 << 16), r1 = z1 - __builtin_bit_cast(float, h01 & 0xffff0000u);
                    hp[q] = h01;
                    lp[q] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r0, r1}, vqc_bf16x2));
                }
                Zh[t] = (vqc_u32x4){hp[0], hp[1], hp[2], hp[3]};
                Zl[t] = (vqc_u32x4){lp[0], lp[1], lp[2], lp[3]};
            }
        }
        VQC_STAMP(0)                                           // latents loaded, operands split
        float cm[NT][8], m1[NT], m2[NT];
        int G1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { m1[t] = 3.4028235e38f; m2[t] = 3.4028235e38f; G1[t] = 0; }
        vqc_f32x16 acc[NT];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[NT - 1][r] = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) cm[NT - 1][p] = 0.f;

        // the cell minima of tile t from its accumulators: one instruction per pair of scores
        auto cells = [&](int t, int p0, int p1, bool first) {
#pragma unroll
            for (int p = p0; p < p1; ++p)
                cm[t][p] = first ? vq2_min(acc[t][2 * p], acc[t][2 * p + 1]) : vq2_min3(cm[t][p], acc[t][2 * p], acc[t][2 * p + 1]);
        };
        // once per group: cell numbers into the low 3 bits, the lane's two best cells, the group minimum of the owned positions
        auto group_end = [&](int g) {
            float gm[NT];
            // (two tiles at a time: the stages of one tile are a dependent chain, two chains interleave; four would need
            //  70 more registers than the loop has)
#pragma unroll
            for (int t0 = 0; t0 < NT; t0 += 2) {
                float v[2][8], n0[2], n1[2], n2[2], d0[2], d1[2], d2[2], M1[2], q[2], dm[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int p = 0; p < 8; ++p)
                        v[u][p] = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, cm[t0 + u][p]) & ~7u) | (unsigned)p);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = t0 + u;
                    n0[u] = vq2_min3(v[u][0], v[u][1], v[u][2]); d0[u] = vq2_med3(v[u][0], v[u][1], v[u][2]);
                    n1[u] = vq2_min3(v[u][3], v[u][4], v[u][5]); d1[u] = vq2_med3(v[u][3], v[u][4], v[u][5]);
                    n2[u] = vq2_min3(v[u][6], v[u][7], m1[t]);   d2[u] = vq2_med3(v[u][6], v[u][7], m1[t]);
                    gm[t] = vq2_min(v[u][6], v[u][7]);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = t0 + u;
                    M1[u] = vq2_min3(n0[u], n1[u], n2[u]); q[u] = vq2_med3(n0[u], n1[u], n2[u]);
                    dm[u] = vq2_min3(d0[u], d1[u], d2[u]); gm[t] = vq2_min3(n0[u], n1[u], gm[t]);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = t0 + u;
                    // the second best cell: the runner-up of the triple minima or the median of the winner's triple (every
                    // other median or minimum is some cell other than the best: no smaller than the runner-up)
                    m2[t] = vq2_min3(q[u], dm[u], m2[t]);
                    G1[t] = __builtin_bit_cast(unsigned, M1[u]) != __builtin_bit_cast(unsigned, m1[t]) ? g : G1[t];
                    m1[t] = M1[u];
                }
            }
            // both halves' minima of the two owned positions (slot s = tile 2 kh + s), top 24 bits, toward -inf, NaN kept
            unsigned t24[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float x = gm[s], y = gm[s + 2];
                vq2_swap<32>(x, y);
                const float pmin = vq2_min(x, y);
                const unsigned pb = __builtin_bit_cast(unsigned, pmin);
                t24[s] = (pb >> 8) + ((pmin < 0.f && (pb & 0xffu)) ? 1u : 0u);
            }
            const unsigned packed = t24[0] | (t24[1] << 24);
            s_pmh[wave][g][lane] = (unsigned short)(t24[1] >> 8);
            s_pm[wave][g][lane] = packed;
        };

        // ---- the code stream: per chunk and tile 3 (4) matrix instructions; the cell minima of the PREVIOUS tile go between
        // them (its accumulators were completed two matrix instructions ago: no wait; inline-asm consumers are not covered
        // by the compiler's matrix -> vector hazard padding, so the distance is kept by construction, see `drain` below)
        // the norms of a chunk (the initial accumulator of its first product) are read from LDS one chunk ahead
        vqc_f32x16 nrb[2];
        auto read_norms = [&](vqc_f32x16 &nr, int cc) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(s_nrm + cc * 32 + 8 * j + 4 * kh);
                nr[4 * j] = v.x; nr[4 * j + 1] = v.y; nr[4 * j + 2] = v.z; nr[4 * j + 3] = v.w;
            }
        };
        read_norms(nrb[0], 0);
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int ccl = 0; ccl < VQC_GCH; ++ccl) {
                const int cc = g * VQC_GCH + ccl;
                const vqc_f32x16 nr = nrb[ccl & 1];
                const vqc_bf16x8 Ah = __builtin_bit_cast(vqc_bf16x8, ah[ccl]), Al = __builtin_bit_cast(vqc_bf16x8, al[ccl]);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    // the tile whose accumulators are complete: the previous one (the last tile of the previous chunk for t = 0)
                    const int pt = t == 0 ? NT - 1 : t - 1;
                    const bool pfirst = t == 0 ? ccl == 1 : ccl == 0;                // was ITS chunk the first of its group
                    // (the very first tile of a pass folds the stale accumulators of tile NT - 1 into cm[NT - 1]: harmless, that
                    //  cell row restarts from its own first chunk before anything reads it)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, __builtin_bit_cast(vqc_bf16x8, Zh[t]), nr, 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, __builtin_bit_cast(vqc_bf16x8, Zl[t]), acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!VQC_DBG(32)) cells(pt, 0, 4, pfirst);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, __builtin_bit_cast(vqc_bf16x8, Zh[t]), acc[t], 0, 0, 0);
                    if constexpr (NPROD == 4)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, __builtin_bit_cast(vqc_bf16x8, Zl[t]), acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!VQC_DBG(32)) cells(pt, 4, 8, pfirst);
                    __builtin_amdgcn_sched_barrier(0);
                    // the previous group is complete once the last tile of its last chunk has been folded in
                    if (t == 0 && ccl == 0 && g > 0 && !VQC_DBG(16) && !(VQC_DBG(256) && (g & 1))) group_end(g - 1);
                    if (t == 1 && !VQC_DBG(128)) read_norms(nrb[(ccl + 1) & 1], cc + 1 < NCH ? cc + 1 : 0);
                }
                // this ring slot's next chunk (the stream wraps into the next pass)
                const int nxt = cc + VQC_GCH < NCH ? cc + VQC_GCH : cc + VQC_GCH - NCH;
                if (!VQC_DBG(64)) {
                    ah[ccl] = aP[(nxt * 2) * 64];
                    al[ccl] = aP[(nxt * 2 + 1) * 64];
                }
            }
        }
        VQC_STAMP(1)                                           // the code stream
        // the owned positions' complete latent vectors are requested here: they land under the drain and the last group end
        const long long own = (long long)pw * 128 + 4 * n + 2 * kh;                 // first owned position inside the sample
        f32x2 zf[D];
        if (VQC_ZF_EARLY) {
#pragma unroll
            for (int d = 0; d < D; ++d) zf[d] = *reinterpret_cast<const f32x2 *>(z + ((long long)b * D + d) * HW + own);
        }
        // drain: the last tile's accumulators have no matrix instruction behind them -- wait them out explicitly
        // (16 passes of the last instruction; s_nop counts issue cycles)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        cells(NT - 1, 0, 8, VQC_GCH == 1);
        if (!VQC_DBG(16)) group_end(NG - 1);

        VQC_STAMP(2)                                           // drain + last group end
        if (VQC_ZPREFETCH) load_z(pass + gridDim.x * 4u);
        // ---- the tail: everything of a position is in its owner's registers ----
        if (!VQC_ZF_EARLY) {
#pragma unroll
            for (int d = 0; d < D; ++d) zf[d] = *reinterpret_cast<const f32x2 *>(z + ((long long)b * D + d) * HW + own);
        }
        float a1[2], thrv[2];
        int kown[2];
        bool flagged[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // both halves' best cells of tile 2 kh + s: after the swap x is the lower half's value, y the upper half's
            float x1 = m1[s], y1 = m1[s + 2], x2 = m2[s], y2 = m2[s + 2];
            float xg = __builtin_bit_cast(float, G1[s]), yg = __builtin_bit_cast(float, G1[s + 2]);
            vq2_swap<32>(x1, y1);
            vq2_swap<32>(x2, y2);
            vq2_swap<32>(xg, yg);
            const bool up = y1 < x1;
            a1[s] = vq2_min(x1, y1);
            const float a2 = vq2_min3(x2, y2, vq2_max(x1, y1));
            const int gw = __builtin_bit_cast(int, up ? yg : xg);
            float zz = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) zz = fmaf(zf[d][s], zf[d][s], zz);
            const float tol = TOL_A * (zz + 2.f * emax) + TOL_D * fmaxf(zz + a1[s], 0.f) + 1e-30f;
            flagged[s] = !((a2 - a1[s]) > tol);                                     // also true when anything is NaN / inf
            thrv[s] = a1[s] + tol;
            // the best cell's 8 codes in the reference's arithmetic, ascending: first minimum, argmax(-dist) NaN rule
            const unsigned pc = __builtin_bit_cast(unsigned, a1[s]) & 7u;
            const int cell = (int)(((pc >> 1) * 2 + (up ? 1u : 0u)) * 2 + (pc & 1u));
            const int base = gw * VQC_GROUP + cell * 8;
            float zv[D];
#pragma unroll
            for (int d = 0; d < D; ++d) zv[d] = zf[d][s];
            float bd = __builtin_inff();
            int bk = 0x7fffffff;
            if (VQC_DBG(8)) bk = min(base, K - 1);
#pragma unroll
            for (int c0 = 0; c0 < (VQC_DBG(8) ? 0 : 8); c0 += VQC_CELL_BATCH) {
                float er[VQC_CELL_BATCH][D];
#pragma unroll
                for (int u = 0; u < VQC_CELL_BATCH; ++u) {
                    const int k = min(base + c0 + u, K - 1);
#pragma unroll
                    for (int q4 = 0; q4 < D / 4; ++q4) {
                        const f32x4 v4 = reinterpret_cast<const f32x4 *>(cb + (long long)k * D)[q4];
                        er[u][4 * q4] = v4.x; er[u][4 * q4 + 1] = v4.y; er[u][4 * q4 + 2] = v4.z; er[u][4 * q4 + 3] = v4.w;
                    }
                }
#pragma unroll
                for (int u = 0; u < VQC_CELL_BATCH; ++u) {
                    const int k = base + c0 + u;
                    const float dk = vq_exact_dist<D>(zv, er[u]);
                    const bool bt = k < K && ((bk == 0x7fffffff) | vq_better(dk, bd));
                    bd = bt ? dk : bd; bk = bt ? k : bk;
                }
            }
            kown[s] = bk == 0x7fffffff ? 0 : bk;
        }
        VQC_STAMP(3)                                           // owned latents, merge, exact evaluation of the best cells
        // ---- exact re-check of the positions the filter could not settle: the whole wave, one position at a time, over the
        // groups whose minimum lies within the tolerance of the best score (a NaN on either side keeps the group)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            unsigned long long fm = VQC_DBG(4) ? 0ull : __ballot(flagged[s]);
            while (fm) {
                const int fl = __builtin_ctzll(fm);
                fm &= fm - 1;
                float zv[D];
#pragma unroll
                for (int d = 0; d < D; ++d)
                    zv[d] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (float)zf[d][s]), fl));
                const float thr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, thrv[s]), fl));
                float bd = __builtin_inff();
                int bk = 0x7fffffff;
                // lane g looks at group g's minimum (one LDS read for all groups); the groups to visit are the set bits
                unsigned gmask;
                {
                    const int gq = lane < NG ? lane : 0;
                    const unsigned pk = s_pm[wave][gq][fl];
                    const float pg = __builtin_bit_cast(float, (s ? (pk >> 24) | ((unsigned)s_pmh[wave][gq][fl] << 8) : pk & 0xffffffu) << 8);
                    gmask = (unsigned)__ballot(lane < NG && !(pg > thr));
                }
#ifdef VQ2_STAMPS
                st_sum[6] += __builtin_popcount(gmask);        // (counts, not cycles: groups visited / positions re-checked)
                st_sum[7] += 1;
#endif
                while (gmask) {
                    const int g = __builtin_ctz(gmask);
                    gmask &= gmask - 1;
                    float er[2][D];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int k = min(g * VQC_GROUP + lane + 64 * u, K - 1);
#pragma unroll
                        for (int q4 = 0; q4 < D / 4; ++q4) {
                            const f32x4 v4 = reinterpret_cast<const f32x4 *>(cb + (long long)k * D)[q4];
                            er[u][4 * q4] = v4.x; er[u][4 * q4 + 1] = v4.y; er[u][4 * q4 + 2] = v4.z; er[u][4 * q4 + 3] = v4.w;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int k = g * VQC_GROUP + lane + 64 * u;
                        const float dk = vq_exact_dist<D>(zv, er[u]);
                        const bool bt = k < K && ((bk == 0x7fffffff) | vq_better(dk, bd));      // (a lane's codes ascend)
                        bd = bt ? dk : bd; bk = bt ? k : bk;
                    }
                }
                // wave-wide first minimum: NaN beats numbers, equal distances (and two NaNs) go to the smaller code
                auto combine = [&](float od, int ok) {
                    const int an = bd != bd, bn = od != od, lk = ok < bk;
                    const int other = (bn & ((an ^ 1) | lk)) | ((bn ^ 1) & (an ^ 1) & ((od < bd) | ((od == bd) & lk)));
                    bd = other ? od : bd; bk = other ? ok : bk;
                };
                combine(lane_xor1(bd), __builtin_bit_cast(int, lane_xor1(__builtin_bit_cast(float, bk))));
                combine(lane_xor2(bd), __builtin_bit_cast(int, lane_xor2(__builtin_bit_cast(float, bk))));
                combine(lane_xor4(bd), __builtin_bit_cast(int, lane_xor4(__builtin_bit_cast(float, bk))));
                combine(lane_xor8(bd), __builtin_bit_cast(int, lane_xor8(__builtin_bit_cast(float, bk))));
                {
                    float dl, dh; unsigned kl, kh2;
                    vq2_pair<16>(bd, dl, dh); vq2_pair<16>((unsigned)bk, kl, kh2);
                    bd = dl; bk = (int)kl; combine(dh, (int)kh2);
                    vq2_pair<32>(bd, dl, dh); vq2_pair<32>((unsigned)bk, kl, kh2);
                    bd = dl; bk = (int)kl; combine(dh, (int)kh2);
                }
                kown[s] = lane == fl ? bk : kown[s];
                ++nflag;
            }
        }
        VQC_STAMP(4)                                           // exact re-checks
        // ---- gather, straight-through value z + (q - z) (vq_vae.py:71), squared error, stores, counters ----
        float ssef = 0.f;
        f32x2 o[D];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float ev[D];
#pragma unroll
            for (int q4 = 0; q4 < D / 4; ++q4) {
                const f32x4 v4 = reinterpret_cast<const f32x4 *>(cb + (long long)kown[s] * D)[q4];
                ev[4 * q4] = v4.x; ev[4 * q4 + 1] = v4.y; ev[4 * q4 + 2] = v4.z; ev[4 * q4 + 3] = v4.w;
            }
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float diff = ev[d] - zf[d][s];
                o[d][s] = zf[d][s] + diff;
                ssef += diff * diff;
            }
        }
        sse += (double)ssef;
        if (out) {
#pragma unroll
            for (int d = 0; d < D; ++d) *reinterpret_cast<f32x2 *>(out + ((long long)b * D + d) * HW + own) = o[d];
        }
        if (idx) {
            typedef long long vqc_i64x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<vqc_i64x2 *>(idx + (long long)b * HW + own) = (vqc_i64x2){(long long)kown[0], (long long)kown[1]};
        }
        atomicAdd(&hist[kown[0]], 1);
        atomicAdd(&hist[kown[1]], 1);
        VQC_STAMP(5)                                           // gather, value, stores, counters
    }

    const double tot = block_sum(sse, s_red);
    if (threadIdx.x == 0) sse_slabs[blockIdx.x] = tot;
    if (lane == 0 && nflag) atomicAdd(hdr, nflag);
#ifdef VQ2_STAMPS
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(reinterpret_cast<unsigned long long *>(hdr + 4) + i, st_sum[i]);
#endif
}
