// pairwise.hip -- the pairwise mean-squared latent distance of the time-matching loss.
//
// Reference: HiddenStateExtractor/vq_vae.py:324-329 and vae.py:322-326
//     z = z_before.reshape(B, -1);  sim_mat = pow(z.reshape(1, B, n) - z.reshape(B, 1, n), 2).mean(2)
// which materialises a (B, B, n) tensor (n = 4096: 268 MB at B = 128, 68 GB at B = 2048).  Here the (B, B)
// result is formed directly; differences are taken before squaring (no |a|^2 + |b|^2 - 2ab cancellation).
// The weighting / hinge / mean of the (B, B) matrix stays in torch (it is B*B elements); its autograd hands
// g = dL/dsim back to dm_pair_msd_backward:
//     dz[i] = (2/n) * sum_j (g[i][j] + g[j][i]) * (z[i] - z[j])
#include "dm_common.h"

namespace {

constexpr int PM_T = 16;          // pairs block: 16 x 16
constexpr int PM_CH = 64;         // latent elements staged per step

// sim[i][j] = (1/n) sum_d (z[i][d] - z[j][d])^2 ; one workgroup per 16 x 16 block of pairs, thread = one pair
__global__ __launch_bounds__(256) void pair_msd_kernel(const float *__restrict__ z, float *__restrict__ sim, int B, int n)
{
    __shared__ float sI[PM_T][PM_CH + 1], sJ[PM_T][PM_CH + 1];
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    const int i0 = blockIdx.y * PM_T, j0 = blockIdx.x * PM_T;
    double acc = 0.0;
    for (int d0 = 0; d0 < n; d0 += PM_CH) {
        __syncthreads();
        for (int e = threadIdx.x; e < PM_T * PM_CH; e += 256) {
            const int r = e / PM_CH, c = e % PM_CH;
            const int d = d0 + c;
            sI[r][c] = (i0 + r < B && d < n) ? z[(long long)(i0 + r) * n + d] : 0.f;
            sJ[r][c] = (j0 + r < B && d < n) ? z[(long long)(j0 + r) * n + d] : 0.f;
        }
        __syncthreads();
        float part = 0.f;
#pragma unroll 16
        for (int c = 0; c < PM_CH; ++c) {
            const float df = sJ[tj][c] - sI[ti][c];
            part += df * df;
        }
        acc += (double)part;
    }
    if (i0 + ti < B && j0 + tj < B) sim[(long long)(i0 + ti) * B + j0 + tj] = (float)(acc / (double)n);
}

// dz[i][d] = (2/n) * sum_j s_ij * (z[i][d] - z[j][d]),  s_ij = g[i][j] + g[j][i]
// one workgroup per (16 rows i) x (256 elements d); j walks in chunks of 16 through LDS
__global__ __launch_bounds__(256) void pair_msd_backward_kernel(const float *__restrict__ z, const float *__restrict__ g,
                                                                float *__restrict__ dz, int B, int n)
{
    __shared__ float sS[PM_T][PM_T + 1];            // s_ij for the block's 16 rows i and the current 16 columns j
    __shared__ float sZ[PM_T][256];                 // z[j][d chunk]
    const int i0 = blockIdx.y * PM_T, d = blockIdx.x * 256 + threadIdx.x;
    float zi[PM_T];
    double acc[PM_T];
#pragma unroll
    for (int r = 0; r < PM_T; ++r) {
        zi[r] = (i0 + r < B && d < n) ? z[(long long)(i0 + r) * n + d] : 0.f;
        acc[r] = 0.0;
    }
    for (int j0 = 0; j0 < B; j0 += PM_T) {
        __syncthreads();
        {
            const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
            const int i = i0 + r, j = j0 + c;
            sS[r][c] = (i < B && j < B) ? g[(long long)i * B + j] + g[(long long)j * B + i] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < PM_T; ++r) sZ[r][threadIdx.x] = (j0 + r < B && d < n) ? z[(long long)(j0 + r) * n + d] : 0.f;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PM_T; ++r) {
            float part = 0.f;
#pragma unroll
            for (int c = 0; c < PM_T; ++c) part += sS[r][c] * (zi[r] - sZ[c][threadIdx.x]);
            acc[r] += (double)part;
        }
    }
    if (d < n) {
        const double sc = 2.0 / (double)n;
#pragma unroll
        for (int r = 0; r < PM_T; ++r)
            if (i0 + r < B) dz[(long long)(i0 + r) * n + d] = (float)(acc[r] * sc);
    }
}

}  // namespace

extern "C" int dm_pair_msd(const float *z, float *sim, int B, int n, void *stream)
{
    DM_REQUIRE(z && sim && B > 0 && n > 0, "dm_pair_msd: bad argument");
    DM_REQUIRE((long long)B * n < (1LL << 31), "dm_pair_msd: tensor too large for 32-bit offsets");
    const int nb = (B + PM_T - 1) / PM_T;
    hipLaunchKernelGGL(pair_msd_kernel, dim3(nb, nb), dim3(256), 0, (hipStream_t)stream, z, sim, B, n);
    return dm_launch_status("dm_pair_msd");
}

extern "C" int dm_pair_msd_backward(const float *z, const float *g_sim, float *dz, int B, int n, void *stream)
{
    DM_REQUIRE(z && g_sim && dz && B > 0 && n > 0, "dm_pair_msd_backward: bad argument");
    DM_REQUIRE((long long)B * n < (1LL << 31), "dm_pair_msd_backward: tensor too large for 32-bit offsets");
    hipLaunchKernelGGL(pair_msd_backward_kernel, dim3((n + 255) / 256, (B + PM_T - 1) / PM_T), dim3(256), 0,
                       (hipStream_t)stream, z, g_sim, dz, B, n);
    return dm_launch_status("dm_pair_msd_backward");
}

// ================================================================================================================
// The whole time-matching term on the matrix pipe (vq_vae.py:324-332, vae.py:322-336).
//
//   sim_ij = mean_d (z_i - z_j)^2 = (G_ii + G_jj - 2 G_ij) / n,   G = Z Z^T        (B x B x n GEMM, f32 MFMA)
//   loss   = sum_ij sim_ij tm_ij                                                   (mode 0, vq_vae.py:331)
//          = mean_ij v_ij, v = sim * w(tm), tm == 0: v = max(v + margin, 0)        (mode 1, vae.py:327-336)
//   S_ij   = dloss/dsim_ij + dloss/dsim_ji          (formed by the epilogue: the backward needs nothing else)
//   dz_i   = (2/n) (rowsum(S)_i z_i - sum_j S_ij z_j)                              (B x n x B GEMM, f32 MFMA)
//
// The reference materialises a (B, B, n) tensor for this (9.7 GB at its example batch of 768); the first version here
// formed the differences on the VALU (3 B^2 n operations, two LDS reads per multiply).  The Gram form costs 2 B^2 n on
// v_mfma_f32_16x16x4_f32; its cancellation error is ~ c u (|z_i|^2 + |z_j|^2) / n ABSOLUTE on sim (K split into chunks
// whose partial sums are added in double): harmless for a pair of unrelated cells, whose distance is of the order of the
// norms, but the pairs this loss exists for -- adjacent frames of ONE cell, tm in {1, 2}, weight_matching = 100 in the
// reference's example configuration (config_example.yml:164) -- lie close together, where |a|^2 + |b|^2 - 2ab has
// cancelled most of its bits (a pair at 1e-3 relative distance keeps none).  The reference takes differences first
// (vae.py:441-455).  So the Gram value is a FILTER, as in the VectorQuantizer kernel: a pair whose Gram distance is below
// TM_NEAR x (|z_i|^2 + |z_j|^2) (relative distance under 25 %) is re-evaluated from differences -- sum_d (z_i - z_j)^2
// by the whole wave, fp32 products added in double -- in the epilogue, and its gradient term S_ij (z_i - z_j) is taken
// from differences too (tm_near_backward_kernel) instead of the second GEMM, where rowsum(S) z_i - sum_j S_ij z_j
// cancels the same way.  Every other pair keeps a relative error <= c u / TM_NEAR ~ 3e-6 on sim.
namespace {

constexpr int TM_T = 64;          // 64 x 64 output tile per workgroup: 4 waves, 32 x 32 each (2 x 2 MFMA tiles)
constexpr int TM_KC = 32;         // K staged per step
constexpr int TM_LDA = TM_KC + 2; // LDS row stride of a K-major tile: == 2 (mod 32) banks, conflict-free operand reads

// ---- the sparse form (mode 0 only).  vq_vae.py:331 sums sim * time_matching_mat: a pair with a zero entry adds nothing to the
// loss or to its gradient, and the relation matrix of a batch holds a handful of entries per row (the frames of one
// trajectory).  state[0] = number of nonzero entries of tm (tm_count_kernel); when it is at most TM_SPARSE_ROW per row
// every kernel below reads that and the two GEMMs fall away: the Gram kernel returns, the epilogue takes EVERY related
// pair from differences (the path of the near pairs: exact, as the reference computes them), the gradient kernel writes its
// `add` operand through, and tm_near_backward_kernel adds the pairs' gradient row by row.  The decision is taken on the
// device: nothing for the host to wait for, the same launches in a captured step whatever the matrix holds.
constexpr int TM_SPARSE_ROW = 32;
__device__ __forceinline__ bool tm_sparse(const int *state, int B) { return state && state[0] <= TM_SPARSE_ROW * B; }
// The state block: 4 header words ([0] = the count above) and a map of the far part of S, one int per (panel of 64 rows,
// chunk of 32 columns), nonzero where the block holds a nonzero -- written by the epilogue (plain stores of 1: every writer
// writes the same value, so no read-modify-write on words that a dense S has thousands of workgroups aiming at), read by the
// gradient product, which only multiplies those chunks.  In the z16 / z32 form (mode 1) every pair has a distance to evaluate
// (the hinge on unrelated pairs), but an unrelated pair beyond the margin carries no gradient: S is as sparse as the relation
// matrix again once training has pushed such pairs apart.  A skipped chunk is all zeros: the sums are the same to the bit.
constexpr int TM_STATE_HDR = 4;
__host__ __device__ inline int tm_map_chunks(int B) { return (B + TM_KC - 1) / TM_KC; }
__device__ __forceinline__ void tm_map_set(int *map, int B, int row, int col)
{
    map[(row / TM_T) * tm_map_chunks(B) + col / TM_KC] = 1;
}

__global__ __launch_bounds__(256) void tm_state_clear_kernel(int *__restrict__ state, int ints)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < ints) state[i] = 0;
}

__global__ __launch_bounds__(256) void tm_count_kernel(const float *__restrict__ tm, long long BB, int *__restrict__ state)
{
    __shared__ int s_cnt[4];
    int c = 0;
    if ((BB & 3) == 0) {                                   // 16 bytes per lane and step
        const f32x4 *__restrict__ t4 = reinterpret_cast<const f32x4 *>(tm);
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (BB >> 2); e += (long long)gridDim.x * 256) {
            const f32x4 v = t4[e];
            c += (int)(v.x != 0.f) + (int)(v.y != 0.f) + (int)(v.z != 0.f) + (int)(v.w != 0.f);
        }
    } else {
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < BB; e += (long long)gridDim.x * 256) c += tm[e] != 0.f;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(state, (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]));      // (integers: any order)
}

// P[ks][i][j] = sum_{d in split ks} z[i][d] z[j][d] for the tiles ON AND ABOVE the diagonal: G is symmetric and the epilogue
// reads one orientation of every entry (P[min][max]), so the nt (nt - 1) / 2 tiles below the diagonal are never formed --
// at B = 2048 that is 496 of 1024 workgroups (blockIdx.x walks the upper triangle row by row)
__global__ __launch_bounds__(256) void tm_gram_kernel(const float *__restrict__ z, float *__restrict__ P, int B, int n, int klen,
                                                      int nt, const int *__restrict__ state)
{
    __shared__ float sA[TM_T * TM_LDA], sB[TM_T * TM_LDA];
    if (tm_sparse(state, B)) return;                       // (uniform) the epilogue will not read P
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
    int tcol = blockIdx.x, trow = 0;
    for (int len = nt; tcol >= len; --len) { tcol -= len; ++trow; }    // (uniform: at most nt scalar steps)
    tcol += trow;
    const int i0 = trow * TM_T, j0 = tcol * TM_T;
    const int k_lo = blockIdx.z * klen, k_hi = min(n, k_lo + klen);
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lr = threadIdx.x >> 3, lq = threadIdx.x & 7;     // staging: 8 threads per row, 32 rows per pass
    // register-staged prefetch: the loads of K chunk c + 1 are in flight during the products of chunk c (round 4; the
    // load -> barrier -> products form left every chunk's round trip exposed: 54 TFLOP/s at B = 768, n = 4096)
    f32x4 va[2], vb[2];
    auto issue = [&](int k0) {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = lr + 32 * pass;
            va[pass] = (f32x4){0.f, 0.f, 0.f, 0.f}; vb[pass] = va[pass];
            if (k0 < k_hi && i0 + r < B) va[pass] = *reinterpret_cast<const f32x4 *>(z + (long long)(i0 + r) * n + k0 + 4 * lq);
            if (k0 < k_hi && j0 + r < B) vb[pass] = *reinterpret_cast<const f32x4 *>(z + (long long)(j0 + r) * n + k0 + 4 * lq);
        }
    };
    issue(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += TM_KC) {
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = lr + 32 * pass;
            *reinterpret_cast<f32x2 *>(sA + r * TM_LDA + 4 * lq) = (f32x2){va[pass].x, va[pass].y};
            *reinterpret_cast<f32x2 *>(sA + r * TM_LDA + 4 * lq + 2) = (f32x2){va[pass].z, va[pass].w};
            *reinterpret_cast<f32x2 *>(sB + r * TM_LDA + 4 * lq) = (f32x2){vb[pass].x, vb[pass].y};
            *reinterpret_cast<f32x2 *>(sB + r * TM_LDA + 4 * lq + 2) = (f32x2){vb[pass].z, vb[pass].w};
        }
        __syncthreads();
        issue(k0 + TM_KC);
        const float *pa = sA + (wr * 32 + (lane & 15)) * TM_LDA + (lane >> 4);
        const float *pb = sB + (wc * 32 + (lane & 15)) * TM_LDA + (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < TM_KC; ks += 4) {
            const float a0 = pa[ks], a1 = pa[16 * TM_LDA + ks], b0 = pb[ks], b1 = pb[16 * TM_LDA + ks];
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    float *__restrict__ Pk = P + (long long)blockIdx.z * B * B;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wr * 32 + a * 16 + (lane >> 4) * 4 + r, j = j0 + wc * 32 + b * 16 + (lane & 15);
                if (i < B && j < B) Pk[(long long)i * B + j] = acc[a][b][r];
            }
}

struct TmParams { int mode; float w_a, w_t, w_n, margin; };

__device__ __forceinline__ void tm_value(const TmParams &p, float sim, float tm, float inv_count, double &val, float &dsim)
{
    if (p.mode == 0) { val = (double)(sim * tm); dsim = tm; return; }
    const float w = tm == 2.f ? p.w_a : (tm == 1.f ? p.w_t : (tm == 0.f ? p.w_n : tm));
    float v = sim * w, live = 1.f;
    if (tm == 0.f) {                                      // hinge on the non-related pairs (vae.py:333-335)
        v = v + p.margin;
        live = v >= 0.f ? 1.f : 0.f;
        v = v > 0.f ? v : 0.f;
    }
    val = (double)v * (double)inv_count;
    dsim = w * live * inv_count;
}

constexpr float TM_NEAR = 1.f / 16.f;   // Gram distance below this share of |z_i|^2 + |z_j|^2: re-evaluate from differences

// The 64 x 64 tiles ON OR ABOVE the diagonal (the tiles tm_gram_kernel forms), TM_EP workgroups per tile (16 rows i each),
// one thread per 4 of its pairs (i, j), i <= j: sim from the Gram slabs (near pairs: from differences, by the wave), BOTH
// orientations of the loss term -- the pair's two entries of tm -- and S_ij = S_ji written to both places, the mirrored
// entries through LDS.  (The first form, one thread per entry of the (B, B) matrix, read P and tm across the diagonal with a
// stride of B floats between lanes: 0.22 of the 0.38 ms forward at B = 2048.)  S: (2, B, B), the far part (for the gradient
// GEMM) and the near part (for tm_near_backward_kernel); loss partials per workgroup.
// TM_ER rows i per workgroup = TM_T / TM_ER workgroups per tile: 16 rows, or 4 where 16 would leave most of the chip without a
// workgroup (B <= 640: fewer than 66 tiles)
template <int TM_ER>
__global__ __launch_bounds__(256) void tm_epilogue_kernel(const float *__restrict__ z, const float *__restrict__ P, int ksplit,
                                                          const float *__restrict__ tm, int B, int n, TmParams p,
                                                          float *__restrict__ S, double *__restrict__ loss_slabs, int nt,
                                                          const int *__restrict__ state, int *__restrict__ far_map)
{
    __shared__ float sT[TM_T][TM_ER + 1];                // tm[j0 + r][ib + c]
    __shared__ float sF[TM_ER][TM_T + 1], sN[TM_ER][TM_T + 1];   // far / near part of S by (i - ib, j - j0)
    __shared__ double s_gi[TM_ER], s_gj[TM_T];           // Gram diagonal of the workgroup's rows i and the tile's columns j
    __shared__ double s_red[4];
    int tcol = blockIdx.x, trow = 0;
    for (int len = nt; tcol >= len; --len) { tcol -= len; ++trow; }
    tcol += trow;
    constexpr int TM_EP = TM_T / TM_ER;
    const int ib = trow * TM_T + blockIdx.y * TM_ER, j0 = tcol * TM_T;
    const bool diag = trow == tcol;
    const long long BB = (long long)B * B;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, lane = tx;
#pragma unroll
    for (int k = 0; k < TM_T * TM_ER / 256; ++k) {
        const int e = threadIdx.x + 256 * k, r = e / TM_ER, c = e % TM_ER;
        sT[r][c] = (j0 + r < B && ib + c < B) ? tm[(long long)(j0 + r) * B + ib + c] : 0.f;
    }
    const bool sparse = tm_sparse(state, B);
    if (!sparse && threadIdx.x < TM_ER + TM_T) {         // chunk sums added in double, in a fixed order
        const int g = threadIdx.x < TM_ER ? ib + (int)threadIdx.x : j0 + (int)threadIdx.x - TM_ER;
        double d = 0.0;
        if (g < B)
            for (int ks = 0; ks < ksplit; ++ks) d += (double)P[ks * BB + (long long)g * B + g];
        if (threadIdx.x < TM_ER) s_gi[threadIdx.x] = d; else s_gj[threadIdx.x - TM_ER] = d;
    }
    __syncthreads();
    const float inv_count = p.mode == 0 ? 1.f : 1.f / (float)BB;
    const int j = j0 + tx;
    const double gjj = s_gj[tx];
    double val = 0.0;
    bool any_far = false;
#pragma unroll
    for (int q = 0; q < TM_ER / 4; ++q) {
        const int il = ty + 4 * q, i = ib + il;
        const bool have = i < B && j < B && (!diag || i <= j);
        float sim = 0.f;
        bool near = false;
        if (have && sparse) {                              // every related pair from differences, nothing else matters
            near = i != j && (tm[(long long)i * B + j] != 0.f || sT[tx][il] != 0.f);
        } else if (have) {
            double gij = 0.0;
            for (int ks = 0; ks < ksplit; ++ks) gij += (double)P[ks * BB + (long long)i * B + j];
            const double gii = s_gi[il];
            const double d2 = gii + gjj - 2.0 * gij;
            sim = i == j ? 0.f : (float)(d2 / (double)n);
            near = i != j && !(d2 >= (double)TM_NEAR * (gii + gjj));    // (also when the Gram value is not finite)
        }
        // near pairs, one at a time, by the whole wave: sum of squared differences, 16 bytes per lane and step
        unsigned long long todo = __ballot(near);
        while (todo) {
            const int l = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int pj = j0 + l;                         // (the wave shares i: lane l's pair is (i, j0 + l))
            const float *__restrict__ zi = z + (long long)i * n, *__restrict__ zj = z + (long long)pj * n;
            double acc = 0.0;
            int d = 4 * lane;
            for (; d + 768 < n; d += 1024) {                // four steps' loads in flight together (the sparse form lives here)
                f32x4 a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[u] = *reinterpret_cast<const f32x4 *>(zi + d + 256 * u);
                    b[u] = *reinterpret_cast<const f32x4 *>(zj + d + 256 * u);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {               // (the same terms in the same order as the single steps below)
                    const f32x4 df = a[u] - b[u];
                    acc += (double)((df.x * df.x + df.y * df.y) + (df.z * df.z + df.w * df.w));
                }
            }
            for (; d < n; d += 256) {                      // n % 32 == 0: whole float4s
                const f32x4 a = *reinterpret_cast<const f32x4 *>(zi + d), b = *reinterpret_cast<const f32x4 *>(zj + d);
                const f32x4 df = a - b;
                acc += (double)((df.x * df.x + df.y * df.y) + (df.z * df.z + df.w * df.w));
            }
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            if (lane == l) sim = (float)(acc / (double)n);
        }
        float sij = 0.f;
        if (have) {
            float d_ij, d_ji;
            double v_ij, v_ji;
            tm_value(p, sim, tm[(long long)i * B + j], inv_count, v_ij, d_ij);
            tm_value(p, sim, sT[tx][il], inv_count, v_ji, d_ji);
            val += i == j ? v_ij : v_ij + v_ji;
            sij = d_ij + d_ji;
            S[(long long)i * B + j] = near ? 0.f : sij;
            S[BB + (long long)i * B + j] = near ? sij : 0.f;
        }
        sF[il][tx] = near ? 0.f : sij;
        sN[il][tx] = near ? sij : 0.f;
        any_far |= !near && sij != 0.f;
    }
    if (far_map) {                                         // the chunks of the far part this workgroup put a nonzero into
        const unsigned long long bal = __ballot(any_far);  // (a wave's rows share the panel; its lanes span two chunks of columns)
        if (lane == 0) {
            if (bal & 0xffffffffull) tm_map_set(far_map, B, ib, j0);
            if (bal >> 32) tm_map_set(far_map, B, ib, j0 + 32);
            if (bal) tm_map_set(far_map, B, j0, ib);       // the mirrored entries: rows j0 .. j0 + 63, columns ib .. ib + TM_ER - 1
        }
    }
    __syncthreads();
    // the mirrored entries: S[j][i] = S[i][j] (the loss counts both orientations of a pair, so S is symmetric)
#pragma unroll
    for (int k = 0; k < TM_T * TM_ER / 256; ++k) {
        const int e = threadIdx.x + 256 * k, jl = e / TM_ER, c = e % TM_ER;
        const int jj = j0 + jl, ii = ib + c;               // entry (jj, ii) <- the value at (i = ii, j = jj)
        if (jj < B && ii < B && (!diag || ii < jj)) {
            S[(long long)jj * B + ii] = sF[c][jl];
            S[BB + (long long)jj * B + ii] = sN[c][jl];
        }
    }
    const double tot = block_sum(val, s_red);
    if (threadIdx.x == 0) {
        const long long slab = (long long)blockIdx.x * TM_EP + blockIdx.y;
        loss_slabs[2 * slab] = tot; loss_slabs[2 * slab + 1] = 0.0;
    }
}

// dz[i][d] += scale * g * sum_{j near i} S_ij (z[i][d] - z[j][d]): the near pairs' share of the gradient from differences.
// One workgroup per row i; rows without a near pair (the usual case for most of them) leave after reading their S row.
// The row's nonzero columns are compacted first (thread t owns the columns [t c, (t + 1) c): counts, a scan, the list in
// ascending j -- a fixed order), so the sum walks the row's pairs, not its B columns (in the sparse form EVERY related pair
// comes through here).  Dynamic LDS: B floats (the row) + B 16-bit column numbers.
__global__ __launch_bounds__(256) void tm_near_backward_kernel(const float *__restrict__ z, const float *__restrict__ Snear,
                                                               const float *__restrict__ g_dev, float scale,
                                                               float *__restrict__ dz, int B, int n)
{
    extern __shared__ float s_row[];                       // S_near[i][0..B)
    unsigned short *s_col = reinterpret_cast<unsigned short *>(s_row + B);
    __shared__ int s_cnt[256];
    __shared__ int s_any;
    const int i = blockIdx.x;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    bool any = false;
    for (int j = threadIdx.x; j < B; j += 256) {
        const float v = Snear[(long long)i * B + j];
        s_row[j] = v;
        any |= v != 0.f;
    }
    if (any) s_any = 1;
    __syncthreads();
    if (!s_any) return;
    const int c = (B + 255) / 256, j_lo = threadIdx.x * c, j_hi = min(B, j_lo + c);
    int mine = 0;
    for (int j = j_lo; j < j_hi; ++j) mine += s_row[j] != 0.f;
    s_cnt[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {                     // inclusive scan
        const int v = threadIdx.x >= o ? s_cnt[threadIdx.x - o] : 0;
        __syncthreads();
        s_cnt[threadIdx.x] += v;
        __syncthreads();
    }
    int at = s_cnt[threadIdx.x] - mine;
    const int total = s_cnt[255];
    for (int j = j_lo; j < j_hi; ++j)
        if (s_row[j] != 0.f) s_col[at++] = (unsigned short)j;
    __syncthreads();
    const float sc = scale * (g_dev ? g_dev[0] : 1.f);
    const float *__restrict__ zi = z + (long long)i * n;
    for (int d = 4 * threadIdx.x; d < n; d += 1024) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(zi + d);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < total; ++k) {                   // ascending j: a fixed order
            const int j = s_col[k];
            acc += s_row[j] * (a - *reinterpret_cast<const f32x4 *>(z + (long long)j * n + d));
        }
        // (product and sum rounded separately: what is already in dz may be another gradient of the latents -- the `add` operand
        //  of the call -- and the sum has to come out as the separate elementwise add of the unfused path does)
        f32x4 *o = reinterpret_cast<f32x4 *>(dz + (long long)i * n + d);
        const f32x4 cur = *o;
        *o = (f32x4){__fadd_rn(cur.x, __fmul_rn(sc, acc.x)), __fadd_rn(cur.y, __fmul_rn(sc, acc.y)),
                     __fadd_rn(cur.z, __fmul_rn(sc, acc.z)), __fadd_rn(cur.w, __fmul_rn(sc, acc.w))};
    }
}

// dz[i][d] = scale * g * (rowsum(S)_i z[i][d] - sum_j S_ij z[j][d]);  64 rows x 64 columns per workgroup, K = B
constexpr int TM_LDZ = TM_T + 16;  // LDS row stride of the [j][d] tile: == 16 (mod 32), conflict-free B-operand reads
// add (optional): another gradient of the same latents (the quantiser's), added here instead of in a pass of its own
__global__ __launch_bounds__(256) void tm_backward_kernel(const float *__restrict__ z, const float *__restrict__ S,
                                                          const float *__restrict__ g_dev, float scale,
                                                          float *__restrict__ dz, int B, int n, const float *__restrict__ add,
                                                          const int *__restrict__ state)
{
    __shared__ float sS[TM_T * TM_LDA], sZ[TM_KC * TM_LDZ], s_rs[TM_T], s_part[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
    const int i0 = blockIdx.y * TM_T, d0 = blockIdx.x * TM_T;
    // the chunks of 32 columns of S to multiply: all of them, or (state) those the epilogue marked for this panel of rows, in
    // ascending order either way
    __shared__ unsigned short s_list[512];                 // (B <= 16384)
    __shared__ int s_scan[256];
    const int nchunks = (B + TM_KC - 1) / TM_KC;
    int nl;
    {
        const int *__restrict__ map = state ? state + TM_STATE_HDR + blockIdx.y * nchunks : nullptr;
        auto live = [&](int c) { return c < nchunks && (!map || map[c] != 0); };
        const int c_lo = 2 * threadIdx.x;                  // thread t owns chunks 2t, 2t + 1
        const int mine = (int)live(c_lo) + (int)live(c_lo + 1);
        s_scan[threadIdx.x] = mine;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int v = threadIdx.x >= o ? s_scan[threadIdx.x - o] : 0;
            __syncthreads();
            s_scan[threadIdx.x] += v;
            __syncthreads();
        }
        int at = s_scan[threadIdx.x] - mine;
        if (live(c_lo)) s_list[at++] = (unsigned short)c_lo;
        if (live(c_lo + 1)) s_list[at] = (unsigned short)(c_lo + 1);
        nl = s_scan[255];
        __syncthreads();
    }
    {   // row sums of S for the tile's 64 rows: 4 threads per row, fixed order (a skipped chunk would have added zeros)
        const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
        float a = 0.f;
        if (i0 + r < B) {
            const float *__restrict__ row = S + (long long)(i0 + r) * B;
            for (int li = 0; li < nl; ++li) {
                const int k0 = TM_KC * s_list[li];
                if ((B & 3) == 0) {                          // 16 bytes per load: a quarter of the round trips of this serial prologue
#pragma unroll
                    for (int h = 0; h < TM_KC; h += 16) {
                        const int j = k0 + h + 4 * q;
                        if (j < B) {
                            const f32x4 v = *reinterpret_cast<const f32x4 *>(row + j);
                            a += (v.x + v.y) + (v.z + v.w);
                        }
                    }
                } else {
                    for (int j = k0 + q; j < min(B, k0 + TM_KC); j += 4) a += row[j];
                }
            }
        }
        s_part[threadIdx.x] = a;
        __syncthreads();
        if (q == 0) s_rs[r] = (s_part[threadIdx.x] + s_part[threadIdx.x + 1]) + (s_part[threadIdx.x + 2] + s_part[threadIdx.x + 3]);
        __syncthreads();                                    // (with no chunk to multiply nothing else stands before their use)
    }
    if (nl == 0 && ((((uintptr_t)dz | (uintptr_t)add) & 15) == 0)) {
        // (uniform) nothing marked for these rows -- the sparse form, or a panel of unrelated samples: dz = add + 0 * z, the
        // value the general path below would store (a zero row sum times a finite latent), 16 bytes per lane, z not read
        for (int e = threadIdx.x; e < TM_T * (TM_T / 4); e += 256) {
            const int i = i0 + e / (TM_T / 4), d = d0 + 4 * (e % (TM_T / 4));
            if (i < B && d < n) {
                const long long o = (long long)i * n + d;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (add) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4 *>(add + o);
                    v = (f32x4){__fadd_rn(a4.x, 0.f), __fadd_rn(a4.y, 0.f), __fadd_rn(a4.z, 0.f), __fadd_rn(a4.w, 0.f)};
                }
                *reinterpret_cast<f32x4 *>(dz + o) = v;
            }
        }
        return;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lr = threadIdx.x >> 3, lq = threadIdx.x & 7;     // S tile staging: 8 threads per row (32 columns j)
    const int zr = threadIdx.x >> 4, zq = threadIdx.x & 15;    // Z tile staging: 16 threads per row (64 columns d)
    // register-staged prefetch of the next K chunk, as in tm_gram_kernel
    float sv[2][4];
    f32x4 zv[2];
    const bool s_vec = (B & 3) == 0;                           // rows of S are 16-byte aligned: one load instead of four
    auto issue = [&](int k0) {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = lr + 32 * pass;
            if (s_vec) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if (i0 + r < B && k0 + 4 * lq < B) t = *reinterpret_cast<const f32x4 *>(S + (long long)(i0 + r) * B + k0 + 4 * lq);
                sv[pass][0] = t.x; sv[pass][1] = t.y; sv[pass][2] = t.z; sv[pass][3] = t.w;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = k0 + 4 * lq + u;
                    sv[pass][u] = (i0 + r < B && j < B) ? S[(long long)(i0 + r) * B + j] : 0.f;
                }
            }
            const int jr = zr + 16 * pass;
            zv[pass] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (k0 + jr < B && d0 + 4 * zq < n) zv[pass] = *reinterpret_cast<const f32x4 *>(z + (long long)(k0 + jr) * n + d0 + 4 * zq);
        }
    };
    if (nl > 0) issue(TM_KC * s_list[0]);
    for (int li = 0; li < nl; ++li) {
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = lr + 32 * pass;
            *reinterpret_cast<f32x2 *>(sS + r * TM_LDA + 4 * lq) = (f32x2){sv[pass][0], sv[pass][1]};
            *reinterpret_cast<f32x2 *>(sS + r * TM_LDA + 4 * lq + 2) = (f32x2){sv[pass][2], sv[pass][3]};
            *reinterpret_cast<f32x4 *>(sZ + (zr + 16 * pass) * TM_LDZ + 4 * zq) = zv[pass];
        }
        __syncthreads();
        issue(li + 1 < nl ? TM_KC * s_list[li + 1] : B);    // (past the end: every lane's guard fails, nothing is loaded)
        const float *pa = sS + (wr * 32 + (lane & 15)) * TM_LDA + (lane >> 4);
        const float *pb = sZ + (lane >> 4) * TM_LDZ + wc * 32 + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < TM_KC; ks += 4) {
            const float a0 = pa[ks], a1 = pa[16 * TM_LDA + ks], b0 = pb[ks * TM_LDZ], b1 = pb[ks * TM_LDZ + 16];
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    const float sc = scale * (g_dev ? g_dev[0] : 1.f);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int il = wr * 32 + a * 16 + (lane >> 4) * 4 + r, i = i0 + il, d = d0 + wc * 32 + b * 16 + (lane & 15);
                if (i < B && d < n) {
                    const long long o = (long long)i * n + d;
                    const float t = sc * (s_rs[il] * z[o] - acc[a][b][r]);
                    dz[o] = add ? __fadd_rn(add[o], t) : t;            // (rounded like the separate elementwise add it replaces)
                }
            }
}

int tm_ksplit(int B, int n)
{
    const int nt = (B + TM_T - 1) / TM_T, tiles = nt * (nt + 1) / 2;         // the upper triangle of tiles
    int ks = 1024 / tiles;
    const int maxks = n / 256 > 0 ? n / 256 : 1;
    if (ks > maxks) ks = maxks;
    if (ks > 64) ks = 64;
    if (ks < 1) ks = 1;
    return ks;
}

}  // namespace

extern "C" int dm_time_matching_supported(int B, int n) { return (B > 0 && n > 0 && n % TM_KC == 0) ? 1 : 0; }

extern "C" int64_t dm_time_matching_workspace_floats(int B, int n)
{
    // Gram slabs + (double) loss partials of the epilogue
    const long long blocks = ((long long)B * B + 255) / 256;
    return (long long)tm_ksplit(B, n) * B * B + 4 * blocks + 4;
}

static int tm_epilogue_rows(int B)                        // rows per workgroup of tm_epilogue_kernel
{
    const int nt = (B + TM_T - 1) / TM_T;
    return 4 * (nt * (nt + 1) / 2) >= 256 ? 16 : 4;
}

extern "C" int dm_time_matching_state_ints(int B)         // the state block of the _state entries
{
    return TM_STATE_HDR + ((B + TM_T - 1) / TM_T) * tm_map_chunks(B);
}

extern "C" int dm_time_matching_num_slabs(int B)          // one per epilogue workgroup: TM_T / rows per tile on or above the diagonal
{
    const int nt = (B + TM_T - 1) / TM_T;
    return (TM_T / tm_epilogue_rows(B)) * (nt * (nt + 1) / 2);
}

static int tm_forward_launch(const float *z, const float *tm, int B, int n, int mode, float w_a, float w_t, float w_n,
                             float margin, float *workspace, int64_t workspace_floats, float *S, double *loss_slabs,
                             int32_t *state, void *stream);

extern "C" int dm_time_matching_forward(const float *z, const float *tm, int B, int n, int mode, float w_a, float w_t,
                                        float w_n, float margin, float *workspace, int64_t workspace_floats, float *S,
                                        double *loss_slabs, void *stream)
{
    return tm_forward_launch(z, tm, B, n, mode, w_a, w_t, w_n, margin, workspace, workspace_floats, S, loss_slabs, nullptr, stream);
}

extern "C" int dm_time_matching_forward_state(const float *z, const float *tm, int B, int n, int mode, float w_a, float w_t,
                                              float w_n, float margin, float *workspace, int64_t workspace_floats, float *S,
                                              double *loss_slabs, int32_t *state, void *stream)
{
    DM_REQUIRE(state, "dm_time_matching_forward_state: NULL pointer");
    return tm_forward_launch(z, tm, B, n, mode, w_a, w_t, w_n, margin, workspace, workspace_floats, S, loss_slabs, state, stream);
}

static int tm_forward_launch(const float *z, const float *tm, int B, int n, int mode, float w_a, float w_t, float w_n,
                             float margin, float *workspace, int64_t workspace_floats, float *S, double *loss_slabs,
                             int32_t *state, void *stream)
{
    DM_REQUIRE(z && tm && workspace && S && loss_slabs, "dm_time_matching_forward: NULL pointer");
    DM_REQUIRE(dm_time_matching_supported(B, n), "dm_time_matching_forward: latent length %d is not a multiple of %d", n, TM_KC);
    DM_REQUIRE(mode == 0 || mode == 1, "dm_time_matching_forward: mode %d", mode);
    DM_REQUIRE((long long)B * n < (1LL << 31) && (long long)B * B < (1LL << 31), "dm_time_matching_forward: tensor too large");
    DM_REQUIRE(workspace_floats >= dm_time_matching_workspace_floats(B, n), "dm_time_matching_forward: workspace too small");
    // (tm_count_kernel and the epilogue read tm 16 bytes at a time where B * B is a multiple of four)
    DM_REQUIRE((((uintptr_t)tm | (uintptr_t)z | (uintptr_t)S) & 15) == 0, "dm_time_matching_forward: z, tm and S must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int ks = tm_ksplit(B, n), nt = (B + TM_T - 1) / TM_T;
    int klen = (n + ks - 1) / ks;
    klen = (klen + TM_KC - 1) / TM_KC * TM_KC;
    // the sparse form: mode 0 only (in mode 1 the unrelated pairs carry the hinge term), and only where the caller keeps a state
    // word for the backward call to read; the count always starts from zero
    const int *st = nullptr;
    int *far_map = nullptr;
    if (state) {
        // (cleared by a kernel of this stream, ordered like every other launch of the call)
        const int ints = dm_time_matching_state_ints(B);
        hipLaunchKernelGGL(tm_state_clear_kernel, dim3((ints + 255) / 256), dim3(256), 0, s, (int *)state, ints);
        far_map = (int *)state + TM_STATE_HDR;
        if (mode == 0) {
            const long long BB = (long long)B * B;
            const long long units = (BB & 3) == 0 ? BB >> 2 : BB;
            const int grid = (int)((units + 255) / 256 < 2048 ? (units + 255) / 256 : 2048);
            hipLaunchKernelGGL(tm_count_kernel, dim3(grid), dim3(256), 0, s, tm, BB, (int *)state);
            st = (const int *)state;
        }
    }
    hipLaunchKernelGGL(tm_gram_kernel, dim3(nt * (nt + 1) / 2, 1, ks), dim3(256), 0, s, z, workspace, B, n, klen, nt, st);
    const TmParams p{mode, w_a, w_t, w_n, margin};
    if (tm_epilogue_rows(B) == 16)
        hipLaunchKernelGGL(tm_epilogue_kernel<16>, dim3(nt * (nt + 1) / 2, TM_T / 16), dim3(256), 0, s, z, workspace, ks, tm, B, n,
                           p, S, loss_slabs, nt, st, far_map);
    else
        hipLaunchKernelGGL(tm_epilogue_kernel<4>, dim3(nt * (nt + 1) / 2, TM_T / 4), dim3(256), 0, s, z, workspace, ks, tm, B, n,
                           p, S, loss_slabs, nt, st, far_map);
    return dm_launch_status("dm_time_matching_forward");
}

static int tm_backward_launch(const float *z, const float *S, const float *g_loss_dev, float scale, const float *add,
                              float *dz, int B, int n, const int32_t *state, void *stream);

extern "C" int dm_time_matching_backward(const float *z, const float *S, const float *g_loss_dev, float scale, float *dz,
                                         int B, int n, void *stream)
{
    return tm_backward_launch(z, S, g_loss_dev, scale, nullptr, dz, B, n, nullptr, stream);
}

extern "C" int dm_time_matching_backward_add(const float *z, const float *S, const float *g_loss_dev, float scale,
                                             const float *add, float *dz, int B, int n, void *stream)
{
    DM_REQUIRE(add, "dm_time_matching_backward_add: NULL pointer");
    return tm_backward_launch(z, S, g_loss_dev, scale, add, dz, B, n, nullptr, stream);
}

extern "C" int dm_time_matching_backward_state(const float *z, const float *S, const float *g_loss_dev, float scale,
                                               const float *add, float *dz, int B, int n, const int32_t *state, void *stream)
{
    DM_REQUIRE(state, "dm_time_matching_backward_state: NULL pointer");
    return tm_backward_launch(z, S, g_loss_dev, scale, add, dz, B, n, state, stream);
}

static int tm_backward_launch(const float *z, const float *S, const float *g_loss_dev, float scale, const float *add,
                              float *dz, int B, int n, const int32_t *state, void *stream)
{
    DM_REQUIRE(z && S && dz, "dm_time_matching_backward: NULL pointer");
    DM_REQUIRE(dm_time_matching_supported(B, n), "dm_time_matching_backward: latent length %d is not a multiple of %d", n, TM_KC);
    DM_REQUIRE((long long)B * n < (1LL << 31), "dm_time_matching_backward: tensor too large");
    DM_REQUIRE(B <= 16384, "dm_time_matching_backward: batch %d too large (a row of S is staged in LDS)", B);
    // (rows of S are read as 16-byte vectors where B is a multiple of four; z tiles always are)
    DM_REQUIRE((((uintptr_t)S | (uintptr_t)z) & 15) == 0, "dm_time_matching_backward: z and S must be 16-byte aligned");
    const int *st = (const int *)state;
    hipLaunchKernelGGL(tm_backward_kernel, dim3((n + TM_T - 1) / TM_T, (B + TM_T - 1) / TM_T), dim3(256), 0, (hipStream_t)stream,
                       z, S, g_loss_dev, scale * 2.f / (float)n, dz, B, n, add, st);
    const size_t near_lds = (size_t)B * (sizeof(float) + sizeof(unsigned short));
    if (near_lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void *)tm_near_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)near_lds);
        if (e != hipSuccess) {
            dm_set_error("dm_time_matching_backward: cannot reserve %zu bytes of LDS: %s", near_lds, hipGetErrorString(e));
            return (int)e;
        }
    }
    hipLaunchKernelGGL(tm_near_backward_kernel, dim3((unsigned)B), dim3(256), near_lds, (hipStream_t)stream, z,
                       S + (long long)B * B, g_loss_dev, scale * 2.f / (float)n, dz, B, n);
    return dm_launch_status("dm_time_matching_backward");
}
