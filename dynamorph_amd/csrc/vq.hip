// vq.hip -- VectorQuantizer kernels (reference: HiddenStateExtractor/vq_vae.py:52-116).
//
// Compiled with -ffp-contract=off: the reference materialises (z - e) and (z - e)^2 in
// fp32 before summing over d (no FMA), and ATen's CPU reduction adds the squares
// sequentially inside blocks of 16 consecutive d, block sums again sequentially.  The
// kernels reproduce that order so argmin indices are bit-identical to the CPU path.
//
// Layout: one thread per latent position (b,h,w); the D-vector of a position is strided
// by H*W in NCHW, so for every d a wave reads 64 consecutive floats (256 B, coalesced)
// and keeps its D values in registers.  Codes are processed two at a time as packed fp32
// (v_pk_add_f32 / v_pk_mul_f32, IEEE round-to-nearest, same results as scalar ops); the
// codebook is first re-laid out as [K/2][D][2] so a pair's operands are adjacent, staged in
// LDS per workgroup and read with wave-uniform (broadcast) ds_read_b128.
#include "dm_common.h"
#include <type_traits>

namespace {

constexpr int VQ_BLOCK = 256;
constexpr int VQ_MAX_LDS_HIST = 4096;

// Workspace of dm_vq_forward (float offsets; every region starts on a 16-byte boundary):
//   header  32 ints: [0] = positions that went through the exact re-check of the MFMA path (statistics); [4..] = per-phase
//           cycle sums of the diagnostic build (-DVQ2_STAMPS, never the shipped one)
//   cbT     [ceil(K/2)][D][2]   pair-interleaved codebook of the exact kernel (v1)
//   cbA     [K64/64][4 kt][SQ][64 lanes][4]  A operand of v_mfma_f32_16x16x4_f32: -2 * e[64 cc + 16 kt + (lane & 15)]
//           [4 (4 sq + j) + (lane >> 4)], zero beyond K / D  (K64 = K rounded up to 64, SQ = ceil(D / 16))
//   nrm     [K64]  ||e_k||^2 summed in double, rounded once; VQ2_PAD_NORM beyond K (such a code never wins)
//   cbB     [K64/64][4 kt][U][2 (hi, lo)][64 lanes][4 x u32]  A operand of v_mfma_f32_16x16x32_bf16 (the bf16-split filter,
//           U = D / 16 K-groups): the bf16 head (resp. remainder) of -2 * e[64 cc + 16 kt + (lane & 15)][4 (4 u + j) + (lane >> 4)],
//           j = 0..3, packed in pairs and repeated, so that the 8 k-slots of a lane group multiply (z_hi[0..3], z_lo[0..3])
//   cbH     [K][4 h][D / 4]  e[k][4 s + h]: the codebook row in the order a lane of the MFMA kernel holds z
//   hrep    [R][K] int32  replicated code-usage counters (vq_hist_replicas), or -- codebooks of at most 64 codes with
//           embedding_dim 16/32/64 through the MFMA kernel, which then needs no preparation launch at all -- one row of
//           VQ2_SLAB_STRIDE ints per WORKGROUP: its 64 counters and, at [64], the positions it re-evaluated exactly
//           (plain stores: nothing to zero, no atomics).  header[1] = rows in use, header[2] = row stride, header[3] = row
//           format (1: per-workgroup rows, 0: replicas -- an explicit flag: a codebook of exactly VQ2_SLAB_STRIDE codes has
//           that stride on the replica path too), written by whichever kernel filled the counters; the finalisers read
//           them on the device
constexpr int VQ2_HDR = 32;
constexpr int VQ2_SLAB_ROWS = 1024, VQ2_SLAB_STRIDE = 72;
constexpr float VQ2_PAD_NORM = 3.0e38f;
struct Vq2Layout { long long cbT, cbA, cbB, nrm, cbH, hrep, cbP, nrmP, total; int R; };
// (cbP, nrmP: the permuted bf16 operand and norms of vq_cells_kernel -- vq_cells.h -- for 64 < K <= 4096 at embedding_dim 16)

// Code-usage counters: every workgroup flushing its LDS histogram into ONE set of K global counters serialises
// (workgroups x K atomics on K addresses: 8 us of a 28 us kernel at 512 workgroups, K = 64).  The workgroups add into
// R replicas (replica = workgroup % R) and a tiny kernel sums the replicas into `hist`.
int vq_hist_replicas(int K)
{
    int r = 65536 / (K > 0 ? K : 1);
    return r < 1 ? 1 : (r > 64 ? 64 : r);
}

Vq2Layout vq2_layout(int K, int D)
{
    const long long K64 = ((long long)K + 63) / 64 * 64;
    const int SQ = (D / 4 + 3) / 4;
    Vq2Layout L;
    long long o = VQ2_HDR;
    L.cbT = o; o += (long long)((K + 1) / 2) * 2 * D; o = (o + 3) & ~3LL;
    L.cbA = o; o += K64 * 16 * SQ;
    L.cbB = o; o += (D % 16 == 0 && D <= 64) ? K64 * 2 * D : 0;
    L.nrm = o; o += K64;
    L.cbH = o; o += (long long)K * D; o = (o + 3) & ~3LL;
    L.R = vq_hist_replicas(K);
    {
        const long long rep = (long long)L.R * K, slab = K <= 64 ? (long long)VQ2_SLAB_ROWS * VQ2_SLAB_STRIDE : 0;
        L.hrep = o; o += rep > slab ? rep : slab; o = (o + 3) & ~3LL;
    }
    {
        const long long K128 = ((long long)K + 127) / 128 * 128;
        const bool cells = D == 16 && K > 64 && K <= 4096;
        L.cbP = o; o += cells ? K128 * 16 : 0;
        L.nrmP = o; o += cells ? K128 : 0;
    }
    L.total = o;
    return L;
}

// float -> bf16 bits, round to nearest even, exactly what v_cvt_pk_bf16_f32 does for finite values (a NaN stays a NaN, an
// overflow becomes inf: either makes the filter's tolerance non-finite and sends the position down the exact path)
__device__ __forceinline__ unsigned vq_bf16_rne(float v)
{
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t r = __builtin_convertvector((f32x2){v, 0.f}, bf16x2_t);
    return __builtin_bit_cast(unsigned, r) & 0xffffu;
}

// ---- large codebooks at embedding_dim 16: operand layout of vq_cells_kernel (vq_cells.h) ----
constexpr int VQC_NT = 4;                       // position tiles of 32 per wave
constexpr int VQC_GCH = 4;                      // chunks of 32 codes per group
constexpr int VQC_GROUP = 32 * VQC_GCH;         // 128 codes
constexpr int VQC_MAX_K = 4096;                 // norms and group minima are sized for this in LDS

// row m of chunk cc of the permuted operand <-> code: register r = 4 j + i of lane half hf holds row 8 j + 4 hf + i; the
// cell (j, hf, i >> 1) of group g owns the codes g * 128 + 8 * cell .. + 7 = (chunk in group) * 2 + (i & 1)
__host__ __device__ __forceinline__ int vqc_code_of(int cc, int m)
{
    const int g = cc / VQC_GCH, ccl = cc % VQC_GCH, j = m >> 3, hf = (m >> 2) & 1, i = m & 3;
    const int cell = (j * 2 + hf) * 2 + (i >> 1);
    return g * VQC_GROUP + cell * 8 + ccl * 2 + (i & 1);
}

typedef unsigned vqc_u32x4 __attribute__((ext_vector_type(4)));
typedef float vqc_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 vqc_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 vqc_bf16x2 __attribute__((ext_vector_type(2)));

// (cbP, nrmP) of vq_prep_kernel for this kernel: cbP[cc][hi | lo][lane][4 x u32] = the 8 bf16 (heads resp. remainders of
// -2 e) of dimensions 8 (lane >> 5) .. + 7 of row lane & 31; nrmP[cc * 32 + m] = |e|^2 of that row (VQ2_PAD_NORM beyond K)
__device__ __forceinline__ void vqc_prep(const float *__restrict__ cb, unsigned *__restrict__ cbP, float *__restrict__ nrmP,
                                         int K, int t0, int nt)
{
    const int K128 = (K + VQC_GROUP - 1) / VQC_GROUP * VQC_GROUP, NCH = K128 / 32;
    for (int i = t0; i < NCH * 512; i += nt) {
        const int q = i & 3, l = (i >> 2) & 63, hl = (i >> 8) & 1, cc = i >> 9;
        const int code = vqc_code_of(cc, l & 31);
        unsigned u = 0;
        for (int e = 0; e < 2; ++e) {
            const int d = 8 * (l >> 5) + 2 * q + e;
            const float a = code < K ? -2.f * cb[(long long)code * 16 + d] : 0.f;
            const unsigned hi = vq_bf16_rne(a);
            const float rem = a - __builtin_bit_cast(float, hi << 16);
            u |= (hl ? vq_bf16_rne(rem) : hi) << (16 * e);
        }
        cbP[i] = u;
    }
    for (int i = t0; i < K128; i += nt) {
        const int code = vqc_code_of(i >> 5, i & 31);
        double acc = 0.0;
        if (code < K)
            for (int d = 0; d < 16; ++d) { const double e = (double)cb[(long long)code * 16 + d]; acc += e * e; }
        nrmP[i] = code < K ? (float)acc : VQ2_PAD_NORM;
    }
}

__global__ void vq_prep_kernel(const float *__restrict__ cb, float *__restrict__ ws, Vq2Layout L, int K, int D,
                               double *__restrict__ slabs, int nslabs)
{
    // also clears the outputs the forward kernel accumulates into (no separate memset launches)
    const int t0 = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    int *__restrict__ hrep = reinterpret_cast<int *>(ws + L.hrep);
    for (int i = t0; i < L.R * K; i += nt) hrep[i] = 0;
    for (int i = t0; i < nslabs; i += nt) slabs[i] = 0.0;
    if (t0 < VQ2_HDR) reinterpret_cast<int *>(ws)[t0] = t0 == 1 ? L.R : (t0 == 2 ? K : 0);
    // cbT[p][d][j] = cb[2p + j][d]; for odd K the missing partner repeats code K-1 (never selected).
    float *__restrict__ cbT = ws + L.cbT;
    const int npairs = (K + 1) >> 1;
    for (int i = t0; i < npairs * D * 2; i += nt) {
        const int j = i & 1, d = (i >> 1) % D, p = (i >> 1) / D;
        int k = 2 * p + j;
        if (k >= K) k = K - 1;
        cbT[i] = cb[(long long)k * D + d];
    }
    if (D & 3) return;                                     // the MFMA path needs D % 4 == 0
    const int S = D / 4, SQ = (S + 3) / 4;
    const long long K64 = ((long long)K + 63) / 64 * 64;
    float *__restrict__ cbA = ws + L.cbA;
    for (long long i = t0; i < K64 * 16 * SQ; i += nt) {
        const int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
        long long rest = i >> 8;
        const int sq = (int)(rest % SQ); rest /= SQ;
        const int kt = (int)(rest & 3);
        const long long cc = rest >> 2;
        const long long code = cc * 64 + kt * 16 + (lane & 15);
        const int d = 4 * (4 * sq + j) + (lane >> 4);
        cbA[i] = (code < K && d < D) ? -2.f * cb[code * D + d] : 0.f;
    }
    if (D % 16 == 0 && D <= 64) {
        // bf16-split A operand: -2e = hi + lo + r with hi = bf16(-2e), lo = bf16(-2e - hi), |r| <= 2^-17 |2e|
        unsigned *__restrict__ cbB = reinterpret_cast<unsigned *>(ws + L.cbB);
        const int U = D / 16;
        for (long long i = t0; i < K64 * 2 * D / 2; i += nt) {           // one packed pair of k-slots (j = 2 jp, 2 jp + 1) per step
            const int jp = (int)(i & 1), lane = (int)((i >> 1) & 63), half = (int)((i >> 7) & 1);
            long long rest = i >> 8;
            const int u = (int)(rest % U); rest /= U;
            const int kt = (int)(rest & 3);
            const long long cc = rest >> 2;
            const long long code = cc * 64 + kt * 16 + (lane & 15);
            unsigned pair = 0;
            for (int q = 0; q < 2; ++q) {
                const int d = 4 * (4 * u + 2 * jp + q) + (lane >> 4);
                const float a = code < K ? -2.f * cb[code * D + d] : 0.f;
                const unsigned hi = vq_bf16_rne(a);
                const float rem = a - __builtin_bit_cast(float, hi << 16);
                const unsigned part = half ? vq_bf16_rne(rem) : hi;
                pair |= part << (16 * q);
            }
            // u32x4 of a lane = (pair 0, pair 1, pair 0, pair 1): slots 0..3 meet z_hi, slots 4..7 meet z_lo
            const long long base = ((((cc * 4 + kt) * U + u) * 2 + half) * 64 + lane) * 4;
            cbB[base + jp] = pair;
            cbB[base + 2 + jp] = pair;
        }
    }
    float *__restrict__ nrm = ws + L.nrm;
    for (long long k = t0; k < K64; k += nt) {
        double acc = 0.0;
        if (k < K)
            for (int d = 0; d < D; ++d) { const double e = (double)cb[k * D + d]; acc += e * e; }
        nrm[k] = k < K ? (float)acc : VQ2_PAD_NORM;
    }
    float *__restrict__ cbH = ws + L.cbH;
    for (long long i = t0; i < (long long)K * D; i += nt) {
        const int s = (int)(i % S), h = (int)((i / S) & 3);
        const long long k = i / D;
        cbH[i] = cb[k * D + 4 * s + h];
    }
    if (L.nrmP > L.cbP) vqc_prep(cb, reinterpret_cast<unsigned *>(ws + L.cbP), ws + L.nrmP, K, t0, nt);
}

// first-minimum with torch.argmax(-dist) NaN semantics: a NaN distance beats any number,
// the first NaN wins (vq_vae.py:68).
__device__ __forceinline__ bool vq_better(float cand, float best)
{
    // best is a number AND (cand is NaN OR cand < best); bitwise so that no branch splits the loop body
    return (best == best) & !(cand >= best);
}

// PP positions per lane (pos, pos + 256, ...): every codebook operand fetched from LDS is used PP times.
// The pair-interleaved codebook is staged in LDS in chunks of CHUNK_PAIRS pairs; all lanes of a wave read the
// same address (broadcast ds_read_b128, conflict free), and because DS reads return in order hipcc can keep the
// next pair's operands in flight behind a counted lgkmcnt while the current pair's packed math runs -- scalar
// (s_load) operands cannot be pipelined that way (SMEM returns out of order: every wait is lgkmcnt(0)), which
// left the first version of this kernel 58 % parked on s_waitcnt.
template <int D, int PP>
__global__ __launch_bounds__(VQ_BLOCK, (D <= 16 ? 4 : (D <= 64 ? 2 : 1))) void vq_forward_kernel(
    const float *__restrict__ z, const float *__restrict__ cb, const float *__restrict__ cbT,
    long long *__restrict__ idx, float *__restrict__ out, double *__restrict__ sse_slabs,
    int *__restrict__ hrep, int R, int K, int HW, long long P)
{
    int *__restrict__ hist = hrep + (long long)(blockIdx.x % (unsigned)R) * K;     // this workgroup's replica of the counters
    constexpr int CHUNK_PAIRS = 2048 / D;                 // 16 KB of LDS per chunk
    __shared__ __attribute__((aligned(16))) float s_cb[CHUNK_PAIRS * D * 2];
    __shared__ int s_hist[VQ_MAX_LDS_HIST];
    __shared__ double s_red[4];
    const bool lds_hist = K <= VQ_MAX_LDS_HIST;
    if (lds_hist)
        for (int k = threadIdx.x; k < K; k += VQ_BLOCK) s_hist[k] = 0;

    long long pos[PP], base[PP];
    bool active[PP];
    float zr[PP][D];
#pragma unroll
    for (int q = 0; q < PP; ++q) {
        pos[q] = ((long long)blockIdx.x * PP + q) * VQ_BLOCK + threadIdx.x;
        active[q] = pos[q] < P;
        const long long b = active[q] ? pos[q] / HW : 0;
        const long long p = active[q] ? pos[q] - b * HW : 0;
        base[q] = b * (long long)D * HW + p;
#pragma unroll
        for (int d = 0; d < D; ++d) zr[q][d] = active[q] ? z[base[q] + (long long)d * HW] : 0.f;
    }

    float bestd[PP];
    int bi[PP];
#pragma unroll
    for (int q = 0; q < PP; ++q) { bestd[q] = __builtin_inff(); bi[q] = 0; }
    const int npairs = K >> 1;
    for (int c0 = 0; c0 < npairs; c0 += CHUNK_PAIRS) {
        const int cn = min(CHUNK_PAIRS, npairs - c0);
        __syncthreads();                                   // previous chunk consumed (and s_hist zeroed)
        for (int i = threadIdx.x; i < cn * D * 2 / 4; i += VQ_BLOCK)
            reinterpret_cast<f32x4 *>(s_cb)[i] = reinterpret_cast<const f32x4 *>(cbT + (long long)c0 * D * 2)[i];
        __syncthreads();
        for (int kp = 0; kp < cn; ++kp) {
            const float *e = s_cb + kp * (2 * D);          // wave-uniform LDS address: broadcast reads
            f32x2 total[PP];
#pragma unroll
            for (int d0 = 0; d0 < D; d0 += 16) {
                f32x2 acc[PP];
#pragma unroll
                for (int q = 0; q < PP; ++q) acc[q] = (f32x2){0.f, 0.f};
                // the additions of a block are a sequential chain (ATen's order); the subtractions and squares are
                // not, so they are formed eight d at a time ahead of the chain to keep independent work in the pipe
#pragma unroll
                for (int h = d0; h < d0 + 16 && h < D; h += 8) {
                    // (the empty asm statements pin each stage: hipcc otherwise sinks every sub and mul next to its add
                    // and runs sub -> mul -> add per d on two registers, each instruction waiting for the previous one)
                    f32x2 sq[PP][8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const f32x2 e2 = *reinterpret_cast<const f32x2 *>(e + 2 * (h + j));
#pragma unroll
                        for (int q = 0; q < PP; ++q) {
                            const f32x2 zz = {zr[q][h + j], zr[q][h + j]};
                            sq[q][j] = zz - e2;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int q = 0; q < PP; ++q) asm volatile("" : "+v"(sq[q][j]));
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int q = 0; q < PP; ++q) sq[q][j] = sq[q][j] * sq[q][j];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int q = 0; q < PP; ++q) asm volatile("" : "+v"(sq[q][j]));
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int q = 0; q < PP; ++q)
                            acc[q] = (h == d0 && j == 0) ? sq[q][0] : acc[q] + sq[q][j];   // 0 + s == s for a square (never -0)
                }
#pragma unroll
                for (int q = 0; q < PP; ++q) total[q] = (d0 == 0) ? acc[q] : total[q] + acc[q];
            }
            const int k0 = 2 * (c0 + kp);
#pragma unroll
            for (int q = 0; q < PP; ++q) {
                // bestd starts at +inf with index 0: code 0 wins unless a later code is strictly smaller (or NaN)
                const bool b0 = vq_better(total[q].x, bestd[q]);
                bestd[q] = b0 ? total[q].x : bestd[q]; bi[q] = b0 ? k0 : bi[q];
                const bool b1 = vq_better(total[q].y, bestd[q]);
                bestd[q] = b1 ? total[q].y : bestd[q]; bi[q] = b1 ? k0 + 1 : bi[q];
            }
        }
    }
    if (K & 1) {   // odd tail, scalar
        const int k = K - 1;
        const float *__restrict__ e = cb + (long long)k * D;
#pragma unroll
        for (int q = 0; q < PP; ++q) {
            float total = 0.f;
#pragma unroll
            for (int d0 = 0; d0 < D; d0 += 16) {
                float acc = 0.f;
#pragma unroll
                for (int d = d0; d < d0 + 16 && d < D; ++d) {
                    const float diff = zr[q][d] - e[d];
                    acc = acc + diff * diff;
                }
                total = (d0 == 0) ? acc : total + acc;
            }
            const bool bt = vq_better(total, bestd[q]);
            bestd[q] = bt ? total : bestd[q]; bi[q] = bt ? k : bi[q];
        }
    }

    if (npairs == 0) __syncthreads();                      // s_hist zeroing visible even when the pair loop is empty
    double sse = 0.0;
#pragma unroll
    for (int q = 0; q < PP; ++q) {
        if (active[q]) {
            if (idx) idx[pos[q]] = (long long)bi[q];
            const float *__restrict__ qv = cb + (long long)bi[q] * D;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float diff = qv[d] - zr[q][d];
                if (out) out[base[q] + (long long)d * HW] = zr[q][d] + diff;     // z + (q - z), vq_vae.py:71
                const float sq = diff * diff;
                sse += (double)sq;
            }
            if (lds_hist) atomicAdd(&s_hist[bi[q]], 1);
            else atomicAdd(&hist[bi[q]], 1);
        }
    }
    const double tot = block_sum(sse, s_red);
    if (threadIdx.x == 0) sse_slabs[blockIdx.x] = tot;
    if (lds_hist) {
        __syncthreads();
        for (int k = threadIdx.x; k < K; k += VQ_BLOCK) {
            const int c = s_hist[k];
            if (c) atomicAdd(&hist[k], c);
        }
    }
}

// Any embedding_dim (the reference's VectorQuantizer takes any width, vq_vae.py:35-50; num_hiddens = 24 / 48 / 96 ...): the
// widths without a register-resident instantiation.  One wave per workgroup, a position per lane, the lane's z in LDS as
// s_z[d][lane] (conflict-free; D x 256 bytes, dynamic), the code rows through wave-uniform (scalar) loads.  The reference's
// arithmetic and order exactly as vq_forward_kernel: (z - e), its square, sums of 16 consecutive d, block sums in order,
// first minimum with the argmax(-dist) NaN rule.  A coverage path: 3 K D vector operations per position, no matrix cores.
__global__ __launch_bounds__(64) void vq_forward_any_kernel(
    const float *__restrict__ z, const float *__restrict__ cb, long long *__restrict__ idx, float *__restrict__ out,
    double *__restrict__ sse_slabs, int *__restrict__ hrep, int R, int K, int D, int HW, long long P)
{
    extern __shared__ float s_zany[];                       // [D][64]
    int *__restrict__ hist = hrep + (long long)(blockIdx.x % (unsigned)R) * K;
    const int lane = threadIdx.x;
    double sse = 0.0;
    for (long long p0 = (long long)blockIdx.x * 64; p0 < P; p0 += (long long)gridDim.x * 64) {
        const long long pos = p0 + lane;
        const bool active = pos < P;
        const long long b = active ? pos / HW : 0, pp = active ? pos - b * HW : 0;
        const long long base = b * (long long)D * HW + pp;
        for (int d = 0; d < D; ++d) s_zany[d * 64 + lane] = active ? z[base + (long long)d * HW] : 0.f;
        float bestd = __builtin_inff();
        int bi = 0;
        for (int k = 0; k < K; ++k) {
            const float *__restrict__ e = cb + (long long)k * D;
            float total = 0.f;
            for (int d0 = 0; d0 < D; d0 += 16) {
                float acc = 0.f;
                const int d1 = d0 + 16 < D ? d0 + 16 : D;
                for (int d = d0; d < d1; ++d) {
                    const float diff = s_zany[d * 64 + lane] - e[d];
                    const float sq = diff * diff;
                    acc = (d == d0) ? sq : acc + sq;
                }
                total = (d0 == 0) ? acc : total + acc;
            }
            const bool bt = vq_better(total, bestd);
            bestd = bt ? total : bestd; bi = bt ? k : bi;
        }
        if (active) {
            if (idx) idx[pos] = (long long)bi;
            const float *__restrict__ qv = cb + (long long)bi * D;
            for (int d = 0; d < D; ++d) {
                const float zv = s_zany[d * 64 + lane];
                const float diff = qv[d] - zv;
                if (out) out[base + (long long)d * HW] = zv + diff;     // z + (q - z), vq_vae.py:71
                sse += (double)(diff * diff);
            }
            atomicAdd(&hist[bi], 1);
        }
    }
    const double tot = wave_sum(sse);
    if (lane == 0) sse_slabs[blockIdx.x] = tot;
}

// Sum of column k over the counter rows (replicas or per-workgroup slabs; rows and stride from the workspace header).
__device__ __forceinline__ int vq_count_column(const int *__restrict__ hrep, int R, int stride, int k)
{
    int s = 0;
    for (int r0 = 0; r0 < R; r0 += 32) {
        // (a batch of rows requested before the first is used: one memory round trip per 32 rows)
        int v[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) v[r] = r0 + r < R ? hrep[(long long)(r0 + r) * stride + k] : 0;
#pragma unroll
        for (int r = 0; r < 32; ++r) s += v[r];
    }
    return s;
}

// hist[k] = column sum over the counter rows.  One workgroup per 64 codes: its sixteen waves take every sixteenth row (64
// codes side by side = one coalesced 256-byte read per row), 32 rows in flight per wave.
__global__ __launch_bounds__(1024) void vq_hist_reduce_kernel(const int *__restrict__ hrep, int *__restrict__ hdr, int K,
                                                              int *__restrict__ hist)
{
    __shared__ int s_cnt[16][65];
    const int R = hdr[1], stride = hdr[2];
    const int kl = threadIdx.x & 63, g = threadIdx.x >> 6, k = blockIdx.x * 64 + kl;
    const bool slabs = hdr[3] == 1;                        // per-workgroup rows: column 64 = positions re-evaluated exactly
    int h = 0, n = 0;
    for (int r0 = g; r0 < R; r0 += 512) {
        int v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = (r0 + 16 * j < R && k < K) ? hrep[(long long)(r0 + 16 * j) * stride + k] : 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) h += v[j];
        if (slabs && kl == 0 && blockIdx.x == 0) {
#pragma unroll
            for (int j = 0; j < 32; ++j) n += r0 + 16 * j < R ? hrep[(long long)(r0 + 16 * j) * stride + 64] : 0;
        }
    }
    s_cnt[g][kl] = h;
    if (kl == 0) s_cnt[g][64] = n;
    __syncthreads();
    if (g == 0) {
        int th = 0, tn = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { th += s_cnt[w][kl]; tn += s_cnt[w][64]; }
        if (k < K) hist[k] = th;
        if (slabs && kl == 0 && blockIdx.x == 0) hdr[0] = tn;
    }
}

// ================================================================================================================
// v2: MFMA prefilter + exact re-check -- same indices as the kernel above, a third of its instructions.
//
// The exact distance costs 3*K*D non-fused VALU operations per position.  argmin_k |z - e_k|^2 = argmin_k S(k) with
// S(k) = |e_k|^2 - 2 z.e_k, a (positions x D) . (D x K) product: v_mfma_f32_16x16x4_f32 with the codes on the rows
// (A = -2 e, exact), the positions on the columns (B = z) and |e_k|^2 as the initial accumulator gives
// s(k) = fl-chain(S(k)) at 2*K*D FLOP on the matrix pipe.  The result is only an approximation of the reference's
// d_ref(k) = sum_d fl(fl(z_d - e_kd)^2) (summed in ATen's order), so it is used as a FILTER:
//   * per position the two smallest s' are tracked (s' = s with its low 5 bits replaced by the code's number -- 4 bits
//     inside the lane's 16 scores, 1 more in the first merge step -- so the minimum carries its own index: v_and_or,
//     v_med3, v_min per score);
//   * error bounds, u = 2^-24, A = |z|^2 + 2 max_k |e_k|^2:
//       |s'(k) - S(k)| <= (D + 1) u A   (rounded |e|^2 + D fused multiply-adds, every partial sum <= A in magnitude)
//                         + 64 u A      (at most 31 ulp of index bits, 1 ulp <= 2 u |s'|)   =: eta = (D + 65) u A
//       |d_ref(k) - D(k)| <= (D/16 + 18) u D(k) =: rho D(k),  D(k) = |z|^2 + S(k)  (3 roundings per square, then at
//                                                              most 15 + D/16 - 1 additions of non-negative terms);
//     with g' = s'(j) - s'(k*) the reference orders d_ref(k*) < d_ref(j) whenever
//       g' > [2 eta + 2 rho (|z|^2 + s'(k*)) + 2 rho eta] / (1 - rho);
//     the kernel tests g' > tol with tol = 1.25 x (2 eta + 2 rho max(|z|^2 + s'(k*), 0)) = TOL_A A + TOL_D max(..., 0)
//     (TOL_A = 2.5 (D + 65) u, TOL_D = 2.5 (D/16 + 18) u in the kernel: a margin of 1.25 over the first-order bound --
//     the neglected terms 2 rho eta and the 1/(1 - rho) are O(u) relative to it -- |z|^2 from an fp32 dot product) for
//     the runner-up, which then holds for every other code;
//   * a position that fails the test (1e-3 of them on N(0,1) data; every position whose z or codebook is not finite,
//     because tol is then inf or NaN) is re-evaluated EXACTLY by the whole wave: lane l takes codes l, l + 64, ... in the
//     reference's arithmetic and order, first minimum and torch.argmax(-dist) NaN rule included.  Codebooks of more than
//     64 codes (round 3): only over the GROUPS of codes whose smallest filter score lies within tol of the best one -- the
//     bound that clears the runner-up clears, code by code, every group whose minimum is farther away (s_pm in the kernel).
//     4096 codes: 16 groups of 256 (one LDS buffer pair each; until round 5 six groups of 768, the LDS going to a 4096-entry
//     counter array that large codebooks now keep in their global replicas): the re-checks of the 0.5-0.8 % of positions
//     that fail went from 19 % of the kernel to 7 % on N(0,1) data (tools/exp/vq_parts.sh).
// So the index is the reference's for every position; only the work per position differs.
//
// Layout: a wave owns a chunk of 64 consecutive positions of one sample; lane (h = lane >> 4, c = lane & 15) loads
// z[d = 4 s + h][4 c .. 4 c + 3] as one 16-byte load per s (16 lanes x 16 B = 256 B contiguous per row), which is
// exactly the B operand of step s for the four position tiles t = 0..3 (column c of tile t = position 4 c + t): no
// transpose, no LDS.  The 16 x 16 result tile has code 16 kt + 4 h + r in register r, so a position's 64 scores sit in
// 16 registers of 4 lanes: 16 in-lane updates, then two cross-lane steps (lanes l ^ 16, l ^ 32).

// Diagnostic build only (make STAMPS=1): s_memtime at the phase boundaries of the chunk loop, per-wave sums added into the
// workspace header.  The shipped library has no stamp (cdna_hip_programming.md section 7, In-kernel stamps).
#ifdef VQ2_STAMPS
#define VQ2_STAMP(i)                                                                                      \
    {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        unsigned long long t_;                                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        st_sum[i] += t_ - st_prev;                                                                        \
        st_prev = t_;                                                                                     \
    }
#define VQ2_USE(v) asm volatile("; use" ::"v"(v))
#else
#define VQ2_STAMP(i)
#define VQ2_USE(v)
#endif

#ifdef DM_MEASURE
// measurement build only (make measure; DM_VQ_DBG): 1 the codebook piece is staged once (stale operands), 4 no exact re-checks;
// compile time: -DVQ_DBG_NOEMBED no index bits in the scores, -DVQ_DBG_MINONLY the in-lane minimum only (no runner-up).
// Results are then wrong; the time is what is read (tools/exp/vq_parts.sh).
__device__ int vq_dbg_dev;
#define VQ_DBG(bit) (vq_dbg_dev & (bit))
#else
#define VQ_DBG(bit) false
#endif

// 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4: no register in between; the LDS image is
// lane-linear: wave-uniform base + 16 * lane, which is what a contiguous copy wants).  Completion is counted on vmcnt.
__device__ __forceinline__ void vq2_glds16(const void *g, void *l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

__device__ __forceinline__ float vq2_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vq2_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vq2_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vq2_med3(float a, float b, float c) { float r; asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vq2_embed(float v, unsigned code)
{
    return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, v) & ~63u) | code);
}
__device__ __forceinline__ float vq2_or(float v, unsigned bits)
{
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) | bits);
}

// Exchange across the lane pairs (l, l ^ 16) resp. (l, l ^ 32) as ONE VALU instruction (v_permlane16_swap /
// v_permlane32_swap, gfx950) instead of two ds_bpermute round trips through the LDS crossbar: with both operands = x
// the swap leaves the even rows' (lower half's) value in `lo` and the odd rows' (upper half's) in `hi`, in BOTH lanes of
// a pair, so a symmetric combine of (lo, hi) gives every lane the pair's result.
template <int M>
__device__ __forceinline__ void vq2_pair(unsigned x, unsigned &lo, unsigned &hi)
{
    if constexpr (M == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        const unsigned r0 = r[0], r1 = r[1];
        lo = r0; hi = r1;
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        const unsigned r0 = r[0], r1 = r[1];
        lo = r0; hi = r1;
    }
}
template <int M>
__device__ __forceinline__ void vq2_pair(float x, float &lo, float &hi)
{
    unsigned l, h;
    vq2_pair<M>(__builtin_bit_cast(unsigned, x), l, h);
    lo = __builtin_bit_cast(float, l); hi = __builtin_bit_cast(float, h);
}

// exact distance of the reference (vq_vae.py:65): blocks of 16 consecutive d summed sequentially, block sums sequentially
template <int D>
__device__ __forceinline__ float vq_exact_dist(const float (&zv)[D], const float *__restrict__ e)
{
    float total = 0.f;
#pragma unroll
    for (int d0 = 0; d0 < D; d0 += 16) {
        float acc = 0.f;
#pragma unroll
        for (int d = d0; d < d0 + 16 && d < D; ++d) {
            const float diff = zv[d] - e[d];
            const float sq = diff * diff;
            acc = (d == d0) ? sq : acc + sq;
        }
        total = (d0 == 0) ? acc : total + acc;
    }
    return total;
}

// Smallest and second smallest of 16 values in 20 instructions (a chain of v_med3 / v_min updates takes 32): triples give
// (min3, med3); the overall minimum is the minimum of the triple minima, the runner-up is either the second smallest of
// the triple minima (it sits in another triple) or the median of the winner's triple -- and every other triple median or
// triple minimum is some element other than the winner, hence no smaller than the runner-up: m2 = min of both kinds.
__device__ __forceinline__ void vq2_top2_16(const float (&v)[16], float &m1, float &m2)
{
    float n[5], d[5];
#pragma unroll
    for (int g = 0; g < 5; ++g) {
        n[g] = vq2_min3(v[3 * g], v[3 * g + 1], v[3 * g + 2]);
        d[g] = vq2_med3(v[3 * g], v[3 * g + 1], v[3 * g + 2]);
    }
    const float p0 = vq2_min3(n[0], n[1], n[2]), q0 = vq2_med3(n[0], n[1], n[2]);
    const float p1 = vq2_min3(n[3], n[4], v[15]), q1 = vq2_med3(n[3], n[4], v[15]);
    m1 = vq2_min(p0, p1);
    const float second_of_minima = vq2_min3(vq2_max(p0, p1), q0, q1);
    const float least_median = vq2_min3(vq2_min3(d[0], d[1], d[2]), d[3], d[4]);
    m2 = vq2_min(second_of_minima, least_median);
}

// lane-pair exchange of TWO values in one instruction: after vq2_swap<16>(x, y) the even 16-lane rows hold (own x,
// partner's x) and the odd rows (partner's y, own y); in both, x is the even row's value and y the odd row's.  <32>: the
// same for the 32-lane halves.
template <int M>
__device__ __forceinline__ void vq2_swap(float &x, float &y)
{
    unsigned lo, hi;
    if constexpr (M == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
        const unsigned r0 = r[0], r1 = r[1];
        lo = r0; hi = r1;
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
        const unsigned r0 = r[0], r1 = r[1];
        lo = r0; hi = r1;
    }
    x = __builtin_bit_cast(float, lo); y = __builtin_bit_cast(float, hi);
}

#include "vq_cells.h"

// The kernel.  One 256-thread workgroup = 4 waves = 4 chunks of 64 positions per iteration, persistent over the chunks.
// Per chunk and wave: 64 MFMAs (the scores of 64 codes x 64 positions), the in-lane top 2 of every lane's 16 scores per
// position tile, a reduce-scatter over the four lanes that share a position (afterwards lane (h, c) OWNS position tile
// T(h) = {0, 2, 1, 3}[h], column c: its two best scores, |z|^2, the tolerance test, the re-check request), the exact
// re-checks, an all-gather of the four chosen codes, the gather / straight-through value / squared error, the stores.
// On gfx950 the f32 MFMA and the VALU do not overlap (SQ_VALU_MFMA_COEXEC_CYCLES = 0: the kernel's time is the SUM of its
// matrix and vector instructions), so what is left to optimise is the instruction count of everything around the MFMAs.
// BF: the filter product on v_mfma_f32_16x16x32_bf16 with both operands split into a bf16 head and a bf16 remainder
// (z = z_hi + z_lo + r, |r| <= 2^-17 |z|; likewise -2e, split once by vq_prep_kernel).  The 32 k-slots of one instruction
// are (z_hi, z_lo) of 16 dimensions; two instructions per 16 dimensions (A = the heads of -2e, then its remainders) give
// (a_hi + a_lo) . (z_hi + z_lo): a quarter of the matrix-pipe cycles of four v_mfma_f32_16x16x4_f32 steps (2 x 16 against
// 4 x 32), on the real matrix cores -- the f32-input instruction runs at the vector rate and (SQ_VALU_MFMA_COEXEC_CYCLES
// = 0) never beside vector instructions.  Price: 12 vector instructions per position tile to split z, and a wider
// tolerance (below), i.e. more positions on the exact path.  cbA is then the cbB region of the workspace.
// INL (codebooks of at most 64 codes, embedding_dim 16/32/64): no preparation launch and no counter reduction -- the
// workgroup builds its A operands, norms and gather rows from the raw codebook in its prologue (cbA / nrm / cbH are then
// unused), writes its code counters as one row of plain stores (hrep = the slab region, R ignored) and zeroes the squared-
// error slabs no workgroup owns.  dm_vq_forward is ONE launch for every configuration of the reference.
// JOIN (with INL): the encoder's last residual join runs in the load path -- z = fma(c0, rb, c2) + h_in per channel
// (BatchNorm of the block's last convolution applied to its raw output rb, plus the block's input: dm_apply's arithmetic,
// ResidualBlock.forward vq_vae.py:222-224) -- and the kernel writes z (`jz`: the latents the backward pass and the callers
// need) itself: `z` is then rb, `jh` the block input, `jcoef` the [D][4] coefficient table of dm_bn_finalize.  One launch
// (dm_apply) and one round trip of the latents fewer per step.
template <int D, bool SINGLE, int MINW, bool BF, bool INL, bool JOIN = false>
__global__ __launch_bounds__(256, MINW) void vq_forward_mfma_kernel(
    const float *__restrict__ z, const float *__restrict__ cb, const float *__restrict__ cbA,
    const float *__restrict__ nrm, const float *__restrict__ cbH, long long *__restrict__ idx,
    float *__restrict__ out, double *__restrict__ sse_slabs, int *__restrict__ hrep, int R, int *__restrict__ hdr,
    int K, int HW, long long P, int nslabs, const float *__restrict__ jh = nullptr, const float *__restrict__ jcoef = nullptr,
    float *__restrict__ jz = nullptr)
{
    constexpr int BLOCK = 256, NW = 4;
    static_assert(!INL || (SINGLE && D % 16 == 0), "inline preparation: <= 64 codes, embedding_dim 16 / 32 / 64");
    static_assert(!JOIN || INL, "the fused residual join is built for the one-launch form");
    // this workgroup's replica of the counters (INL: its own row)
    int *__restrict__ hist = INL ? hrep + (long long)blockIdx.x * VQ2_SLAB_STRIDE : hrep + (long long)(blockIdx.x % (unsigned)R) * K;
    constexpr int S = D / 4, SQ = (S + 3) / 4;
    constexpr int UG = BF ? D / 16 : 1;                            // K-groups of 16 dimensions (bf16-split filter)
    static_assert(!BF || D % 16 == 0, "bf16-split filter: embedding_dim 16, 32 or 64");
    constexpr int AQ = BF ? 2 * UG : SQ;                           // 16-byte A operands per lane and code tile
    constexpr int CHUNK_F4 = 4 * AQ * 64;                          // 16-byte units of packed A operand per 64-code chunk
    // Large codebooks stream their packed A operands through TWO LDS buffers of PCH code chunks (16 KB each; one chunk each
    // where a chunk is larger): the copy of the next buffer (global_load_lds, no registers) runs under the products of the
    // current one, ONE barrier per buffer, and the stream wraps around from one chunk of positions to the next.  Up to round 5
    // a single 32 KB piece was refilled through registers between two barriers, its load latency exposed 16 times per 256
    // positions at 4096 codes: 12 % of the kernel (tools/exp/vq_parts.sh).  A codebook that fits both buffers is staged once.
    constexpr int PCH = SINGLE ? 1 : (1024 / CHUNK_F4 > 0 ? 1024 / CHUNK_F4 : 1);     // code chunks per LDS buffer
    constexpr int PIECE = 2 * PCH;
    constexpr float U = 5.9604645e-8f;                             // 2^-24
    // the header comment's bound: tol = 1.25 x (2 eta + 2 rho max(|z|^2 + s', 0)), eta = ETA u A, rho = (D/16 + 18) u.
    // f32 filter: ETA = D + 65 (D fused multiply-adds + the rounded norm, 64 of index bits).
    // bf16-split filter: ETA = 66 (norm, index bits)
    //   + 256: the split's remainders.  bf16 keeps 8 significand bits (unit roundoff 2^-8): |z - z_hi| <= 2^-8 2^e for z in
    //          [2^e, 2^(e+1)), that remainder lies in a binade at or below 2^(e-9) (or is exactly 2^(e-8), which bf16 holds),
    //          so |z - z_hi - z_lo| <= 2^-8 2^(e-9) <= 2^-17 |z| (attained: z = 1 + 2^-9 (2 - 2^-8)); hence
    //          |sum_d a z - sum_d (a_hi + a_lo)(z_hi + z_lo)| <= (2^-17 + 2^-17 + 2^-34) sum |a||z|
    //          <= 2^-16 (|e|^2 + |z|^2) <= 2^-16 A = 256 u A  (the products of bf16 values are exact in fp32)
    //   + 100 per matrix instruction: its 32 products + C are added in an order and with intermediate roundings the ISA does
    //          not specify; every partial sum is <= 1.01 A in magnitude, so 33 additions that each lose at most one ulp
    //          (2 u relative: truncation) stay below 67 u A -- taken as 100.
    constexpr float ETA = BF ? 66.f + 256.f + 100.f * (2 * UG) : (float)(D + 65);
    constexpr float TOL_A = 2.5f * ETA * U, TOL_D = 2.5f * (D / 16 + 18) * U;
    __shared__ f32x4 s_A[SINGLE ? 1 : PIECE * CHUNK_F4];
    __shared__ f32x4 s_n[SINGLE ? 1 : PIECE * 16];
    constexpr int HROW = S + 1;                                       // f32x4 per code in s_H: D floats + 16 bytes of padding (banks)
    __shared__ f32x4 s_H[SINGLE ? 64 * HROW : 1];                    // small codebooks: the lane-ordered rows (cbH) for the gather
    // code counters of this workgroup: in LDS up to VQ2_MAX_LDS_HIST codes; larger codebooks add straight into the
    // workgroup's global replica (one wave-wide atomic per 64 positions on scattered addresses, and no K-entry flush at the
    // end) -- their LDS goes to the group minima below
    constexpr int VQ2_MAX_LDS_HIST = 1024;
    __shared__ int s_hist[SINGLE ? 64 : VQ2_MAX_LDS_HIST];
    __shared__ double s_red[NW];
    __shared__ float s_em[NW];
    __shared__ __attribute__((aligned(16))) float s_nrm[INL ? 64 : 4];
    // Large codebooks: the smallest filter score of every GROUP of codes, per position (bf16, rounded down), so that a
    // position that fails the tolerance test is re-evaluated exactly only against the groups that can hold a code within the
    // tolerance of its best score -- the same proven bound that clears the runner-up clears every code of a group whose
    // minimum is farther away.  Groups are whole LDS pieces; their number is what fits beside the piece at this occupancy.
    constexpr int NG = SINGLE ? 1 : (MINW >= 3 ? (BF ? 16 : 8) : 16);     // (3 workgroups per CU: 53 KB each, 512-byte granules)
    __shared__ unsigned short s_pm[SINGLE ? 2 : NG * BLOCK];
    const int lane = threadIdx.x & 63, h = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // provably wave-uniform: scalar branches
    const bool lds_hist = K <= (SINGLE ? 64 : VQ2_MAX_LDS_HIST);

    // Chunk bookkeeping in 32-bit scalars, advanced incrementally (a 64-bit `pos / HW` per iteration is a ~150-instruction
    // software division in front of the loads whose address it feeds).  All global traffic of the loop goes through
    // buffer descriptors rebased per sample: the per-lane offset is a constant, the per-chunk part a scalar offset.
    const unsigned NC = (unsigned)(P >> 6);                    // chunks of 64 positions (HW % 64 == 0: never across samples)
    const unsigned cps = (unsigned)HW >> 6;                    // chunks per sample
    const unsigned qstep = (unsigned)NW * gridDim.x;
    const unsigned step_b = qstep / cps, step_c = qstep - step_b * cps;
    const int ncc = (K + 63) >> 6;                             // 64-code chunks
    const int gch = ((ncc + NG - 1) / NG + PCH - 1) / PCH * PCH;             // code chunks per group (whole buffers)
    const int nhp = (ncc + PCH - 1) / PCH;                     // buffers' worth of code chunks in the codebook
    const bool resident = nhp <= 2;                            // (uniform) the whole codebook sits in the two buffers
    unsigned hp_it = 0;                                        // buffers consumed so far (parity = the buffer in use)
    // code chunks [hn * PCH, ...) into buffer b, asynchronously (whole waves, or the first lanes of wave 0: the LDS image of a
    // global_load_lds is M0 = the first active lane's address + 16 * lane)
    auto vq2_stage = [&](int hn, int b) {
        if constexpr (!SINGLE) {
            const int c0 = hn * PCH, n = min(PCH, ncc - c0);
            const f32x4 *__restrict__ ga = reinterpret_cast<const f32x4 *>(cbA) + (long long)c0 * CHUNK_F4;
            for (int i = threadIdx.x; i < n * CHUNK_F4; i += BLOCK) vq2_glds16(ga + i, s_A + b * PCH * CHUNK_F4 + i);
            const f32x4 *__restrict__ gn = reinterpret_cast<const f32x4 *>(nrm) + (long long)c0 * 16;
            for (int i = threadIdx.x; i < n * 16; i += BLOCK) vq2_glds16(gn + i, s_n + b * PCH * 16 + i);
        }
    };
    double sse = 0.0;
    int nflag = 0;
#ifdef VQ2_STAMPS
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
    const unsigned sample_bytes = (unsigned)D * (unsigned)HW * 4u;
    const unsigned zvoff = ((unsigned)h * (unsigned)HW + 4u * (unsigned)c) * 4u;     // this lane's bytes inside a chunk's rows
    // (a wave without a chunk -- the tail of the last quad -- loads the last chunk and computes nothing)
    auto z_rsrc = [&](unsigned chunk, unsigned b) {
        const unsigned bb = chunk < NC ? b : NC / cps - 1;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(z) + (long long)bb * D * HW, 0, sample_bytes, 0x00020000);
    };
    auto z_soff = [&](unsigned chunk, unsigned cw) { return (chunk < NC ? cw : cps - 1) * 256u; };
    auto z_load = [&](f32x4 (&dst)[S], f32x4 (&dsth)[JOIN ? S : 1], unsigned chunk, unsigned b, unsigned cw) {
        const __amdgpu_buffer_rsrc_t r = z_rsrc(chunk, b);
        const unsigned so = z_soff(chunk, cw);
#pragma unroll
        for (int s = 0; s < S; ++s)
            dst[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, zvoff, so + (unsigned)(4 * s) * (unsigned)HW * 4u, 0));
        if constexpr (JOIN) {
            const unsigned bb = chunk < NC ? b : NC / cps - 1;
            const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(jh) + (long long)bb * D * HW, 0,
                                                                                sample_bytes, 0x00020000);
#pragma unroll
            for (int s = 0; s < S; ++s)
                dsth[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rh, zvoff, so + (unsigned)(4 * s) * (unsigned)HW * 4u, 0));
        }
    };
    // JOIN: z = fma(c0, rb, c2) + h_in for the lane's channels 4 s + h, written to jz for the chunk it belongs to
    // (the (c0, c2) pairs of the D channels wait in LDS: eight more live registers would spill at three waves per SIMD)
    __shared__ __attribute__((aligned(8))) float s_jc[JOIN ? 2 * D : 2];
    if constexpr (JOIN) {
        if (threadIdx.x < D) { s_jc[2 * threadIdx.x] = jcoef[threadIdx.x * 4]; s_jc[2 * threadIdx.x + 1] = jcoef[threadIdx.x * 4 + 2]; }
    }
    auto z_join_store = [&](f32x4 (&zv)[S], const f32x4 (&hv)[JOIN ? S : 1], bool live, unsigned b, unsigned cw) {
        if constexpr (JOIN) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const f32x2 cf = *reinterpret_cast<const f32x2 *>(s_jc + 2 * (4 * s + h));
                f32x4 v;
                v.x = __builtin_fmaf(cf.x, zv[s].x, cf.y) + hv[s].x; v.y = __builtin_fmaf(cf.x, zv[s].y, cf.y) + hv[s].y;
                v.z = __builtin_fmaf(cf.x, zv[s].z, cf.y) + hv[s].z; v.w = __builtin_fmaf(cf.x, zv[s].w, cf.y) + hv[s].w;
                zv[s] = v;
            }
            if (live) {
                const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(jz + (long long)b * D * HW, 0, sample_bytes, 0x00020000);
#pragma unroll
                for (int s = 0; s < S; ++s)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, zv[s]),
                                                           rz, zvoff, cw * 256u + (unsigned)(4 * s) * (unsigned)HW * 4u, 0);
            }
        }
    };
    unsigned chunk = blockIdx.x * (unsigned)NW + (unsigned)wave;
    unsigned cb_ = chunk / cps, cw_ = chunk - cb_ * cps;       // sample and chunk-in-sample of `chunk`
    f32x4 zr[S], hq[JOIN ? S : 1];
    z_load(zr, hq, chunk, cb_, cw_);
    // every global load of the prologue is issued before the first wait: z of the first chunk, the A operand and the
    // norms (small codebooks: registers), the rows for the gather -- one memory round trip instead of four
    f32x4 areg[4][AQ], nreg[4];
    constexpr int HCOPY = SINGLE ? (64 * S + BLOCK - 1) / BLOCK : 1;
    f32x4 hreg[HCOPY];
    if constexpr (SINGLE) {
        static_assert(S % 4 == 0 || S == 2, "embedding_dim 8, 16, 32 or 64");
        if constexpr (!INL) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
                for (int sq = 0; sq < AQ; ++sq) areg[kt][sq] = reinterpret_cast<const f32x4 *>(cbA)[(kt * AQ + sq) * 64 + lane];
                nreg[kt] = *reinterpret_cast<const f32x4 *>(nrm + kt * 16 + h * 4);
            }
        }
#pragma unroll
        for (int j = 0; j < HCOPY; ++j) {
            const int i = threadIdx.x + j * BLOCK;
            if constexpr (INL) {
                // 16 bytes (code k, row hh, dimensions 4 (4 s4 + 0..3) + hh) of the gather image, straight from the codebook
                const int k = i / S, q = i - k * S, hh = q / (S / 4), s4 = q - hh * (S / 4);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < K) {
                    const float *__restrict__ e = cb + (long long)k * D + 16 * s4 + hh;
                    v = (f32x4){e[0], e[4], e[8], e[12]};
                }
                hreg[j] = v;
            } else {
                hreg[j] = i < K * S ? reinterpret_cast<const f32x4 *>(cbH)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    if (lds_hist)
        for (int k = threadIdx.x; k < (SINGLE ? 64 : K); k += BLOCK) s_hist[k] = 0;

    // max_k |e_k|^2 for the tolerance; a non-finite codebook makes it inf: every position takes the exact path
    float emax;
    // small codebooks: the 64 norms are in the wave's registers (16 per lane): in-lane maximum, two lane-pair steps
    auto emax_of_nreg = [&]() {
        float em = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = nreg[kt][r];
                const bool real = kt * 16 + h * 4 + r < K;         // (padding carries VQ2_PAD_NORM)
                const float vv = v < __builtin_inff() ? v : __builtin_inff();    // NaN -> inf
                em = real ? fmaxf(em, vv) : em;
            }
        float lo, hi;
        vq2_pair<16>(em, lo, hi); em = fmaxf(lo, hi);
        vq2_pair<32>(em, lo, hi);
        return fmaxf(lo, hi);
    };
    if constexpr (SINGLE) {
        if constexpr (!INL) emax = emax_of_nreg();
#pragma unroll
        for (int j = 0; j < HCOPY; ++j) {
            const int i = threadIdx.x + j * BLOCK;
            if (INL ? i < 64 * S : i < K * S) s_H[(i / S) * HROW + i % S] = hreg[j];      // (INL: rows beyond K are zeros)
        }
        __syncthreads();
        if constexpr (INL) {
            // |e_k|^2 in double, rounded once (as vq_prep_kernel does), by the first wave; then every lane's operands
            if (threadIdx.x < 64) {
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < S; ++q) {
                    const f32x4 v = s_H[threadIdx.x * HROW + q];
                    acc += (double)v.x * (double)v.x + (double)v.y * (double)v.y + (double)v.z * (double)v.z + (double)v.w * (double)v.w;
                }
                s_nrm[threadIdx.x] = (int)threadIdx.x < K ? (float)acc : VQ2_PAD_NORM;
            }
            __syncthreads();
            typedef __bf16 vq_bf16x2p __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                nreg[kt] = *reinterpret_cast<const f32x4 *>(s_nrm + kt * 16 + h * 4);
#pragma unroll
                for (int u = 0; u < S / 4; ++u) {
                    // dimensions 4 (4 u + j) + h, j = 0..3, of code 16 kt + c: exactly one 16-byte unit of the gather image
                    const f32x4 a = -2.f * s_H[(kt * 16 + c) * HROW + h * (S / 4) + u];
                    if constexpr (BF) {
                        const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a.x, a.y}, vq_bf16x2p));
                        const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a.z, a.w}, vq_bf16x2p));
                        const float r0 = a.x - __builtin_bit_cast(float, h01 << 16), r1 = a.y - __builtin_bit_cast(float, h01 & 0xffff0000u);
                        const float r2 = a.z - __builtin_bit_cast(float, h23 << 16), r3 = a.w - __builtin_bit_cast(float, h23 & 0xffff0000u);
                        const unsigned l01 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r0, r1}, vq_bf16x2p));
                        const unsigned l23 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r2, r3}, vq_bf16x2p));
                        typedef unsigned vq_u32x4p __attribute__((ext_vector_type(4)));
                        areg[kt][2 * u] = __builtin_bit_cast(f32x4, (vq_u32x4p){h01, h23, h01, h23});
                        areg[kt][2 * u + 1] = __builtin_bit_cast(f32x4, (vq_u32x4p){l01, l23, l01, l23});
                    } else {
                        areg[kt][u] = a;
                    }
                }
            }
            emax = emax_of_nreg();
        }
        // every prologue load is waited for HERE: left pending, hipcc's counted waits for them at the first MFMAs of the
        // loop body would also hold every later iteration until its predecessor's stores have completed
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int sq = 0; sq < AQ; ++sq) asm volatile("" ::"v"(areg[kt][sq]));
            asm volatile("" ::"v"(nreg[kt]));
        }
    } else {
        // buffer 0 (and 1, where the codebook fits the two) on its way; the barrier below waits for it
        vq2_stage(0, 0);
        if (nhp == 2) vq2_stage(1, 1);
        float em = 0.f;
        bool bad = false;
        for (int k = threadIdx.x; k < K; k += BLOCK) {
            const float v = nrm[k];
            bad |= !(v < __builtin_inff());
            em = fmaxf(em, v);
        }
        em = bad ? __builtin_inff() : em;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) em = fmaxf(em, __shfl_xor(em, o, 64));
        if (lane == 0) s_em[wave] = em;
        __syncthreads();
        emax = s_em[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) emax = fmaxf(emax, s_em[w]);
    }

    // the position tile this lane owns after the reduce-scatter, and constants of the stores
    const unsigned ovoff = zvoff;                                  // `out` has z's layout
    const unsigned ivoff = (4u * (unsigned)c + 2u * (unsigned)h) * 8u;      // int64 indices of positions 4c + 2h, + 1 (lanes h < 2)

    z_join_store(zr, hq, chunk < NC, cb_, cw_);                   // (JOIN) the first chunk's latents

    for (unsigned q = blockIdx.x; q * (unsigned)NW < NC; q += gridDim.x) {
        const bool act = chunk < NC;                           // wave-uniform
        unsigned nchunk = chunk + qstep, nb = cb_ + step_b, nw = cw_ + step_c;
        if (nw >= cps) { nw -= cps; ++nb; }
        // the next chunk's z is requested now and lands under this chunk's MFMAs
        f32x4 zn[S];
        z_load(zn, hq, nchunk, nb, nw);

        VQ2_STAMP(1)                                           // prefetch issue
        float m1[4], m2[4], pmv[SINGLE ? 1 : 4];
        int c1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { m1[t] = 3.4028235e38f; m2[t] = 3.4028235e38f; c1[t] = 0; }
        if constexpr (!SINGLE) {
#pragma unroll
            for (int t = 0; t < 4; ++t) pmv[t] = 3.4028235e38f;
        }

        // bf16-split filter: the B operands of the chunk's four position tiles, (z_hi[0..3], z_lo[0..3]) of the lane's four
        // dimensions of K-group u as four packed pairs -- built once per chunk of positions, used by every code chunk
        typedef unsigned vq_u32x4 __attribute__((ext_vector_type(4)));
        typedef __bf16 vq_bf16x8 __attribute__((ext_vector_type(8)));
        typedef __bf16 vq_bf16x2 __attribute__((ext_vector_type(2)));
        vq_u32x4 bq[BF ? 4 : 1][UG];
        if constexpr (BF) {
            if (act) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < UG; ++u) {
                        const float z0 = zr[4 * u][t], z1 = zr[4 * u + 1][t], z2 = zr[4 * u + 2][t], z3 = zr[4 * u + 3][t];
                        const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){z0, z1}, vq_bf16x2));
                        const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){z2, z3}, vq_bf16x2));
                        const float r0 = z0 - __builtin_bit_cast(float, h01 << 16), r1 = z1 - __builtin_bit_cast(float, h01 & 0xffff0000u);
                        const float r2 = z2 - __builtin_bit_cast(float, h23 << 16), r3 = z3 - __builtin_bit_cast(float, h23 & 0xffff0000u);
                        const unsigned l01 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r0, r1}, vq_bf16x2));
                        const unsigned l23 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r2, r3}, vq_bf16x2));
                        bq[t][u] = (vq_u32x4){h01, h23, l01, l23};
                    }
            }
        }

        // scores of one 64-code chunk against the 4 position tiles; na: the chunk's |e|^2, aq(kt, sq): 4 K-steps of A
        // (bf16-split: aq(kt, 2 u + half) = heads / remainders of -2e for K-group u)
        auto chunk_scores = [&](const f32x4 (&na)[4], auto &&aq, int ccg) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 acc[4];
                if constexpr (BF) {
#pragma unroll
                    for (int u = 0; u < UG; ++u)
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            f32x4 a4[4];
#pragma unroll
                            for (int kt = 0; kt < 4; ++kt) a4[kt] = aq(kt, 2 * u + half);
                            // code tile inner: consecutive instructions belong to four independent accumulator chains
#pragma unroll
                            for (int kt = 0; kt < 4; ++kt)
                                acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vq_bf16x8, a4[kt]),
                                                                                   __builtin_bit_cast(vq_bf16x8, bq[t][u]),
                                                                                   (u == 0 && half == 0) ? na[kt] : acc[kt], 0, 0, 0);
                        }
                } else {
#pragma unroll
                for (int sq = 0; sq < SQ; ++sq) {
                    f32x4 a4[4];
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) a4[kt] = aq(kt, sq);
                    // K-step outer, code tile inner: consecutive MFMAs belong to four independent accumulator chains
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int s = 4 * sq + j;
                        if (s < S) {
#pragma unroll
                            for (int kt = 0; kt < 4; ++kt)
                                acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[kt][j], zr[s][t], s == 0 ? na[kt] : acc[kt], 0, 0, 0);
                        }
                    }
                }
                }
                // the score's low 4 bits become its number inside the lane (code = 16 kt + 4 h + r): the minimum carries it
                float v[16];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#ifdef VQ_DBG_NOEMBED
                        v[kt * 4 + r] = (float)acc[kt][r];
#else
                        v[kt * 4 + r] = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, (float)acc[kt][r]) & ~15u) | (unsigned)(kt * 4 + r));
#endif
                float t1, t2;
#ifdef VQ_DBG_MINONLY
                t1 = vq2_min3(vq2_min3(vq2_min3(v[0], v[1], v[2]), vq2_min3(v[3], v[4], v[5]), vq2_min3(v[6], v[7], v[8])),
                              vq2_min3(vq2_min3(v[9], v[10], v[11]), vq2_min3(v[12], v[13], v[14]), v[15]), v[15]);
                t2 = t1;
#else
                vq2_top2_16(v, t1, t2);
#endif
                if constexpr (SINGLE) { m1[t] = t1; m2[t] = t2; }
                else {
                    m2[t] = vq2_min3(m2[t], t2, vq2_max(m1[t], t1));
                    c1[t] = t1 < m1[t] ? ccg : c1[t];
                    m1[t] = vq2_min(m1[t], t1);
                    pmv[t] = vq2_min(pmv[t], t1);
                }
            }
        };

        if constexpr (SINGLE) {
            if (act) chunk_scores(nreg, [&](int kt, int sq) { return areg[kt][sq]; }, 0);
        } else {
            for (int hp = 0; hp < nhp; ++hp) {
                const int p0 = hp * PCH, pn = min(PCH, ncc - p0);
                const int buf = resident ? hp : (int)(hp_it & 1u);
                if (!resident && !VQ_DBG(1)) {
                    // the buffer the previous barrier freed takes the next PCH chunks (of the next chunk of positions at the end)
                    const int hn = hp + 1 < nhp ? hp + 1 : 0;
                    vq2_stage(hn, buf ^ 1);
                }
                if (act) {
                    for (int cc = 0; cc < pn; ++cc) {
                        f32x4 na[4];
#pragma unroll
                        for (int kt = 0; kt < 4; ++kt) na[kt] = s_n[(buf * PCH + cc) * 16 + kt * 4 + h];
                        const f32x4 *__restrict__ ab = s_A + (buf * PCH + cc) * CHUNK_F4 + lane;
                        // (conflict-free ds_read_b128: consecutive lanes, consecutive 16-byte slots)
                        chunk_scores(na, [&](int kt, int sq) { return ab[(kt * AQ + sq) * 64]; }, p0 + cc);
                    }
                    const int done = p0 + pn;
                    if (done % gch == 0 || done == ncc) {      // (uniform) a group of codes is complete
                        // the same lane-pair steps as the final reduce-scatter below: the lane ends up with the minimum of
                        // the position it will own there
                        float pj[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            float x1 = pmv[j], y1 = pmv[j + 2];
                            vq2_swap<16>(x1, y1);
                            pj[j] = vq2_min(x1, y1);
                        }
                        float x1 = pj[0], y1 = pj[1];
                        vq2_swap<32>(x1, y1);
                        const float pmin = vq2_min(x1, y1);
                        // bf16, rounded toward -inf (a stored minimum is never above the true one); NaN stays NaN
                        const unsigned pb = __builtin_bit_cast(unsigned, pmin);
                        unsigned ph = pb >> 16;
                        ph += (pmin < 0.f && (pb & 0xffffu)) ? 1u : 0u;
                        ph = pmin != pmin ? 0xffffu : ph;
                        s_pm[((done - 1) / gch) * BLOCK + threadIdx.x] = (unsigned short)ph;
#pragma unroll
                        for (int t = 0; t < 4; ++t) pmv[t] = 3.4028235e38f;
                    }
                }
                if (!resident) {
                    __syncthreads();                           // (drains the copy: vmcnt(0)) the other buffer is complete, this one free
                    ++hp_it;
                }
            }
        }

        VQ2_USE(m1[3]); VQ2_USE(m2[3]);
        VQ2_STAMP(3)                                           // MFMAs + in-lane top-2
        f32x4 o[S];
        long long kpair[2] = {0, 0};
        int kown = 0;
        if (act) {
            // ---- reduce-scatter over the 4 lanes of a column: 9 lane-pair swaps, then this lane owns tile own_tile ----
            float zp[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                zp[t] = 0.f;
#pragma unroll
                for (int s = 0; s < S; ++s) zp[t] = fmaf(zr[s][t], zr[s][t], zp[t]);
            }
            float A1[2], A2[2], ZZ[2];
            int CC[2];
            // step 1, lanes h ^ 1: even rows keep tiles {0, 1}, odd rows tiles {2, 3}; the low embedded bit 4 of the
            // minimum records which row it came from (bit 0 of the winner's h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float x1 = m1[j], y1 = m1[j + 2], x2 = m2[j], y2 = m2[j + 2], xz = zp[j], yz = zp[j + 2];
                vq2_swap<16>(x1, y1);
                vq2_swap<16>(x2, y2);
                vq2_swap<16>(xz, yz);
                const bool odd_wins = y1 < x1;
                if constexpr (!SINGLE) {
                    float xc = __builtin_bit_cast(float, c1[j]), yc = __builtin_bit_cast(float, c1[j + 2]);
                    vq2_swap<16>(xc, yc);
                    CC[j] = __builtin_bit_cast(int, odd_wins ? yc : xc);
                }
                A2[j] = vq2_min3(x2, y2, vq2_max(x1, y1));
                const float w = vq2_min(x1, y1);
                A1[j] = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, w) & ~16u) | (odd_wins ? 16u : 0u));
                ZZ[j] = xz + yz;
            }
            // step 2, lanes h ^ 2: the lower half keeps its first tile, the upper half its second
            float a1, a2, zz;
            int cc1 = 0, hb1;
            {
                float x1 = A1[0], y1 = A1[1], x2 = A2[0], y2 = A2[1], xz = ZZ[0], yz = ZZ[1];
                vq2_swap<32>(x1, y1);
                vq2_swap<32>(x2, y2);
                vq2_swap<32>(xz, yz);
                const bool hi_wins = y1 < x1;
                if constexpr (!SINGLE) {
                    float xc = __builtin_bit_cast(float, CC[0]), yc = __builtin_bit_cast(float, CC[1]);
                    vq2_swap<32>(xc, yc);
                    cc1 = __builtin_bit_cast(int, hi_wins ? yc : xc);
                }
                a2 = vq2_min3(x2, y2, vq2_max(x1, y1));
                a1 = vq2_min(x1, y1);
                zz = xz + yz;
                hb1 = hi_wins ? 2 : 0;
            }
            const float tol = TOL_A * (zz + 2.f * emax) + TOL_D * fmaxf(zz + a1, 0.f) + 1e-30f;
            const bool flagged = !((a2 - a1) > tol);           // also true when anything is NaN / inf
            const float thrv = a1 + tol;                       // no code with a filter score above this can be the reference's
            {
                const unsigned bits = __builtin_bit_cast(unsigned, a1);
                const int kt = (bits >> 2) & 3, r = bits & 3, hw = hb1 | ((bits >> 4) & 1);
                kown = (SINGLE ? 0 : cc1 * 64) + kt * 16 + hw * 4 + r;
            }
            VQ2_USE(kown);
            VQ2_STAMP(4)                                       // reduce-scatter, tolerance
            // ---- exact re-check of the positions that failed the test: the whole wave, one position at a time ----
            unsigned long long fm = VQ_DBG(4) ? 0ull : __ballot(flagged);
            while (fm) {
                const int fl = __builtin_ctzll(fm);
                fm &= fm - 1;
                const int col = fl & 15, fh = fl >> 4, ft = ((fh & 1) << 1) | (fh >> 1);      // the lane's column and tile
                float zv[D];
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    // (scalar temporaries: __builtin_bit_cast applied to a vector-element lvalue reads element 0 -- clang 7.2)
                    const float z0 = zr[d >> 2][0], z1 = zr[d >> 2][1], z2 = zr[d >> 2][2], z3 = zr[d >> 2][3];
                    const float zc = ft == 0 ? z0 : (ft == 1 ? z1 : (ft == 2 ? z2 : z3));
                    zv[d] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zc), (d & 3) * 16 + col));
                }
                float bd = __builtin_inff();
                int bk = 0x7fffffff;
                if constexpr (SINGLE) {
                    // the codebook row of code `lane` from the LDS copy (no vector-memory operation in this loop)
                    float er[D];
                    const f32x4 *__restrict__ hr = s_H + lane * HROW;
                    if constexpr (S % 4 == 0) {
#pragma unroll
                        for (int hh = 0; hh < 4; ++hh)
#pragma unroll
                            for (int s4 = 0; s4 < S / 4; ++s4) {
                                const f32x4 v = hr[hh * (S / 4) + s4];
                                er[4 * (4 * s4) + hh] = v.x; er[4 * (4 * s4 + 1) + hh] = v.y;
                                er[4 * (4 * s4 + 2) + hh] = v.z; er[4 * (4 * s4 + 3) + hh] = v.w;
                            }
                    } else {
#pragma unroll
                        for (int d = 0; d < D; ++d) er[d] = reinterpret_cast<const float *>(hr)[(d & 3) * S + (d >> 2)];
                    }
                    if (lane < K) { bd = vq_exact_dist<D>(zv, er); bk = lane; }
                } else {
                    // only the groups whose minimum is within the tolerance of the best score (a NaN on either side of the
                    // comparison -- non-finite latents or codes -- keeps the group); the best code's own group always is
                    const float thr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, thrv), fl));
                    const int ng = (ncc + gch - 1) / gch;
                    for (int g = 0; g < ng; ++g) {
                        const float pg = __builtin_bit_cast(float, (unsigned)s_pm[g * BLOCK + wave * 64 + fl] << 16);
                        if (pg > thr) continue;
                        const int kend = min((g + 1) * gch * 64, K);
                        // (the rows of two codes per lane are requested before either is used: the loop is bound by the
                        //  latency of these loads, one round trip per iteration)
                        constexpr int UNR = D <= 16 ? 2 : 1;
                        for (int k0 = g * gch * 64 + lane; k0 < kend; k0 += 64 * UNR) {
                            float er[UNR][D];
#pragma unroll
                            for (int u = 0; u < UNR; ++u) {
                                const int k = min(k0 + 64 * u, K - 1);
#pragma unroll
                                for (int q = 0; q < D / 4; ++q) {
                                    const f32x4 v = reinterpret_cast<const f32x4 *>(cb + (long long)k * D)[q];
                                    er[u][4 * q] = v.x; er[u][4 * q + 1] = v.y; er[u][4 * q + 2] = v.z; er[u][4 * q + 3] = v.w;
                                }
                            }
#pragma unroll
                            for (int u = 0; u < UNR; ++u) {
                                const int k = k0 + 64 * u;
                                const float dk = vq_exact_dist<D>(zv, er[u]);
                                const bool bt = k < kend && ((bk == 0x7fffffff) | vq_better(dk, bd));      // (a lane's first code is taken as it is)
                                bd = bt ? dk : bd; bk = bt ? k : bk;
                            }
                        }
                    }
                }
                // wave-wide first minimum: DPP moves inside the 16-lane rows, lane-pair swaps across them (no LDS round trips)
                auto combine = [&](float od, int ok) {
                    // the other lane's candidate wins: NaN beats numbers, equal distances (and two NaNs) go to the smaller code
                    const int an = bd != bd, bn = od != od, lk = ok < bk;
                    const int other = (bn & ((an ^ 1) | lk)) | ((bn ^ 1) & (an ^ 1) & ((od < bd) | ((od == bd) & lk)));
                    bd = other ? od : bd; bk = other ? ok : bk;
                };
                combine(lane_xor1(bd), __builtin_bit_cast(int, lane_xor1(__builtin_bit_cast(float, bk))));
                combine(lane_xor2(bd), __builtin_bit_cast(int, lane_xor2(__builtin_bit_cast(float, bk))));
                combine(lane_xor4(bd), __builtin_bit_cast(int, lane_xor4(__builtin_bit_cast(float, bk))));
                combine(lane_xor8(bd), __builtin_bit_cast(int, lane_xor8(__builtin_bit_cast(float, bk))));
                {
                    float dl, dh; unsigned kl, kh;
                    vq2_pair<16>(bd, dl, dh); vq2_pair<16>((unsigned)bk, kl, kh);
                    bd = dl; bk = (int)kl; combine(dh, (int)kh);
                    vq2_pair<32>(bd, dl, dh); vq2_pair<32>((unsigned)bk, kl, kh);
                    bd = dl; bk = (int)kl; combine(dh, (int)kh);
                }
                kown = lane == fl ? bk : kown;
                ++nflag;
            }
            // ---- all-gather of the four codes of the column (3 swaps) ----
            int kb[4];
            {
                float lo = __builtin_bit_cast(float, kown), hi = lo;
                vq2_swap<32>(lo, hi);                          // lo: the lower half's code (tile 0 or 2), hi: the upper half's (1 or 3)
                float e0 = lo, o0 = lo, e1 = hi, o1 = hi;
                vq2_swap<16>(e0, o0);                          // even row's: tile 0, odd row's: tile 2
                vq2_swap<16>(e1, o1);                          // tile 1, tile 3
                kb[0] = __builtin_bit_cast(int, e0); kb[2] = __builtin_bit_cast(int, o0);
                kb[1] = __builtin_bit_cast(int, e1); kb[3] = __builtin_bit_cast(int, o1);
            }
            VQ2_USE(kb[3]);
            VQ2_STAMP(5)                                       // exact re-checks, all-gather
            // ---- gather + straight-through value + squared error, in the layout the lane already holds ----
            float ssef = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float *__restrict__ eh = SINGLE ? reinterpret_cast<const float *>(s_H) + kb[t] * (4 * HROW) + h * S
                                                      : cbH + ((long long)kb[t] * 4 + h) * S;
                float ev[S];
                if constexpr (S % 4 == 0) {
#pragma unroll
                    for (int s4 = 0; s4 < S / 4; ++s4) {
                        const f32x4 v = reinterpret_cast<const f32x4 *>(eh)[s4];
                        ev[4 * s4] = v.x; ev[4 * s4 + 1] = v.y; ev[4 * s4 + 2] = v.z; ev[4 * s4 + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < S; ++s) ev[s] = eh[s];
                }
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float diff = ev[s] - zr[s][t];
                    o[s][t] = zr[s][t] + diff;                 // z + (q - z), vq_vae.py:71
                    ssef += diff * diff;
                }
            }
            sse += (double)ssef;
            kpair[0] = (long long)(h ? kb[2] : kb[0]); kpair[1] = (long long)(h ? kb[3] : kb[1]);
            VQ2_USE(o[S - 1]); VQ2_USE(ssef);
            VQ2_STAMP(6)                                       // gather, straight-through value, squared error
        }
        // The prefetched z replaces this chunk's BEFORE the stores are issued: the wait for the prefetch then counts loads
        // only (vmcnt retires loads and stores in issue order; after the stores it would be a wait for them too).
#pragma unroll
        for (int s = 0; s < S; ++s) {
            asm volatile("" : "+v"(zn[s]));                    // (an opaque use pins the wait here; a plain copy is only renaming)
            zr[s] = zn[s];
        }
        if constexpr (JOIN) {
#pragma unroll
            for (int s = 0; s < S; ++s) asm volatile("" : "+v"(hq[s]));
        }
        __builtin_amdgcn_sched_barrier(0);
        VQ2_STAMP(0)                                           // wait for the prefetched z
        z_join_store(zr, hq, nchunk < NC, nb, nw);             // (JOIN) the next chunk's latents: formed and written here
        if (act) {
            if (out) {
                const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + (long long)cb_ * D * HW, 0, sample_bytes, 0x00020000);
#pragma unroll
                for (int s = 0; s < S; ++s)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o[s]),
                                                           ro, ovoff, cw_ * 256u + (unsigned)(4 * s) * (unsigned)HW * 4u, 0);
            }
            if (idx && h < 2) {
                const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(idx + (long long)cb_ * HW, 0, (unsigned)HW * 8u, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned *>(kpair),
                                                       ri, ivoff, cw_ * 512u, 0);
            }
            if (lds_hist) atomicAdd(&s_hist[kown], 1);
            else atomicAdd(&hist[kown], 1);
        }
        VQ2_STAMP(7)                                           // stores issued
        chunk = nchunk; cb_ = nb; cw_ = nw;
    }

    const double tot = block_sum(sse, s_red);
    if (threadIdx.x == 0) sse_slabs[blockIdx.x] = tot;
    if constexpr (INL) {
        // nothing was zeroed for this launch: the slabs no workgroup owns are zeroed here, the counters are plain stores of
        // this workgroup's row (its 64 counters, its re-evaluated positions at [64]), the header says how to read them
        for (int t2 = blockIdx.x + gridDim.x; t2 < nslabs; t2 += gridDim.x)
            if (threadIdx.x == 0) sse_slabs[t2] = 0.0;
        if (blockIdx.x == 0 && threadIdx.x < 4)
            hdr[threadIdx.x] = threadIdx.x == 1 ? (int)gridDim.x : (threadIdx.x == 2 ? VQ2_SLAB_STRIDE : (threadIdx.x == 3 ? 1 : 0));
        if (lane == 0) s_em[wave] = __builtin_bit_cast(float, nflag);
        __syncthreads();                                       // (also: every wave's s_hist adds are done)
        if (threadIdx.x < 64) hist[threadIdx.x] = s_hist[threadIdx.x];
        if (threadIdx.x == 64) {
            int n = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) n += __builtin_bit_cast(int, s_em[w]);
            hist[64] = n;
        }
        return;
    }
    if (lane == 0 && nflag) atomicAdd(hdr, nflag);
#ifdef VQ2_STAMPS
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(reinterpret_cast<unsigned long long *>(hdr + 4) + i, st_sum[i]);
#endif
    if (lds_hist) {
        __syncthreads();
        for (int k = threadIdx.x; k < K; k += BLOCK) {
            const int cnt = s_hist[k];
            if (cnt) atomicAdd(&hist[k], cnt);
        }
    }
}

__global__ void vq_decode_kernel(const long long *__restrict__ idx, const float *__restrict__ cb,
                                 float *__restrict__ q, int D, int K, int HW, long long P)
{
    const long long pos = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= P) return;
    long long k = idx[pos];
    if (k < 0) k = 0;
    if (k >= K) k = K - 1;
    const long long b = pos / HW, p = pos - b * HW;
    for (int d = 0; d < D; ++d) q[(b * D + d) * HW + p] = cb[k * D + d];
}

__global__ void vq_finalize_kernel(const double *__restrict__ sse_slabs, int nslabs,
                                   const int *__restrict__ hist, int K, long long P, int D,
                                   float cc, float *__restrict__ scalars)
{
    __shared__ double s_red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nslabs; i += blockDim.x) s += sse_slabs[i];
    const double sse = block_sum(s, s_red);
    double e = 0.0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float pk = (float)hist[k] / (float)P;
        e += (double)(pk * logf(pk + 1e-10f));
    }
    const double ent = block_sum(e, s_red);
    if (threadIdx.x == 0) {
        const float mse = (float)(sse / ((double)P * (double)D));
        scalars[0] = mse + cc * mse;           // q_latent_loss + commitment_cost * e_latent_loss
        scalars[1] = expf(-(float)ent);
        scalars[2] = mse;
    }
}

// The training step's last scalar launch: vq_finalize_kernel + the reconstruction-loss finaliser in one, reading the
// code counters straight from their replicas (no vq_hist_reduce launch before it).  out = (recon, commitment, total,
// perplexity), same arithmetic as dm_vq_finalize followed by dm_loss_finalize.
__global__ __launch_bounds__(1024) void vq_loss_finalize_kernel(const double *__restrict__ sse_slabs, int nslabs, const int *__restrict__ hrep,
                                        const int *__restrict__ hdr,
                                        int K, long long P, int D, float cc, const double *__restrict__ loss_slabs, int nloss,
                                        long long count, float w_recon, float w_commit, float *__restrict__ out,
                                        const double *__restrict__ tm_slabs, int ntm, float w_matching)
{
    __shared__ double s_red[16];
    __shared__ int s_cnt[16][64];
    const int R = hdr[1], stride = hdr[2];
    double s = 0.0, l = 0.0, e = 0.0, m = 0.0;
    // the pairwise term's partial losses (dm_time_matching_forward: pairs of doubles, the loss in the first)
    for (int i = threadIdx.x; i < ntm; i += blockDim.x) m += tm_slabs[2 * i];
    // (eight slabs per thread requested together: one memory round trip for up to 2048 slabs)
    for (int i0 = threadIdx.x; i0 < nslabs; i0 += 8 * (int)blockDim.x) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int i = i0 + j * (int)blockDim.x; v[j] = i < nslabs ? sse_slabs[i] : 0.0; }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (int i0 = threadIdx.x; i0 < nloss; i0 += 8 * (int)blockDim.x) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int i = i0 + j * (int)blockDim.x; v[j] = i < nloss ? loss_slabs[i] : 0.0; }
#pragma unroll
        for (int j = 0; j < 8; ++j) l += v[j];
    }
    if (hdr[3] == 1 && blockDim.x == 1024) {
        // per-workgroup slabs (<= 64 codes, up to 1024 rows): the sixteen waves take every sixteenth row, 64 codes side by
        // side, 32 rows in flight per wave: two memory round trips for 1024 rows
        const int k = threadIdx.x & 63, g = threadIdx.x >> 6;
        int h = 0;
        for (int r0 = g; r0 < R; r0 += 512) {
            int v[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) v[j] = r0 + 16 * j < R ? hrep[(long long)(r0 + 16 * j) * stride + k] : 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) h += v[j];
        }
        s_cnt[g][k] = h;
        __syncthreads();
        if (threadIdx.x < K) {
            int tot_k = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) tot_k += s_cnt[w][k];
            const float pk = (float)tot_k / (float)P;
            e += (double)(pk * logf(pk + 1e-10f));
        }
    } else {
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            const float pk = (float)vq_count_column(hrep, R, stride, k) / (float)P;
            e += (double)(pk * logf(pk + 1e-10f));
        }
    }
    const double sse = block_sum(s, s_red);
    const double tot = block_sum(l, s_red);
    const double ent = block_sum(e, s_red);
    const double tml = ntm > 0 ? block_sum(m, s_red) : 0.0;
    if (threadIdx.x == 0) {
        const float mse = (float)(sse / ((double)P * (double)D));
        const float commit = mse + cc * mse;
        const float recon = (float)(tot / (double)count);
        out[0] = recon;
        out[1] = commit;
        const float total = w_recon * recon + w_commit * commit;
        out[2] = total;
        out[3] = expf(-(float)ent);
        if (ntm > 0) {        // total + weight_matching * time_matching_loss (vq_vae.py:332; vae.py:470), the term as a fifth value
            const float tl = (float)tml;
            out[2] = __fadd_rn(total, __fmul_rn(w_matching, tl));
            out[4] = tl;
        }
    }
}

// 1024-thread workgroups (16 waves) walking the positions with a grid stride: the codebook gradient is accumulated in LDS
// over ALL of a workgroup's positions and flushed once, so the global float atomics (K*D addresses that every
// workgroup hits) number grid*K*D instead of (P/256)*K*D -- at B = 2048 that flush, not the streaming, was the cost.
// Codebooks larger than the LDS window (512 x 64, 4096 x 16) are split into windows of codes over grid.y.
constexpr int VQ_BWD_BLOCK = 1024;
constexpr int VQ_BWD_LDS = 136 * 1024;     // codebook-gradient window per workgroup (gfx950: 160 KB of LDS per CU)
// rows of the window are D + 1 floats apart: a wave adds into cell (code, d) for 64 positions at once -- the same d, 64 different
// codes -- and with rows of D = 16 / 64 floats every lane hit the same one or two LDS banks (a 32- to 64-way conflict per
// instruction: at 512 x 64 a third of the kernel's time)
__host__ __device__ constexpr int vq_bwd_row(int D) { return D + 1; }
// grid (x: positions, grid stride; y: windows of Kc codes).  A workgroup only touches the positions whose code falls
// into its window, so z / g_out / dz are still streamed once; idx is read once per window.
template <int D>
__global__ __launch_bounds__(VQ_BWD_BLOCK) void vq_backward_kernel(
    const float *__restrict__ z, const float *__restrict__ cb, const long long *__restrict__ idx,
    const float *__restrict__ g_out, const float *__restrict__ g_loss_dev, float cc,
    float *__restrict__ dz, float *__restrict__ dw, float *__restrict__ dw_slabs, int K, int HW, long long P,
    int Kc)
{
    extern __shared__ float s_dw[];    // [Kc][D + 1]
    constexpr int RS = vq_bwd_row(D);
    const long long k_lo = (long long)blockIdx.y * Kc;
    const int kn = K - k_lo < Kc ? (int)(K - k_lo) : Kc;
    for (int i = threadIdx.x; i < kn * RS; i += VQ_BWD_BLOCK) s_dw[i] = 0.f;
    __syncthreads();
    const float g_loss = g_loss_dev ? g_loss_dev[0] : 1.f;
    const double N = (double)P * (double)D;
    const float sz = (float)(2.0 * (double)cc / N) * g_loss;   // d/dz of cc * mse(q.detach(), z)
    const float sw = (float)(2.0 / N) * g_loss;                // d/dq of mse(q, z.detach())
    // Every (code, d) cell of the window belongs to ONE wave: wave w owns the dimensions [w DPW, (w + 1) DPW) and walks ALL of
    // the workgroup's positions, 64 per chunk (lane = position), in a fixed order.  The cell's additions are then program-ordered
    // inside one wave (chunk by chunk; the lanes of one ds_add_f32 that hit the same cell are served in lane order), so the
    // window -- and with the slab form the whole gradient -- is the same from launch to launch.  (Until round 6 a thread owned a
    // position and added into all D cells of its code: 16 waves raced on every cell and 4-6 % of the gradient's elements
    // differed in the last bits between launches, which Adam turns into +- lr on elements whose sign that decides.)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int DPW = (D + 15) / 16, CH = DPW >= 4 ? 2 : 4;   // dimensions per wave; chunks of 64 positions per stage
    static_assert(D < 16 || D % 16 == 0, "the waves split the dimensions evenly");
    const int d_lo = wave * DPW;
    // Two stages in flight.  Only the codebook row, the window test and the addresses of the store and the LDS addition depend
    // on a position's code; the codes themselves, z and the upstream gradient depend on the position alone and are requested
    // one stage ahead -- a wave used to wait out two dependent round trips (codes, then everything else) per 256 positions
    // with nothing else in flight, which was the whole kernel at D = 16.
    // (sample, offset) of a stage's first position advance without a division: with H W a multiple of a stage's positions all
    // of them lie in one sample (`whole`).
    struct Stage {
        long long k[CH], base[CH];
        float zv[DPW][CH], gv[DPW][CH];
    };
    const bool whole = HW % (64 * CH) == 0;
    const long long stride = (long long)gridDim.x * (64 * CH), stride_b = stride / HW, stride_r = stride - stride_b * HW;
    long long sb = ((long long)blockIdx.x * (64 * CH)) / HW, sr = ((long long)blockIdx.x * (64 * CH)) - sb * HW;
    auto request = [&](Stage &S, long long c0) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const long long pos = c0 + 64 * c + lane;
            const bool in = pos < P;
            const long long pp = in ? pos : 0;
            const long long kk = idx[pp];
            S.k[c] = in ? kk : -1;
            if (whole) {
                S.base[c] = in ? sb * (long long)D * HW + sr + 64 * c + lane : 0;
            } else {
                const long long b = pp / HW;
                S.base[c] = b * (long long)D * HW + (pp - b * HW);
            }
#pragma unroll
            for (int j = 0; j < DPW; ++j) {
                const long long o = S.base[c] + (long long)(d_lo + j) * HW;
                S.zv[j][c] = z[o];
                S.gv[j][c] = g_out ? g_out[o] : 0.f;
            }
        }
        sb += stride_b; sr += stride_r;
        if (sr >= HW) { sr -= HW; ++sb; }
    };
    auto process = [&](const Stage &S) {
        int kl[CH];
        bool hit[CH];
        // the wave's DPW dimensions of each position's code: ONE vector load per position where DPW is 2 / 4 / 8
        float qd[CH][DPW];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            hit[c] = S.k[c] >= k_lo && S.k[c] < k_lo + kn;
            kl[c] = hit[c] ? (int)(S.k[c] - k_lo) : 0;
            const float *q = cb + (hit[c] ? S.k[c] : 0) * D + d_lo;
            if constexpr (DPW % 4 == 0) {
#pragma unroll
                for (int j = 0; j < DPW; j += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(q + j);
                    qd[c][j] = v.x; qd[c][j + 1] = v.y; qd[c][j + 2] = v.z; qd[c][j + 3] = v.w;
                }
            } else if constexpr (DPW == 2) {
                const f32x2 v = *reinterpret_cast<const f32x2 *>(q);
                qd[c][0] = v.x; qd[c][1] = v.y;
            } else {
                qd[c][0] = q[0];
            }
        }
#pragma unroll
        for (int j = 0; j < DPW; ++j)
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (hit[c]) {
                    const int d = d_lo + j;
                    if (dz) dz[S.base[c] + (long long)d * HW] = S.gv[j][c] + sz * (S.zv[j][c] - qd[c][j]);
                    atomicAdd(&s_dw[kl[c] * RS + d], sw * (qd[c][j] - S.zv[j][c]));
                }
    };
    if (d_lo < D) {
        Stage sa, sbuf;
        long long c0 = (long long)blockIdx.x * (64 * CH);
        if (c0 < P) request(sa, c0);
        while (c0 < P) {
            const long long c1 = c0 + stride, c2 = c1 + stride;
            if (c1 < P) request(sbuf, c1);
            process(sa);
            if (c2 < P) request(sa, c2);
            if (c1 < P) process(sbuf);
            c0 = c2;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kn * D; i += VQ_BWD_BLOCK) {
        const float v = s_dw[(i / D) * RS + i % D];
        if (dw_slabs) dw_slabs[(long long)blockIdx.x * K * D + k_lo * D + i] = v;   // dm_reduce_slabs adds them in slab order
        else if (v != 0.f) atomicAdd(&dw[k_lo * D + i], v);
    }
}

// Any embedding_dim: vq_backward_kernel with the width at run time (the same LDS window of code gradients, the same
// arithmetic per element; a coverage path for widths without an instantiation).
__global__ __launch_bounds__(VQ_BWD_BLOCK) void vq_backward_any_kernel(
    const float *__restrict__ z, const float *__restrict__ cb, const long long *__restrict__ idx,
    const float *__restrict__ g_out, const float *__restrict__ g_loss_dev, float cc,
    float *__restrict__ dz, float *__restrict__ dw, float *__restrict__ dw_slabs, int K, int D, int HW, long long P, int Kc)
{
    extern __shared__ float s_dw[];    // [Kc][D + 1]
    const int RS = vq_bwd_row(D);
    const long long k_lo = (long long)blockIdx.y * Kc;
    const int kn = K - k_lo < Kc ? (int)(K - k_lo) : Kc;
    for (int i = threadIdx.x; i < kn * RS; i += VQ_BWD_BLOCK) s_dw[i] = 0.f;
    __syncthreads();
    const float g_loss = g_loss_dev ? g_loss_dev[0] : 1.f;
    const double N = (double)P * (double)D;
    const float sz = (float)(2.0 * (double)cc / N) * g_loss;
    const float sw = (float)(2.0 / N) * g_loss;
    // (waves own dimensions, as in vq_backward_kernel: the cells' additions are ordered)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dpw = (D + 15) / 16, d_lo = wave * dpw, d_hi = d_lo + dpw < D ? d_lo + dpw : D;
    if (d_lo < D)
        for (long long c0 = (long long)blockIdx.x * 64; c0 < P; c0 += (long long)gridDim.x * 64) {
            const long long pos = c0 + lane;
            const long long k = pos < P ? idx[pos] : -1;
            if (k < k_lo || k >= k_lo + kn) continue;
            const long long b = pos / HW;
            const long long base = b * (long long)D * HW + (pos - b * HW);
            const float *__restrict__ q = cb + k * D;
            const int kl = (int)(k - k_lo);
            for (int d = d_lo; d < d_hi; ++d) {
                const long long o = base + (long long)d * HW;
                const float zv = z[o], qv = q[d], gv = g_out ? g_out[o] : 0.f;
                if (dz) dz[o] = gv + sz * (zv - qv);
                atomicAdd(&s_dw[kl * RS + d], sw * (qv - zv));
            }
        }
    __syncthreads();
    for (int i = threadIdx.x; i < kn * D; i += VQ_BWD_BLOCK) {
        const float v = s_dw[(i / D) * RS + i % D];
        if (dw_slabs) dw_slabs[(long long)blockIdx.x * K * D + k_lo * D + i] = v;
        else if (v != 0.f) atomicAdd(&dw[k_lo * D + i], v);
    }
}


// ---- codebooks of at most 64 codes (every configuration of the reference): the codebook gradient as a one-hot product
// on the matrix cores.  dW[k][d] = sw * sum_p [idx[p] == k] (e_k[d] - z[p][d]) is (codes x positions) . (positions x D):
// v_mfma_f32_16x16x4_f32 with A[i = code][kk = position] = 1.0 or 0.0 built from the indices and
// B[kk = position][j = d] = sw * (e - z) -- the value the lane has just formed for dz.  The accumulators live in
// registers over all of a wave's positions, the waves of a workgroup are combined in wave order and the workgroup
// writes ONE slab, so the sum over positions has a fixed order: no float atomics anywhere (the LDS-atomic kernel
// above stays for larger codebooks, where a one-hot product would cost K/16 matrix instructions per four positions).
// Lane (i = lane & 15, g = lane >> 4) holds row d = 16 dt + i of four position quads: z[d][16 u + 4 g .. + 3].
constexpr int VQ_BWD2_BLOCK = 512;
template <int D, int WGS>
__global__ __launch_bounds__(VQ_BWD2_BLOCK, WGS) void vq_backward_mfma_kernel(
    const float *__restrict__ z, const float *__restrict__ cb, const long long *__restrict__ idx,
    const float *__restrict__ g_out, const float *__restrict__ g_loss_dev, float cc,
    float *__restrict__ dz, float *__restrict__ dw_slabs, int K, int HW, unsigned NC)
{
    constexpr int DT = D / 16, HROW = D + 1, NW = VQ_BWD2_BLOCK / 64;
    __shared__ float s_cb[64 * HROW];                        // codes as rows, one pad word: the gather is conflict-free
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float g_loss = g_loss_dev ? g_loss_dev[0] : 1.f;
    const double N = (double)NC * 64.0 * (double)D;
    const float sz = (float)(2.0 * (double)cc / N) * g_loss;   // d/dz of cc * mse(q.detach(), z)
    const float sw = (float)(2.0 / N) * g_loss;                // d/dq of mse(q, z.detach())
    const unsigned cps = (unsigned)HW >> 6;                    // chunks of 64 positions per sample
    f32x4 acc[4][DT];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[kt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned sample_bytes = (unsigned)D * (unsigned)HW * 4u;
    const unsigned voff = ((unsigned)i * (unsigned)HW + 4u * (unsigned)g) * 4u;      // this lane's bytes inside a row group
    typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
    auto rsrc = [&](const float *p, unsigned b) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p) + (long long)b * D * HW, 0, sample_bytes, 0x00020000);
    };
    // (row group dt, position quad u) of chunk cw: scalar byte offset inside the sample
    auto soff = [&](unsigned cw, int dt, int u) { return cw * 256u + (unsigned)(16 * dt) * (unsigned)HW * 4u + 64u * u; };
    auto load = [&](f32x4 (&dst)[DT][4], const float *p, unsigned b, unsigned cw) {
        const __amdgpu_buffer_rsrc_t r = rsrc(p, b);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                dst[dt][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff(cw, dt, u), 0));
    };
    // (the low word of the int64 index: a 64-bit load leaves a dead high register whose reuse stalls on the load)
    const int *__restrict__ idx32 = reinterpret_cast<const int *>(idx);
    const unsigned step = gridDim.x * NW;
    unsigned chunk = blockIdx.x * NW + wave;
    f32x4 zr[DT][4], gr[DT][4], zn[DT][4], gn[DT][4];
    int kv = 0, kn = 0;
    constexpr bool PF = D <= 32;                               // (embedding_dim 64: the second set of rows would spill)
    if (PF && chunk < NC) {
        const unsigned b = chunk / cps, cw = chunk - b * cps;
        load(zr, z, b, cw);
        if (g_out) load(gr, g_out, b, cw);
        kv = idx32[2 * ((long long)chunk * 64 + lane)];
    }
    // (the codebook after the first rows have been requested: one memory round trip for both)
    for (int e = threadIdx.x; e < 64 * D; e += VQ_BWD2_BLOCK) {
        const int k = e / D, d = e - k * D;
        s_cb[k * HROW + d] = k < K ? cb[e] : 0.f;
    }
    __syncthreads();
    for (; chunk < NC; chunk += step) {
        const unsigned b = chunk / cps, cw = chunk - b * cps;
        const unsigned nxt = chunk + step;
        if constexpr (!PF) {
            load(zr, z, b, cw);
            if (g_out) load(gr, g_out, b, cw);
            kv = idx32[2 * ((long long)chunk * 64 + lane)];
        }
        if constexpr (PF) {
            // the next chunk's rows are in flight during this one's products (past the end: the last chunk again, from L2 --
            // an unconditional load keeps the wait counts exact)
            const unsigned nc = nxt < NC ? nxt : NC - 1;
            const unsigned nb = nc / cps, ncw = nc - nb * cps;
            load(zn, z, nb, ncw);
            if (g_out) load(gn, g_out, nb, ncw);
            kn = idx32[2 * ((long long)nc * 64 + lane)];
            __builtin_amdgcn_sched_barrier(0);                 // (the scheduler otherwise sinks the index load below the products)
        }
        if (!g_out) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int u = 0; u < 4; ++u) gr[dt][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // the sixteen codes of this lane's positions, then their rows of the codebook: two batched LDS round trips
        int kj[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = __builtin_amdgcn_ds_bpermute(4 * (16 * u + 4 * g + t), kv);
                kj[u][t] = (unsigned)k < 64u ? k : 63;         // (indices come from dm_vq_forward: always in range)
            }
        float ev[DT][4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) ev[dt][u][t] = s_cb[kj[u][t] * HROW + 16 * dt + i];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int rel = kj[u][t] - i;
                float bv[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const float zc = zr[dt][u][t], gc = gr[dt][u][t];
                    const float diff = ev[dt][u][t] - zc;
                    bv[dt] = sw * diff;
                    gr[dt][u][t] = gc - sz * diff;            // g + sz * (z - e), the bits of the kernel above
                }
                // (all four code tiles: a tile past K holds no index, its products are zeros)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const float a = rel == 16 * kt ? 1.f : 0.f;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        acc[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[dt], acc[kt][dt], 0, 0, 0);
                }
            }
        }
        if constexpr (PF) {
            // pin the wait for the prefetched rows here, ahead of the stores (else it lands after them as vmcnt(0))
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    asm volatile("" : "+v"(zn[dt][u]));
                    if (g_out) asm volatile("" : "+v"(gn[dt][u]));
                }
            asm volatile("" : "+v"(kn));
        }
        if (dz) {
            const __amdgpu_buffer_rsrc_t r = rsrc(dz, b);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, gr[dt][u]), r, voff, soff(cw, dt, u), 0);
        }
        if constexpr (PF) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    zr[dt][u] = zn[dt][u];
                    if (g_out) gr[dt][u] = gn[dt][u];
                }
            kv = kn;
        }
    }
    // ---- the waves of the workgroup in wave order, then one slab [K][D].  acc[kt][dt][r] is code 16 kt + 4 g + r, d 16 dt + i
    __syncthreads();                                           // every gather from s_cb is done: reuse it as the sum
    float *s_sum = s_cb;
    for (unsigned w = 0; w < NW; ++w) {
        if (wave == w) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float *dst = s_sum + (16 * kt + 4 * g + r) * HROW + 16 * dt + i;
                        *dst = w == 0 ? acc[kt][dt][r] : *dst + acc[kt][dt][r];
                    }
        }
        __syncthreads();
    }
    float *slab = dw_slabs + (long long)blockIdx.x * K * D;
    for (int e = threadIdx.x; e < K * D; e += VQ_BWD2_BLOCK) {
        const int k = e / D, d = e - k * D;
        slab[e] = s_sum[k * HROW + d];
    }
}

// widths with register-resident instantiations of the exact kernels; every other width 1 .. 512 takes vq_forward_any_kernel /
// vq_backward_any_kernel (run-time width, the same arithmetic)
bool vq_dim_built(int D) { return D == 8 || D == 16 || D == 32 || D == 64 || D == 128; }
bool vq_dim_supported(int D) { return D >= 1 && D <= 512; }

}  // namespace

extern "C" size_t dm_vq_workspace_bytes(int K, int D)
{
    if (K <= 0 || D <= 0) return 0;
    return (size_t)vq2_layout(K, D).total * sizeof(float);
}

constexpr int VQ_PP = 2;      // positions per lane in the exact kernel (1 for embedding_dim 64: registers)

extern "C" int dm_vq_num_blocks(int64_t positions)
{
    return (int)((positions + VQ_BLOCK - 1) / VQ_BLOCK);     // upper bound over all variants; unused slabs are zeroed
}

namespace {
// DM_VQ_AUTO's choice of filter; DM_VQ_FILTER=f32 in the environment keeps the f32-input one (A/B measurements)
bool vq2_auto_bf16()
{
    static const bool on = [] { const char *e = getenv("DM_VQ_FILTER"); return !(e && e[0] == 'f'); }();
    return on;
}
// measurement knobs of the headline shape (embedding_dim 16, <= 64 codes): DM_VQ_OCC=4 takes the build bounded to 128
// registers (4 waves per SIMD), DM_VQ_WGS=n launches n workgroups per CU instead of the occupancy's
// 0: off, 3 / 4: products of the bf16 split in vq_cells_kernel (default 3)
int vq2_cells()
{
    static const int v = [] {
        const char *e = getenv("DM_VQ_CELLS"), *p = getenv("DM_VQ_CELLS_PROD");
        if (e && e[0] == '0') return 0;
        return (p && p[0] == '4') ? 4 : 3;
    }();
    return v;
}
// start delay of the CUs' second workgroups in percent of one pass's matrix time (DM_VQ_CELLS_STAGGER; 0 = none)
int vq2_cells_stagger_pct() { static const int v = [] { const char *e = getenv("DM_VQ_CELLS_STAGGER"); return e ? atoi(e) : 90; }(); return v; }
bool vq2_force_prep() { static const bool v = [] { const char *e = getenv("DM_VQ_PREP"); return e && e[0] == '1'; }(); return v; }
int vq2_occ() { static const int v = [] { const char *e = getenv("DM_VQ_OCC"); return e ? atoi(e) : 3; }(); return v; }
int vq2_wgs(int dflt) { static const int v = [] { const char *e = getenv("DM_VQ_WGS"); return e ? atoi(e) : 0; }(); return v > 0 ? v : dflt; }
bool vq2_applicable(const float *z, const int64_t *idx, const float *out, const void *ws, int D, int HW)
{
    const uintptr_t al = (uintptr_t)z | (uintptr_t)idx | (uintptr_t)out | (uintptr_t)ws;
    return (D == 8 || D == 16 || D == 32 || D == 64) && HW % 64 == 0 && (al & 15) == 0;
}
}  // namespace

namespace {
int vq_forward_launch(const float *z, const float *codebook, int64_t *idx, float *out, double *sse_slabs, int32_t *hist,
                      int B, int D, int K, int H, int W, void *workspace, size_t workspace_bytes, int variant, int repeats,
                      void *stream, const float *jh = nullptr, const float *jcoef = nullptr, float *jz = nullptr);
}

extern "C" int dm_vq_forward_variant(const float *z, const float *codebook, int64_t *idx, float *out,
                                     double *sse_slabs, int32_t *hist, int B, int D, int K, int H, int W,
                                     void *workspace, size_t workspace_bytes, int variant, void *stream)
{
    return vq_forward_launch(z, codebook, idx, out, sse_slabs, hist, B, D, K, H, W, workspace, workspace_bytes, variant, 1,
                             stream);
}

extern "C" int dm_vq_forward_repeat(const float *z, const float *codebook, int64_t *idx, float *out,
                                    double *sse_slabs, int32_t *hist, int B, int D, int K, int H, int W,
                                    void *workspace, size_t workspace_bytes, int variant, int repeats, void *stream)
{
    DM_REQUIRE(repeats >= 1 && repeats <= 1000, "dm_vq_forward_repeat: repeats %d", repeats);
    return vq_forward_launch(z, codebook, idx, out, sse_slabs, hist, B, D, K, H, W, workspace, workspace_bytes, variant,
                             repeats, stream);
}

namespace {
int vq_forward_launch(const float *z, const float *codebook, int64_t *idx, float *out, double *sse_slabs, int32_t *hist,
                      int B, int D, int K, int H, int W, void *workspace, size_t workspace_bytes, int variant, int repeats,
                      void *stream, const float *jh, const float *jcoef, float *jz)
{
    DM_REQUIRE(z && codebook && sse_slabs, "dm_vq_forward: NULL pointer");     // hist == NULL: the counters stay in their replicas
    DM_REQUIRE(B > 0 && H > 0 && W > 0 && K > 0, "dm_vq_forward: bad shape B=%d K=%d H=%d W=%d", B, K, H, W);
    DM_REQUIRE(vq_dim_supported(D), "dm_vq_forward: embedding_dim %d outside 1 .. 512", D);
    DM_REQUIRE(workspace && workspace_bytes >= dm_vq_workspace_bytes(K, D), "dm_vq_forward: workspace too small");
    DM_REQUIRE(variant >= DM_VQ_AUTO && variant <= DM_VQ_BF16, "dm_vq_forward: bad variant %d", variant);
    DM_REQUIRE(variant != DM_VQ_BF16 || D % 16 == 0, "dm_vq_forward: the bf16-split filter needs embedding_dim 16, 32 or 64");
    hipStream_t s = (hipStream_t)stream;
    const long long P = (long long)B * H * W;
    const Vq2Layout L = vq2_layout(K, D);
    float *ws = (float *)workspace;
    const bool can2 = vq2_applicable(z, idx, out, workspace, D, H * W) && ((uintptr_t)codebook & 15) == 0;
    DM_REQUIRE((variant != DM_VQ_MFMA && variant != DM_VQ_BF16) || can2,
               "dm_vq_forward: the MFMA variant needs embedding_dim 8/16/32/64, H*W %% 64 == 0 and 16-byte aligned tensors");
    const bool use2 = variant == DM_VQ_MFMA || variant == DM_VQ_BF16 || (variant == DM_VQ_AUTO && can2);
    const long long n = L.total;
    const int nslabs = dm_vq_num_blocks(P);
    int pgrid = (int)((n + 255) / 256);
    if (pgrid > 1024) pgrid = 1024;
    // <= 64 codes (every configuration of the reference), embedding_dim 16 / 32 / 64: the MFMA kernel prepares its own
    // operands and writes its counters as per-workgroup rows -- no preparation launch, no counter reduction
    // (DM_VQ_PREP=1 in the environment keeps the separate preparation: A/B measurements)
    const bool inl = use2 && K <= 64 && D % 16 == 0 && !vq2_force_prep();
    DM_REQUIRE(!jz || (inl && D == 16), "dm_vq_forward_join: built for the one-launch form (<= 64 codes, embedding_dim 16, H*W %% 64 == 0)");
    if (!inl) hipLaunchKernelGGL(vq_prep_kernel, dim3(pgrid), dim3(256), 0, s, codebook, ws, L, K, D, sse_slabs, nslabs);
    // (vq_prep_kernel cleared the counter replicas and all slabs: there are fewer workgroups than slabs)
    int *hrep = reinterpret_cast<int *>(ws + L.hrep);
    if (use2) {
#ifdef DM_MEASURE
        {
            static const int dbg = [] { const char *e = getenv("DM_VQ_DBG"); return e ? atoi(e) : 0; }();
            static bool sent = false;
            if (!sent) { (void)hipMemcpyToSymbol(HIP_SYMBOL(vq_dbg_dev), &dbg, sizeof(int)); sent = true; }
        }
#endif
        const long long groups = ((P >> 6) + 3) / 4;
#define DM_VQ2K(DD, SINGLE_, MINW, WGS, BF_, INL_, JOIN_)                                                            \
    {                                                                                                                \
        const int wgs = (WGS);                                                                                       \
        long long g_ = groups < 256 * wgs ? groups : 256 * wgs;                                                      \
        if (INL_ && g_ > VQ2_SLAB_ROWS) g_ = VQ2_SLAB_ROWS;          /* one counter row per workgroup */              \
        hipLaunchKernelGGL((vq_forward_mfma_kernel<DD, SINGLE_, MINW, BF_, INL_, JOIN_>),                            \
                           dim3((unsigned)g_), dim3(256), 0, s, z, codebook,                                         \
                           ws + (BF_ ? L.cbB : L.cbA), ws + L.nrm, ws + L.cbH, (long long *)idx, out, sse_slabs, hrep, \
                           L.R, (int *)ws, K, H * W, P, nslabs, jh, jcoef, jz);                                      \
    }
#define DM_VQ2(DD, SINGLE_, MINW, WGS)                                                                               \
    {                                                                                                                \
        if constexpr (DD % 16 == 0 && SINGLE_) {                                                                     \
            if (inl && bf && jz) DM_VQ2K(DD, SINGLE_, MINW, WGS, true, true, DD == 16)                               \
            else if (inl && jz) DM_VQ2K(DD, SINGLE_, MINW, WGS, false, true, DD == 16)                               \
            else if (inl && bf) DM_VQ2K(DD, SINGLE_, MINW, WGS, true, true, false)                                          \
            else if (inl) DM_VQ2K(DD, SINGLE_, MINW, WGS, false, true, false)                                               \
            else if (bf) DM_VQ2K(DD, SINGLE_, MINW, WGS, true, false, false)                                                \
            else DM_VQ2K(DD, SINGLE_, MINW, WGS, false, false, false)                                                       \
        } else if constexpr (DD % 16 == 0) {                                                                         \
            if (bf) DM_VQ2K(DD, SINGLE_, MINW, WGS, true, false, false) else DM_VQ2K(DD, SINGLE_, MINW, WGS, false, false, false)  \
        } else DM_VQ2K(DD, SINGLE_, MINW, WGS, false, false, false)                                                         \
    }
        const bool single = K <= 64;
        // bf16-split filter (DM_VQ_BF16; DM_VQ_AUTO takes it where it applies) or the f32 one (DM_VQ_MFMA)
        const bool bf = D % 16 == 0 && (variant == DM_VQ_BF16 || (variant == DM_VQ_AUTO && vq2_auto_bf16()));
        // 64 < K <= 4096 at embedding_dim 16 (BASELINE configs[4]): the cell kernel of vq_cells.h -- DM_VQ_CELLS=0 keeps the
        // streamed 16x16x32 kernel, DM_VQ_CELLS_PROD=4 all four products of the split (A/B measurements)
        const bool cells = bf && D == 16 && K > 64 && K <= VQC_MAX_K && (H * W) % 128 == 0 && vq2_cells() > 0;
        for (int rep = 0; rep < repeats; ++rep) {
            if (cells) {
                const long long passes = P >> 7;
                long long g_ = (passes + 3) / 4;
                if (g_ > 512) g_ = 512;                              // two workgroups per CU, persistent over the passes
                // half a pass in units of 64 cycles (s_sleep): a pass streams K / 32 chunks of 12 (16) matrix instructions of 32
                // cycles; only where both slots of the CUs are taken and every wave has more than one pass
                const int prod = vq2_cells();
                const long long chunk_cycles = (long long)(prod == 4 ? 16 : 12) * 32;
                int stagger = (g_ > 256 && passes >= 2 * g_ * 4) ? (int)(((K + 31) / 32) * chunk_cycles * vq2_cells_stagger_pct() / 100 / 64) : 0;
                if (prod == 4)
                    hipLaunchKernelGGL((vq_cells_kernel<4>), dim3((unsigned)g_), dim3(256), 0, s, z, codebook,
                                       reinterpret_cast<const vqc_u32x4 *>(ws + L.cbP), ws + L.nrmP, ws + L.nrm, (long long *)idx,
                                       out, sse_slabs, hrep, L.R, (int *)ws, K, H * W, P, stagger);
                else
                    hipLaunchKernelGGL((vq_cells_kernel<3>), dim3((unsigned)g_), dim3(256), 0, s, z, codebook,
                                       reinterpret_cast<const vqc_u32x4 *>(ws + L.cbP), ws + L.nrmP, ws + L.nrm, (long long *)idx,
                                       out, sse_slabs, hrep, L.R, (int *)ws, K, H * W, P, stagger);
                continue;
            }
            switch (D) {
            case 8: if (single) DM_VQ2(8, true, 3, 3) else DM_VQ2(8, false, 3, 3) break;
            case 16:
                if (single && vq2_occ() == 4) DM_VQ2(16, true, 4, vq2_wgs(4))
                else if (single) DM_VQ2(16, true, 3, vq2_wgs(3))
                else DM_VQ2(16, false, 3, 3)
                break;
            case 32: if (single) DM_VQ2(32, true, 2, 2) else DM_VQ2(32, false, 2, 2) break;
            default: if (single) DM_VQ2(64, true, 1, 1) else DM_VQ2(64, false, 1, 2) break;
            }
        }
#undef DM_VQ2
#undef DM_VQ2K
        if (hist) hipLaunchKernelGGL(vq_hist_reduce_kernel, dim3((K + 63) / 64), dim3(1024), 0, s, hrep, (int *)ws, K, (int *)hist);
        return dm_launch_status("dm_vq_forward");
    }
    if (!vq_dim_built(D)) {
        long long g_ = (P + 63) / 64;
        if (g_ > nslabs) g_ = nslabs;                            // one squared-error slab per workgroup (the rest were zeroed)
        if (g_ > 4096) g_ = 4096;
        const size_t lds = (size_t)D * 64 * sizeof(float);
        static DmPerDeviceOnce any_attr;
        if (lds > 48 * 1024 && any_attr.need()) {
            const hipError_t ea = hipFuncSetAttribute((const void *)vq_forward_any_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 512 * 64 * 4);
            if (ea != hipSuccess) { dm_set_error("dm_vq_forward: cannot reserve LDS: %s", hipGetErrorString(ea)); return (int)ea; }
            any_attr.mark();
        }
        for (int rep = 0; rep < repeats; ++rep)
            hipLaunchKernelGGL(vq_forward_any_kernel, dim3((unsigned)g_), dim3(64), lds, s, z, codebook, (long long *)idx, out,
                               sse_slabs, hrep, L.R, K, D, H * W, P);
        if (hist) hipLaunchKernelGGL(vq_hist_reduce_kernel, dim3((K + 63) / 64), dim3(1024), 0, s, hrep, (int *)ws, K, (int *)hist);
        return dm_launch_status("dm_vq_forward");
    }
    const float *cbT = ws + L.cbT;
#define DM_VQ_FWD(DD, PP_)                                                                                   \
    hipLaunchKernelGGL((vq_forward_kernel<DD, PP_>), dim3((unsigned)((P + VQ_BLOCK * PP_ - 1) / (VQ_BLOCK * PP_))), \
                       dim3(VQ_BLOCK), 0, s, z, codebook, cbT, (long long *)idx, out, sse_slabs, hrep, L.R, K, H * W, P)
    for (int rep = 0; rep < repeats; ++rep) {
        switch (D) {
        case 8: DM_VQ_FWD(8, VQ_PP); break;
        case 16: DM_VQ_FWD(16, VQ_PP); break;
        case 32: DM_VQ_FWD(32, VQ_PP); break;
        case 64: DM_VQ_FWD(64, 1); break;
        default: DM_VQ_FWD(128, 1); break;        // VectorQuantizer's own default embedding_dim (vq_vae.py:35)
        }
    }
#undef DM_VQ_FWD
    if (hist) hipLaunchKernelGGL(vq_hist_reduce_kernel, dim3((K + 63) / 64), dim3(1024), 0, s, hrep, (int *)ws, K, (int *)hist);
    return dm_launch_status("dm_vq_forward");
}
}  // namespace

extern "C" int dm_vq_forward(const float *z, const float *codebook, int64_t *idx, float *out,
                             double *sse_slabs, int32_t *hist, int B, int D, int K, int H, int W,
                             void *workspace, size_t workspace_bytes, void *stream)
{
    return dm_vq_forward_variant(z, codebook, idx, out, sse_slabs, hist, B, D, K, H, W, workspace, workspace_bytes,
                                 DM_VQ_AUTO, stream);
}

extern "C" int dm_vq_forward_join_supported(int D, int K, int H, int W)
{
    // (embedding_dim 16 = the reference's num_hiddens: at 32 / 64 the second prefetched tensor does not fit the registers)
    return (K > 0 && K <= 64 && D == 16 && H > 0 && W > 0 && (H * W) % 64 == 0 && !vq2_force_prep()) ? 1 : 0;
}

extern "C" int dm_vq_forward_join(const float *rb, const float *h_in, const float *coef, float *z_out, const float *codebook,
                                  int64_t *idx, float *out, double *sse_slabs, int32_t *hist, int B, int D, int K, int H, int W,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    DM_REQUIRE(rb && h_in && coef && z_out, "dm_vq_forward_join: NULL pointer");
    DM_REQUIRE(rb != z_out && h_in != z_out, "dm_vq_forward_join: z_out must not alias an input (chunks are read ahead)");
    DM_REQUIRE(dm_vq_forward_join_supported(D, K, H, W), "dm_vq_forward_join: shape not built (D %d, K %d, %dx%d)", D, K, H, W);
    DM_REQUIRE((((uintptr_t)h_in | (uintptr_t)z_out) & 15) == 0, "dm_vq_forward_join: 16-byte aligned tensors");
    return vq_forward_launch(rb, codebook, idx, out, sse_slabs, hist, B, D, K, H, W, workspace, workspace_bytes, DM_VQ_AUTO, 1,
                             stream, h_in, coef, z_out);
}

extern "C" int dm_vq_decode(const int64_t *idx, const float *codebook, float *q,
                            int B, int D, int K, int H, int W, void *stream)
{
    DM_REQUIRE(idx && codebook && q, "dm_vq_decode: NULL pointer");
    DM_REQUIRE(B > 0 && D > 0 && K > 0 && H > 0 && W > 0, "dm_vq_decode: bad shape");
    const long long P = (long long)B * H * W;
    hipLaunchKernelGGL(vq_decode_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long *)idx, codebook, q, D, K, H * W, P);
    return dm_launch_status("dm_vq_decode");
}

extern "C" int dm_vq_finalize(const double *sse_slabs, int nslabs, const int32_t *hist, int K,
                              int64_t positions, int D, float commitment_cost, float *scalars, void *stream)
{
    DM_REQUIRE(sse_slabs && hist && scalars && nslabs > 0 && K > 0 && positions > 0, "dm_vq_finalize: bad argument");
    hipLaunchKernelGGL(vq_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream,
                       sse_slabs, nslabs, (const int *)hist, K, (long long)positions, D, commitment_cost, scalars);
    return dm_launch_status("dm_vq_finalize");
}

extern "C" int dm_vq_loss_finalize(const double *sse_slabs, int nslabs, const void *workspace, int K, int D,
                                   int64_t positions, float commitment_cost, const double *loss_slabs, int nloss,
                                   int64_t count, float weight_recon, float weight_commitment, float *scalars_out,
                                   void *stream)
{
    DM_REQUIRE(sse_slabs && workspace && loss_slabs && scalars_out && nslabs > 0 && nloss > 0 && K > 0 && D > 0 &&
                   positions > 0 && count > 0, "dm_vq_loss_finalize: bad argument");
    const Vq2Layout L = vq2_layout(K, D);
    const int *hrep = reinterpret_cast<const int *>(reinterpret_cast<const float *>(workspace) + L.hrep);
    hipLaunchKernelGGL(vq_loss_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sse_slabs, nslabs, hrep,
                       reinterpret_cast<const int *>(workspace), K,
                       (long long)positions, D, commitment_cost, loss_slabs, nloss, (long long)count, weight_recon,
                       weight_commitment, scalars_out, (const double *)nullptr, 0, 0.f);
    return dm_launch_status("dm_vq_loss_finalize");
}

extern "C" int dm_vq_loss_finalize_tm(const double *sse_slabs, int nslabs, const void *workspace, int K, int D,
                                      int64_t positions, float commitment_cost, const double *loss_slabs, int nloss,
                                      int64_t count, float weight_recon, float weight_commitment, const double *tm_slabs,
                                      int ntm, float weight_matching, float *scalars_out, void *stream)
{
    DM_REQUIRE(sse_slabs && workspace && loss_slabs && scalars_out && tm_slabs && nslabs > 0 && nloss > 0 && ntm > 0 && K > 0 &&
                   D > 0 && positions > 0 && count > 0, "dm_vq_loss_finalize_tm: bad argument");
    const Vq2Layout L = vq2_layout(K, D);
    const int *hrep = reinterpret_cast<const int *>(reinterpret_cast<const float *>(workspace) + L.hrep);
    hipLaunchKernelGGL(vq_loss_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sse_slabs, nslabs, hrep,
                       reinterpret_cast<const int *>(workspace), K,
                       (long long)positions, D, commitment_cost, loss_slabs, nloss, (long long)count, weight_recon,
                       weight_commitment, scalars_out, tm_slabs, ntm, weight_matching);
    return dm_launch_status("dm_vq_loss_finalize_tm");
}

namespace {
// workgroups = slabs of K*D floats each: two 1024-thread workgroups per CU, fewer for large codebooks so that the slab
// tensor stays below 32 MB (4096 x 16: 128 slabs instead of 512 = 128 MB written and read back per step)
int vq_backward_grid(long long P, int K, int D)
{
    const long long want = (P + VQ_BWD_BLOCK - 1) / VQ_BWD_BLOCK;
    long long cap = (8LL << 20) / ((long long)K * D);
    cap = cap > 512 ? 512 : (cap < 32 ? 32 : cap);
    return (int)(want < cap ? want : cap);
}

int vq_backward_launch(const char *who, const float *z, const float *codebook, const int64_t *idx, const float *g_out,
                       const float *g_loss_dev, float commitment_cost, float *dz, float *dw, float *dw_slabs,
                       int B, int D, int K, int H, int W, void *stream)
{
    DM_REQUIRE(z && codebook && idx && (dw || dw_slabs), "%s: NULL pointer", who);
    DM_REQUIRE(vq_dim_supported(D), "%s: embedding_dim %d outside 1 .. 512", who, D);
    const long long P = (long long)B * H * W;
    int Kc = VQ_BWD_LDS / (vq_bwd_row(D) * (int)sizeof(float));
    if (Kc > K) Kc = K;
    const size_t lds = (size_t)Kc * vq_bwd_row(D) * sizeof(float);
    const int grid = vq_backward_grid(P, K, D);
    const dim3 g3((unsigned)grid, (unsigned)((K + Kc - 1) / Kc));
    hipStream_t s = (hipStream_t)stream;
    const uintptr_t al = (uintptr_t)z | (uintptr_t)g_out | (uintptr_t)dz;
    if (dw_slabs && K <= 64 && (D == 16 || D == 32 || D == 64) && (H * W) % 64 == 0 && (al & 15) == 0) {
        const unsigned NC = (unsigned)(P >> 6);
#define DM_VQ_BWD2(DD, WGS)                                                                                    \
    hipLaunchKernelGGL((vq_backward_mfma_kernel<DD, WGS>), dim3((unsigned)grid), dim3(VQ_BWD2_BLOCK), 0, s, z, codebook, \
                       (const long long *)idx, g_out, g_loss_dev, commitment_cost, dz, dw_slabs, K, H * W, NC)
        switch (D) {
        case 16: DM_VQ_BWD2(16, 4); break;
        case 32: DM_VQ_BWD2(32, 2); break;
        default: DM_VQ_BWD2(64, 2); break;
        }
#undef DM_VQ_BWD2
        return dm_launch_status(who);
    }
#define DM_VQ_BWD(DD)                                                                                          \
    if (lds > 48 * 1024) {                                                                                     \
        const hipError_t ea = hipFuncSetAttribute((const void *)vq_backward_kernel<DD>,                        \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
        if (ea != hipSuccess) {                                                                                \
            dm_set_error("%s: cannot reserve %zu bytes of LDS: %s", who, (size_t)lds, hipGetErrorString(ea));  \
            return (int)ea;                                                                                    \
        }                                                                                                      \
    }                                                                                                          \
    hipLaunchKernelGGL(vq_backward_kernel<DD>, g3, dim3(VQ_BWD_BLOCK), lds, s, z, codebook,                    \
                       (const long long *)idx, g_out, g_loss_dev, commitment_cost, dz, dw, dw_slabs, K, H * W, P, Kc)
    if (!vq_dim_built(D)) {
        if (lds > 48 * 1024) {
            const hipError_t ea = hipFuncSetAttribute((const void *)vq_backward_any_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (ea != hipSuccess) {
                dm_set_error("%s: cannot reserve %zu bytes of LDS: %s", who, (size_t)lds, hipGetErrorString(ea));
                return (int)ea;
            }
        }
        hipLaunchKernelGGL(vq_backward_any_kernel, g3, dim3(VQ_BWD_BLOCK), lds, s, z, codebook, (const long long *)idx, g_out,
                           g_loss_dev, commitment_cost, dz, dw, dw_slabs, K, D, H * W, P, Kc);
        return dm_launch_status(who);
    }
    switch (D) {
    case 8: DM_VQ_BWD(8); break;
    case 16: DM_VQ_BWD(16); break;
    case 32: DM_VQ_BWD(32); break;
    case 64: DM_VQ_BWD(64); break;
    default: DM_VQ_BWD(128); break;
    }
#undef DM_VQ_BWD
    return dm_launch_status(who);
}
}  // namespace

extern "C" int dm_vq_backward(const float *z, const float *codebook, const int64_t *idx,
                              const float *g_out, const float *g_loss_dev, float commitment_cost,
                              float *dz, float *dw, int B, int D, int K, int H, int W, void *stream)
{
    return vq_backward_launch("dm_vq_backward", z, codebook, idx, g_out, g_loss_dev, commitment_cost, dz, dw, nullptr,
                              B, D, K, H, W, stream);
}

extern "C" int dm_vq_backward_num_slabs(int64_t positions, int K, int D)
{
    DM_REQUIRE(positions > 0 && K > 0 && D > 0, "dm_vq_backward_num_slabs: bad argument");
    return vq_backward_grid(positions, K, D);
}

extern "C" int dm_vq_backward_slabs(const float *z, const float *codebook, const int64_t *idx,
                                    const float *g_out, const float *g_loss_dev, float commitment_cost,
                                    float *dz, float *dw_slabs, int B, int D, int K, int H, int W, void *stream)
{
    return vq_backward_launch("dm_vq_backward_slabs", z, codebook, idx, g_out, g_loss_dev, commitment_cost, dz, nullptr,
                              dw_slabs, B, D, K, H, W, stream);
}
