// feed.hip -- feeding the training step from a dataset that lives in HBM.
//
// Reference: run_training.py:504-532 (the batch loop: `dataset[ids][0].to(device)`, get_relation_tensor, get_mask),
// :396-403 (per-sample flip / rot90 drawn from numpy's global generator), :335-374 (relation block, mask plane).
// The reference gathers every batch on the host and copies it over PCIe; on a 288 GB part the whole dataset (and its
// masks, and the CSR relation matrix) is uploaded once and a batch is ONE kernel that gathers the chosen samples,
// applies each sample's flip / rotation and writes straight into the input buffer the captured step replays on.
#include "dm_common.h"
#include <stdlib.h>

namespace {

constexpr int FT = 64;          // a workgroup moves one 64 x 64 tile of one (sample, channel) plane

// Source coordinates of output (y, x) under out = rot90(flip(in, f), k): the dihedral map of run_training.py:396-403
// (torch.flip dims (1,) / (2,) of a (C, H, W) patch, torch.rot90 k times counter-clockwise over dims [1, 2]) is affine,
//   k = 0: (y, x)   1: (x, H-1-y)   2: (H-1-y, H-1-x)   3: (H-1-x, y);   f = 1: sy -> H-1-sy;   f = 2: sx -> H-1-sx,
// kept as six coefficients in {-1, 0, 1, H-1} (no branch, no table in memory: a switch over k ended up in scratch).
struct Dihedral { int ay, by, cy, ax, bx, cx; };

__device__ __forceinline__ Dihedral dihedral(int k, int f, int H)
{
    Dihedral d;
    d.ay = (k == 0) - (k == 2); d.by = (k == 1) - (k == 3); d.cy = k >= 2 ? H - 1 : 0;
    d.ax = (k == 3) - (k == 1); d.bx = (k == 0) - (k == 2); d.cx = (k == 1 || k == 2) ? H - 1 : 0;
    if (f == 1) { d.ay = -d.ay; d.by = -d.by; d.cy = H - 1 - d.cy; }
    if (f == 2) { d.ax = -d.ax; d.bx = -d.bx; d.cx = H - 1 - d.cx; }
    return d;
}

__device__ __forceinline__ void source_of(int y, int x, int H, int k, int f, int &sy, int &sx)
{
    const Dihedral d = dihedral(k, f, H);
    sy = d.ay * y + d.by * x + d.cy;
    sx = d.ax * y + d.bx * x + d.cx;
}

// H % 64 == 0.  A dihedral map sends 64 x 64 tiles to 64 x 64 tiles, so the source tile of an output tile is read in
// full rows (16 B per lane, 256 B per row segment, all four loads of a thread in flight together), parked in LDS (row
// stride 65: the transposing reads of the odd rotations hit 64 different banks) and written out in full rows again.
__global__ __launch_bounds__(256) void gather_augment_tiled_kernel(const float *__restrict__ src, float *__restrict__ out,
                                                                   const int *__restrict__ ids,
                                                                   const int *__restrict__ flip_code,
                                                                   const int *__restrict__ rot_code, int C, int H,
                                                                   long long n_src)
{
    __shared__ float tile[FT * (FT + 1)];
    const int tiles = H / FT;
    int w = blockIdx.x;
    const int tx = w % tiles; w /= tiles;
    const int ty = w % tiles; w /= tiles;
    const int c = w % C;
    const int b = w / C;
    const long long s = ids ? (long long)ids[b] : (long long)b;
    const int k = rot_code ? (rot_code[b] & 3) : 0;
    const int f = flip_code ? flip_code[b] : 0;
    float *dst = out + ((long long)b * C + c) * H * H + (long long)(ty * FT) * H + tx * FT;
    const int col = (threadIdx.x & 15) * 4, row0 = threadIdx.x >> 4;
    if (s < 0 || s >= n_src) {           // an id outside the dataset: zeros, never a stray read
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4 *>(dst + (long long)(row0 + 16 * j) * H + col) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const Dihedral d = dihedral(k, f, H);
    // where the output tile's corner and its opposite corner come from: the source tile is the box between them
    const int y0 = ty * FT, x0 = tx * FT;
    const int ay = d.ay * y0 + d.by * x0 + d.cy, ax = d.ax * y0 + d.bx * x0 + d.cx;
    const int by = ay + (d.ay + d.by) * (FT - 1), bx = ax + (d.ax + d.bx) * (FT - 1);
    const int sy0 = ay < by ? ay : by, sx0 = ax < bx ? ax : bx;
    const float *pl = src + (s * C + c) * H * H + (long long)sy0 * H + sx0;
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4 *>(pl + (long long)(row0 + 16 * j) * H + col);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float *t = tile + (row0 + 16 * j) * (FT + 1) + col;
        t[0] = v[j].x; t[1] = v[j].y; t[2] = v[j].z; t[3] = v[j].w;
    }
    __syncthreads();
    // LDS address of output (r, col + e) inside the tile: linear in r and e
    const int base = (ay - sy0) * (FT + 1) + (ax - sx0);              // output (0, 0) of the tile
    const int dr = d.ay * (FT + 1) + d.ax, dc = d.by * (FT + 1) + d.bx;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = row0 + 16 * j;
        const float *t = tile + base + r * dr + col * dc;
        f32x4 o;
        o.x = t[0]; o.y = t[dc]; o.z = t[2 * dc]; o.w = t[3 * dc];
        *reinterpret_cast<f32x4 *>(dst + (long long)r * H + col) = o;
    }
}

// Any H: one element per thread and trip (coverage form; the odd rotations read with a stride of one row).
__global__ __launch_bounds__(256) void gather_augment_kernel(const float *__restrict__ src, float *__restrict__ out,
                                                             const int *__restrict__ ids, const int *__restrict__ flip_code,
                                                             const int *__restrict__ rot_code, int C, int H,
                                                             long long n_src, long long total)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % H), y = (int)((i / H) % H);
        const long long plane = i / ((long long)H * H);
        const int b = (int)(plane / C), c = (int)(plane % C);
        const long long s = ids ? (long long)ids[b] : (long long)b;
        int sy, sx;
        source_of(y, x, H, rot_code ? (rot_code[b] & 3) : 0, flip_code ? flip_code[b] : 0, sy, sx);
        out[i] = (s < 0 || s >= n_src) ? 0.f : src[(s * C + c) * H * H + (long long)sy * H + sx];
    }
}

// out[b] = src[ids[b]] for rows of `row` floats (row % 4 == 0, 16-byte aligned): the cell masks of a batch.
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src, float *__restrict__ out,
                                                          const int *__restrict__ ids, long long row4, long long n_src,
                                                          long long total4)
{
    const f32x4 *s4 = reinterpret_cast<const f32x4 *>(src);
    f32x4 *o4 = reinterpret_cast<f32x4 *>(out);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / row4, e = i - b * row4;
        const long long s = ids ? (long long)ids[b] : b;
        o4[i] = (s < 0 || s >= n_src) ? f32x4{0.f, 0.f, 0.f, 0.f} : s4[s * row4 + e];
    }
}

// pos[ids[j]] = stamp * 2^32 + j: which column of the block a sample id lands in, valid for this call only (a stale
// entry carries an older stamp), so the N-entry table is never cleared.
__global__ void csr_mark_kernel(long long *__restrict__ pos, const int *__restrict__ ids, int B, long long n,
                                long long stamp)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < B) {
        const long long s = ids[j];
        if (s >= 0 && s < n) pos[s] = (stamp << 32) | (long long)j;
    }
}

// One workgroup per row of the block: zero the row, then walk CSR row ids[i] and drop every entry whose column is in
// the batch at its block column.  (relation_mat[ids, :][:, ids].todense(), run_training.py:348-351; duplicate entries
// of a column are summed on the host before the upload, as scipy's todense() does.)
__global__ __launch_bounds__(256) void csr_block_kernel(const long long *__restrict__ indptr, const int *__restrict__ indices,
                                                        const float *__restrict__ data, const long long *__restrict__ pos,
                                                        const int *__restrict__ ids, int B, long long n, long long stamp,
                                                        float *__restrict__ out)
{
    const int i = blockIdx.x;
    float *row = out + (long long)i * B;
    for (int j = threadIdx.x; j < B; j += blockDim.x) row[j] = 0.f;
    __syncthreads();
    const long long s = ids[i];
    if (s < 0 || s >= n) return;
    const long long lo = indptr[s], hi = indptr[s + 1];
    for (long long e = lo + threadIdx.x; e < hi; e += blockDim.x) {
        const long long col = indices[e];
        if (col < 0 || col >= n) continue;
        const long long p = pos[col];
        if ((p >> 32) == stamp) row[(int)(p & 0xffffffffLL)] = data[e];
    }
}

}  // namespace

extern "C" int dm_gather_augment(const float *src, int64_t n_src, const int32_t *ids, const int32_t *flip_code,
                                 const int32_t *rot_code, float *out, int B, int C, int H, void *stream)
{
    DM_REQUIRE(src && out && B > 0 && C > 0 && H > 0 && n_src > 0, "dm_gather_augment: bad argument");
    DM_REQUIRE((const void *)src != (const void *)out, "dm_gather_augment: in-place not supported");
    DM_REQUIRE(ids || (int64_t)B <= n_src, "dm_gather_augment: B exceeds the number of source samples");
    if (H % FT == 0) {
        // (the tiled form moves whole 16-byte row segments on both sides)
        DM_REQUIRE((((uintptr_t)src | (uintptr_t)out) & 15) == 0, "dm_gather_augment: src and out must be 16-byte aligned");
        const long long blocks = (long long)B * C * (H / FT) * (H / FT);
        DM_REQUIRE(blocks < (1LL << 31), "dm_gather_augment: batch too large for one launch");
        hipLaunchKernelGGL(gather_augment_tiled_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, out,
                           (const int *)ids, (const int *)flip_code, (const int *)rot_code, C, H, (long long)n_src);
    } else {
        const long long total = (long long)B * C * H * H;
        const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL(gather_augment_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, out, (const int *)ids,
                           (const int *)flip_code, (const int *)rot_code, C, H, (long long)n_src, total);
    }
    return dm_launch_status("dm_gather_augment");
}

extern "C" int dm_gather_rows(const float *src, int64_t n_src, const int32_t *ids, float *out, int B, int64_t row_floats,
                              void *stream)
{
    DM_REQUIRE(src && out && B > 0 && row_floats > 0 && n_src > 0, "dm_gather_rows: bad argument");
    DM_REQUIRE(row_floats % 4 == 0, "dm_gather_rows: rows must be a multiple of 4 floats");
    DM_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)out & 15) == 0, "dm_gather_rows: pointers must be 16-byte aligned");
    DM_REQUIRE(ids || (int64_t)B <= n_src, "dm_gather_rows: B exceeds the number of source rows");
    const long long total4 = (long long)B * (row_floats / 4);
    const int grid = (int)((total4 + 255) / 256 < 16384 ? (total4 + 255) / 256 : 16384);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, out, (const int *)ids,
                       (long long)(row_floats / 4), (long long)n_src, total4);
    return dm_launch_status("dm_gather_rows");
}

extern "C" int dm_csr_block(const int64_t *indptr, const int32_t *indices, const float *data, int64_t n, const int32_t *ids,
                            int B, int64_t *pos, int64_t stamp, float *out, void *stream)
{
    DM_REQUIRE(indptr && ids && pos && out && n > 0 && B > 0, "dm_csr_block: bad argument");
    DM_REQUIRE(stamp > 0 && stamp < (1LL << 31), "dm_csr_block: stamp must be in [1, 2^31)");
    hipLaunchKernelGGL(csr_mark_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, (long long *)pos,
                       (const int *)ids, B, (long long)n, (long long)stamp);
    hipLaunchKernelGGL(csr_block_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const long long *)indptr,
                       (const int *)indices, data, (const long long *)pos, (const int *)ids, B, (long long)n, (long long)stamp,
                       out);
    return dm_launch_status("dm_csr_block");
}

// Host only.  run_training.py:396-403 draws, per sample and in this order, np.random.choice([0, 1, 2]) and
// np.random.choice([0, 1, 2, 3]).  numpy's legacy generator serves both from 32-bit words of the Mersenne twister by
// masked rejection: word & 3, redrawn while it exceeds the largest value (so only the flip draw ever rejects).  `raw` is
// a run of such words (np.random.randint(0, 2**32, dtype=uint32) hands them out one per value); the codes of n samples
// are parsed from it exactly as n interleaved choice() calls would have consumed them.  Returns the number of words
// used (the caller rewinds the generator and skips that many), or -1 when `raw` is too short.
extern "C" int64_t dm_augment_codes(const uint32_t *raw, int64_t n_raw, int64_t n, int32_t *flip_code, int32_t *rot_code)
{
    if (!raw || !flip_code || !rot_code || n < 0 || n_raw < 0) return -1;
    int64_t i = 0;
    for (int64_t s = 0; s < n; ++s) {
        while (i < n_raw && (raw[i] & 3u) == 3u) ++i;
        if (i + 1 >= n_raw) return -1;
        flip_code[s] = (int32_t)(raw[i++] & 3u);
        rot_code[s] = (int32_t)(raw[i++] & 3u);
    }
    return i;
}

// Host only.  run_training.py:97-140 (reorder_with_trajectories) puts the samples of a trajectory next to each other:
// while samples remain, np.random.choice(list(pool)) picks one -- the pool is a Python set of small ints, whose list is
// the remaining ids in ascending order, and choice() is one masked-rejection draw below len(pool) from the legacy
// generator's 32-bit words (none when one sample remains) -- and the pick is followed by everything reachable from it over
// the ADJACENT pairs (breadth first, neighbours in the order the relation dict lists them).  The reference rebuilds the
// list for every pick (quadratic in the sample count: hours for 10^5 patches); here the k-th remaining id comes from a
// Fenwick tree of counts.  adj_ptr / adj_idx: the adjacency of the value-2 pairs as CSR over the first id, insertion
// order kept.  Returns the words consumed; -1: `raw` too short; -2: a reached sample has no adjacency row (the
// reference's KeyError on relation_dict[elem]), -3: a reached sample had left the pool already (its KeyError on
// inds_pool.remove); *err_id names it.
extern "C" int64_t dm_reorder_with_trajectories(const uint32_t *raw, int64_t n_raw, int64_t n, const int64_t *adj_ptr,
                                                const int64_t *adj_idx, int64_t *order, int64_t *err_id)
{
    if (!raw || !adj_ptr || !order || n < 0 || n_raw < 0) return -4;
    if (n == 0) return 0;
    int sh = 0;
    while ((1LL << sh) < n) ++sh;                       // tree over [1, 2^sh]
    const int64_t top = 1LL << sh;
    int64_t *tree = (int64_t *)malloc((size_t)(top + 1) * sizeof(int64_t));
    unsigned char *gone = (unsigned char *)calloc((size_t)n, 1);
    int64_t *stamp = (int64_t *)calloc((size_t)n, sizeof(int64_t));
    int64_t *traj = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    if (!tree || !gone || !stamp || !traj) { free(tree); free(gone); free(stamp); free(traj); return -4; }
    for (int64_t i = 1; i <= top; ++i) tree[i] = 0;
    for (int64_t i = 1; i <= top; ++i) {                // linear build: every node passes its sum to its parent
        if (i <= n) tree[i] += 1;
        const int64_t p = i + (i & -i);
        if (p <= top) tree[p] += tree[i];
    }
    int64_t used = 0, out = 0, left = n, rc = 0, pick_no = 0;
    while (left > 0 && rc == 0) {
        uint32_t k = 0;
        const uint32_t rng = (uint32_t)(left - 1);
        if (rng != 0) {
            uint32_t mask = rng;
            mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
            for (;;) {
                if (used >= n_raw) { rc = -1; break; }
                k = raw[used++] & mask;
                if (k <= rng) break;
            }
            if (rc) break;
        }
        // the (k + 1)-th remaining id: descend the tree
        int64_t pos = 0, need = (int64_t)k + 1;
        for (int64_t step = top; step > 0; step >>= 1)
            if (pos + step <= top && tree[pos + step] < need) { pos += step; need -= tree[pos]; }
        const int64_t first = pos;                       // 0-based id
        ++pick_no;
        int64_t len = 0;
        traj[len++] = first;
        stamp[first] = pick_no;
        if (adj_ptr[first + 1] > adj_ptr[first]) {       // in relation_dict: follow the trajectory (the list IS the queue)
            for (int64_t q = 0; q < len && rc == 0; ++q) {
                const int64_t e = traj[q];
                if (adj_ptr[e + 1] == adj_ptr[e]) { rc = -2; if (err_id) *err_id = e; break; }
                for (int64_t a = adj_ptr[e]; a < adj_ptr[e + 1]; ++a) {
                    const int64_t v = adj_idx[a];
                    if (v < 0 || v >= n) { rc = -3; if (err_id) *err_id = v; break; }   // never in the pool
                    if (stamp[v] == pick_no) continue;   // `if not e in traj`
                    stamp[v] = pick_no;
                    traj[len++] = v;                     // (distinct ids below n: len <= n)
                }
            }
        }
        for (int64_t q = 0; q < len && rc == 0; ++q) {   // inds_in_order.extend(traj); inds_pool.remove(e)
            const int64_t e = traj[q];
            if (gone[e]) { rc = -3; if (err_id) *err_id = e; break; }
            gone[e] = 1;
            order[out++] = e;
            for (int64_t i = e + 1; i <= top; i += i & -i) tree[i] -= 1;
            --left;
        }
    }
    free(tree); free(gone); free(stamp); free(traj);
    return rc ? rc : used;
}
