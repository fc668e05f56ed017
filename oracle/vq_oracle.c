/*
 * oracle/vq_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference VectorQuantizer arithmetic
 * (reference: HiddenStateExtractor/vq_vae.py:52-116, identical text in
 * HiddenStateExtractor/vae.py:39-103).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (dynamorph_amd/) never does.
 *
 * Pinned against tests/golden/g4_vq_indices.npz, g5_vq_forward.npz,
 * g6_vq_backward.npz, g9_vq_*.npz (vectors produced by importing the reference
 * in the build container, tests/golden/make_golden.py).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no FMA contraction, the
 * reference materialises (z-e) and (z-e)^2 in fp32 before summing.)
 *
 * Layouts are the reference's: z / quantized are NCHW contiguous
 * (B, D, H, W); codebook is (K, D) row-major; indices are int64 (B, H, W).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* vq_vae.py:65 -- dist[b,k,h,w] = sum_d (z[b,d,h,w] - e[k,d])^2.
 * ATen's CPU reduction over the (non-innermost) d axis of the materialised
 * (B,K,D,H,W) tensor adds the squares sequentially inside blocks of 16
 * consecutive d, then adds the block sums sequentially (SURVEY.md section 7;
 * verified bit-equal against g4/g9 dist_sample0 for D = 16 and D = 64). */
static inline float vq_dist(const float *zpos, int64_t zstride, const float *e, int D)
{
    float total = 0.0f;
    int first_block = 1;
    for (int d0 = 0; d0 < D; d0 += 16) {
        int d1 = d0 + 16 < D ? d0 + 16 : D;
        float acc = 0.0f;
        int first = 1;
        for (int d = d0; d < d1; ++d) {
            float diff = zpos[(int64_t)d * zstride] - e[d];
            float sq = diff * diff;
            if (first) { acc = sq; first = 0; } else { acc = acc + sq; }
        }
        if (first_block) { total = acc; first_block = 0; } else { total = total + acc; }
    }
    return total;
}

/* vq_vae.py:68 -- argmax(-dist, 1): first index of the maximum; a NaN compares
 * as the maximum (torch.argmax), the first NaN wins. */
static inline int vq_better(float cand, float best)
{
    /* candidate = -dist_k, best = -dist_best */
    if (isnan(best)) return 0;
    if (isnan(cand)) return 1;
    return cand > best;
}

/* Distances for one sample, (K,H,W) layout, for pinning against dist_sample0. */
void oracle_vq_distances(const float *z, const float *cb, float *dist,
                         int D, int K, int H, int W)
{
    const int64_t hw = (int64_t)H * W;
    for (int k = 0; k < K; ++k)
        for (int64_t p = 0; p < hw; ++p)
            dist[(int64_t)k * hw + p] = vq_dist(z + p, hw, cb + (int64_t)k * D, D);
}

/* vq_vae.py:90-103 encode_inputs (V1+V2). */
void oracle_vq_encode(const float *z, const float *cb, int64_t *idx,
                      int B, int D, int K, int H, int W)
{
    const int64_t hw = (int64_t)H * W;
    for (int b = 0; b < B; ++b) {
        const float *zb = z + (int64_t)b * D * hw;
        for (int64_t p = 0; p < hw; ++p) {
            float best = -vq_dist(zb + p, hw, cb, D);
            int64_t bi = 0;
            for (int k = 1; k < K; ++k) {
                float cand = -vq_dist(zb + p, hw, cb + (int64_t)k * D, D);
                if (vq_better(cand, best)) { best = cand; bi = k; }
            }
            idx[(int64_t)b * hw + p] = bi;
        }
    }
}

/* vq_vae.py:105-116 decode_inputs (V3), written out contiguous NCHW. */
void oracle_vq_decode(const int64_t *idx, const float *cb, float *q,
                      int B, int D, int K, int H, int W)
{
    (void)K;
    const int64_t hw = (int64_t)H * W;
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d)
            for (int64_t p = 0; p < hw; ++p)
                q[((int64_t)b * D + d) * hw + p] = cb[idx[(int64_t)b * hw + p] * D + d];
}

/* vq_vae.py:52-84 forward.  out = z + (q - z) (V4, NOT bit-equal to q);
 * loss = mse + commitment_cost * mse evaluated in fp32 as the reference does
 * (two identical mse values, V5); perplexity from the code histogram (V6).
 * The mse mean itself is accumulated in double: ATen's own fp32 cascade order
 * is not reproduced, the comparison tolerance in tests is 1e-6 relative. */
void oracle_vq_forward(const float *z, const float *cb, float commitment_cost,
                       int64_t *idx, float *out, float *loss, float *perplexity,
                       int64_t *hist, int B, int D, int K, int H, int W)
{
    const int64_t hw = (int64_t)H * W;
    const int64_t P = (int64_t)B * hw;
    oracle_vq_encode(z, cb, idx, B, D, K, H, W);
    memset(hist, 0, sizeof(int64_t) * (size_t)K);
    double sse = 0.0;
    for (int b = 0; b < B; ++b) {
        for (int64_t p = 0; p < hw; ++p) {
            int64_t k = idx[(int64_t)b * hw + p];
            hist[k] += 1;
            for (int d = 0; d < D; ++d) {
                int64_t o = ((int64_t)b * D + d) * hw + p;
                float qv = cb[k * D + d];
                float diff = qv - z[o];
                out[o] = z[o] + diff;
                float sq = diff * diff;
                sse += (double)sq;
            }
        }
    }
    float mse = (float)(sse / (double)(P * D));
    *loss = mse + commitment_cost * mse;
    float ent = 0.0f;
    for (int k = 0; k < K; ++k) {
        float pk = (float)hist[k] / (float)P;
        ent = ent + pk * logf(pk + 1e-10f);
    }
    *perplexity = expf(-ent);
}

/* Gradients of (sum(out * g_out) + g_loss * loss) as autograd derives them from
 * vq_vae.py:71-76:  d/dz = g_out + g_loss * 2*cc*(z-q)/N ;
 * d/dw[k] = sum_{pos: idx=k} g_loss * 2*(q-z)/N  (embedding scatter-add),
 * N = B*D*H*W.  The decoder-side gradient g_out does not reach w (detach). */
void oracle_vq_backward(const float *z, const float *cb, const int64_t *idx,
                        const float *g_out, float g_loss, float commitment_cost,
                        float *dz, float *dw, int B, int D, int K, int H, int W)
{
    const int64_t hw = (int64_t)H * W;
    const double N = (double)B * D * (double)hw;
    /* double accumulators for the scatter-add: order-independent reference */
    for (int64_t i = 0; i < (int64_t)K * D; ++i) dw[i] = 0.0f;
    for (int k = 0; k < K; ++k) {
        for (int d = 0; d < D; ++d) {
            double acc = 0.0;
            for (int b = 0; b < B; ++b)
                for (int64_t p = 0; p < hw; ++p)
                    if (idx[(int64_t)b * hw + p] == k) {
                        int64_t o = ((int64_t)b * D + d) * hw + p;
                        acc += 2.0 * ((double)cb[(int64_t)k * D + d] - (double)z[o]) / N;
                    }
            dw[(int64_t)k * D + d] = (float)(acc * (double)g_loss);
        }
    }
    const float scale = (float)(2.0 * (double)commitment_cost / N) * g_loss;
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d)
            for (int64_t p = 0; p < hw; ++p) {
                int64_t o = ((int64_t)b * D + d) * hw + p;
                float qv = cb[idx[(int64_t)b * hw + p] * D + d];
                dz[o] = g_out[o] + scale * (z[o] - qv);
            }
}
