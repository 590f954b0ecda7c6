/* Plain-C restatement of the reference's 2D->3D lift predictors and postprocess_masks.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/ as a checker and by
 * bench.py's cpu_baseline leg ("kind": "port").  Never linked into the product.
 *
 * Follows (citations into /root/reference):
 *   orc_lift_mesh_soft    model/components.py:220-277   (HumanContact3DPredictor)
 *   orc_lift_mesh_thresh  model/components.py:392-424,445-489 (ObjectMeshContact3DPredictor)
 *   orc_lift_points       model/components.py:289-347   (ObjectPCAfford3DPredictor)
 *   orc_postprocess       model/segment_anything/modeling/sam.py:137-172
 * Summation order = the reference's: three scatter passes (k = 0,1,2), pixels ascending.
 * Build: make -C oracle   ->  oracle/_build/liboracle.so   (gcc -O2 -ffp-contract=off)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* one view: votes/cnt are scratch [nv]; accumulates into pred/nviews [nv] */
static void bary_view(const float *m /*[hw] or NULL*/, const float *logit, int mode, float param,
                      const int32_t *vid, const float *bary, long hw, int nv,
                      float *votes, float *cnt, float *pred, float *nviews, float *mbuf) {
    /* mode 0: soft  (clamp +-param, sigmoid, all pixels)
     * mode 1: thresh (sigmoid, keep p > param) */
    long any = 0;
    (void)m;
    for (long p = 0; p < hw; ++p) {
        float x = logit[p];
        float s;
        if (mode == 0) {
            x = x < -param ? -param : (x > param ? param : x);
            s = sigmoidf_(x);
        } else {
            s = sigmoidf_(x);
            if (!(s > param)) s = -1.0f; /* deselected */
        }
        mbuf[p] = s;
    }
    memset(votes, 0, sizeof(float) * nv);
    memset(cnt, 0, sizeof(float) * nv);
    for (int k = 0; k < 3; ++k) {
        for (long p = 0; p < hw; ++p) {
            float s = mbuf[p];
            if (mode == 1 && s < 0.0f) continue;
            int32_t a = vid[3 * p], b = vid[3 * p + 1], c = vid[3 * p + 2];
            if (a < 0 || a >= nv || b < 0 || b >= nv || c < 0 || c >= nv) continue;
            int32_t id = vid[3 * p + k];
            float w = bary[3 * p + k];
            float t = w * s;
            votes[id] += t;
            cnt[id] += w;
            any = 1;
        }
    }
    if (!any) return;
    for (int i = 0; i < nv; ++i) {
        if (cnt[i] > 0.0f) {
            pred[i] += votes[i] / cnt[i];
            nviews[i] += 1.0f;
        }
    }
}

int orc_lift_mesh_soft(const float *logits /*[B,V,hw]*/, const int32_t *vid /*[V,hw,3]*/,
                       const float *bary /*[V,hw,3]*/, int B, int V, long hw, int nv, float clampv,
                       float *pred /*[B,nv]*/, float *nviews /*[B,nv]*/) {
    float *votes = (float *)malloc(sizeof(float) * nv * 2);
    float *mbuf = (float *)malloc(sizeof(float) * hw);
    if (!votes || !mbuf) return -1;
    memset(pred, 0, sizeof(float) * (size_t)B * nv);
    memset(nviews, 0, sizeof(float) * (size_t)B * nv);
    for (int b = 0; b < B; ++b) {
        for (int v = 0; v < V; ++v)
            bary_view(NULL, logits + ((size_t)b * V + v) * hw, 0, clampv, vid + (size_t)v * hw * 3,
                      bary + (size_t)v * hw * 3, hw, nv, votes, votes + nv, pred + (size_t)b * nv,
                      nviews + (size_t)b * nv, mbuf);
        for (int i = 0; i < nv; ++i) {
            float *p = pred + (size_t)b * nv + i;
            float n = nviews[(size_t)b * nv + i];
            if (n > 0.0f) *p = *p / n;
            *p = *p < 0.0f ? 0.0f : (*p > 1.0f ? 1.0f : *p);
        }
    }
    free(votes);
    free(mbuf);
    return 0;
}

int orc_lift_mesh_thresh(const float *logits /*[V,hw]*/, const int32_t *vid, const float *bary, int V,
                         long hw, int nv, float thr, float *pred /*[nv]*/, float *nviews /*[nv]*/) {
    float *votes = (float *)malloc(sizeof(float) * nv * 2);
    float *mbuf = (float *)malloc(sizeof(float) * hw);
    if (!votes || !mbuf) return -1;
    memset(pred, 0, sizeof(float) * nv);
    memset(nviews, 0, sizeof(float) * nv);
    for (int v = 0; v < V; ++v)
        bary_view(NULL, logits + (size_t)v * hw, 1, thr, vid + (size_t)v * hw * 3, bary + (size_t)v * hw * 3,
                  hw, nv, votes, votes + nv, pred, nviews, mbuf);
    for (int i = 0; i < nv; ++i)
        if (nviews[i] > 0.0f) pred[i] /= nviews[i];
    free(votes);
    free(mbuf);
    return 0;
}

int orc_lift_points(const float *probs /*[B,V,hw]*/, const int32_t *pid /*[B,V,hw]*/, int B, int V, long hw,
                    int np_, float *pred /*[B,np]*/, float *nviews /*[B,np]*/) {
    float *votes = (float *)malloc(sizeof(float) * np_ * 2);
    if (!votes) return -1;
    float *cnt = votes + np_;
    memset(pred, 0, sizeof(float) * (size_t)B * np_);
    memset(nviews, 0, sizeof(float) * (size_t)B * np_);
    for (int b = 0; b < B; ++b) {
        for (int v = 0; v < V; ++v) {
            const float *pr = probs + ((size_t)b * V + v) * hw;
            const int32_t *mp = pid + ((size_t)b * V + v) * hw;
            memset(votes, 0, sizeof(float) * np_ * 2);
            for (long p = 0; p < hw; ++p) {
                int32_t id = mp[p];
                if (id == -1) continue;
                votes[id] += pr[p];
                cnt[id] += 1.0f;
            }
            for (int i = 0; i < np_; ++i)
                if (cnt[i] > 0.0f) {
                    pred[(size_t)b * np_ + i] += votes[i] / cnt[i];
                    nviews[(size_t)b * np_ + i] += 1.0f;
                }
        }
        for (int i = 0; i < np_; ++i)
            if (nviews[(size_t)b * np_ + i] > 0.0f) pred[(size_t)b * np_ + i] /= nviews[(size_t)b * np_ + i];
    }
    free(votes);
    return 0;
}

/* F.interpolate(mode="bilinear", align_corners=False), fp32 */
static void resize_plane(const float *src, int h, int w, float *dst, int oh, int ow) {
    if (h == oh && w == ow) {
        memcpy(dst, src, sizeof(float) * (size_t)h * w);
        return;
    }
    const float sy = (float)h / (float)oh, sx = (float)w / (float)ow;
    for (int y = 0; y < oh; ++y) {
        float fy = ((float)y + 0.5f) * sy - 0.5f;
        if (fy < 0.0f) fy = 0.0f;
        int y0 = (int)fy;
        if (y0 > h - 1) y0 = h - 1;
        int y1 = y0 + 1 < h ? y0 + 1 : h - 1;
        float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
        for (int x = 0; x < ow; ++x) {
            float fx = ((float)x + 0.5f) * sx - 0.5f;
            if (fx < 0.0f) fx = 0.0f;
            int x0 = (int)fx;
            if (x0 > w - 1) x0 = w - 1;
            int x1 = x0 + 1 < w ? x0 + 1 : w - 1;
            float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
            float t = src[(size_t)y0 * w + x0] * lx0 + src[(size_t)y0 * w + x1] * lx1;
            float b = src[(size_t)y1 * w + x0] * lx0 + src[(size_t)y1 * w + x1] * lx1;
            dst[(size_t)y * ow + x] = t * ly0 + b * ly1;
        }
    }
}

int orc_postprocess(const float *low /*[n,h,w]*/, int n, int h, int w, int img, int in_h, int in_w, int oh,
                    int ow, float *out /*[n,oh,ow]*/) {
    float *big = (float *)malloc(sizeof(float) * (size_t)img * img);
    float *crop = (float *)malloc(sizeof(float) * (size_t)in_h * in_w);
    if (!big || !crop) return -1;
    for (int i = 0; i < n; ++i) {
        resize_plane(low + (size_t)i * h * w, h, w, big, img, img);
        for (int y = 0; y < in_h; ++y) memcpy(crop + (size_t)y * in_w, big + (size_t)y * img, sizeof(float) * in_w);
        resize_plane(crop, in_h, in_w, out + (size_t)i * oh * ow, oh, ow);
    }
    free(big);
    free(crop);
    return 0;
}
