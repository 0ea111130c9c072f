/* ORACLE -- test infrastructure only (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
 * Never linked, imported or called by the product path.
 *
 * Plain-C restatement of the reference's binary morphology and selection arithmetic.
 * Each function cites the reference lines it follows (paths relative to
 * /root/reference/rsuper_train/training/losses_foundation.py).
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* Ball structuring element, create_ball_kernel(kernel_size) as used by dilate_volume_conv
 * (:81-85, :1161-1230): diameter_odd = k (k odd), radius = k/2.0, a voxel offset (dz,dy,dx)
 * belongs to the ball iff dz^2+dy^2+dx^2 <= radius^2.  The conv kernel edge is 1.2x larger
 * but everything outside the ball is zero, so only the ball offsets matter.
 * 4*(dz^2+dy^2+dx^2) <= k^2 keeps the test in integers. */
static int ball_offsets(int k, int **out) {
    int r = k / 2 + 1, n = 0;
    int cap = (2 * r + 1) * (2 * r + 1) * (2 * r + 1);
    int *o = (int *)malloc(sizeof(int) * 3 * cap);
    for (int dz = -r; dz <= r; ++dz)
        for (int dy = -r; dy <= r; ++dy)
            for (int dx = -r; dx <= r; ++dx)
                if (4 * (dz * dz + dy * dy + dx * dx) <= k * k) {
                    o[3 * n] = dz; o[3 * n + 1] = dy; o[3 * n + 2] = dx; ++n;
                }
    *out = o;
    return n;
}

int oracle_ball_nnz(int k) {
    int *o; int n = ball_offsets(k, &o); free(o); return n;
}

/* One pass of dilate_volume_conv (:50-99): depthwise conv with the binary ball, then >0.
 * Sums of 0/1 are positive iff any covered voxel is set, so an OR is exact.
 * vol/out: nvol independent volumes of D*H*W bytes (0/1). */
static void dilate_pass(const uint8_t *vol, uint8_t *out, long nvol, int D, int H, int W, int k) {
    int *off; int n = ball_offsets(k, &off);
    long V = (long)D * H * W;
    memset(out, 0, (size_t)(nvol * V));
    for (long v = 0; v < nvol; ++v) {
        const uint8_t *src = vol + v * V; uint8_t *dst = out + v * V;
        for (int z = 0; z < D; ++z) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
            if (!src[((long)z * H + y) * W + x]) continue;
            for (int i = 0; i < n; ++i) {   /* scatter form of the same OR */
                int zz = z + off[3 * i], yy = y + off[3 * i + 1], xx = x + off[3 * i + 2];
                if (zz < 0 || zz >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                dst[((long)zz * H + yy) * W + xx] = 1;
            }
        }
    }
    free(off);
}

/* dilate_volume (:22-46): even sizes are bumped to odd; sizes <= 7 are one pass; larger sizes
 * are radius//3 passes of k=7 followed by one pass of 2*(radius%3)+1 when the remainder is > 0. */
void oracle_dilate_volume(const uint8_t *vol, uint8_t *out, long nvol, int D, int H, int W, int kernel_size) {
    long V = (long)D * H * W * nvol;
    if (kernel_size % 2 == 0) kernel_size += 1;
    const int full = 3;
    if (kernel_size <= 2 * full + 1) { dilate_pass(vol, out, nvol, D, H, W, kernel_size); return; }
    int radius = (kernel_size - 1) / 2, num_full = radius / full, rem = radius % full;
    uint8_t *a = (uint8_t *)malloc((size_t)V), *b = (uint8_t *)malloc((size_t)V);
    memcpy(a, vol, (size_t)V);
    for (int i = 0; i < num_full; ++i) { dilate_pass(a, b, nvol, D, H, W, 2 * full + 1); uint8_t *t = a; a = b; b = t; }
    if (rem > 0) { dilate_pass(a, b, nvol, D, H, W, 2 * rem + 1); uint8_t *t = a; a = b; b = t; }
    memcpy(out, a, (size_t)V);
    free(a); free(b);
}

/* Canonical top-k used by the restatement of isolate_tumor (:1478-1492): the k largest values,
 * ties broken by LOWER linear index first (torch.topk's tie order is implementation-defined;
 * SURVEY.md section 7 "hard parts").  mask[i]=1 for selected voxels. */
typedef struct { float v; long i; } vi_t;
static int cmp_vi(const void *a, const void *b) {
    const vi_t *x = (const vi_t *)a, *y = (const vi_t *)b;
    if (x->v > y->v) return -1; if (x->v < y->v) return 1;
    return (x->i < y->i) ? -1 : (x->i > y->i);
}
void oracle_topk_mask(const float *x, long n, long k, uint8_t *mask) {
    vi_t *a = (vi_t *)malloc(sizeof(vi_t) * (size_t)n);
    for (long i = 0; i < n; ++i) { a[i].v = x[i]; a[i].i = i; }
    qsort(a, (size_t)n, sizeof(vi_t), cmp_vi);
    memset(mask, 0, (size_t)n);
    if (k > n) k = n;
    for (long i = 0; i < k; ++i) mask[a[i].i] = 1;
    free(a);
}

/* Descending rank of every element (value desc, index asc) -- the permutation
 * GlobalWeightedRankPooling builds with sort + argsort (:469,:525-529). */
void oracle_rank_desc(const float *x, long n, int64_t *rank) {
    vi_t *a = (vi_t *)malloc(sizeof(vi_t) * (size_t)n);
    for (long i = 0; i < n; ++i) { a[i].v = x[i]; a[i].i = i; }
    qsort(a, (size_t)n, sizeof(vi_t), cmp_vi);
    for (long i = 0; i < n; ++i) rank[a[i].i] = i;
    free(a);
}
