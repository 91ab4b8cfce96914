/* neuconw_hip.h -- C ABI of libneuconw_hip.so: the MI355X (gfx950) native hot path of
 * NeuralRecon-W's volume renderer.
 *
 * The reference (zju3dv/NeuralRecon-W) has no FFI: its hot path is Python calling torch ops.  The
 * entry points below are the functions a maintainer would bind (ctypes -- see INTEGRATION.md) to
 * replace, one for one, the torch-op bodies of
 *     rendering/renderer.py  (sample_pdf :15-48, up_sample :257-341, cat_z_vals :343-363,
 *                             render_core_outside :157-228, render_core :570-783,
 *                             get_near_far_* :380-456)
 *     models/neuconw.py      (SDFNetwork :183-296, RenderingNetwork :59-170, NeuconW :299-376)
 *     models/nerf.py         (NeRF :86-183)
 *     tools/prepare_data/generate_voxel.py  (get_near_far :311-439, kaolin raytrace)
 *
 * Conventions: all pointers are DEVICE pointers unless a parameter is documented as a host
 * struct; `stream` is a hipStream_t passed as void*; every function is asynchronous on `stream`
 * and returns 0 on success or a hipError_t / negative NCW_E_* code.  No torch types appear here.
 * prec: 0 = exact-fp32 MFMA (v_mfma_f32_32x32x2_f32; parity mode), 1 = bf16 MFMA with f32
 * accumulation (v_mfma_f32_32x32x16_bf16; throughput mode), 2 = fp16 MFMA with f32 accumulation
 * (v_mfma_f32_32x32x16_f16: the same kernels compiled a second time with the 16-bit type switched -- 8x the mantissa
 * of bf16 at the same speed; the library also exports every MLP entry point as <name>_f16, which is what prec 2
 * dispatches to).  Packed weights and stashes of a network must be built with the prec they are used with.
 */
#ifndef NEUCONW_HIP_H
#define NEUCONW_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NCW_PREC_F32 0
#define NCW_PREC_BF16 1
#define NCW_PREC_F16 2   /* fp16 operands (10 mantissa bits), f32 accumulate; the caller loss-scales the backward */
#define NCW_MAX_LAYERS 12
#define NCW_MAX_SEGS 4

#define NCW_E_BADARG (-1)
#define NCW_E_UNSUPPORTED (-2)

/* library / device info; returns the ABI version (bumped on any signature change) */
int ncw_abi_version(void);
/* sha256 (hex) of the sources this library was built from (every .hip / .h file of csrc/ and this header); the Python binding
 * refuses a library whose hash differs from the tree's (neuralrecon-w_amd/build.py source_hash) */
const char* ncw_source_hash(void);
/* writes "gfx950" style arch name of device 0 into buf; returns CU count (0 if no device) */
int ncw_device_info(char* buf, int buflen);

/* ------------------------------------------------------------------------------------------
 * Weight packing (replaces nn.utils.weight_norm's reparametrisation + the implicit cuBLAS
 * operand layout; models/neuconw.py:104-105,256-257).  One descriptor = one nn.Linear (or a row
 * range of it) scattered into one zero-padded matrix in MFMA fragment order.
 * ---------------------------------------------------------------------------------------- */
typedef struct NcwSeg {
    int32_t col0;   /* first source column of the segment                       */
    int32_t ncols;  /* number of source columns                                 */
    int32_t dcol0;  /* destination (padded, logical) input-feature index        */
    int32_t _pad;
} NcwSeg;

typedef struct NcwPackDesc {
    const float* src;    /* weight (or weight_v) [rows_total, ld] row-major     */
    const float* g;      /* weight_g [rows_total] or NULL (plain Linear)        */
    const float* bias;   /* bias [rows_total] or NULL                            */
    void* dst_w;         /* packed matrix (pre-zeroed once; padding never rewritten) */
    float* dst_b;        /* packed bias [rb_out*32] (C-layout order) or NULL    */
    int32_t ld;          /* source leading dimension = in_features               */
    int32_t row0, nrows; /* source row range packed by this descriptor           */
    int32_t drow0;       /* destination (padded, logical) output-feature index   */
    int32_t rb_out, rb_in; /* destination block dims (x32) in its own orientation */
    int32_t transpose;   /* 1: destination is the transpose (out<->in swapped)    */
    int32_t prec;        /* element type of dst_w                                 */
    float scale;         /* e.g. 1/sqrt(2) folded into the skip layer             */
    int32_t nseg;
    NcwSeg seg[NCW_MAX_SEGS];
    int32_t residual;    /* 16-bit only: 1 = store the ROUNDING RESIDUAL lo = h16(w - h16(w)) instead of h16(w): the
                          * second half of a split (hi + lo) weight matrix, NcwSdfNet.w_lo */
    int32_t _pad;
} NcwPackDesc;

/* descs: device array of n descriptors; row_prefix: device int32[n+1] exclusive prefix sum of
 * nrows (grid = row_prefix[n] blocks, passed as total_rows). */
int ncw_pack_weights(const NcwPackDesc* descs, const int32_t* row_prefix, int n, int total_rows,
                     void* stream);

/* Reverse of ncw_pack_weights for gradients: reads the dense padded gradient matrices produced by
 * ncw_wgrad and writes d(weight_g), d(weight_v) / d(weight), d(bias) of the original parameters
 * (weight-norm backward: g_bar = sum(Wbar * v/|v|), v_bar = g/|v| (Wbar - g_bar v/|v|)). */
typedef struct NcwUnpackDesc {
    const float* dw;     /* dense padded gradient [rb_out*32, ldw], forward orientation */
    const float* db;     /* dense padded bias gradient [rb_out*32] or NULL               */
    const float* src;    /* weight_v / weight                                            */
    const float* g;      /* weight_g or NULL                                             */
    float* d_src;        /* out: grad of weight_v / weight  [rows_total, ld]             */
    float* d_g;          /* out: grad of weight_g or NULL                                */
    float* d_bias;       /* out: grad of bias or NULL                                    */
    int32_t ld, ldw;
    int32_t row0, nrows, drow0;
    float scale;
    float grad_mul;      /* multiplies every gradient written (0 = 1): 1 / loss scale in the fp16 mode */
    int32_t accumulate;  /* 1: add into d_* (torch .grad accumulation), 0: overwrite     */
    int32_t nseg;
    NcwSeg seg[NCW_MAX_SEGS];
    const float* grad_mul_dev; /* device scalar multiplied on top of grad_mul (the dynamic 1 / loss scale) or NULL */
} NcwUnpackDesc;
int ncw_unpack_grads(const NcwUnpackDesc* descs, const int32_t* row_prefix, int n, int total_rows,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * SDF network (models/neuconw.py:183-296).  HOST struct of device pointers to packed weights.
 * Layer l in [0, n_layers): w[l] = packed [32*rb (or 32 for the sdf row) x K_l]; the skip layer's
 * K is [rb blocks of h | 2 blocks of gamma]; wt[l] = packed transpose (adjoint / backward).
 * The last Linear is split: w[n_layers-1] = sdf row (1 out-block), w_feat = feature rows.
 * ---------------------------------------------------------------------------------------- */
typedef struct NcwSdfNet {
    const void* w[NCW_MAX_LAYERS];
    const float* b[NCW_MAX_LAYERS];
    const void* wt[NCW_MAX_LAYERS];
    const void* w_feat;   /* last layer rows 1..W   [32*rb x 32*rb]        */
    const float* b_feat;
    const void* wt_feat;  /* its transpose                                  */
    int32_t n_layers;     /* number of Linear layers (9 for the 8x256 net)  */
    int32_t skip_layer;   /* index l whose input is cat([h, gamma])/sqrt2, or -1 */
    int32_t rb;           /* hidden width / 32 (2, 8 or 16)                 */
    int32_t multires;     /* 6                                              */
    float scale;          /* SDFNetwork.scale                               */
    int32_t adj_mode;     /* with wt_lo present: 2 = the adjoint sweep's t_l as a hi + lo pair too (rb = 16: see wt_lo below); 0 / 1 = weights only */
    /* Split-precision VALUE path (fp16 mode, W = 256): w_lo[l] = the rounding residuals of w[l] in the same packed layout
     * (NcwPackDesc.residual), or all NULL.  When present, ncw_sdf_infer* and the forward sweep of ncw_sdf_fwd evaluate
     * sdf with activations AND weights as fp16 hi + lo pairs, three MFMAs per product (hi.hi + lo.hi + hi.lo, f32
     * accumulate): ~2^-22 relative, i.e. fp32-like SDF values at fp16 MFMA rate.  sigmoid(sdf * inv_s) multiplies an SDF
     * error by inv_s = exp(10 variance) -- 20 at initialisation, several hundred where NeuS trains (models/neuconw.py:
     * 173-179, rendering/renderer.py:624-632) -- so the 5e-4 of a plain fp16 evaluation is 0.2 in the sigmoid's argument there.
     * The adjoint / backward passes and the feature rows stay plain fp16. */
    const void* w_lo[NCW_MAX_LAYERS];
    /* wt_lo[l] = the rounding residuals of the TRANSPOSED matrices wt[l] (same packed layout), or all NULL.  When present (fp16,
     * W = 256) the analytic adjoint sweep of ncw_sdf_fwd -- the normals -- takes its WEIGHTS as hi + lo pairs (W_hi^T t + W_lo^T t,
     * two MFMAs per product; t stays single fp16): the compositor multiplies the normal's component along the ray by
     * dist * inv_s inside the sigmoid (rendering/renderer.py:600-632), and on trained weights the plain-fp16 adjoint sweep was
     * what put single rays above 1e-4 (scripts/diag/emul_timed_batch.py: worst rays 1.5e-4 -> 3.7e-5).  Forward only: the stash
     * t_l, ncw_sdf_bwd and the weight gradients are unchanged.  adj_mode = 2 (rb = 16, the shipped W = 512): t_l is a hi + lo pair
     * as well (three MFMAs per product, the value chain's accuracy): with 8 + 16 samples per ray one sample carries a ray and
     * the colour network reads ITS normal -- the weights alone as pairs were not enough there (profiles/r06/emul_timed_batch_shipped_tangent*.log). */
    const void* wt_lo[NCW_MAX_LAYERS];
} NcwSdfNet;

/* a-2 `sdf(x)` (neuconw.py:281-282): x [n,3] f32 -> sdf [n] f32.  No grad, last layer 1 row. */
int ncw_sdf_infer(const NcwSdfNet* net, int prec, const float* x, int64_t n, float* sdf, void* stream);

/* Same network evaluated at ray samples x = o[r] + d[r] * z[r,i]  (renderer.py:519-522, 350-352):
 * rays_o, rays_d [R,3], z [R,n] -> sdf [R,n].  Fuses the point generation into the MLP prologue. */
int ncw_sdf_infer_rays(const NcwSdfNet* net, int prec, const float* rays_o, const float* rays_d, const float* z,
                       int R, int n, float* sdf, void* stream);

/* ------------------------------------------------------------------------------------------
 * Point source shared by the fused MLP kernels: explicit points or ray samples (the point
 * generation of renderer.py:586-597 / :176-186 is fused into the MLP prologue).
 * ---------------------------------------------------------------------------------------- */
typedef struct NcwPoints {
    const float* x;            /* mode 0: [n,3] explicit points                              */
    const float* rays_o;       /* modes 1,2: [R,3]                                           */
    const float* rays_d;       /* modes 1,2: [R,3]                                           */
    const float* z;            /* modes 1,2: [R,per_ray]                                     */
    const float* sample_dist;  /* mode 2: [R]                                                */
    int32_t per_ray;
    int32_t mode;              /* 0: x;  1: o + d z;  2: section mid-point o + d (z_i + dist_i/2);
                                  3: regular grid generated on chip (utils/visualization.py:46-50);
                                  4: a SELECTION of the mode-2 points: point k of the launch is ray sample idx[k]
                                     (= ray * per_ray + i), only the first min(n, *count) are processed, outputs
                                     and cotangents stay addressed by the ray sample (dense [R, per_ray] arrays), the
                                     stashes by k (ncw_bg_select; background NeRF kernels at W = 256, 16-bit only) */
    /* mode 3: point p (+ grid_start) of linspace(gmin, gmax, gdim)^3, meshgrid 'ij' (x slowest),
     * mapped to the unit sphere as (x - gorigin) / gradius. */
    float gmin[3], gmax[3], gorigin[3], gradius;
    int32_t gdim;
    int32_t _gpad;
    int64_t gstart;
    const int32_t* idx;        /* mode 4: DEVICE int32[n] ray-sample indices                               */
    const int32_t* count;      /* mode 4: DEVICE int32[1] number of valid entries of idx                  */
} NcwPoints;

/* config 5 (tools/extract_mesh.py / utils/visualization.py:37-85): sdf of `count` points of the regular grid
 * described by pts (mode 3) starting at linear index pts->gstart; no coordinate array ever exists. */
int ncw_sdf_infer_points(const NcwSdfNet* net, int prec, const NcwPoints* pts, int64_t count, float* sdf, void* stream);


/* Activation stash of the SDF net in fragment-native layout [tile32][blocks][4][64 lanes][4]
 * (element = f32 for prec 0, bf16 for prec 1).  HOST struct of device pointers.
 * Written by ncw_sdf_fwd (gamma, h, s, t, feat), ncw_color_bwd (dfeat) and ncw_sdf_bwd
 * (qbar, zbar, zsdf, one); consumed by ncw_sdf_bwd and ncw_wgrad. */
typedef struct NcwSdfStash {
    void* gamma;                 /* encoding, 2 blocks                                        */
    void* h[NCW_MAX_LAYERS];     /* h[l], l>=1: input of layer l = Softplus(z_{l-1}), rb blocks */
    void* s[NCW_MAX_LAYERS];     /* Softplus'(z_l) is recomputed as 1 - exp(-100 h[l+1]): no stash of its own.  fp16 mode with
                                  * NcwSdfNet.adj_mode 2 (rb = 16): s[l], l >= 1 = the fp16 RESIDUALS of h[l] (h_lo = h16(h - h16(h))),
                                  * written by the split value chain and read back by the adjoint sweep of the SAME launch, whose
                                  * phi' = 1 - exp(-100 (h + h_lo)) then carries no fp16 rounding of h; NULL = phi' from h[l] alone */
    void* t[NCW_MAX_LAYERS];     /* t[l] = a_l * s[l] (adjoint pass), l <= L-2                 */
    void* feat;                  /* feature vector z_{L-1}[1:], rb blocks                      */
    void* dfeat;                 /* upstream d(feat), rb blocks                                */
    void* qbar[NCW_MAX_LAYERS];  /* qbar[0]: 2 blocks (J_gamma nbar); qbar[l]: rb blocks       */
    void* zbar[NCW_MAX_LAYERS];  /* zbar[l] = dL/dz_l, l <= L-2                                */
    void* zsdf;                  /* 1 block: feature 0 = d_sdf                                 */
    void* one;                   /* 1 block: feature 0 = 1                                     */
} NcwSdfStash;

/* a-2/a-5 forward of the SDF net WITH its analytic input gradient (neuconw.py:263-296):
 * sdf [n], grad [n,3] (= d sdf / d x), plus the stash for the backward.  One launch.
 * FORWARD-ONLY form (the reference's validation / novel-view render and vertex colours, rendering/renderer.py:785-916
 * under no_grad, :951-961; models/neuconw.py:353-376 called without a backward): pass a stash whose t[0] is NULL.  The
 * outputs are the same bit for bit; of the stash only h[1 .. L-1] (re-read by the adjoint sweep of the SAME launch: it must
 * exist as scratch, W x 2 B per point and layer) and feat (ncw_color_fwd's input) are written -- no gamma, no t. */
int ncw_sdf_fwd(const NcwSdfNet* net, int prec, const NcwPoints* pts, int64_t n, float* sdf, float* grad,
                const NcwSdfStash* stash, void* stream);
/* First- and second-order backward (SURVEY 8a-2 "parameter-gradient specification"): consumes
 * d_sdf [n], d_grad [n,3] and stash->dfeat; fills stash->qbar/zbar/zsdf/one for ncw_wgrad. */
int ncw_sdf_bwd(const NcwSdfNet* net, int prec, const NcwPoints* pts, int64_t n, const float* d_sdf,
                const float* d_grad, const NcwSdfStash* stash, void* stream);

/* ------------------------------------------------------------------------------------------
 * Colour network -- RenderingNetwork, models/neuconw.py:59-170 (encode_apperence, mode "idr").
 *   f  = xyz_encoding_final(feat)                       [rbf x rbf], no activation
 *   e0 = relu(static_linear_0([f | gamma4(dir) | a]))   K = [rbf blocks | AUX1 (3 blocks)]
 *   e_i= relu(static_linear_i(e_{i-1}))                 [rbh x rbh]
 *   x0 = relu(lin0([points | normals | e]))             K = [rbh blocks | AUX2 (1 block: pts, normals)]
 *   x_l= relu(lin_l(x_{l-1})) ... rgb = sigmoid(lin_last(x))
 * ---------------------------------------------------------------------------------------- */
typedef struct NcwColorNet {
    const void* w_f; const void* wt_f; const float* b_f;
    const void* w_e[4]; const void* wt_e[4]; const float* b_e[4];
    const void* w_l[8]; const void* wt_l[8]; const float* b_l[8];
    int32_t n_head;  /* static_head_layers                       */
    int32_t n_lin;   /* trunk Linear count (n_layers + 1)        */
    int32_t rbf, rbh, rbc, n_a;
    /* fp16 mode, FORWARD only: the residual matrices h16(W - h16(W)) of the forward matrices above (packed with
     * NcwPackDesc.residual); NULL = one rounding per weight.  With them ncw_color_fwd evaluates W_hi x + W_lo x: on trained
     * weights the colour network's weight rounding was the largest remaining term of the fp16 mode's colour error
     * (scripts/diag/emul_color16.py: 5 % of the rays above 1e-4 -> none). */
    const void* w_f_lo; const void* w_e_lo[4]; const void* w_l_lo[8];
    /* with the residual matrices present: 1 = the ACTIVATIONS of every layer as fp16 hi + lo pairs too (a third pass W_hi x_lo; rbf
     * 8 / 16 only).  Default at d_feature = 512, the shipped width: with 8 + 16 samples one sample carries a ray and the
     * activations' rounding was, with the normals', what kept rays above 1e-4 on trained weights (ncw_color.hip).  Forward only. */
    int32_t act_split;
    int32_t _pad;
} NcwColorNet;

typedef struct NcwColorStash {
    void* aux1;      /* 3 blocks */
    void* aux2;      /* 1 block  */
    void* f;         /* rbf      */
    void* e[4];      /* rbh, post-relu head activations */
    void* x[8];      /* rbc, post-relu trunk activations x[0..n_lin-2] */
    void* zf;        /* rbf : dL/d f (pre-activation of xyz_encoding_final) */
    void* ze[4];     /* rbh */
    void* zx[8];     /* rbc */
    void* zo;        /* 1 block: dL/d(pre-sigmoid rgb), features 0..2 */
    const float* aux_bias; /* NULL, or [R, 32 rbh] fp32 from ncw_aux_ray_bias: the per-ray part of the head's first layer,
                            * W_e0[:, view-dir | appearance columns] . [gamma_4(d) | a], added to that layer's bias in the
                            * FORWARD instead of being multiplied from 16-bit operands (the backward is unchanged) */
} NcwColorStash;

/* normals [n,3] (the SDF gradient), a [R or n, n_a] appearance rows indexed by the point's ray,
 * feat: stash (rbf blocks) written by ncw_sdf_fwd  ->  rgb [n,3].
 * FORWARD-ONLY form: a stash whose aux1 is NULL (aux_bias stays an input): the same rgb bit for bit, nothing is stored. */
int ncw_color_fwd(const NcwColorNet* net, int prec, const NcwPoints* pts, int64_t n, const float* normals,
                  const float* a, const void* feat_stash, float* rgb, const NcwColorStash* stash, void* stream);
/* d_rgb [n,3] -> d_grad[n,3] += d(normals), d_a [R,n_a] += (atomics; zero it first), dfeat stash (rbf),
 * d_a_rows: NULL, or [n,n_a] -- then every point's appearance-code adjoint is STORED as its row instead of being
 * added to d_a with atomics, and ncw_ray_sum_rows reduces the rows of a ray in sample order (reproducible; the
 * fp32 parity mode uses it).
 * and the z-stashes for ncw_wgrad. */
int ncw_color_bwd(const NcwColorNet* net, int prec, const NcwPoints* pts, int64_t n, const float* rgb,
                  const float* d_rgb, float* d_grad, float* d_a, float* d_a_rows, void* dfeat_stash,
                  const NcwColorStash* stash, void* stream);

/* ------------------------------------------------------------------------------------------
 * Background NeRF -- models/nerf.py:86-183 (use_viewdirs, encode_appearance) evaluated on the
 * inverted-sphere points of renderer.py:176-186 (fused into the prologue).
 * ---------------------------------------------------------------------------------------- */
typedef struct NcwNerfNet {
    const void* w_p[8]; const void* wt_p[8]; const float* b_p[8];
    const void* w_alpha; const void* wt_alpha; const float* b_alpha;
    const void* w_feat; const void* wt_feat; const float* b_feat;
    const void* w_a[4]; const void* wt_a[4]; const float* b_a[4];
    const void* w_rgb; const void* wt_rgb; const float* b_rgb;
    int32_t D;       /* trunk depth (8)                                         */
    int32_t skip;    /* layer index after whose ReLU gamma(p) is concatenated    */
    int32_t rbn, rbh, n_head, n_a;
    /* fp16 mode: the rounding residuals h16(W - h16(W)) of the FORWARD matrices above (NcwPackDesc.residual), or all NULL: the
     * operands of ncw_nerf_refine. */
    const void* w_p_lo[8]; const void* w_alpha_lo; const void* w_feat_lo; const void* w_a_lo[4]; const void* w_rgb_lo;
} NcwNerfNet;

typedef struct NcwNerfStash {
    void* gp;        /* 3 blocks: gamma_10(p4) (84)    */
    void* aux1;      /* 3 blocks                        */
    void* h[9];      /* h[i], i=1..D: post-relu trunk   */
    void* featn;     /* rbn: feature_linear output      */
    void* e[4];      /* rbh                             */
    void* zp[8];     /* rbn: dL/dz of trunk layer i     */
    void* zalpha;    /* 1 block: feature 0 = d density  */
    void* zfeat;     /* rbn                             */
    void* ze[4];     /* rbh                             */
    void* zrgb;      /* 1 block: features 0..2          */
    const float* aux_bias; /* NULL, or [R, 32 rbh] fp32 from ncw_aux_ray_bias (as NcwColorStash.aux_bias): the per-ray part of
                            * apperence_encoding.static_linear_0 (models/nerf.py:131-139,173-174), forward only */
} NcwNerfStash;

/* pts: mode 2 on z_feed (section mid-points, inverted-sphere reparametrisation applied inside), or
 * x4 != NULL: explicit [n,4] points with pts->rays_d / a indexed per point (NeRF.forward API).
 * -> density [n] (raw), rgb [n,3] (raw, no sigmoid).
 * FORWARD-ONLY form: a stash whose gp is NULL (aux_bias stays an input): the same outputs bit for bit, nothing is stored. */
int ncw_nerf_fwd(const NcwNerfNet* net, int prec, const NcwPoints* pts, const float* x4, int64_t n, const float* a,
                 float* density, float* rgb, const NcwNerfStash* stash, void* stream);
int ncw_nerf_bwd(const NcwNerfNet* net, int prec, const NcwPoints* pts, int64_t n, const float* d_density,
                 const float* d_rgb, float* d_a, float* d_a_rows, const NcwNerfStash* stash, void* stream);
/* FORWARD REFINEMENT of the background NeRF in split precision (fp16 mode, W = 256; models/nerf.py:156-182 on the points of
 * rendering/renderer.py:176-186): density / raw rgb of the SELECTED samples (pts: NcwPoints mode 4, the list of ncw_bg_select --
 * the samples whose background the compositor can use) are re-evaluated with gamma_10(p4), weights (net->w_*_lo) and hidden
 * activations as fp16 hi + lo pairs (three MFMAs per product, f32 accumulate: fp32-level outputs) and written over the
 * plain-fp16 results of ncw_nerf_fwd at those samples.  aux_bias: the per-ray fp32 rows of ncw_aux_ray_bias ([R, 128]) -- the
 * view-direction / appearance-code columns of the head.  Forward only (the stash and the backward are the plain launch's).
 * Rays whose colour is all background carry the plain NeRF's error undiluted: 1.1e-4 .. 1.6e-4 on trained weights without this. */
int ncw_nerf_refine(const NcwNerfNet* net, int prec, const NcwPoints* pts, int64_t n, const float* aux_bias, float* density,
                    float* rgb, void* stream);
/* Batch assembly from the HBM-resident ray cache.  Replaces, per batch, PhototourismDataset.__getitem__
 * for split "train" (datasets/phototourism.py:709-726: rays = row[0:8] ++ row[10:13] (with semantics) / row[9:12],
 * ts = long(row[8]), semantics = row[9]), the DataLoader's collate + pinned H2D copy, and the black-list test of
 * NeuconWSystem.training_step (lightning_modules/neuconw_system.py:345-349): keep[b] = 0 iff label[b] is one of
 * mask_ids (<= 4 ids; the shipped RAY_MASK_LIST has 4).  all_rays [n_rows, ncols] (ncols 13 with semantics, 12 without),
 * all_rgbs [n_rows,3], idx [B] int64 or NULL (= identity); outputs rays [B,11], ts [B] i64, label [B] i64 or NULL,
 * rgbs [B,3] or NULL, keep [B] u8 or NULL.  mask_ids is a HOST array. */
int ncw_batch_assemble(const float* all_rays, int ncols, const float* all_rgbs, const int64_t* idx, int64_t n_rows,
                       int64_t B, int with_semantics, float* rays, int64_t* ts, int64_t* label, float* rgbs,
                       const int* mask_ids, int n_ids, uint8_t* keep, void* stream);
/* ------------------------------------------------------------------------------------------
 * Per-step glue of render() / the loss as single launches (each replaces 6-25 tiny torch kernels; the step is
 * GPU-bound, so every 3-5 us launch counts).
 * ---------------------------------------------------------------------------------------- */
/* models/neuconw.py:131-140: out[r, j] = sum_k w[j, col0 + k] * [gamma_4(rays_d[r]) (27) | a[r] (n_a)][k], fp32, once per RAY
 * (w: the row-major fp32 weight of static_linear_0, row stride ldw; out row stride ldo >= n_out).  The appearance code and
 * the view direction are per-ray constants: their fp16 rounding is coherent along a ray and was the largest term of the fp16
 * mode's colour error on trained weights (scripts/diag/emul_color16.py). */
int ncw_aux_ray_bias(const float* w, int ldw, int col0, int n_out, const float* rays_d, const float* a, int n_a, int64_t R,
                     float* out, int ldo, void* stream);
/* renderer.py:793-806: rays [R,ncols >= 8] -> rays_o = (rays[:,0:3] - origin) / radius, rays_d = rays[:,3:6],
 * near = rays[:,6] / radius, far = rays[:,7] / radius, depth_gt = rays[:,8] / radius, depth_weight = rays[:,9]
 * (zeros when ncols < 10).  origin: HOST float[3]. */
int ncw_ray_prologue(const float* rays, int ncols, int64_t R, const float* origin_host, float radius, float* rays_o,
                     float* rays_d, float* near, float* far, float* depth_gt, float* depth_weight, void* stream);
/* SingleVarianceNetwork (models/neuconw.py:173-179) as used by render_core (renderer.py:624-632):
 * inv_s = clamp(exp(10 variance), 1e-6, 1e6), s_val = 1 / inv_s; backward: d_variance = 10 inv_s [1e-6 < inv_s < 1e6]
 * sum_r d_inv_s[r] (the per-ray terms of ncw_composite_bwd, summed in a fixed order). */
int ncw_inv_s_fwd(const float* variance, float* inv_s, float* s_val, void* stream);
int ncw_inv_s_bwd(const float* d_inv_s_rays, int64_t R, const float* inv_s, float* d_variance, void* stream);
/* NeuconWLoss (losses.py:21-43): loss = coef * ( sum|color - rgbs| / (R + 1e-5) + igr_w * gradient_error
 *   + mask_w * mean(mask_error[n_mask]) + depth_w * mean(sfm[n_sfm]) ); NULL / 0 switches a term off.
 * bwd: d_color [R,3], d_gradient_error [1], d_mask_error [n_mask], d_sfm [n_sfm] for an upstream d_loss [1]. */
int ncw_loss_fwd(const float* color, const float* rgbs, int64_t R, const float* gradient_error, const float* mask_error,
                 int64_t n_mask, const float* sfm, int64_t n_sfm, float coef, float igr_w, float mask_w, float depth_w,
                 float* loss, void* stream);
int ncw_loss_bwd(const float* d_loss, const float* color, const float* rgbs, int64_t R, int64_t n_mask, int64_t n_sfm,
                 float coef, float igr_w, float mask_w, float depth_w, float* d_color, float* d_gradient_error,
                 float* d_mask_error, float* d_sfm, void* stream);
/* out[idx[r], :] += rows[r, :] with f32 atomics (out [n_out, n_cols], zero-filled or accumulating; indices outside
 * [0, n_out) are skipped): the backward of the appearance-embedding lookup `embeddings["a"](ts)` (renderer.py:808). */
int ncw_scatter_add_rows(const float* rows, const int64_t* idx, int64_t R, int n_cols, int64_t n_out, float* out,
                         void* stream);
/* out[R,n_cols] (+)= per-ray sums of rows[R*per_ray, n_cols] in sample order (see d_a_rows above). */
int ncw_ray_sum_rows(const float* rows, int64_t R, int per_ray, int n_cols, float* out, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Weight-gradient GEMMs: dense[32*rbx, ld] (f32, forward orientation) += X^T Y over all points,
 * X, Y in stash layout; optional dbias[32*rbx] += column sums of X.  Batched: one launch runs a
 * device table of products (split-K + f32 atomics).
 * ---------------------------------------------------------------------------------------- */
typedef struct NcwWgradDesc {
    const void* x;     /* stash, rbx blocks  (output-feature side)  */
    const void* y;     /* stash, rby blocks  (input-feature side)   */
    float* dense;      /* [32*rbx, ld] (+ column offset already applied) */
    float* dbias;      /* [32*rbx] or NULL                           */
    int32_t rbx, rby, ld;
    int32_t ksplit;    /* ncw_wgrad_tiled only: > 0 overrides the launch-wide split-K for this product */
    int64_t n_points;  /* ncw_wgrad_tiled only: > 0 overrides the launch-wide point count            */
    const int32_t* n_points_dev; /* ncw_wgrad_tiled only: DEVICE int32[1] or NULL; the product covers the first
                                  * min(n_points, *n_points_dev) points of its stashes (a selection made on the
                                  * device, NcwPoints mode 4); the K-slices re-divide that count */
} NcwWgradDesc;
/* descs: DEVICE array; wg_prefix: device int32[n_desc+1] exclusive prefix of
 * ceil(rbx/4)*ceil(rby/4)*ksplit workgroups per product; total_wgs = wg_prefix[n_desc]. */
int ncw_wgrad(const NcwWgradDesc* descs, const int32_t* wg_prefix, int n_desc, int total_wgs, int ksplit,
              int prec, int64_t n_points, void* stream);
/* Run-to-run reproducible variant (the fp32 parity mode uses it): every K-slice writes its 128 x 128 partial into its
 * own slab of `partials` (DEVICE scratch of ncw_wgrad_ordered_scratch_floats(total_wgs) floats, contents
 * irrelevant on entry) and a second launch adds the slabs of a quadrant in slice order.  The autograd GEMMs it
 * replaces (torch `mm` backward of F.linear, models/neuconw.py:269-278) are deterministic on the reference's CPU path. */
int64_t ncw_wgrad_ordered_scratch_floats(int total_wgs);
int ncw_wgrad_ordered(const NcwWgradDesc* descs, const int32_t* wg_prefix, int n_desc, int total_wgs, int ksplit,
                      int prec, int64_t n_points, float* partials, void* stream);
/* bf16 only, explicit workgroup tile: tile 0 = 128 x 256 features (wg_prefix counts ceil(rbx/4)*ceil(rby/8)*ksplit),
 * tile 1 = 256 x 256 (ceil(rbx/8)*ceil(rby/8)*ksplit): every stash element is read once per product.
 * A product's own ksplit / n_points (when > 0) replace the launch-wide values, so products of different
 * networks and sizes share ONE launch with a work-proportional split. */
int ncw_wgrad_tiled(const NcwWgradDesc* descs, const int32_t* wg_prefix, int n_desc, int total_wgs, int ksplit,
                    int tile, int64_t n_points, void* stream);
/* the same for fp16 stashes (prec NCW_PREC_F16; ncw_wgrad / ncw_wgrad_ordered take it through `prec`) */
int ncw_wgrad_tiled_f16(const NcwWgradDesc* descs, const int32_t* wg_prefix, int n_desc, int total_wgs, int ksplit,
                        int tile, int64_t n_points, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser step over flat fp32 buffers (16-byte aligned): global-norm clip + Adam in one launch.
 * Replaces torch.nn.utils.clip_grad_norm_ (train.py:61) + torch.optim.Adam(eps=1e-7).step()
 * (utils/__init__.py:23-31) for parameters re-seated into one flat buffer:
 *   coef = total_norm ? min(max_norm / (total_norm[0] + 1e-6), 1) : 1;   g *= coef  (written back)
 *   m += (1-beta1)(g - m);  v = beta2 v + (1-beta2) g g;
 *   p -= step_size * m / (sqrt(v) / bias_correction2_sqrt + eps)
 * with step_size = lr / (1 - beta1^t), bias_correction2_sqrt = sqrt(1 - beta2^t) computed by the caller.
 * total_norm: DEVICE scalar (the 2-norm of grad) or NULL for no clipping.  A non-finite total_norm skips the step
 * (nothing is written): one overflowed fp16 step must not poison the parameters and the moments.
 * ---------------------------------------------------------------------------------------- */
int ncw_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float step_size,
                  float beta1, float beta2, float eps, float bias_correction2_sqrt, const float* total_norm,
                  float max_norm, void* stream);

/* The same update with every step-dependent quantity on the DEVICE (trainer.FlatAdam): `state` holds the applied-step
 * counter and the scalars derived from it, so a skipped step does not advance Adam's bias correction, nothing is a
 * per-step kernel argument (HIP-graph capturable), and the fp16 loss scale adapts without a device->host round trip.
 * Two launches: a one-thread prologue that reads total_norm (device scalar or NULL) and updates `state` / `loss_scale`,
 * then the elementwise update.
 *   finite norm:  step += 1; coef = max_norm > 0 ? min(max_norm / (norm + 1e-6), 1) : 1; step_size = lr / (1 - beta1^step)
 *                 and bc2_sqrt = sqrt(1 - beta2^step) in double; after `growth_interval` consecutive finite steps the loss
 *                 scale doubles (<= scale_max);
 *   non-finite:   nothing is written to param / grad / moments, step stays, skipped += 1, the loss scale halves (>= scale_min).
 * loss_scale: device float[2] = {scale, 1 / scale} (read by NcwCompositeGrad.grad_scale_dev / NcwUnpackDesc.grad_mul_dev) or
 * NULL; growth_interval 0 = never grow.  lr_dev: device scalar overriding `lr` (schedulers under graph replay) or NULL.
 * The reference's recipe has no counterpart for the scale (fp32 training, train.py:48-62). */
/* total_norm of torch.nn.utils.clip_grad_norm_ (train.py:61) over ONE flat fp32 buffer (16-byte aligned): norm[0] = ||grad||_2.
 * One launch, fixed summation order (run-to-run reproducible), non-finite entries propagate.  scratch: device floats,
 * ncw_grad_norm_scratch_floats() of them; it holds the partial sums and the blocks' arrival ticket, which a stream-ordered memset
 * arms in front of every launch (capture-safe).  One scratch buffer serves ONE stream at a time. */
int64_t ncw_grad_norm_scratch_floats(void);
int ncw_grad_norm(const float* grad, int64_t n, float* scratch, float* norm, void* stream);

typedef struct NcwAdamState {
    int32_t step;      /* applied (non-skipped) steps: Adam's t */
    int32_t good;      /* consecutive finite steps since the last scale change */
    int32_t skipped;   /* total skipped steps */
    int32_t skip_now;  /* 1: the step being applied was skipped */
    float coef, step_size, bc2_sqrt, last_norm;
} NcwAdamState;
int ncw_adam_step_dev(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, NcwAdamState* state,
                      const float* total_norm, float lr, const float* lr_dev, float beta1, float beta2, float eps,
                      float max_norm, float* loss_scale, int growth_interval, float scale_min, float scale_max,
                      void* stream);

/* Layout converters between row-major f32 [n, F] and the stash layout (rb = ceil(F/32) blocks,
 * element type by prec).  Used at the module boundary (NeuconW.forward returning feature vectors,
 * tests); padded lanes / features are written as zero. */
int ncw_stash_from_rows(int prec, const float* rows, int64_t n, int F, int rb, void* stash, void* stream);
int ncw_stash_to_rows(int prec, const void* stash, int64_t n, int F, int rb, float* rows, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-ray sampler kernels (one wavefront per ray).  All arrays f32, row-major [R, n].
 * ---------------------------------------------------------------------------------------- */
/* renderer.py:488-514: coarse z (+ optional whole-ray jitter rand_shift[R]), inverse-depth outside
 * samples (+ optional stratified jitter rand_out[R,O]), sample_dist.  rand_* may be NULL. */
int ncw_sample_coarse(const float* near, const float* far, const float* s_near, const float* s_far, int R,
                      int n_samples, int n_outside, const float* rand_shift, const float* rand_out,
                      float* z, float* z_out, float* sample_dist, void* stream);
/* renderer.py:257-341 (up_sample) + :15-48 (sample_pdf, det=True): -> z_new [R, n_new] */
int ncw_upsample(const float* rays_o, const float* rays_d, const float* z, const float* sdf, int R, int n,
                 float inv_s, int n_new, float* z_new, void* stream);
/* renderer.py:343-363, :566, :835-836: stable sort of cat([a,b]) per ray, optional payload
 * (pa/pb/pout may be NULL). */
int ncw_sort_merge(const float* a, int na, const float* b, int nb, const float* pa, const float* pb, int R,
                   float* out, float* pout, void* stream);
/* renderer.py:549-565 boundary samples: zb [R, nb] (unsorted; feed to ncw_sort_merge) */
int ncw_boundary(const float* near, const float* far, const float* z, int n, int R, int nb, float* zb,
                 void* stream);

/* ------------------------------------------------------------------------------------------
 * Voxel guidance (config 3): native replacement of kaolin's SPC ray tracing used by
 * tools/prepare_data/generate_voxel.py:311-439 / renderer.py:380-456.  Occupancy = bit-packed dense
 * grid of side 2^level over [-1,1]^3 (x slowest) + 8^3-brick mask.  3 <= level <= 10.
 * ---------------------------------------------------------------------------------------- */
/* set the voxels containing the given normalised points (atomicOr; occ/brick must be zeroed first):
 * occ  uint32[(2^level)^3 / 32], brick uint32[ceil((2^level / 8)^3 / 32)] */
int ncw_voxel_build(const float* pts_normalised, int64_t n, int level, uint32_t* occ, uint32_t* brick, void* stream);
/* per ray: entry depth (SfM units) of the first / last occupied voxel; 0 where the ray misses or
 * near <= 1e-4 (generate_voxel.py:397).  scene_origin_host: HOST float[3]. */
int ncw_ray_voxel_near_far(const float* rays_o_sfm, const float* rays_d, int R, const float* scene_origin_host,
                           float scale, int level, const uint32_t* occ, const uint32_t* brick, float* near_sfm,
                           float* far_sfm, void* stream);
/* kaolin.render.spc.unbatched_raytrace(octree, points, pyramid, prefix, origin, direction, level, return_depth=True,
 * with_exit=...) as tools/prepare_data/generate_voxel.py:358-368 calls it: EVERY intersection of a ray with an occupied
 * level-`level` voxel (a "nugget"), ordered by ray, then by depth.  Origins are already normalised to the cube [-1,1]^3
 * (generate_voxel.py:345) and nothing is added to them.  Two calls: offsets == NULL writes counts[R] (nuggets per ray); with
 * offsets[R] (their exclusive prefix sum) ray r writes nug_ray / nug_voxel (linear index (x 2^level + y) 2^level + z) /
 * nug_depth [N,2] = (entry, exit) depth along the direction as given, from offsets[r] on.  (compat/kaolin maps the voxel
 * index to kaolin's point-hierarchy index.) */
int ncw_ray_voxel_trace(const float* rays_o_norm, const float* rays_d, int R, int level, const uint32_t* occ,
                        const uint32_t* brick, const int32_t* offsets, int32_t* counts, int32_t* nug_ray,
                        int32_t* nug_voxel, float* nug_depth, void* stream);

/* ------------------------------------------------------------------------------------------
 * Compositor: renderer.py:205-216 (background alpha) + :586-783 (render_core after the networks).
 * HOST structs of device pointers.
 * ---------------------------------------------------------------------------------------- */
typedef struct NcwCompositeIn {
    const float* rays_o;       /* [R,3] unit-sphere units                         */
    const float* rays_d;       /* [R,3]                                           */
    const float* z;            /* [R,S] sorted inside samples                     */
    const float* z_feed;       /* [R,S+O] sort(cat(z, z_out)) or NULL (no bg)     */
    const float* sample_dist;  /* [R]                                             */
    const float* sdf;          /* [R,S]                                           */
    const float* grad;         /* [R,S,3] d sdf / d x                             */
    const float* rgb;          /* [R,S,3]                                         */
    const float* density;      /* [R,S+O] raw NeRF density or NULL                */
    const float* bg_rgb;       /* [R,S+O,3] or NULL                               */
    const float* inv_s;        /* [1] device scalar exp(10 variance)              */
    const float* background_rgb; /* [3] or NULL                                   */
    const float* cos_anneal_dev; /* [1] device scalar or NULL: when set it replaces cos_anneal (so that a
                                    captured HIP graph of the step can be replayed with a new ratio)  */
    float cos_anneal;
    int32_t R, S, O, has_bg, trim_sphere;
} NcwCompositeIn;

typedef struct NcwCompositeOut {
    float* color;        /* [R,3] */
    float* color_sphere; /* [R,3] */
    float* color_bg;     /* [R,3] */
    float* weights;      /* [R,S+O] */
    float* weights_sum;  /* [R]   */
    float* cdf;          /* [R,S] prev_cdf */
    float* inside;       /* [R,S] */
    float* depth;        /* [R]   */
    float* normals;      /* [R,3] */
    float* eik;          /* [2,R] per-ray eikonal partials: row 0 = sum relax*(|g|-1)^2, row 1 = sum relax */
    float* mid_z;        /* [R,S] */
    float* dists;        /* [R,S] */
    float* bg_alpha;     /* [R,S+O] or NULL when !has_bg */
    float* weights_max;  /* [R] max of weights over the ray (render()'s "weights_max", renderer.py:905) or NULL */
} NcwCompositeOut;

typedef struct NcwCompositeGrad {
    const float* d_color;       /* [R,3] upstream */
    const float* d_weights_sum; /* [R]            */
    const float* d_depth;       /* [R]            */
    const float* d_eik_num;     /* [R]            */
    float* d_sdf;      /* [R,S]     */
    float* d_grad;     /* [R,S,3]   */
    float* d_rgb;      /* [R,S,3]   */
    float* d_density;  /* [R,S+O]   */
    float* d_bg_rgb;   /* [R,S+O,3] */
    float* d_inv_s;    /* [R]: every ray's term of d loss / d inv_s; the caller sums them (order-fixed) */
    float grad_scale;  /* multiplies every upstream cotangent on load (0 = 1): the fp16 mode's loss scale, so that the
                        * per-point adjoints the MLP backward kernels round to fp16 stay in its normal range; the caller
                        * divides it back out of the parameter gradients (NcwUnpackDesc.scale), d_a and d_inv_s */
    const float* grad_scale_dev; /* device scalar multiplied on top of grad_scale (the dynamic loss scale) or NULL */
} NcwCompositeGrad;

/* Dead-background elimination (renderer.py:637,693-708 with trim_sphere): the background NeRF's density / colour in
 * column i < S of the [R, S + O] background arrays is multiplied by 1 - inside_sphere[i] in the compositor, forward and
 * backward, where inside_sphere[i] belongs to PRIMARY sample i (section mid-point of z [R, S], the last section ending
 * sample_dist further) -- an index pairing, wherever the i-th point of z_feed lies; only the other columns need the NeRF
 * at all.  Writes the ray-sample indices (r * M + i, M = S + O, ray-major, ascending) of the columns to evaluate -- i < S
 * with inside_sphere[i] == 0 and every i >= S (the n_outside samples) -- into idx[R * M], the exclusive prefix of the
 * per-ray counts into ray_offsets[R + 1] and their number into count[1]. */
int ncw_bg_select(const float* rays_o, const float* rays_d, const float* z, const float* sample_dist, int R, int S,
                  int O, int32_t* idx, int32_t* ray_offsets, int32_t* count, void* stream);

int ncw_composite_fwd(const NcwCompositeIn* in, const NcwCompositeOut* out, void* stream);
int ncw_composite_bwd(const NcwCompositeIn* in, const NcwCompositeGrad* g, void* stream);

/* Per-ray loss terms that render() itself returns (renderer.py:763-765 gradient_error, :869-877 mask_error,
 * :892-897 sfm_depth_loss) and their backward.  All arrays [R] f32 (label int64, NULL = no masking);
 * mask_ids (HOST, <= 4 ids): labels whose rays get mask 0.  mask_error / sfm_depth_loss may be NULL (term off).
 * sfm_depth_loss is the sync-free form (depth-gt)^2 w [w>0] R / max(#{w>0},1): its mean over R equals the
 * reference's mean over the selected rays.  scalars: DEVICE float[3] = {gradient_error, sum(eik_den), count}. */
int ncw_ray_tail_fwd(const float* weights_sum, const int64_t* label, const int* mask_ids, int n_ids,
                     const float* depth, const float* depth_gt, const float* depth_weight, const float* eik_num,
                     const float* eik_den, int R, float* mask_error, float* sfm_depth_loss, float* scalars,
                     void* stream);
/* cotangents may be NULL (treated as zero); writes d_weights_sum, d_depth, d_eik_num [R] */
int ncw_ray_tail_bwd(const float* weights_sum, const int64_t* label, const int* mask_ids, int n_ids,
                     const float* depth, const float* depth_gt, const float* depth_weight, int R,
                     const float* scalars, const float* d_mask_error, const float* d_sfm_depth_loss,
                     const float* d_gradient_error, float* d_weights_sum, float* d_depth, float* d_eik_num,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * Iso-surface extraction: replaces `skimage.measure.marching_cubes(sdf, level, mask=...)` of
 * utils/visualization.py:114 by MARCHING CUBES on the same grid: one vertex per sign-changing grid edge at its linear
 * zero crossing (the vertex rule of every marching-cubes variant), triangles per cube from the 256-entry case tables
 * generated by neuralrecon-w_amd/mc_tables.py (tri [256][16] int8 edge triples, -1 padded; ntri [256]; edges [12][2]
 * corner pairs; corner k = offset (k&1, k>>1&1, k>>2&1), configuration bit k = value < level).  skimage's triangulation of
 * the ambiguous configurations is not reproducible without skimage: the vertex set and the geometry are what is pinned.
 * sdf: [Dx,Dy,Dz] f32 (x slowest).  mask: uint8 [Dx,Dy,Dz] or NULL; the cube with minimum corner (i,j,k) is processed
 * iff mask[i+1,j+1,k+1] (utils/visualization.py:103-112 builds the mask that way).
 *   ncw_mc_count: counts[(Dx-1)(Dy-1)(Dz-1)] int32 triangles per cube (cube index x slowest)
 *   ncw_mc_emit : offsets = EXCLUSIVE prefix sum of counts (int64); writes tri_pos [T,3,3] f32 in grid-index
 *                 coordinates and tri_key [T,3] int64 vertex ids (lo_point * Dx*Dy*Dz + hi_point: weld by key).
 *                 Faces are wound so that normals point towards increasing values.
 * ---------------------------------------------------------------------------------------- */
int ncw_mc_count(const float* sdf, const uint8_t* mask, int Dx, int Dy, int Dz, float level, const int32_t* ntri,
                 int32_t* counts, void* stream);
int ncw_mc_emit(const float* sdf, const uint8_t* mask, int Dx, int Dy, int Dz, float level, const int8_t* tri,
                const int32_t* ntri, const int32_t* edges, const int64_t* offsets, float* tri_pos, int64_t* tri_key,
                void* stream);

#ifdef __cplusplus
}
#endif
#endif
