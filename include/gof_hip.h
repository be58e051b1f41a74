/*
 * gof_hip.h -- C ABI of libgof_hip.so, the MI355X (gfx950) Gaussian-opacity-field rasterizer.
 *
 * This is the drop-in boundary for the reference's torch/pybind11 module
 * `diff_gaussian_rasterization._C`
 * (reference: submodules/diff-gaussian-rasterization/ext.cpp:15-20).  Each entry point
 * below names the reference interface it replaces.  Conventions:
 *
 *   - extern "C", plain pointers and sizes, no torch / HIP types in the signatures
 *     (`stream` is a hipStream_t passed as void*; NULL = the default stream).
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`.
 *   - all buffers are caller-owned; the library keeps no state between calls except a
 *     thread-local error string.
 *   - an absent optional input is NULL ("compute it"), exactly like the empty-tensor
 *     convention of the reference binding (rasterize_points.cu:98-118 passes the
 *     null data_ptr of `torch.Tensor([])`).
 *   - return value: 0 = ok, negative = error (see GOF_E_*); gof_last_error() gives text.
 *     No C++ exception crosses the boundary.
 *   - all launches are asynchronous on `stream` except where a function is documented
 *     as synchronising (gof_forward_prepare returns num_rendered to the host, like the
 *     reference's cudaMemcpy at rasterizer_impl.cu:336).
 *
 * Workspaces are opaque byte buffers whose sizes come from the gof_*_bytes() queries
 * (they replace GeometryState/ImageState/BinningState/PointState chunk sub-allocation,
 * rasterizer_impl.cu:188-243, rasterizer_impl.h:22-88).  The Python layer keeps them as
 * uint8 tensors so ctx.save_for_backward works as in the reference.
 */
#ifndef GOF_HIP_H_INCLUDED
#define GOF_HIP_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GOF_OK            0
#define GOF_E_INVALID    -1   /* bad argument (shape / NULL / size)                  */
#define GOF_E_WORKSPACE  -2   /* a workspace is smaller than the gof_*_bytes() query  */
#define GOF_E_DEVICE     -3   /* HIP runtime error (text in gof_last_error())         */
#define GOF_E_PREFILTER  -4   /* prefiltered=1 but a Gaussian was culled (debug only) */
#define GOF_E_CAPACITY   -5   /* gof_forward_fused: more instances than the caller's capacity (redo via prepare/render) */

#define GOF_OUTPUT_CHANNELS 9  /* auxiliary.h:21-24: rgb 0-2, normal 3-5, depth 6, alpha 7, distortion 8 */

/* All rasterization inputs of one view.  Mirrors the argument list of
 * RasterizeGaussiansCUDA (rasterize_points.cu:36-58) / Rasterizer::forward
 * (rasterizer_impl.cu:247-272).  POD, passed by pointer (host memory). */
typedef struct GofRasterArgs {
    int32_t P;              /* number of Gaussians                                            */
    int32_t D;              /* active SH degree (0..3)                                        */
    int32_t M;              /* SH coefficients per Gaussian in `shs` (sh.size(1)); 0 if none   */
    int32_t W, H;           /* image width / height                                           */
    float tan_fovx, tan_fovy;
    float kernel_size;      /* 2D mip low-pass added to the screen covariance                 */
    float scale_modifier;
    int32_t prefiltered;
    int32_t debug;          /* 1: synchronise + check after every launch (CHECK_CUDA, auxiliary.h:204-211) */
    const float* background;            /* [3]                                   */
    const float* means3D;               /* [P,3]                                 */
    const float* shs;                   /* [P,M,3] or NULL                       */
    const float* colors_precomp;        /* [P,3]   or NULL                       */
    const float* opacities;             /* [P]                                   */
    const float* scales;                /* [P,3]   or NULL (then cov3D_precomp AND view2gaussian_precomp) */
    const float* rotations;             /* [P,4]   or NULL, used as given (NOT re-normalised, forward.cu:138) */
    const float* cov3D_precomp;         /* [P,6]   or NULL                       */
    const float* view2gaussian_precomp; /* [P,10]  or NULL                       */
    const float* viewmatrix;            /* [16] row-vector convention (transposed), scene/cameras.py:56-58 */
    const float* projmatrix;            /* [16] full projection, same convention */
    const float* campos;                /* [3]                                   */
    const float* subpixel_offset;       /* [H,W,2]; must be valid (integrateCUDA reads it, forward.cu:845) */
    const float* shs_rest;              /* NULL (default): `shs` is [P,M,3].  Non-NULL: the SH coefficients as the reference STORES
                                           them (scene/gaussian_model.py:351-352) -- `shs` = _features_dc [P,1,3], `shs_rest` =
                                           _features_rest [P,M-1,3] -- so that the caller need not concatenate 192 B per Gaussian
                                           (GaussianModel.get_features, gaussian_model.py:173-176) every iteration.  M == 16 only. */
    /* PER-CALL modes (ABI 12).  0 = the process-wide default (gof_set_forward_exact / gof_set_tight_tile_rects /
     * gof_set_integrate_pixel_pass below: what a zero-initialised struct has always meant), > 0 = on for this call, < 0 = off for this
     * call.  The library keeps no state between calls (rasterize_points.cu:36-58 is stateless too): two threads, or two streams of one
     * thread, may run different modes side by side. */
    int32_t forward_exact;              /* the forward blend's verification mode (gof_set_forward_exact)                          */
    int32_t tight_tile_rects;           /* tile rectangles intersected with the footprint box (gof_set_tight_tile_rects)          */
    int32_t integrate_pixel_pass;       /* the opacity-field query's pixel pass in its pixel-centric form (gof_set_integrate_pixel_pass) */
    int32_t reserved0;                  /* 0 */
} GofRasterArgs;

/* ---- error text ----------------------------------------------------------------------- */
const char* gof_last_error(void);
/* library / ABI version, bumped when a signature or workspace layout changes */
int gof_abi_version(void);
/* Verification mode of the forward blend: the process-wide DEFAULT that calls with args->forward_exact == 0 use (returns the previous
 * setting; initial value: 1 if the environment variable GOF_FW_EXACT=1 is set when the library is loaded, else 0).
 *   0 (default): the arithmetic of the reference's renderCUDA (forward.cu:504-557) WITHOUT its two fp64 divisions per (pixel,
 *      Gaussian) pair: the quotient BB / AA comes from the fp32 reciprocal corrected twice in fp64 (faithfully rounded; everything
 *      behind it -- t, min_value, alpha, T, every threshold decision -- is evaluated as in mode 1), the mapped depth in fp32.
 *      n_contrib, contributor masks, colour / depth / alpha channels: those of mode 1 on every scene tested (a last bit of one pair
 *      in ~1e7 may differ: csrc/gof_common.h, pair_nodiv_cc); distortion channel within a few 1e-7.
 *   1: every pair in the reference's arithmetic as written (fp64 where forward.cu widens to double): every output bit is the oracle's. */
int gof_set_forward_exact(int on);
/* Tile lists: the process-wide default for calls with args->tight_tile_rects == 0 (returns the previous setting; initial value from
 * the environment variable GOF_TIGHT_RECTS=1).
 *   0 (default): a Gaussian is binned into every tile of the square of its 3-sigma radius, as getRect does (auxiliary.h:64-74):
 *      tiles_touched, the sorted lists and the ranges are the reference's entry for entry.
 *   1: that rectangle intersected with the conservative pixel box of the Gaussian's alpha >= 1/255 footprint -- tiles it cannot reach
 *      are dropped (R -21 % at 1M Gaussians @ 1600x1063).  Image, final_T, radii, the opacity-field query: unchanged bit for bit;
 *      gradients equal up to the summation order of the per-Gaussian gather; the intermediate lists are no longer the reference's. */
int gof_set_tight_tile_rects(int on);
/* Pixel pass of the opacity-field query, integrateCUDA's first half (forward.cu:886-993): the process-wide default for calls with
 * args->integrate_pixel_pass == 0 (returns the previous setting; initial value from the environment variable GOF_INT_PIXELS=1).
 *   0 (default, round 5): ray-centric -- the centre and corner sub-rays of neighbouring pixels that are the same ray bit for bit
 *      (pixf +- 0.5f is exact) are evaluated once per tile: 545 rays where thread = pixel evaluates 1280; tiles in which a pixel
 *      meets the 1024-contributor cap (forward.cu:986-990) are rendered by the pixel-centric kernel behind it.
 *   1: pixel-centric everywhere (rounds 1-4).  Every output bit is the same in both forms. */
int gof_set_integrate_pixel_pass(int on);

/* ---- workspace size queries (host only) ------------------------------------------------ */
/* replaces required<GeometryState>(P)  (rasterizer_impl.cu:277, 188-204) */
size_t gof_geom_bytes(int32_t P);
/* ... for a forward / backward pair (gof_forward_prepare / _render / _fused, gof_backward*, gof_sh_grad_pack): the same layout without its
 * tail, the footprints' pixel boxes that only the opacity-field query reads (16 B per Gaussian less; ABI 12).  The query's entry points
 * (gof_integrate_prepare / _view / _run / _points, gof_integrate_pack_geom) need gof_geom_bytes; a buffer of that size serves both. */
size_t gof_geom_bytes_forward(int32_t P);
/* replaces required<ImageState>(W*H)   (rasterizer_impl.cu:290, 218-228) */
size_t gof_image_bytes(int32_t W, int32_t H);
/* replaces required<BinningState>(num_rendered) (rasterizer_impl.cu:338, 230-243) */
size_t gof_binning_bytes(uint32_t num_rendered, int32_t W, int32_t H);
/* replaces required<PointState>(PN)    (rasterizer_impl.cu:676, 206-216) */
size_t gof_point_bytes(int32_t PN);
/* replaces required<BinningState>(num_integrated) for the query points (rasterizer_impl.cu:706): the sort state only,
 * without the contributor masks of gof_binning_bytes (a gof_binning_bytes-sized buffer is accepted too) */
size_t gof_point_binning_bytes(uint32_t num_integrated, int32_t W, int32_t H);
/* A SMALLER binning workspace for the forward / backward pair (round 4).  The contributor masks the forward blend leaves for the
 * backward -- one bit per (pixel of the tile, list entry) -- are stored in a pool of 2 KB sub-chunks (one wave's 64 pixels x one staged
 * batch of 256 entries), taken as the blend reaches them; the forward only reaches ~40 % of the lists.  gof_binning_bytes(R, W, H)
 * sizes the pool for the worst case, 4 (R / 256 + tiles + 2) sub-chunks = 32 B per instance (and is also what the opacity-field query
 * needs); gof_binning_bytes_for(R, W, H, n) sizes it for n sub-chunks (never below the sort state).  gof_forward_render /
 * gof_forward_fused / gof_backward derive the pool's capacity from the binning_bytes they are given.  A forward that needs more
 * sub-chunks than the pool holds still renders its image exactly, counts its requests and stores no masks beyond the capacity:
 * BEFORE that frame's backward, gof_backward_query reports requested vs. capacity, and the caller repeats the forward with
 * gof_binning_bytes_for(R, W, H, requested) bytes (the shipped binding sizes the pool 1.25 x the largest request seen so far, starting
 * from the worst case). */
size_t gof_binning_bytes_for(uint32_t num_rendered, int32_t W, int32_t H, uint32_t mask_subchunks);

/* ---- forward (replaces _C.rasterize_gaussians, rasterize_points.cu:36-122) ------------- */
/* Stage 1: preprocess (forward.cu:283-404) + inclusive scan of tiles_touched
 * (rasterizer_impl.cu:332) + read-back of the instance count (rasterizer_impl.cu:336).
 * Writes radii[P] (int32; 0 for culled Gaussians) and *num_rendered_host.  SYNCHRONISES
 * `stream` (one 4-byte D2H, as the reference). */
int gof_forward_prepare(const GofRasterArgs* args,
                        void* geom_ws, size_t geom_bytes,
                        void* image_ws, size_t image_bytes,
                        int32_t* radii,
                        uint32_t* num_rendered_host,
                        void* stream);
/* The same stage in front of the opacity-field query (gof_integrate_view / gof_integrate_run; rasterizer_impl.cu:530-700 runs the same
 * preprocessCUDA there): same arguments, same outputs, and the per-Gaussian footprints COMPLETE -- besides the ray-space conic the forward
 * blend culls with, the conservative pixel box and the front depth the query's pixel and point passes prefilter with (ABI 12: since round 6
 * gof_forward_prepare / gof_forward_fused leave those two at "no statement", which costs a training forward ~40 % fewer fp64
 * instructions per Gaussian; a query run on a workspace of theirs is still exact, only slower). */
int gof_integrate_prepare(const GofRasterArgs* args,
                          void* geom_ws, size_t geom_bytes,
                          void* image_ws, size_t image_bytes,
                          int32_t* radii,
                          uint32_t* num_rendered_host,
                          void* stream);
/* Stage 2: duplicateWithKeys + stable radix sort + identifyTileRanges + forward blend
 * (rasterizer_impl.cu:344-402, forward.cu:409-612).  out_color is [9,H,W]; every pixel of
 * every channel is written.  geom_ws must be the one gof_forward_prepare of THIS frame filled; image_ws need only be large
 * enough (the tile ranges are cleared here, as the reference's cudaMemset at rasterizer_impl.cu:365 does). */
int gof_forward_render(const GofRasterArgs* args,
                       uint32_t num_rendered,
                       const int32_t* radii,          /* [P] as written by gof_forward_prepare */
                       void* geom_ws, size_t geom_bytes,
                       void* binning_ws, size_t binning_bytes,
                       void* image_ws, size_t image_bytes,
                       float* out_color,
                       void* stream);

/* The two stages in ONE call without the pipeline bubble of the mid-forward read-back: the caller sizes the
 * binning workspace for `capacity` instances (e.g. 1.25 x the previous frame's count), every launch after the scan is sized for
 * the capacity and reads the actual count on the device, the count goes to PINNED host memory right after the scan and the host
 * waits for that copy only (~0.3 ms into the call, the rest already queued).  Returns GOF_E_CAPACITY -- with nothing written out
 * of bounds and *num_rendered_pinned_host = the required count -- when the capacity was too small: redo the frame with
 * gof_forward_prepare / gof_forward_render.  The backward and the introspection calls take `capacity` as their num_rendered
 * (it fixes the workspace layout).  Not for P == 0, prefiltered or debug calls.
 * usage_pinned_host (nullable): GOF_USAGE_WORDS uint32 of PINNED host memory that receive the frame's raw pool counters at the end of
 * the forward -- what gof_forward_usage_async would copy there; the caller records its own event behind the call and hands the words
 * to gof_usage_decode after waiting for it.  Both host buffers are written by the producing kernels themselves when the memory is
 * device-mapped (hipHostMalloc / torch's pin_memory()): no copy launches in the stream; otherwise by hipMemcpyAsync. */
int gof_forward_fused(const GofRasterArgs* args, uint32_t capacity,
                      void* geom_ws, size_t geom_bytes, void* binning_ws, size_t binning_bytes,
                      void* image_ws, size_t image_bytes,
                      int32_t* radii, float* out_color, uint32_t* num_rendered_pinned_host, uint32_t* usage_pinned_host, void* stream);

/* ---- backward (replaces _C.rasterize_gaussians_backward, rasterize_points.cu:124-211) --- */
/* Scratch the backward needs besides the outputs: per tile instance (num_rendered of them) one slot word, and a POOL of
 * 64-byte partial gradient records -- one per (tile, Gaussian) instance the per-pixel backward actually stages -- which the
 * per-Gaussian gather adds up (no atomics).  num_rendered = the value the backward is called with.
 *   gof_backward_scratch_bytes(P, R):                a pool of R records: always enough (every instance staged).
 *   gof_backward_query(...):                         [0] the number of entries the forward of this frame staged (~30 % of R at 1M
 *                                                    Gaussians @ 1600x1063), read from the image workspace; [1], [2]: the forward's
 *                                                    contributor-mask pool, see gof_binning_bytes_for.  SYNCHRONISES `stream`.
 *   gof_backward_scratch_bytes_for(P, R, staged):    the size for a pool of exactly that many records (4 + 64 x staged / R bytes per
 *                                                    instance instead of 72).
 * The backward derives the pool's capacity from the scratch_bytes it is given.  A pool smaller than what the frame stages (impossible
 * with either size above) drops the records that do not fit: see gof_forward_usage_async for the caller that sizes it from earlier frames. */
size_t gof_backward_scratch_bytes(int32_t P, uint32_t num_rendered);
size_t gof_backward_scratch_bytes_for(int32_t P, uint32_t num_rendered, uint32_t staged_entries);
int gof_backward_query(const GofRasterArgs* args, uint32_t num_rendered, size_t binning_bytes, const void* image_ws, size_t image_bytes,
                       uint32_t* out3_host /* [0] staged entries, [1] mask sub-chunks requested, [2] mask sub-chunks binning_bytes holds */,
                       void* stream);
/* The same three numbers without a synchronisation inside the library, for a caller that launches the backward OPTIMISTICALLY: pools
 * sized from earlier frames, the backward queued at once, the check afterwards.  A backward whose record pool was too small drops the
 * records that did not fit; one whose frame lacks masks skips those batches -- both write incomplete gradients and nothing out of
 * bounds, and the caller, who compares [0] with its pool and [1] with [2] before it uses the gradients, simply repeats what was short.
 *   gof_forward_usage_async: enqueue, behind the frame's forward on `stream`, the copy of the raw counters (GOF_USAGE_WORDS uint32)
 *                            into PINNED host memory; returns at once.  The caller records its own event behind it.
 *   gof_usage_decode:        host arithmetic only: the raw counters -> out3 as above (after the caller has waited for its event). */
#define GOF_USAGE_WORDS 66
int gof_forward_usage_async(const GofRasterArgs* args, const void* image_ws, size_t image_bytes, uint32_t* words_pinned_host, void* stream);
int gof_usage_decode(const uint32_t* words_host, uint32_t num_rendered, int32_t W, int32_t H, size_t binning_bytes, uint32_t* out3_host);
/* dL_dout is [9,H,W].  All gradient outputs are fully written by the call (the library
 * zero-fills them itself; the reference binding allocates them with torch::zeros,
 * rasterize_points.cu:161-170).  dL_dcov3D [P,6] is all zero in the reference (its producer
 * kernel is disabled, backward.cu:991-1007) and is zero-filled here if non-NULL. */
int gof_backward(const GofRasterArgs* args,
                 uint32_t num_rendered,
                 const int32_t* radii,
                 const void* geom_ws, size_t geom_bytes,
                 const void* binning_ws, size_t binning_bytes,
                 const void* image_ws, size_t image_bytes,
                 const float* dL_dout,
                 float* dL_dmeans2D,        /* [P,3] x,y signed; z = sum |.| (backward.cu:905-909) */
                 float* dL_dcolors,         /* [P,3]  */
                 float* dL_dopacity,        /* [P]    */
                 float* dL_dmeans3D,        /* [P,3]  */
                 float* dL_dcov3D,          /* [P,6] or NULL */
                 float* dL_dsh,             /* [P,M,3] or NULL when M == 0; [P,1,3] when args->shs_rest is given */
                 float* dL_dsh_rest,        /* [P,M-1,3] when args->shs_rest is given, else NULL */
                 float* dL_dscales,         /* [P,3]  */
                 float* dL_drotations,      /* [P,4]  */
                 float* dL_dview2gaussian,  /* [P,10] */
                 void* scratch, size_t scratch_bytes,  /* gof_backward_scratch_bytes(P, num_rendered) */
                 void* stream);

/* The two stages of gof_backward as separate calls with the SAME argument list: gof_backward_blend runs the per-pixel backward
 * and the per-Gaussian gather of its partial sums (afterwards dL_dmeans2D, dL_dcolors, dL_dopacity and dL_dview2gaussian are final),
 * gof_backward_preprocess turns them into the parameter gradients.  A data-parallel trainer starts the exchange of the colour
 * gradient between the two (dp/reducer.py); gof_backward = one after the other. */
#define GOF_BACKWARD_ARGS                                                                                                          \
    const GofRasterArgs* args, uint32_t num_rendered, const int32_t* radii, const void* geom_ws, size_t geom_bytes,               \
    const void* binning_ws, size_t binning_bytes, const void* image_ws, size_t image_bytes, const float* dL_dout,                 \
    float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,               \
    float* dL_dsh_rest, float* dL_dscales, float* dL_drotations, float* dL_dview2gaussian, void* scratch, size_t scratch_bytes,   \
    void* stream
int gof_backward_blend(GOF_BACKWARD_ARGS);
int gof_backward_preprocess(GOF_BACKWARD_ARGS);
#undef GOF_BACKWARD_ARGS

/* ---- integrate (replaces _C.integrate_gaussians_to_points, rasterize_points.cu:234-343) - */
/* Stage 1 for the query points: preprocessPointsCUDA + scan + count read-back
 * (forward.cu:722-766, rasterizer_impl.cu:681-702).  SYNCHRONISES `stream`. */
int gof_integrate_prepare_points(const GofRasterArgs* args,
                                 int32_t PN, const float* points3D,
                                 void* point_ws, size_t point_bytes,
                                 uint32_t* num_integrated_host,
                                 void* stream);
/* Stage 2: Gaussian binning (as gof_forward_render, without the blend), point binning
 * (createWithKeys + sort + ranges, rasterizer_impl.cu:720-752) and integrateCUDA
 * (forward.cu:803-1218).  out_alpha_integrated [PN] must be pre-filled with 1 and
 * out_color_integrated [PN,3] with 0 by the caller (rasterize_points.cu:277-278);
 * only points that project inside the image are written.  out_color is [9,H,W]; it must
 * be zero-filled by the caller (channels 3-5 are never written, forward.cu:1002-1007). */
int gof_integrate_run(const GofRasterArgs* args,
                      uint32_t num_rendered,
                      const int32_t* radii,           /* [P] as written by gof_forward_prepare */
                      int32_t PN, uint32_t num_integrated,
                      void* geom_ws, size_t geom_bytes,
                      void* binning_ws, size_t binning_bytes,
                      void* image_ws, size_t image_bytes,
                      void* point_ws, size_t point_bytes,
                      void* point_binning_ws, size_t point_binning_bytes,
                      float* out_color,
                      float* out_alpha_integrated,
                      float* out_color_integrated,
                      void* stream);

/* The two halves of gof_integrate_run, exposed so that a mesh-extraction driver runs the Gaussian side ONCE per view
 * for the 9-10 point sets it queries against unchanged Gaussians (extract_mesh.py:23-31, 88-100: evaluage_alpha is called
 * once per bisection step and loops over all views):
 *   gof_integrate_view:   Gaussian binning + the pixel pass (5 sub-rays per pixel, forward.cu:886-1007).  Leaves records,
 *                         sorted list, tile ranges and per-pixel contributor masks in the three workspaces and the base
 *                         image (channels 0-2, 6, 7; zero-filled by the caller beforehand) in out_color.  As for
 *                         gof_forward_render: geom_ws from gof_forward_prepare of this view, image_ws any buffer of the
 *                         stated size (its tile ranges are cleared here).
 *   gof_integrate_points: point binning + the point pass (forward.cu:1138-1217) for one point set, after
 *                         gof_integrate_prepare_points.  Reads the workspaces a previous gof_integrate_view of the SAME
 *                         args filled (they are not modified except image_ws's point ranges); base_color is that call's
 *                         image, out_color [9,H,W] receives it plus channel 8 (points per pixel) and may alias it.
 * Late errors: a point / tile sort whose bounded look-back poll expires (GPU heavily oversubscribed) leaves its ranges empty -- the
 * launches behind it then write NOTHING, so out_alpha_integrated / out_color_integrated keep what the caller put there (pre-fill
 * them: 1 and 0, as the reference's torch.ones / zeros at rasterize_points.cu:285-286) -- and raises a status word that the NEXT
 * call into the library (forward, backward or integrate entry point) returns as GOF_E_DEVICE. */
int gof_integrate_view(const GofRasterArgs* args, uint32_t num_rendered, const int32_t* radii,
                       void* geom_ws, size_t geom_bytes, void* binning_ws, size_t binning_bytes,
                       void* image_ws, size_t image_bytes, float* out_color, void* stream);
int gof_integrate_points(const GofRasterArgs* args, uint32_t num_rendered, int32_t PN, uint32_t num_integrated,
                         const void* geom_ws, size_t geom_bytes, const void* binning_ws, size_t binning_bytes,
                         void* image_ws, size_t image_bytes,
                         void* point_ws, size_t point_bytes, void* point_binning_ws, size_t point_binning_bytes,
                         const float* base_color, float* out_color,
                         float* out_alpha_integrated, float* out_color_integrated, void* stream);

/* A cached view only needs the 64-byte records and the front depths of the geometry workspace: gof_integrate_pack_geom copies
 * them into a buffer of gof_integrate_packed_geom_bytes(P) (68 B per Gaussian instead of ~220), gof_integrate_points_packed is
 * gof_integrate_points reading that buffer in place of geom_ws. */
size_t gof_integrate_packed_geom_bytes(int32_t P);
int gof_integrate_pack_geom(const GofRasterArgs* args, const void* geom_ws, size_t geom_bytes,
                            void* packed_geom, size_t packed_bytes, void* stream);
int gof_integrate_points_packed(const GofRasterArgs* args, uint32_t num_rendered, int32_t PN, uint32_t num_integrated,
                                const void* packed_geom, size_t packed_bytes, const void* binning_ws, size_t binning_bytes,
                                void* image_ws, size_t image_bytes,
                                void* point_ws, size_t point_bytes, void* point_binning_ws, size_t point_binning_bytes,
                                const float* base_color, float* out_color,
                                float* out_alpha_integrated, float* out_color_integrated, void* stream);

/* gof_integrate_points / _packed (packed != 0) with the reduction over views of extract_mesh.py:17-34 (evaluage_alpha) fused into
 * the store: alpha_min_inout [PN] holds the running minimum of alpha_integrated over the views queried so far (the caller fills it
 * with 1 before the first view) and color_min_inout [PN,3] (NULL: not wanted) the colour of the view that attained it:
 *     color = where(alpha < alpha_min, color_view, color);  alpha_min = min(alpha_min, alpha)      (NaN propagates as in torch.min)
 * A point outside this view is left untouched (its alpha_integrated would be 1).  out_color (the [9,H,W] image + channel 8) may be
 * NULL: a mesh-extraction driver does not read it. */
int gof_integrate_points_min(const GofRasterArgs* args, uint32_t num_rendered, int32_t PN, uint32_t num_integrated, int32_t packed,
                             const void* geom_ws_or_packed, size_t geom_bytes, const void* binning_ws, size_t binning_bytes,
                             void* image_ws, size_t image_bytes,
                             void* point_ws, size_t point_bytes, void* point_binning_ws, size_t point_binning_bytes,
                             const float* base_color, float* out_color,
                             float* alpha_min_inout, float* color_min_inout, void* stream);

/* ---- mark_visible (replaces _C.mark_visible, rasterize_points.cu:213-232) -------------- */
int gof_mark_visible(int32_t P, const float* means3D,
                     const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/* ---- data-parallel training: compressed exchange of the SH gradient --------------------
 * No reference counterpart (the reference trains on one GPU).  The SH gradient of one view,
 * computeColorFromSH's backward (backward.cu:20-139), is an outer product dL_dsh[k] = basis_k(dir) * dL_dRGB with
 * dL_dRGB = the blend's colour gradient masked by the forward's clamp flags: 12 bytes of information per Gaussian expanded
 * to 192.  View-sharded ranks all-gather dL_dRGB and expand the sum over views locally instead of all-reducing dL_dsh.
 * gof_sh_grad_pack: packed[P,3] = dL_dcolors (gof_backward's output) masked as backward.cu:36-38, zero for radii <= 0.
 * gof_sh_grad_expand: for every Gaussian i and coefficient k < M
 *     out[i][k] = scale * sum_{v < n_views} basis_k(normalize(means3D[i] - campos[v])) * packed[v][i]   (k < (D+1)^2, else 0)
 * views summed in index order; coefficient 0 is written to out_dc + i*stride_dc, coefficient k >= 1 to
 * out_rest + i*stride_rest + 3(k-1) (strides in floats).  One [P,M,3] tensor: out_dc = t, out_rest = t + 3, both strides 3M;
 * the reference's separate _features_dc [P,1,3] / _features_rest [P,M-1,3] (gaussian_model.py:351-352): strides 3 and 3(M-1).
 * View v reads packed + v*packed_stride ([P,3]) and campos + v*campos_stride ([3]); device arrays, strides in floats (an
 * all-gathered buffer of [P+1,3] rows per rank, the last row being the rank's camera centre: both strides 3(P+1)). */
int gof_sh_grad_pack(int32_t P, const float* dL_dcolors, const void* geom_ws, size_t geom_bytes, const int32_t* radii,
                     float* packed, void* stream);
int gof_sh_grad_expand(int32_t P, int32_t D, int32_t M, int32_t n_views, const float* means3D,
                       const float* campos, int64_t campos_stride, const float* packed, int64_t packed_stride, float scale,
                       float* out_dc, int64_t stride_dc, float* out_rest, int64_t stride_rest, void* stream);

/* ---- marching tetrahedra (replaces utils/tetmesh.py:47-138, pure torch in the reference) */
/* Three calls over two caller-owned workspaces.
 * Phase 0, gof_mtets_classify: the 4-bit case of every tet (1 byte per tet) and the number of tets the surface crosses, returned
 *   to the host (SYNCHRONISES `stream`).  tet_ws: gof_mtets_tet_ws_bytes(num_tets) -- about 1 byte per tet.
 * Phase 1, gof_mtets_count: collect the unique crossing edges (sorted ascending by (min vertex, max vertex), the order
 *   torch.unique(dim=0) produces, tetmesh.py:110) and count faces; SYNCHRONISES, returns the counts.  edge_ws:
 *   gof_mtets_edge_ws_bytes(num_valid_tets of phase 0) -- about 200 bytes per VALID tet (may be NULL when there is none).
 * Phase 2, gof_mtets_emit: edge end-point ids [E,2] (int64), end-point positions [E,2,3], end-point sdf [E,2], end-point scales
 *   [E,2] and faces [F,3] (int64) in the reference's order (per 32 Mi-tet chunk all 1-triangle tets first, then the 2-triangle
 *   tets; tetmesh.py:55-95, 126-136).
 * Vertex ids must be < 2^32 and 6 * num_tets < 2^32. */
size_t gof_mtets_tet_ws_bytes(int64_t num_tets);
size_t gof_mtets_edge_ws_bytes(int64_t num_valid_tets);
int gof_mtets_classify(int64_t num_verts, int64_t num_tets, const int64_t* tets /* [Tt,4] */, const float* sdf /* [V] */,
                       void* tet_ws, size_t tet_ws_bytes, int64_t* num_valid_tets_host, void* stream);
int gof_mtets_count(int64_t num_verts, int64_t num_tets, const int64_t* tets, const float* sdf,
                    void* tet_ws, size_t tet_ws_bytes, void* edge_ws, size_t edge_ws_bytes,
                    int64_t* num_edges_host, int64_t* num_faces_host, void* stream);
int gof_mtets_emit(int64_t num_verts, int64_t num_tets,
                   const int64_t* tets, const float* vertices /* [V,3] */,
                   const float* sdf, const float* scales /* [V] */,
                   const void* tet_ws, size_t tet_ws_bytes, const void* edge_ws, size_t edge_ws_bytes,
                   int64_t num_edges, int64_t num_faces,
                   int64_t* edge_ids, float* edge_pos, float* edge_sdf, float* edge_scales,
                   int64_t* faces, void* stream);

/* ---- per-kernel timing (replaces the reference's torch.cuda.Event pair, train.py:103-104,126,191) ---- */
/* When enabled, every kernel launch of the library is bracketed by HIP events on the launch stream. */
int gof_profile_enable(int on);
/* Waits for the recorded events, writes {"kernel": {"calls": n, "total_ms": t}, ...} (JSON, NUL-terminated)
 * into buf and clears the records. */
int gof_profile_report(char* buf, size_t cap);

/* ---- introspection for tests / benchmarks (no reference counterpart) ------------------- */
/* Copies one named intermediate array out of the workspaces into `dst` (device memory).
 * names: "depths" f32[P], "means2D" f32[P,2], "conic_opacity" f32[P,4], "rgb" f32[P,3],
 * "view2gaussian" f32[P,10], "tiles_touched" u32[P],
 * "clamped" u8[P,3], "point_list" u32[R], "point_list_keys" u64[R], "ranges" u32[T,2],
 * "final_T" f32[4,H,W], "n_contrib" u32[2,H,W].  Returns the element count or <0. */
int64_t gof_debug_fetch(const char* name, const GofRasterArgs* args, uint32_t num_rendered,
                        const void* geom_ws, const void* binning_ws, const void* image_ws,
                        void* dst, size_t dst_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GOF_HIP_H_INCLUDED */
