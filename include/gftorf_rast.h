/*
 * gftorf_rast.h -- C ABI of libgftorf_rast.so: the MI355X (gfx950) differentiable
 * ToF Gaussian rasterizer.
 *
 * This is the drop-in boundary for the reference's native extension
 * `diff_gaussian_rasterization_w_tof._C` (RAST/ext.cpp:15-19, where RAST/ =
 * submodules/diff-gaussian-rasterization-w-tof/ of brownvc/gftorf):
 *
 *   _C.rasterize_gaussians           (RAST/rasterize_points.cu:42-165)
 *        -> gft_forward_preprocess() + gft_forward_render(), or gft_forward()
 *   _C.rasterize_gaussians_backward  (RAST/rasterize_points.cu:167-281)
 *        -> gft_backward()
 *   _C.mark_visible                  (RAST/rasterize_points.cu:283-304)
 *        -> gft_mark_visible()
 *
 * Plain pointers and sizes only, no torch types.  All pointers are DEVICE
 * pointers unless a comment says "host".  The library keeps no RESULT between
 * calls -- what it owns is plumbing: a pinned mailbox through which the device
 * posts counts to gft_forward / gft_forward_preprocess, one side stream with its
 * events per device (the opt-in gradient fill), the event pool of the opt-in
 * profiler, the process-wide mode switches below --; the caller owns every buffer,
 * schedules included (tile_hints, tile_weights, cell_sched), and the three scratch buffers
 * (reference: geomBuffer / binningBuffer / imgBuffer, RAST/rasterize_points.cu:
 * 94-101), which must survive from forward to backward.  A NULL input pointer
 * means "tensor absent" (reference: `.data<float>()` of an empty tensor is
 * nullptr and is tested at RAST/cuda_rasterizer/forward.cu:310,346,353,365,389).
 * Every entry point returns 0 on success, non-zero on error; the message is
 * available from gft_last_error() (thread local).
 */
#ifndef GFTORF_RAST_H
#define GFTORF_RAST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFT_ABI_VERSION 14

/* compile-time constants of the reference (RAST/cuda_rasterizer/config.h:15-23) */
#define GFT_NUM_CHANNELS 3
#define GFT_NUM_CHANNELS_PHASOR 7
#define GFT_NUM_CHANNELS_CWTOF 2
#define GFT_TILE_X 16
#define GFT_TILE_Y 16
/* floats per Gaussian in the backward accumulator table (15 used + 1 pad = one 64-byte row) */
#define GFT_ACC_STRIDE 16

/* Scalar arguments of one rasterizer call: the non-tensor fields of
 * GaussianRasterizationSettings (RAST/diff_gaussian_rasterization_w_tof/
 * __init__.py:22-40) plus the sizes RasterizeGaussiansCUDA derives
 * (RAST/rasterize_points.cu:73-75,106-122). */
typedef struct gft_config {
    int32_t P;       /* number of Gaussians */
    int32_t D;       /* active SH degree (0..3) */
    int32_t M;       /* colour SH coefficients per Gaussian in memory (0 = shs absent) */
    int32_t M_p;     /* phasor SH coefficients per Gaussian in memory (0 = shs_p absent) */
    int32_t W, H;    /* image size */
    float tanfovx, tanfovy;
    float scale_modifier;
    float near_n, far_n;
    float depth_range;
    float phase_offset, dc_offset;
    int32_t use_view_dependent_phase;
    int32_t prefiltered;
    int32_t debug;   /* synchronise + check after every stage (reference CHECK_CUDA) */
    /* forward: also store the 64-byte per-Gaussian record d(colour, phase, amplitude)/d(view
     * direction) so the backward does not have to re-read the 320 B of SH coefficients;
     * backward: that record is present in `geom` (same value as in the forward call) */
    int32_t want_backward;
    /* backward only: 0 = the library clears `acc` first; 1 = `acc` is zero (the forward cleared it, gft_forward_io.acc):
     * skip the clear; 2 = `acc` is zero AND is to be left zero: the backward's preprocess kernel zeroes every accumulator
     * row it has read (the rows of the blended Gaussians are the only non-zero ones), so a caller that keeps the buffer
     * hands the next forward no `acc` to clear (64 B per Gaussian of HBM writes per frame saved) */
    int32_t acc_zeroed;
    int32_t grads_zeroed;   /* gft_backward: 1 = the gradient outputs were zeroed by the forward (gft_forward_io.grads_zero),
                               2 = they are zero because the caller kept them and re-zeroed the rows the previous
                               backward wrote (gft_grads_rezero, gft_backward_io.dirty_rows): only the rows of Gaussians
                               that some pixel blended are written; 3 = the caller kept them as the previous backward left
                               them (zero but for the rows marked in dirty_rows): this backward zeroes the marked rows it does
                               not rewrite, writes its own and leaves their marks */
    int32_t grads_accumulate; /* gft_backward: the per-Gaussian gradient outputs already hold the gradients of another view
                               of the same Gaussians (the colour / ToF camera pair of one training iteration,
                               gaussian_renderer/__init__.py:107-128): the rows of the Gaussians this view blended are
                               ADDED to, all other rows are left alone -- no second 376 B/Gaussian of zeros, no separate
                               sum of two dense tensors.  dL_dphase_offset / dL_ddc_offset are written, not added.
                               Needs the forward's direction-gradient records (want_backward). */
    /* background [7,H,W] addressed as bg[c*sc + y*sy + x*sx] (element strides), so
     * the reference's expanded constant background (train.py:127) needs no copy */
    int64_t bg_stride_c, bg_stride_y, bg_stride_x;
} gft_config;

/* Tensors of the forward call (argument order of RAST/rasterize_points.cu:42-67). */
typedef struct gft_forward_io {
    /* inputs */
    const float* bg;               /* [7,H,W] via strides in gft_config */
    const float* means3D;          /* [P,3] */
    const float* colors_precomp;   /* [P,3] or NULL */
    const float* phasors_precomp;  /* [P,2] or NULL */
    const float* opacities;        /* [P] */
    const float* scales;           /* [P,3] or NULL */
    const float* rotations;        /* [P,4] (r,x,y,z) or NULL */
    const float* cov3D_precomp;    /* [P,6] or NULL */
    const float* viewmatrix;       /* 16 floats, transposed storage */
    const float* projmatrix;       /* 16 floats, transposed storage (full projection) */
    const float* campos;           /* 3 floats */
    const float* shs;              /* [P,M,3] or NULL */
    const float* shs_p;            /* [P,M_p,2] or NULL */
    /* scratch, caller-owned, sizes from gft_geom_bytes / gft_image_bytes /
     * gft_binning_bytes; contents are the forward->backward hand-off */
    void* geom;
    void* img;
    void* binning;                 /* may be NULL for gft_forward_preprocess */
    /* outputs (all written in full by the library; no pre-zeroing needed) */
    float* out_color;              /* [3,H,W] */
    float* out_phasor;             /* [7,H,W] */
    float* out_depth;              /* [1,H,W] */
    float* out_normal;             /* [3,H,W] always 0 (reference never writes it) */
    float* out_acc;                /* [1,H,W] */
    float* out_entropy;            /* [1,H,W] always 0 */
    float* out_depth_distortion;   /* [1,H,W] */
    float* out_amp_distortion;     /* [1,H,W] always 0 */
    float* pixels;                 /* [P,1] contributing-pixel count per Gaussian */
    float* out_distribution;       /* [3,H,W] first-hit (alpha, dist, amplitude/d^2) */
    int32_t* radii;                /* [P] */
    /* optional: the backward's accumulator (gft_acc_bytes(P)).  With want_backward the forward
     * clears it as a side job of the LDS-bound tile sort (64 B/Gaussian of HBM writes that
     * would otherwise be a separate pass of the backward); pass cfg.acc_zeroed to gft_backward.
     * NULL: nothing is cleared (the caller's buffer is zero already, e.g. left so by a backward with
     * cfg.acc_zeroed = 2). */
    float* acc;
    /* optional: the buffer that holds the backward's per-Gaussian gradient tensors (any layout, `grads_zero_bytes`
     * bytes).  The forward zero-fills it on a library-owned side stream while its render kernel (bound by VALU issue,
     * little HBM traffic) runs; with cfg.grads_zeroed the backward then writes only the rows of the Gaussians that
     * were blended -- in a dense frame most are not -- instead of streaming ~376 B of zeros per Gaussian. */
    void* grads_zero;
    size_t grads_zero_bytes;
    /* optional (tile-pull binning): uint32[T] kept by the caller from one forward of this image size to the next (zero
     * before the first; T = tiles).  A SCHEDULE, never a result: every forward stores, per 8x8 pixel quadrant (one byte
     * each), whether that quadrant walked past where the sorted head of its tile's list ends (about 940 entries); the next
     * forward sorts the WHOLE list of a tile with a non-zero word up front -- one pass of k_tile_pull in chunks of whole
     * depth bins (gft_forward_hints.whole_lists) -- or, with the heads-only build of that kernel, gives it the longest head
     * one placement holds (2047 entries), instead of a head now and the rest on demand (flag, k_tail_build, resume pass:
     * the silhouette tiles of a dense view, whose quadrants mostly end inside such a head).  A frame in which
     * nothing saturates (the reference's scenes right after an opacity reset, arguments/__init__.py:99) has every tile
     * take the on-demand route: 0.57 ms of a 1.8 ms step at 1 M Gaussians.  Images, counts and gradients do not depend
     * on the hints (the blend walks the same entries in the same order either way).  Frames whose forward blend is
     * segment-parallel (fewer than 768 tiles, gft_set_render_mode) neither read nor write it: that kernel cuts a list by
     * the length of its sorted part.  NULL: no schedule. */
    uint32_t* tile_hints;
    /* optional, beside tile_hints (same owner, same lifetime, zero before the first use): uint32[4 T + 4].  The forward
     * leaves every 8x8 quadrant's walk length there (and sets word 4 T to 1); the next forward of this image size and
     * camera deals its quadrant waves to the chip heaviest tile first by those lengths -- and so does that frame's
     * backward, which otherwise goes by the lengths of its own frame -- all quadrant waves of a 640x480 frame are resident together, so the order decides which
     * waves share a SIMD, and equal shares end together.  The order is derived anew on the device from whatever the words
     * hold (any contents give a permutation of the tiles): a schedule, never a result.  Neither read nor written on frames
     * of more than 4096 tiles (their many rounds of waves balance by themselves, and image order keeps neighbouring tiles on
     * one XCD's L2: measured) or with the segment-parallel forward.  NULL: tiles in image order. */
    uint32_t* tile_weights;
    /* optional, tile-pull binning (same owner and lifetime as tile_hints, zero before the first use):
     * uint32[gft_cell_sched_words(W, H)] -- where every (supertile, depth slab) list of the camera's next frame starts in the
     * entry array and how many entries it may take: this frame's counts plus a quarter plus 64, written by whichever pass
     * closes the frame's binning front end.  With gft_forward_hints.use_cell_sched the forward appends its entries to
     * those lists directly and runs no count pass (14 us of a 0.38 ms step at 1 M Gaussians); a list that outgrows its
     * capacity, or words that are no schedule, are found on the device: nothing is rendered from them, gft_forward runs the
     * counted flow in the same call.  Any contents are safe.  NULL: count, then scatter, every frame. */
    uint32_t* cell_sched;
} gft_forward_io;

/* Tensors of the backward call (RAST/rasterize_points.cu:167-198). */
typedef struct gft_backward_io {
    /* forward inputs again */
    const float* bg;
    const float* means3D;
    const int32_t* radii;
    const float* scales;
    const float* rotations;
    const float* cov3D_precomp;
    const float* viewmatrix;
    const float* projmatrix;
    const float* campos;
    const float* shs;
    const float* shs_p;
    const float* opacities;                  /* [P] as given to the forward */
    const float* pixels;                     /* [P] the forward's `pixels` output (contributing pixels per Gaussian), or NULL:
                                                Gaussians nobody blended get their zero gradients without their records being read */
    /* upstream gradients, contiguous [C,H,W]; NULL = all zeros.  Gradients of
     * normal / entropy / amp_distortion / pixels / distribution are accepted by
     * the reference and ignored by its kernels, so they are not part of the ABI */
    const float* dL_dout_color;              /* [3,H,W] */
    const float* dL_dout_phasor;             /* [7,H,W] */
    const float* dL_dout_depth;              /* [1,H,W] */
    const float* dL_dout_acc;                /* [1,H,W] */
    const float* dL_dout_depth_distortion;   /* [1,H,W] */
    /* forward scratch */
    const void* geom;
    const void* img;
    const void* binning;
    /* backward scratch of gft_acc_bytes(P) bytes: [P, GFT_ACC_STRIDE] floats (zeroed by the
     * library, here or -- cfg.acc_zeroed -- already in the forward) followed by two partial
     * sums per 64-Gaussian block */
    float* acc;
    /* outputs, written in full (zeros for culled Gaussians); NULL = not wanted */
    float* dL_dmeans3D;     /* [P,3] */
    float* dL_dmeans2D;     /* [P,3] (x,y in NDC-derivative units, z = 0) */
    float* dL_dcolors;      /* [P,3]  grad of colors_precomp, may be NULL */
    float* dL_dopacity;     /* [P,1] */
    float* dL_dcov3D;       /* [P,6]  grad of cov3D_precomp, may be NULL */
    float* dL_dsh;          /* [P,M,3]   required iff shs   != NULL */
    float* dL_dsh_p;        /* [P,M_p,2] required iff shs_p != NULL */
    float* dL_dscales;      /* [P,3]  required iff scales != NULL */
    float* dL_drotations;   /* [P,4]  required iff scales != NULL */
    float* dL_dphase_offset;/* [1] (written by the library); NULL together with dL_ddc_offset = not wanted (saves a reduction launch) */
    float* dL_ddc_offset;   /* [1] */
    /* optional: deterministic reduction (tests).  A buffer of gft_det_partials_bytes(binning_instances) bytes: every
     * (list entry, pixel quadrant) then stores its 16-float partial row there instead of adding it with float atomics,
     * and one workgroup adds the rows tile by tile, entries in list order, quadrants 0..3: two runs give bit-identical
     * gradients (the reference's atomicAdd sums, backward.cu:795-886, have no defined order).  NULL = float atomics. */
    float* det_partials;
    /* optional: uint8[P (rounded up to a multiple of 4)] marks of the rows a backward writes (non-zero gradient rows: the
     * Gaussians some pixel blended).  A caller that keeps its gradient tensors from one backward to the next passes the
     * same array every time: gft_grads_rezero() zeroes exactly the marked rows (and clears the marks), after which the
     * tensors are all zero again and the next backward may run with cfg.grads_zeroed = 2: it then writes the rows of the
     * blended Gaussians only, instead of streaming ~376 B of zeros for every other Gaussian. */
    uint8_t* dirty_rows;
    /* optional, with grads_zeroed = 2 / 3: pinned host memory (one uint32) into which the backward stores the number of
     * gradient rows it wrote -- the Gaussians some pixel blended -- when it ends (no wait, no copy: a caller reads it
     * whenever it likes and finds the most recent backward's count).  dirty_rows must then be followed by 144 more bytes
     * (device counter + tickets, zero before the first use).  The rows-only backward pays when few rows are written (a
     * dense frame: 7 %); a frame that blends most of its Gaussians is better served by a full write (grads_zeroed = 0
     * into the same tensors: marks every row): gftorf_amd/api.py switches on this count. */
    uint32_t* rows_report;
} gft_backward_io;

/* Byte offsets of the sub-arrays inside the scratch buffers (the forward <->
 * backward layout contract; reference: GeometryState/ImageState/BinningState::
 * fromChunk, RAST/cuda_rasterizer/rasterizer_impl.cu:161-211).  Exposed so
 * tests can inspect intermediate state stage by stage. */
typedef struct gft_layout {
    /* geom */
    size_t geom_rec_a;        /* float[P][8]  {x,y, conic a,b,c, opacity, dist_ndc, dist} of every visible Gaussian */
    size_t geom_rec_b;        /* float[P][8]  {r,g,b, R,I,Am (ToF phasor basis: cos,sin,1 times A/d^2), phase_sh, amplitude}; with
                                 tile-pull binning only for the Gaussians marked in geom_need (like dirgrad, clamped) */
    size_t geom_depth;        /* float[P]     view-space z (sort key bits) */
    size_t geom_tiles;        /* uint32[P]    tiles touched */
    size_t geom_rect;         /* uint16[P][4] tile rectangle {x0,y0,x1,y1} (all 0 when culled) */
    size_t geom_dirgrad;      /* float[P][16] d rgb/d dir (9), d (phase,amp)/d dir (6), pad; only with want_backward */
    size_t geom_clamped;      /* uint8[P]     bit0..2 rgb clamped, bit3 amplitude clamped */
    size_t geom_need;         /* uint8[P]     tile-pull binning: 1 = the Gaussian stands in the sorted part of a tile list and has
                                 its appearance records */
    size_t geom_blockhist;    /* uint16[ceil(P/4096)][2048] instances per (4096-Gaussian block, tile); whole-frame binning, T <= 2048 */
    size_t geom_total;
    /* img */
    size_t img_pix_state;     /* float[N][4]  {final_T, n_contrib(bits), w_z, w_z2} */
    size_t img_ranges;        /* uint32[T][2] [first,last) of the tile's list */
    size_t img_tile_max;      /* uint32[T][4] max n_contrib over each 8x8 quadrant of the tile */
    size_t img_ctrl;          /* uint32[16]   {R, flags, max tile list length, ...}; 32 ticket counters, then tile_cnt, tile_cut, super_tab follow directly */
    size_t img_tile_cnt;      /* uint32[T]    instances per tile (tile-pull binning with depth slabs: of the slabs a tile scanned) */
    size_t img_tile_cut;      /* uint32[T]    tile-pull binning: first depth bin behind the sorted head of the tile's list;
                                 0xffffffff: the head is the whole list */
    size_t img_super_tab;     /* uint32[4][16384] tile-pull binning: per (counter copy, supertile of S x S tiles, depth slab) entry
                                 counters and scatter cursors (up to 8 copies spread the same-address atomics), per (supertile,
                                 slab) entry count and list start */
    size_t img_tile_cursor;   /* uint32[T]    scatter cursors */
    size_t img_tile_order;    /* uint32[T]    tiles by backward weight, heaviest first */
    size_t img_front_len;     /* uint32[T]    length of the sorted head of the tile's id list */
    size_t img_unit_flag;     /* uint32[4T]   non-zero: the quadrant ran out of sorted ids before saturating (bit 31 + the bounding box
                                 of its unsaturated pixels, 4 x 3 bits) */
    size_t img_resume_state;  /* float[N][16] blend state of such quadrants' pixels */
    size_t img_pix_sums;      /* float[N][8]  final blend sums {C0,C1,C2,R | I,Am,dist,A}: the split backward starts mid-list from them */
    size_t img_snaps;         /* float[4T][S-1][12][64] blend state of every 8x8 quadrant in front of list entries 256, 512, ...
                                 (S = up to 8 segments): where the other waves of a split backward walk start */
    size_t img_total;
    /* binning */
    size_t bin_keys;          /* uint64[R]    whole-frame binning: (depth bits << 32 | Gaussian id) grouped by tile, unsorted;
                                 tile-pull binning: (Gaussian, supertile) entries id | rectangle << 32 | depth bin << 52 */
    size_t bin_point_list;    /* uint32[2048 T + R] Gaussian ids, per tile ascending (depth bits, id).  Whole-frame binning: [0, R);
                                 tile-pull binning: the sorted head of tile t at [2048 t, ...), lists completed on demand
                                 (head + the part of the tail that reaches a quadrant that asked) from 2048 T on */
    size_t bin_total;
} gft_layout;

/* per-stage GPU time in milliseconds, accumulated while profiling is enabled */
typedef struct gft_profile {
    double preprocess_fwd_ms, tile_count_ms, tile_scatter_ms, tile_sort_ms, render_fwd_ms;
    double render_bwd_ms, preprocess_bwd_ms, memset_ms;
    int64_t forward_calls, backward_calls;
} gft_profile;

int gft_abi_version(void);
/* 1: tile lists are sorted head first, tail on demand (default); 0: whole (GFT_LAZY_SORT=0) */
int gft_lazy_sort(void);
const char* gft_last_error(void);   /* host string, thread local */

size_t gft_geom_bytes(int32_t P);
size_t gft_image_bytes(int32_t W, int32_t H);
/* words of gft_forward_io.cell_sched for this image size (tile-pull binning; 0 where it does not apply) */
size_t gft_cell_sched_words(int32_t W, int32_t H);
size_t gft_binning_bytes(int64_t R, int32_t W, int32_t H);
size_t gft_acc_bytes(int32_t P);
size_t gft_det_partials_bytes(int64_t binning_instances, int32_t W, int32_t H);   /* gft_backward_io.det_partials: 256 bytes per list slot
                                                                                   * (tiles x head slots + instances: 4.3 GB at 1080p) -- a test mode */
/* instances a binning buffer of `bytes` bytes holds (inverse of gft_binning_bytes for multiples of 64) */
int64_t gft_binning_capacity(size_t bytes, int32_t W, int32_t H);
/* 1: tile-pull binning (default): ids to supertiles, every tile pulls and sorts the head of its own list, lists are
 * completed on demand.  0: whole-frame binning, the structure of the reference (every instance counted, keyed, sorted);
 * also GFT_LAZY_BIN=0 in the environment.  Results are identical.  Process-wide; meant for tests and tuning. */
int gft_set_binning_mode(int mode);
int gft_binning_mode(const gft_config* cfg);   /* the mode a forward with this config runs in */
/* Forward blend kernel.  -1 (default): on frames with fewer than 768 tiles -- where one wave per 8x8 pixel quadrant
 * leaves most of the chip idle -- every quadrant's list is cut into segments that several waves blend side by side
 * (k_render_fwd_seg); larger frames run one wave per quadrant.  0: always one wave per quadrant;
 * 1: segments wherever the tile count allows more than one wave.  Also GFT_FWD_SEG=0 / 1 in the environment.  The two
 * kernels agree to fp32 rounding of the transmittance products (not bit for bit); each is deterministic.
 * Process-wide; meant for tests and tuning. */
int gft_set_render_mode(int mode);
int gft_get_layout(int32_t P, int32_t W, int32_t H, int64_t R, gft_layout* out /*host*/);

/* The forward in two stages, shaped like the reference's resize callbacks
 * (RAST/rasterize_points.cu:27-33): the binning buffer is sized after stage 1.
 *
 * Stage 1: preprocess (reference K1), per-tile instance counts and their scan, then
 * the one blocking read of the number of (Gaussian, tile) instances, where the
 * reference blocks too (RAST/cuda_rasterizer/rasterizer_impl.cu:311).  The device
 * posts the totals into pinned host memory which this call polls.
 * *num_rendered (host) receives R; the caller then sizes `binning` for R instances. */
int gft_forward_preprocess(void* hip_stream, const gft_config* cfg,
                           const gft_forward_io* io, int64_t* num_rendered /*host*/,
                           int64_t* max_tile_list /*host, may be NULL: longest per-tile list*/);

/* Stage 2: per-tile key scatter, per-tile sort, per-tile blend.
 * `binning_instances` = the instance count `binning` was sized for (>= R); the same
 * value goes to gft_backward, it fixes the layout inside the buffer. */
int gft_forward_render(void* hip_stream, const gft_config* cfg,
                       const gft_forward_io* io, int64_t binning_instances,
                       int64_t max_tile_list /* from stage 1; <= 0 = unknown */);

/* What a caller knows before the forward (from the previous frame of the same kind), for gft_forward(). */
typedef struct gft_forward_hints {
    int64_t binning_instances;   /* instances `io->binning` holds (the caller's guess of R plus headroom) */
    int64_t max_tile_list;       /* whole-frame binning without the lazy sort: guess of the longest per-tile list (<= 0 = unknown) */
    int64_t whole_lists;         /* tile-pull binning with io->tile_hints: non-zero = run the build of the pull kernel that sorts hinted
                                  * tiles' whole lists (30.8 KB of LDS, five workgroups per CU); 0 = the heads-only build (20.5 KB, seven
                                  * per CU), which ignores the schedule.  A caller sets it when the previous frame of the shape
                                  * reported enough hinted tiles (gft_forward_report.hinted_tiles; api.py: a sixteenth of the tiles) --
                                  * a schedule like the hints themselves: results do not depend on it */
    int64_t use_cell_sched;      /* non-zero = io->cell_sched was left by an earlier frame of this camera and image size: bin by it
                                  * without a count pass (gft_forward only; gft_forward_enqueue, which cannot fall back, counts) */
} gft_forward_hints;

/* What the device reported while the forward was running. */
typedef struct gft_forward_report {
    int64_t num_rendered;        /* R, the number of (Gaussian, tile) instances of the frame (reference: num_rendered) */
    int64_t max_tile_list;       /* longest per-tile list (whole-frame binning; 0 with tile-pull binning) */
    int64_t list_entries;        /* entries that were scattered: (Gaussian, supertile) pairs with tile-pull binning, else R */
    int64_t hinted_tiles;        /* non-zero words of io->tile_hints as this frame found them (tile-pull binning; else 0) */
    int64_t sched_misses;        /* 1: hints->use_cell_sched was set and the schedule did not hold the frame's lists -- the counted flow ran */
} gft_forward_report;

/* The forward in one call, for callers that can guess R (a training loop: R of the
 * previous iteration plus headroom).  `io->binning` holds `hints->binning_instances`
 * instances.  Both stages are queued back to back, so the device never waits for the
 * host; the call returns as soon as the device has posted R (stage 2 may still run).
 * If report->num_rendered > hints->binning_instances the stage-2 kernels have done nothing (they
 * compare the device-side count themselves): allocate for report->num_rendered and call
 * gft_forward_render().  The size of the buffer is the only thing taken from earlier frames: what is binned,
 * sorted and given an appearance is decided from the frame itself (tile-pull binning).
 *
 * Results are identical to the two-stage flow in every case (same lists, same arithmetic order). */
int gft_forward(void* hip_stream, const gft_config* cfg, const gft_forward_io* io,
                const gft_forward_hints* hints, gft_forward_report* report /*host*/);

/* The forward queued only: both stages back to back and NO read of anything the device computes -- the call is launch
 * work alone, so it can be captured in a HIP graph (torch.cuda.graphs around a fixed-shape training iteration) and never
 * stalls the host.  What gft_forward() reads from the device while it runs, the device posts into `status` instead:
 * GFT_STATUS_WORDS uint32 of pinned host memory or device memory owned by the caller (may be NULL), cleared by this call
 * on the stream and written when stage 1 ends: status[0] = R (num_rendered), status[1] bit 0 = "prefiltered point was
 * culled" (the reference's error, rasterizer_impl.cu: trap), status[2] = longest tile list (whole-frame binning),
 * status[3] = 1 once posted, status[5] = list entries, status[8] = hinted tiles.  If R > hints->binning_instances the stage-2 kernels have done
 * nothing (they compare the count on the device): the outputs of that frame are undefined and the caller, reading
 * `status` once the stream has passed (in front of its next call, say), renders again with a larger buffer.
 * A caller that runs ahead of the device (the purpose of the mode) may find the posting of frame N overwritten by frame
 * N + 1's before it looks: the words from GFT_STATUS_STICKY on are therefore NEVER cleared by the library -- the owner zeroes
 * the block once -- and only ever grow: status[GFT_STATUS_OVERFLOWS] counts the frames that posted R > their
 * binning_instances, status[GFT_STATUS_MAX_R] is the largest such R; status[GFT_STATUS_CAP] is the binning_instances of the
 * frame that posted status[0] (so the pair is self-consistent whichever frame it belongs to).  Same kernels, same results as
 * gft_forward().  cfg->debug is refused (it synchronises). */
#define GFT_STATUS_WORDS 16
#define GFT_STATUS_STICKY 12     /* words [12, 16) survive gft_forward_enqueue's clear */
#define GFT_STATUS_CAP 12
#define GFT_STATUS_OVERFLOWS 13
#define GFT_STATUS_MAX_R 14
int gft_forward_enqueue(void* hip_stream, const gft_config* cfg, const gft_forward_io* io,
                        const gft_forward_hints* hints, uint32_t* status /*device-accessible, may be NULL*/);

/* The backward of the forward whose scratch buffers `io` carries.  Gradient sums are added with float atomics
 * (as in the reference), so two runs agree to rounding, not bit for bit.  GFT_BWD_SPLIT=0 in the environment
 * keeps one wave per pixel quadrant (default: deep quadrants are walked by up to eight waves).
 * With io->det_partials the sums are formed in a fixed order instead (bit-reproducible, slower: a test mode). */
int gft_backward(void* hip_stream, const gft_config* cfg,
                 const gft_backward_io* io, int64_t binning_instances);

/* Zeroes the rows of the gradient tensors in `io` (dL_d* pointers; P, M, M_p from cfg) that io->dirty_rows marks and
 * clears the marks: what a caller that reuses its gradient tensors runs before the next backward (see dirty_rows). */
int gft_grads_rezero(void* hip_stream, const gft_config* cfg, const gft_backward_io* io);

int gft_mark_visible(void* hip_stream, int32_t P, const float* means3D,
                     const float* viewmatrix, const float* projmatrix,
                     float near_n, float far_n, uint8_t* present);

/* HIP-event timing of every stage on the stream it is launched on (bench.py
 * roofline leg).  Enabling it adds two event records per stage. */
int gft_profile_enable(int on);
int gft_profile_reset(void);
int gft_profile_read(gft_profile* out /*host*/);   /* synchronises pending events */

#ifdef __cplusplus
}
#endif
#endif
