/*
 * esahrnet.h — C-ABI of libesahrnet.so: the MI355X (gfx950) HRNet keypoint-heatmap path.
 *
 * The reference (bonjour-l/esa-pose-estimation) has no FFI seam for this path: its seam is
 * the torch.nn.Module protocol.  This header is therefore what a binding for
 * the path would bind, entry point by entry point:
 *
 *   esahrnet_create / _conv_count / _conv_desc   <- models/seg_hrnet.py:260-340, 343-423
 *                                                   (HighResolutionNet.__init__ and the _make_*
 *                                                   builders: cfg -> list of Conv2d/BatchNorm2d)
 *   esahrnet_set_conv / _commit                  <- models/seg_hrnet.py:475-493 init_weights /
 *                                                   val.py:64-66 load_model -> load_state_dict
 *   esahrnet_forward                             <- models/seg_hrnet.py:425-473
 *                                                   HighResolutionNet.forward (val.py:146 call)
 *   esahrnet_keypoints                           <- demo.py:172-185 / val.py:151-164 two-stage
 *                                                   torch.max + inference.py:136-152 get_final
 *                                                   (inference.py:75-94 my_taylor)
 *
 * Conventions: plain C types only; every function returns 0 on success, non-zero on error
 * with a message in esahrnet_last_error() (thread-local); nothing throws across the ABI.
 * All device buffers are CALLER-OWNED (in the Python host: torch tensors, so the caching
 * allocator and stream semantics stay intact).  The handle owns only the packed weights.
 * Kernels are enqueued on the stream passed in and never synchronise it.  One handle per
 * device; calls on one handle are not re-entrant.
 */
#ifndef ESAHRNET_H
#define ESAHRNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESAHRNET_MAX_BRANCHES 4
#define ESAHRNET_ABI_VERSION 5      /* 5: esahrnet_cfg.precision 2 (fp32-grade bf16x6) */

typedef struct esahrnet_ctx* esahrnet_handle;
typedef void* esahrnet_stream; /* hipStream_t */

/* Stage table of config/default.py:39-74 as plain ints. */
typedef struct esahrnet_cfg {
    int32_t cin;                 /* 1 (seg_hrnet2.py:265) or 3 (seg_hrnet.py:265)            */
    int32_t num_keypoints;       /* 11 (seg_hrnet2.py:324) or 32 (seg_hrnet.py:324)           */
    int32_t stem_width;          /* 64 (seg_hrnet.py:265-270)                                  */
    int32_t widths[ESAHRNET_MAX_BRANCHES];          /* NUM_CHANNELS of STAGE4: 32,64,128,256   */
    int32_t blocks[4][ESAHRNET_MAX_BRANCHES];       /* NUM_BLOCKS per stage (stage1 uses [0][0]) */
    int32_t modules[4];          /* NUM_MODULES per stage (stage1 entry unused)                */
    int32_t final_conv_kernel;   /* FINAL_CONV_KERNEL, must be 1 (config/default.py:43)        */
    int32_t variant;             /* 0: seg_hrnet.py / seg_hrnet2.py;  1: seg_hrnet3.py (CBAM in every
                                    BasicBlock and on the 64-ch pre-BN stem skip, 3x3 last_layer[0],
                                    output_layer over [heatmaps, skip]; models/seg_hrnet3.py)         */
    int32_t precision;           /* arithmetic of the convolutions (not a reference knob: BASELINE.json configs):
                                    2: "bf16x6" — fp32-grade, THE MODE OF configs[1] / configs[2] (the reference computes in
                                       fp32, models/seg_hrnet.py:425-473): f32 NHWC activations, every operand split exactly
                                       into three bf16 terms, 6 MFMAs per product, f32 accumulate; error vs fp64 at or below
                                       that of an f32 FMA chain; both variants;
                                    0: "bf16x3" — split-bf16 hi/lo operands (~16 significand bits), 3 MFMAs per product, f32
                                       accumulate: heatmap L_inf ~1e-5 of the heat-map scale; opt-in fast mode, NOT fp32;
                                    1: bf16 activations and weights stored ONCE (half the bytes, one MFMA per
                                       product), f32 accumulate, f32 folded-BN bias epilogue — configs[3]
                                       (heatmap L_inf ~1e-2); variant 0 only                                   */
} esahrnet_cfg;

/* A parameter tensor that is not a convolution of the main graph (variant 1: the CBAM weights). */
typedef struct esahrnet_aux_desc {
    char name[96];               /* state_dict key, e.g. "layer1.0.ca.fc.0.weight"               */
    int32_t shape[4];            /* Conv2d weight shape [out][in][kh][kw]                          */
} esahrnet_aux_desc;

/* One Conv2d of the reference module tree, by its state_dict prefix. */
typedef struct esahrnet_conv_desc {
    char name[96];               /* e.g. "stage3.0.fuse_layers.2.0.1.0" (conv: name + ".weight") */
    char bn[96];                 /* partner BatchNorm2d prefix, "" if none (output_layer.0)    */
    int32_t cin, cout, k, stride;
    int32_t has_bias;            /* conv carries its own bias (last_layer.0/3, output_layer.0) */
    int32_t relu;                /* a ReLU follows conv(+BN) directly                           */
} esahrnet_conv_desc;

const char* esahrnet_last_error(void);
int esahrnet_abi_version(void);

/* Build the static plan for a stage table.  `device` is the HIP device ordinal the weights
 * will live on; nothing touches the device until esahrnet_commit. */
int esahrnet_create(const esahrnet_cfg* cfg, int device, esahrnet_handle* out);
int esahrnet_destroy(esahrnet_handle h);
/* Device ordinal the handle was created for. */
int esahrnet_handle_device(esahrnet_handle h);

int esahrnet_conv_count(esahrnet_handle h);
int esahrnet_conv_desc_get(esahrnet_handle h, int index, esahrnet_conv_desc* out);

/* Hand over one convolution with BatchNorm ALREADY FOLDED (eval mode, eps 1e-5):
 * w: host f32 [cout][cin][k][k], b: host f32 [cout] (never NULL). */
int esahrnet_set_conv(esahrnet_handle h, int index, const float* w, const float* b);
int esahrnet_aux_count(esahrnet_handle h);
int esahrnet_aux_desc_get(esahrnet_handle h, int index, esahrnet_aux_desc* out);
/* w: host f32, prod(shape) elements, exactly the state_dict tensor (no folding applies). */
int esahrnet_set_aux(esahrnet_handle h, int index, const float* w);
/* Pack (split-bf16, MFMA fragment order) and upload every convolution.  Synchronous. */
int esahrnet_commit(esahrnet_handle h);

/* Bytes of caller-owned device scratch esahrnet_forward needs for a batch of n crops of h x w. */
int esahrnet_workspace_bytes(esahrnet_handle h, int n, int height, int width, size_t* bytes);

/* x_dev: f32 [n][cin][height][width] NCHW contiguous (already normalised, data_load_val.py:86).
 * heat_dev: f32 [n][K][height][width] NCHW, fully overwritten.  x_dev is not modified. */
int esahrnet_forward(esahrnet_handle h, const void* x_dev, int n, int height, int width,
                     void* heat_dev, void* ws_dev, size_t ws_bytes, esahrnet_stream stream);

/* heat_dev: f32 [n][k][height][width] -> kp_dev: f32 [n][k][3] = (x, y, peak):
 * first-occurrence arg-max, log-quadratic sub-pixel refine, raw peak value. */
int esahrnet_keypoints(const void* heat_dev, int n, int k, int height, int width,
                       void* kp_dev, esahrnet_stream stream);
/* Same, and additionally idx_dev: int32 [n][k] = row * width + column of the arg-max (the integer coordinates
 * inference.py:22-51 get_max_preds returns; NULL: not written).  A NaN in a plane is the maximum, as for
 * np.argmax / torch.max: first NaN's index, peak NaN, no refinement. */
int esahrnet_keypoints_ex(const void* heat_dev, int n, int k, int height, int width,
                          void* kp_dev, void* idx_dev, esahrnet_stream stream);

/* ---- forward + keypoints without re-reading the heat-maps -------------------------------------------------
 * The output-layer kernel can leave, beside the heat-maps, the first row-major maximum of each of its tiles:
 * part_dev = 8 bytes x [n * K][ntiles] (f32 value, int32 index row * width + column).  esahrnet_keypoints_finish reduces
 * those instead of sweeping n*K*height*width floats; its results are bit-identical to esahrnet_keypoints_ex on the same
 * heat-maps (same ordering of ties and NaNs, same refinement).  Replaces: `net(x)` followed by get_final / the host peak
 * search (val.py:151-166, inference.py:136-152).
 * esahrnet_partial_tiles: ntiles for a crop size; 0 = this handle cannot (seg_hrnet3, a VALU output layer):
 * use esahrnet_forward + esahrnet_keypoints.  part_dev == NULL makes esahrnet_forward_partials plain esahrnet_forward. */
int esahrnet_partial_tiles(esahrnet_handle h, int height, int width, int* ntiles);
int esahrnet_forward_partials(esahrnet_handle h, const void* x_dev, int n, int height, int width, void* heat_dev,
                              void* part_dev, void* ws_dev, size_t ws_bytes, esahrnet_stream stream);
int esahrnet_keypoints_finish(const void* heat_dev, const void* part_dev, int ntiles, int n, int k, int height, int width,
                              void* kp_dev, void* idx_dev, esahrnet_stream stream);

/* Loader stage in front of the path (data_load_val.py:139-187): for each of n 8-bit frames
 * [frame_h][frame_w] take the clamped box boxes[i] = (x0, y0, x1, y1) (int32, device), edge-pad it the
 * way the reference does, resize to scale x scale (OpenCV 8-bit INTER_LINEAR arithmetic) and write
 * (v/255 - mean)/std into out_dev f32 [n][1][scale][scale] — the tensor esahrnet_forward takes. */
int esahrnet_crops(const void* frames_dev, int n, int frame_h, int frame_w, const void* boxes_dev, int scale,
                   float mean, float stdv, void* out_dev, esahrnet_stream stream);

/* Host pose solve behind the path, for a batch (pnp.py:46-90 + cpnp.cpnp_m of val.py:194-209 + val.py:172-180,
 * 221-224): kp = host f32 [n][k][3] keypoint rows (x, y, peak) in crop coordinates as esahrnet_keypoints wrote
 * them; kp3d = f64 [k][3] model points; K9 = f64 row-major camera matrix; boxes_xy = int32 [n][2] crop origins;
 * rates = f64 [n] crop scale factors.  Per image: keypoints with peak > thresh (at least min_k, largest first),
 * mapped back to image pixels, EPnP + RANSAC (5 px, 100 iterations, 0.99), peak-weighted LM refinement,
 * -> q_out f64 [n][4] = [w, x, y, z], t_out f64 [n][3] (NaN when fewer than 4 keypoints or no solution).
 * Pure host code, `threads` worker threads; needs no GPU.  Returns 0, or 1 on a bad argument. */
int esahrnet_pnp_batch(const float* kp, int n, int k, const double* kp3d, const double* K9, const int* boxes_xy,
                       const double* rates, double thresh, int min_k, int threads, double* q_out, double* t_out);

/* ---- introspection / per-operator entry points (used by the parity tests) ------------- */

/* Algorithmic (direct-convolution) FLOPs of one forward of one crop: 2 * MACs of every conv. */
int esahrnet_flops_per_crop(esahrnet_handle h, int height, int width, double* flops);
/* Number of kernel launches one forward enqueues. */
int esahrnet_launch_count(esahrnet_handle h);
/* One kernel launch of the forward plan, for measurement (bench.py roofline leg). */
typedef struct esahrnet_op_desc {
    char kernel[64];             /* kernel template instance, e.g. "conv_mfma<3,1,16,2>"        */
    char label[96];              /* reference layer it implements, e.g. "stage4.0.branches.3.1.conv2" */
    double flops;                /* algorithmic FLOPs (2*MAC of the direct convolution) for n crops */
    double bytes;                /* compulsory HBM bytes of this launch: inputs + outputs + weights */
} esahrnet_op_desc;
int esahrnet_op_desc_get(esahrnet_handle h, int index, int n, int height, int width,
                         esahrnet_op_desc* out);
/* Same as esahrnet_forward but brackets every launch with hipEvents ON `stream` and returns the
 * per-launch durations in milliseconds (ms_out[esahrnet_launch_count]), with the duration of an
 * empty event bracket on the same stream subtracted.  Synchronises `stream`. */
int esahrnet_forward_timed(esahrnet_handle h, const void* x_dev, int n, int height, int width,
                           void* heat_dev, void* ws_dev, size_t ws_bytes, esahrnet_stream stream,
                           float* ms_out);
/* Names of the intermediate tensors that can be dumped ("stem2", "layer1", "stage3.1", ...). */
int esahrnet_tap_count(esahrnet_handle h);
int esahrnet_tap_name(esahrnet_handle h, int index, char* out, size_t cap);
/* keep != 0: every intermediate tensor gets its own workspace region (no recycling) so that
 * esahrnet_tap_read can be used after a forward; changes esahrnet_workspace_bytes. */
int esahrnet_set_debug_keep(esahrnet_handle h, int keep);
/* After a forward on (n,height,width) with the same workspace: convert intermediate tensor
 * `name` from the internal split-bf16 NHWC layout to f32 NCHW [n][c][th][tw] in out_dev. */
int esahrnet_tap_shape(esahrnet_handle h, const char* name, int height, int width,
                       int* c, int* th, int* tw);
int esahrnet_tap_read(esahrnet_handle h, const char* name, int n, int height, int width,
                      const void* ws_dev, void* out_dev, esahrnet_stream stream);

/* Stand-alone convolution on f32 NCHW device tensors through the same MFMA kernels:
 * y = [relu]( conv_{k,stride,pad=(k-1)/2}(x; w) + b [+ res] ).  w,b are HOST pointers
 * (packed and uploaded on the spot; synchronous; test use only). */
int esahrnet_op_conv(const void* x_dev, int n, int cin, int height, int width,
                     const float* w, const float* b, int cout, int k, int stride, int relu,
                     const void* res_dev, void* y_dev, esahrnet_stream stream);
/* y = [relu]( sum_i up_bilinear_align_corners_false(x_i -> (height,width)) ), f32 NCHW. */
int esahrnet_op_fuse(const void* const* xs_dev, const int* hs, const int* ws, int nterms,
                     int n, int c, int height, int width, int relu, void* y_dev,
                     esahrnet_stream stream);
/* The same two operators in the arithmetic of esahrnet_cfg.precision (0: split-bf16, 1: single bf16, 2: bf16x6): inputs are
 * converted to the internal format, the kernel of that mode runs, the result is converted back to f32. */
int esahrnet_op_conv_ex(const void* x_dev, int n, int cin, int height, int width,
                        const float* w, const float* b, int cout, int k, int stride, int relu,
                        const void* res_dev, void* y_dev, int precision, esahrnet_stream stream);
int esahrnet_op_fuse_ex(const void* const* xs_dev, const int* hs, const int* ws, int nterms,
                        int n, int c, int height, int width, int relu, void* y_dev, int precision,
                        esahrnet_stream stream);

/* ---- test hooks ------------------------------------------------------------------------------------ */
/* Launch state is kept per DEVICE (dynamic-LDS limits raised per kernel and device, CU counts), never in
 * process-wide flags: number of (kernel, device) entries and of devices seen so far. */
int esahrnet_debug_devstate(int* kernel_device_entries, int* devices);
/* Bytes one launch of the stream convolution kernels may address (default and maximum 2^31 - 1); batches beyond it
 * are cut into image ranges on the host.  Lowered by the tests to exercise the cut at small sizes; 0 restores. */
int esahrnet_debug_set_launch_limit(long long bytes);
/* Where launch `index` (0 .. esahrnet_launch_count-1) sits in the wave schedule: launches of one wave on different
 * lanes run concurrently (lane 0 = the caller's stream), waves one after another.  All zero on a one-lane handle. */
int esahrnet_debug_op_schedule(esahrnet_handle h, int index, int* wave, int* lane);

#ifdef __cplusplus
}
#endif
#endif /* ESAHRNET_H */
