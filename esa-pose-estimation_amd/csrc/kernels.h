// kernels.h — host-side launchers of the HIP kernels (internal; the public ABI is include/esahrnet.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace esa {

// tensor formats of the internal activations (sb.h): split-bf16 NHWC, single bf16 NHWC, plain f32 NHWC
enum { FMT_SB = 0, FMT_BF = 1, FMT_F32 = 2 };

// thread-local error text behind esahrnet_last_error() (plan.hip); returns 1 so that `return set_error(...)` reads well
int set_error(const char* fmt, ...);

// ---- implicit-GEMM convolution on MFMA (conv_mfma.hip) -------------------------------------
struct ConvParams {
    const char* x;      // SB input  [N][H][W][Cinp]
    char* y;            // SB output [N][OH][OW][Coutp]
    const char* res;    // SB residual (same shape as y) or nullptr
    const uint4* w;     // packed split-bf16 weights, fragment order (see pack_conv_weights)
    const float* bias;  // f32 [Coutp]
    int N, H, W, OH, OW;
    int Cinp, Coutp;    // multiples of 32
    int relu;
    int out_f32;        // 0: y is SB; 1: y is plain f32 NHWC [N][OH][OW][Coutp] (head terms t_b)
    int fmt;            // FMT_SB; FMT_BF: x, y, res are BF tensors (single bf16, sb.h), Cinp / Coutp multiples of 64, weights
                        //    packed by pack_conv_weights_bf, served by the stream kernel (3x3) and conv1x1 only; FMT_F32: plain
                        //    f32 NHWC tensors, weights packed by pack_conv_weights_x6, served by conv_x6.hip (bf16x6 arithmetic)
    // multi-head form (stream kernel, stride 2 only): several convolutions of the SAME input evaluated in one launch —
    // w / bias are the members' packed weights / biases concatenated along cout, Coutp their total, and head h
    // (couts hb[h] .. hb[h+1]-1 of the concatenation, multiples of 32) goes to its own SB tensor yh[h] of
    // hb[h+1]-hb[h] channels with its own ReLU flag.  nheads <= 1: the plain form above (y, relu).
    int nheads;
    char* yh[3];
    int hb[4];
    int hrelu[3];
};
// multi-head stride-2 3x3 (see ConvParams); every head must satisfy conv_s2c32_supported on its own
bool conv_s2c32_multi_supported(const ConvParams& p);
int launch_conv_s2c32_multi(const ConvParams& p, hipStream_t stream);
// k in {1,3}, stride in {1,2}, pad = (k-1)/2.  Returns hipError_t as int.
int launch_conv(const ConvParams& p, int k, int stride, hipStream_t stream);
// name of the kernel launch_conv() runs for these parameters (only shapes / flags / res != nullptr are looked at)
const char* conv_kernel_name(const ConvParams& p, int k, int stride);
// 3x3 stride-2, Cinp == 32, no residual: weights in registers, one cout tile per wave (conv_s2c32.hip)
bool conv_s2c32_supported(const ConvParams& p);
int launch_conv_s2c32(const ConvParams& p, hipStream_t stream);
int launch_conv_s1w(const ConvParams& p, hipStream_t stream);
bool conv_is_stream_s1(const ConvParams& p);       // launch_conv(p, 3, 1) runs the stride-1 stream kernel
// 2..4 independent 3x3 convolutions of one stride (each one a stream-kernel launch on its own) as ONE launch
bool conv_jobs_supported(const ConvParams* ps, int n, int stride);
int launch_conv_jobs(const ConvParams* ps, int n, int stride, hipStream_t stream);
bool use_th16(const ConvParams& p);               // stride-1 stream kernel: 16-row tile, one workgroup per CU (conv_s2c32.hip)
void set_stream_launch_limit(long long bytes);     // test hook, see conv_s2c32.hip images_per_launch
// 1x1, Cinp in {64..384}: all input channels of 16 pixels in registers, weights streamed through LDS (conv1x1.hip)
bool conv1x1_supported(const ConvParams& p);
int launch_conv1x1(const ConvParams& p, hipStream_t stream);
// 2..6 independent 1x1 convolutions (split format, 64 / 128 / 256 input channels) as ONE launch
bool conv1x1_jobs_supported(const ConvParams* ps, int n);
int launch_conv1x1_jobs(const ConvParams* ps, int n, hipStream_t stream);
// bytes of the packed weight image for a conv with padded channel counts
size_t packed_weight_bytes(int coutp, int cinp, int k);
// host-side packing: w f32 [cout][cin][k][k] -> dst (packed_weight_bytes), zero padded
void pack_conv_weights(const float* w, int cout, int cin, int k, int coutp, int cinp, void* dst);
// BF mode: [cout16 tile][cin64 block][tap][K-step 0|1][lane][8 x bf16], lane l holding
// W[tile*16 + (l&15)][block*64 + step*32 + 8*(l>>4) + j] rounded to bf16 — same fragment count and order as the
// split format has for twice the channels per block
size_t packed_weight_bytes_bf(int coutp, int cinp, int k);
void pack_conv_weights_bf(const float* w, int cout, int cin, int k, int coutp, int cinp, void* dst);

// bf16x6 mode (conv_x6.hip): [cout16 tile][cin32 chunk][tap][term 0..2][lane][8 x bf16], every weight split exactly into
// three bf16 terms, K order inside a chunk as the staging threads load it (x6_chan_of_k)
size_t packed_weight_bytes_x6(int coutp, int cinp, int k);
void pack_conv_weights_x6(const float* w, int cout, int cin, int k, int coutp, int cinp, void* dst);
// 3x3 stride 1 / 2 and 1x1 on FMT_F32 tensors in bf16x6 arithmetic
bool conv_x6_supported(const ConvParams& p, int k, int stride);
int launch_conv_x6(const ConvParams& p, int k, int stride, hipStream_t stream);
const char* conv_x6_kernel_name(const ConvParams& p, int k, int stride);
// 2..6 independent convolutions of one kernel size and stride on FMT_F32 tensors as ONE launch (conv_x6_jobs_kernel)
bool conv_x6_jobs_supported(const ConvParams* ps, int n, int k, int stride);
int launch_conv_x6_jobs(const ConvParams* ps, int n, int k, int stride, hipStream_t stream);
// 1x1 with all input channels of a wave's pixels in registers (<= 256 input channels, no residual); launch_conv_x6 routes here
bool conv1x1_x6_supported(const ConvParams& p);
int launch_conv1x1_x6(const ConvParams& p, hipStream_t stream);
int launch_conv1x1_x6_jobs(const ConvParams* ps, int n, hipStream_t stream);
// does launch_conv_x6_jobs(k = 1) run the register-resident 1x1 kernel (conv1x1_x6_jobs_kernel) for these members?
bool conv1x1_x6_jobs_supported(const ConvParams* ps, int n);

// ---- stem conv1: f32 NCHW -> SB, 3x3 s1, cin in {1..4}, cout = multiple of 32 (stem.hip) ---
struct StemParams {
    const float* x;     // f32 [N][cin][H][W]
    char* y;            // SB [N][H][W][cout]
    const float* w;     // f32 [cout/8][cin][9][8]  (repacked: 8 consecutive couts innermost)
    const float* bias;  // f32 [cout]
    int N, H, W, cin, cout;
    int relu;           // 0: raw conv output (seg_hrnet3 keeps the pre-BN conv1 tensor for its skip)
    int fmt;            // format of y (FMT_BF: cout = padded channel count, multiple of 64)
};
int launch_stem(const StemParams& p, hipStream_t stream);
// the same convolution + the per-channel (sum, max) of the output over slabs of pixels, in pool_partial's layout
// [N][slabs][cout][2] (cbam.hip); stem_pool_slabs: slabs per image, 0 = not available (needs cin 1, cout 64)
int stem_pool_slabs(int cin, int cout, int H, int W);
int launch_stem_pool(const StemParams& p, float* pool, hipStream_t stream);

// ---- fused BasicBlock of the 32-channel branch (bblock32.hip) -------------------------------------
struct BlockParams {
    const char* x;      // SB [N][H][W][32]
    char* y;            // SB [N][H][W][32]
    const uint4* w1;    // packed 3x3 32->32 weights of conv1 / conv2 (pack_conv_weights)
    const uint4* w2;
    const float* bias1; // f32 [32]
    const float* bias2; // f32 [32]
    int N, H, W;
};
int launch_bblock32(const BlockParams& p, hipStream_t stream);

// ---- fused stem: conv1 (VALU, recomputed per tile) -> conv2 3x3 s2 (MFMA) (stem_fused.hip) -------
struct StemFusedParams {
    const float* x;     // f32 [N][cin][H][W]
    char* y;            // SB [N][OH][OW][Coutp]
    const float* w1;    // f32 [Cmid/8][cin][9][8]   (same image as StemParams::w)
    const float* bias1; // f32 [Cmid]
    const uint4* w2;    // packed conv2 weights (pack_conv_weights, k = 3)
    const float* bias2; // f32 [Coutp]
    int N, H, W, OH, OW, cin, Cmid, Coutp;
};
int launch_stem_fused(const StemFusedParams& p, hipStream_t stream);
// fp32-grade mode (conv_x6.hip): x f32 NCHW, y f32 NHWC, w1 = pack_stem_w1_x6 image (conv1 folded weights + bias, 640 floats),
// w2 = pack_conv_weights_x6 of conv2; cin == 1 only
bool stem_fused_x6_supported(int cin, int cmid, int coutp);
void pack_stem_w1_x6(const float* w, const float* b, float* dst);
int launch_stem_fused_x6(const StemFusedParams& p, hipStream_t stream);

// ---- cross-resolution fuse: y = relu?(sum_i up(x_i)) on SB tensors (fuse.hip) ---------------
struct FuseParams {
    const char* x[4];
    int h[4], w[4];     // source resolution of each term (== H,W for same-resolution terms)
    int nterms;
    char* y;
    int N, H, W, Cp;
    int relu;
    int fmt;            // format of all tensors
};
int launch_fuse(const FuseParams& p, hipStream_t stream);

// ---- head tail: up x2 (align_corners=True) + concat raw input + 3x3 conv -> f32 NCHW (head.hip)
struct FinalParams {
    const char* h3;     // SB [N][H/2][W/2][Cp]  (last_layer.3 output, K valid channels)
    const float* x0;    // f32 [N][cin][H][W]
    float* out;         // f32 [N][K][H][W]
    const float* w;     // f32 [K+cin][9][K]   (repacked: output channel innermost)
    const float* bias;  // f32 [K] (padded with zeros to 32)
    const uint4* wpk;   // MFMA path: split-bf16 fragments (pack_final_mfma), or nullptr -> VALU kernel
    int N, H, W, h, wd; // h,wd = resolution of h3
    int K, cin, Cp;
    int fmt;            // format of h3 (FMT_F32: VALU kernel only)
    float2* part;       // optional (MFMA kernel only): [N*K][final_part_tiles] (value, index bits) of every tile's first maximum
};
int launch_final(const FinalParams& p, hipStream_t stream);
// tiles per heat-map the MFMA output-layer kernel reports partial maxima for; 0: that kernel does not serve (K, cin)
int final_part_tiles(int K, int cin, int H, int W);
// (v, i) beats (bv, bi) if it is larger — NaN counting as larger than every number — or equal with a lower index
// (np.argmax / torch.max: first row-major maximum, first NaN)
__device__ __forceinline__ void argmax_take(float v, int i, float& bv, int& bi) {
    const bool vn = v != v, bn = bv != bv;
    if (v > bv || (vn && !bn) || ((v == bv || (vn && bn)) && i < bi)) { bv = v; bi = i; }
}
bool final_mfma_supported(int K, int cin);
size_t final_mfma_bytes(int K, int cin);
void pack_final_mfma(const float* w, int K, int cin, void* dst);

// ---- fused head: W0*x0 + sum up(t_b) + bias, ReLU, W3*(.) + bias, ReLU (head_fused.hip) ----------
struct HeadParams {
    const char* x0;     // SB [N][H][W][C0p]            stage-4 branch 0
    const char* t[3];   // f32 NHWC [N][th][tw][Ctp]    W_b * x_b on branch b's grid, b = 1..3
    char* y;            // SB [N][H][W][C3p]            last_layer[3..5] output
    const uint4* w0;    // packed [Ctp/16][C0p/32][hi|lo][64]   (pack_conv_weights, k = 1)
    const uint4* w3;    // packed [M3][Ctp/32][hi|lo][64]       (pack_head_w3, permuted K order)
    const float* bias0; // f32 [Ctp]
    const float* bias3; // f32 [C3p]
    int N, H, W;
    int th[3], tw[3];
    int C0p, Ctp, C3p, K;
};
int launch_head(const HeadParams& p, hipStream_t stream);
bool head_fused_supported(int H, int W, const int th[3], const int tw[3], int C0p, int K);
size_t head_w3_bytes(int K, int Ctp);
void pack_head_w3(const float* w, int K, int Ct, int Ctp, void* dst);
// bf16 mode (head_fused_bf.hip): x0, t[b] and y are BF tensors; w0 = pack_conv_weights_bf(k = 1) of the branch-0 slice
// [Ctp/16][C0p/64][K-step][64], w3 = pack_head_w3_bf [M3][Ctp/32][64] (one bf16 fragment per chunk)
int launch_head_bf(const HeadParams& p, hipStream_t stream);
bool head_fused_bf_supported(int H, int W, const int th[3], const int tw[3], int C0p, int K);
size_t head_w3_bf_bytes(int K, int Ctp);
void pack_head_w3_bf(const float* w, int K, int Ct, int Ctp, void* dst);

// fp32-grade mode (head_x6.hip): x0, t[b] and y are f32 NHWC tensors; w0 / w3 = pack_conv_weights_x6(k = 1) of the branch-0
// slice of last_layer[0] and of last_layer[3] (cout padded to 16 or 32)
int launch_head_x6(const HeadParams& p, hipStream_t stream);
bool head_x6_supported(int H, int W, const int th[3], const int tw[3], int C0p, int K);

// ---- second-generation fused head: bilinear up-sampling on the matrix cores (head_t.hip, head_fused2.hip)
constexpr int HT_PAD = 1;       // T layout: stored column = x + HT_PAD
struct HeadTParams {
    const char* x;      // SB [N][h][w][Cinp]                      branch b activations
    char* t;            // T layout [N][h][Ctp/32][hi|lo][32][XP] bf16   t_b = W_b * x_b
    const uint4* wt;    // packed [Ctp/16][Cinp/32][hi|lo][64]     (pack_conv_weights, k = 1)
    int N, h, w, Cinp, Ctp, XP;
};
int launch_head_t(const HeadTParams& p, hipStream_t stream);
int head_t_xp(int w);                       // row pitch (pixels) of the T layout for a w-pixel row
bool head_t_supported(int Cinp);
struct Head2Params {
    const char* x0;     // SB [N][H][W][C0p]            stage-4 branch 0
    const char* x1;     // SB [N][th0][tw0][C1p]        stage-4 branch 1
    const char* t2;     // T layout, branch 2
    const char* t3;     // T layout, branch 3
    char* y;            // SB [N][H][W][C3p]
    const uint4* w0;    // packed [Ctp/16][C0p/32][hi|lo][64]
    const uint4* w1;    // packed [Ctp/16][C1p/32][hi|lo][64]
    const uint4* w3;    // packed [M3][Ctp/32][hi|lo][64]  (pack_head_w3)
    const float* bias0; // f32 [Ctp]
    const float* bias3; // f32 [C3p]
    int N, H, W;
    int th[3], tw[3];   // grids of branches 1..3
    int xp2, xp3;       // T-layout row pitches
    int C0p, C1p, Ctp, C3p, K;
};
int launch_head2(const Head2Params& p, bool ulo, hipStream_t stream);
bool head_fused2_supported(int H, int W, const int th[3], const int tw[3], int C0p, int C1p, int K, bool* ulo);

// ---- CBAM attention of the seg_hrnet3 variant + slice re-sampling (cbam.hip) --------------------
int launch_pool_partial(const char* x, float* partial, int N, int HW, int Cp, int P, hipStream_t s, int fmt = FMT_SB);
int launch_ca_mlp(const float* partial, const float* w0, const float* w2, float* ca, int N, int HW, int C,
                  int Cp, int Cr, int P, hipStream_t s);
int launch_cbam_maps(const char* x, const float* ca, float* maps, int N, int HW, int C, int Cp, hipStream_t s, int fmt = FMT_SB);
struct CbamApplyParams {
    const char* x;      // SB [N][H][W][Cp]
    const char* res;    // SB same shape or nullptr
    const float* ca;    // f32 [N][Cp]   channel attention
    const float* maps;  // f32 [N][H][W][2]  (mean_c, max_c) of ca*x
    const float* w_sa;  // f32 [2][7][7]
    char* y;            // SB, pixel pitch y_pix_bytes, written at channel offset y_c0 (multiple of 8)
    int N, H, W, Cp, y_pix_bytes, y_c0, relu;
    int C;              // real channels (cbam_spatial only: it forms the maps itself and ignores `maps`)
    int fmt = FMT_SB;   // FMT_SB or FMT_F32 (x, res and y alike)
};
int launch_cbam_apply(const CbamApplyParams& p, hipStream_t s);
// maps + apply in one pass (no `maps` tensor); Cp / 8 must be a power of two <= 32
bool cbam_spatial_supported(int Cp);
int launch_cbam_spatial(const CbamApplyParams& p, hipStream_t s);
// one launch for the same CBAM step of up to four tensors (the branches of an HRModule at one depth)
enum { CBAM_POOL = 0, CBAM_MLP = 1, CBAM_MAPS = 2, CBAM_APPLY = 3, CBAM_SPATIAL = 4 };
constexpr int CBAM_MAXJOBS = 4;
struct CbamJob {
    int kind;
    CbamApplyParams ap;         // every kind: x, N, H, W, Cp, C (+ the rest for APPLY / SPATIAL; ca for MAPS)
    float* partial;             // POOL (out), MLP (in)
    const float* w0; const float* w2; float* ca;        // MLP
    float* maps;                // MAPS (out)
    int HW, P, Cr;
    int tiles_x, tiles_y;       // filled by the launcher
};
struct CbamJobs { CbamJob j[CBAM_MAXJOBS]; int start[CBAM_MAXJOBS + 1]; int n; };
int launch_cbam_jobs(const CbamJob* jobs, int n, hipStream_t s);
struct ResampleParams {
    const char* x;      // SB [N][h][w][Cp_src]
    char* y;            // SB [N][H][W][..], pixel pitch y_pix_bytes, channel offset y_c0 (multiple of 8)
    int N, h, w, H, W, C, Cp_src, y_pix_bytes, y_c0;
    int align;          // 1: align_corners=True, 0: False (irrelevant when h==H && w==W: copy)
    int fmt = FMT_SB;   // FMT_SB or FMT_F32 (x and y alike)
};
int launch_resample_slice(const ResampleParams& p, hipStream_t s);
int launch_zero_slice(char* y, long long npix, int y_pix_bytes, int c0, int nchan, hipStream_t s);

// ---- seg_hrnet3 head: low-resolution branches' share of last_layer[0] (head_gather.hip) ----
struct GatherParams {
    const char* z[2];   // SB [N][h][w][zpix / 4]: the nine 1x1 products per output channel, channel (cout/8)*72 + tap*8 + cout%8
    char* y;            // SB [N][H][W][Cp]
    int N, H, W, C, Cp;                 // C real output channels, Cp padded (the padding is written as zeros)
    int h[2], w[2], zpix[2];
    int R[2], Cc[2], ngroups, nreal, gpw;      // filled by the launcher
};
bool head_gather_supported(int H, int W, const int* h, const int* w, int Cp);
int launch_head_gather(GatherParams p, hipStream_t stream, int fmt = FMT_SB);      // fmt of y: FMT_SB or FMT_F32 (z is plain f32 in both)

// ---- arg-max + log-quadratic refine (keypoints.hip) ----------------------------------------
// idx_out: optional int32 [planes], the flat index (row * W + column) of the arg-max
int launch_keypoints(const float* heat, int planes, int H, int W, float* kp, int* idx_out, hipStream_t stream);
// same result from the per-tile maxima of the output-layer kernel (FinalParams::part, `ntiles` pairs per plane)
int launch_keypoints_finish(const float* heat, const float2* part, int ntiles, int planes, int H, int W, float* kp, int* idx_out,
                            hipStream_t stream);

// ---- crop + edge-pad + 8-bit bilinear resize + normalise: u8 frames -> f32 [N][1][S][S] (crops.hip) ----
int launch_crops(const unsigned char* frames, const int* boxes, float* out, int N, int FH, int FW, int S,
                 float mean, float std_, hipStream_t s);

// ---- layout conversion f32 NCHW <-> SB (layout.hip) ----------------------------------------
int launch_nchw_to_sb(const float* x, int N, int C, int H, int W, char* y, int Cp, hipStream_t s);
int launch_sb_to_nchw(const char* x, int N, int C, int H, int W, int Cp, float* y, hipStream_t s);
// the same for BF tensors (single bf16 NHWC, sb.h)
int launch_nchw_to_bf(const float* x, int N, int C, int H, int W, char* y, int Cp, hipStream_t s);
int launch_bf_to_nchw(const char* x, int N, int C, int H, int W, int Cp, float* y, hipStream_t s);
// and for F32 tensors (plain f32 NHWC)
int launch_nchw_to_f32(const float* x, int N, int C, int H, int W, char* y, int Cp, hipStream_t s);
int launch_f32_to_nchw(const char* x, int N, int C, int H, int W, int Cp, float* y, hipStream_t s);
// by format code
int launch_nchw_to_fmt(int fmt, const float* x, int N, int C, int H, int W, char* y, int Cp, hipStream_t s);
int launch_fmt_to_nchw(int fmt, const char* x, int N, int C, int H, int W, int Cp, float* y, hipStream_t s);

}  // namespace esa
