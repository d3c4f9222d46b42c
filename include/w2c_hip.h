/*
 * w2c_hip.h -- C ABI of libw2c_hip.so: the MI355X (gfx950) kernels under the
 * When2com forward path of GT-RIPL/MultiAgentPerception.
 *
 * The reference has no FFI: its boundary is the Python package path
 * ptsemseg.models (get_model -> nn.Module.forward, models/__init__.py:8-86).
 * This library sits UNDER that boundary.  Each entry point replaces the stock
 * torch.nn call(s) the reference module bottoms out in; the reference
 * file:line each one stands in for is cited per function.  Plain pointers and
 * sizes only -- no torch types.  Every pointer is a DEVICE pointer on the
 * current HIP device unless stated; `stream` is a hipStream_t passed as void*.
 * All functions are re-entrant, keep no global mutable state (scratch such as the split-K workspace is passed in
 * by the caller, one per stream), launch asynchronously on `stream` and return 0 (W2C_OK) or a negative W2C_E_* code
 * (never abort).  w2c_status_string() maps a code to text.
 *
 * Data layout inside the path: activations are bf16 NHWC ("pixel-major":
 * [image][y][x][channel]) with an explicit per-pixel channel stride so that
 * the u_encoder trunk and the query_key_net trunk (same shapes, different
 * weights) live channel-interleaved in ONE tensor and run as a 2-group conv.
 * Images are agent-major: row = agent * B + sample (agent.py:1105-1108).
 * bf16 values are passed as uint16_t.
 */
#ifndef W2C_HIP_H
#define W2C_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define W2C_OK 0
#define W2C_E_ARG (-1)    /* invalid argument / unsupported shape */
#define W2C_E_LAUNCH (-2) /* HIP launch error (hipGetLastError) */

typedef void* w2c_stream_t;

int w2c_version(void);
const char* w2c_status_string(int code);
/* HIP error name + text of the calling thread's last W2C_E_LAUNCH (empty string if none) */
const char* w2c_last_error_string(void);
/* name of the device the calling thread is on, gcnArchName (e.g. "gfx950:...") */
int w2c_device_arch(char* buf, int buflen);

/* ---- a3 + K1: divide_inputs/cat (agent.py:1088-1096,1105-1108) fused with
 * resnet conv1 7x7/2 p3 (no bias) + bn1 + relu (backbone.py:65-66,76-80).
 * x   : f32 [B, 3*N, H, W] (NCHW, agent i in channels 3i..3i+2)
 * w   : bf16 [Cout][7 ky][8 kx][4 ci]  (kx==7 and ci==3 are zero padding)
 * scale/shift : f32 [Cout]  (eval BN folded: y = relu(conv * scale + shift))
 * y   : bf16 NHWC [N*B, H/2, W/2, Cout], agent-major.  Cout in {64, 128}
 *       (128 = both trunks' stems side by side).  H, W multiples of 32. */
int w2c_stem_conv7x7_bn_relu(const float* x, int B, int N, int H, int W,
                             const uint16_t* w, const float* scale, const float* shift, int Cout,
                             uint16_t* y, w2c_stream_t stream);

/* ---- K1 + K1b fused: the same stem followed by maxpool 3x3 s2 p1 (backbone.py:66), the half-resolution
 * conv map never leaves the chip.  y : bf16 NHWC [N*B, H/4, W/4, Cout].  Same argument rules as above. */
int w2c_stem_conv7x7_bn_relu_maxpool(const float* x, int B, int N, int H, int W,
                                     const uint16_t* w, const float* scale, const float* shift, int Cout,
                                     uint16_t* y, w2c_stream_t stream);

/* ---- SURVEY 8f row 4 (input side): the same fused stem fed with the camera frames as the reference's loader reads them
 * -- u8 RGB [B, N, H, W, 3] -- applying the loader's transform (airsim_loader.py:521-527: RGB->BGR,
 * float64 (v - mean)/255, cast to f32) on the fly, bit-identically to feeding the transformed f32 frames.
 * mean_b/g/r : the loader's BGR means (103.939, 116.779, 123.68). */
int w2c_stem_u8_conv7x7_bn_relu_maxpool(const uint8_t* frames, double mean_b, double mean_g, double mean_r,
                                        int B, int N, int H, int W,
                                        const uint16_t* w, const float* scale, const float* shift, int Cout,
                                        uint16_t* y, w2c_stream_t stream);

/* ---- K1b: maxpool 3x3 s2 p1 (backbone.py:66 via resnet.maxpool), bf16 NHWC.
 * x [M, H, W, C] -> y [M, H/2, W/2, C]; C multiple of 8. */
int w2c_maxpool3x3s2(const uint16_t* x, int M, int H, int W, int C, uint16_t* y, w2c_stream_t stream);

/* ---- K2/K3/K4/K8: conv (3x3 p1 or 1x1 p0, stride 1|2) as an MFMA implicit
 * GEMM, with the eval-mode BatchNorm / bias folded into a per-channel f32
 * scale/shift epilogue, optional residual add, optional ReLU.  Stands in for
 * nn.Conv2d+BatchNorm2d+ReLU (models/utils.py:87-120), the BasicBlock convs
 * and 1x1 downsample (backbone.py:66-69 via the third-party resnet18) and the
 * decoder convs (backbone.py:150-154).
 * x        : bf16 NHWC, pixel stride x_cstride elements; group g reads
 *            channels [g*Cin, (g+1)*Cin)
 * w        : bf16 [groups][Cout][ksize*ksize][Cin]
 * scale, shift : f32 [groups*Cout]
 * residual : bf16, same geometry as y (nullable)
 * y        : bf16 (y_is_f32=0) or f32 (y_is_f32=1) NHWC, pixel stride
 *            y_cstride; group g writes channels [g*Cout, (g+1)*Cout)
 * y_group_stride : element offset between the groups' output slabs (0 = Cout: side by side in one NHWC tensor).
 *            Any multiple of 8 is allowed, so group 0 can be written contiguously (y_cstride = Cout) straight into
 *            another buffer -- the agent-parallel path lets the u_encoder squeezer write V into its rank's slot of the
 *            all-gather buffer while the policy encoder's map goes to a private one.  `residual` uses the same geometry.
 * zero_page: >= 256 bytes of zeros (source for padded taps)
 * Cin multiple of 64; Cout multiple of 32.  x and w are addressed through 32-bit buffer descriptors: an x tensor of
 * >= 2 GiB (M*H*W*x_cstride*2 bytes) or >= 2^29 output pixels returns W2C_E_ARG -- split M (results do not depend on
 * M, see the _variant entry point). */
int w2c_conv_igemm_bf16(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                        const uint16_t* w, int Cout, int ksize, int stride, int groups,
                        const float* scale, const float* shift,
                        const uint16_t* residual, int relu,
                        void* y, int y_cstride, int y_is_f32,
                        const void* zero_page, long long y_group_stride, w2c_stream_t stream);

/* Same contract, with the tile/pipeline variant forced (index into the table in csrc/conv_igemm.hip).
 * For tuning (tools/bench_conv.py) and tests.  Every variant walks K as (64-channel chunk, tap) and issues the same
 * v_mfma_f32_32x32x16_bf16 sequence per output element, so results are bit-identical across variants -- and therefore
 * independent of the image count M, which is what w2c_conv_igemm_bf16 keys its variant choice on
 * (tests/test_kernels_gpu.py::test_conv_result_is_independent_of_tile_variant_and_image_count). */
int w2c_conv_igemm_bf16_variant(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                                const uint16_t* w, int Cout, int ksize, int stride, int groups,
                                const float* scale, const float* shift,
                                const uint16_t* residual, int relu,
                                void* y, int y_cstride, int y_is_f32,
                                const void* zero_page, int variant, long long y_group_stride, w2c_stream_t stream);

/* ---- K2w: 3x3 / stride-1 / pad-1 convolution, "weights to registers" form (csrc/conv_wreg.inl) -- the BasicBlock and policy /
 * decoder convs on maps of at least 8 x 16 pixels with Cin >= 256 (backbone.py:66-69 layer3 / layer4, agent.py:126-127,
 * backbone.py:150): same operation and epilogue as w2c_conv_igemm_bf16 (y = relu?(conv * scale + shift (+ residual)), bf16
 * NHWC, groups side by side or y_group_stride apart), different schedule: 128-pixel x 64-channel wave tiles, K split over the
 * waves of a workgroup, weights loaded straight into registers from a fragment-ordered copy.  Results agree with
 * w2c_conv_igemm_bf16 to f32 summation order (the K groups' partial sums are added at the end), not bit for bit; they are
 * independent of M (one workgroup per 8 x 16-pixel tile, fixed order).
 *   w2c_pack_wfrag_bf16      : w [groups][Cout][9*Cin] (the layout w2c_conv_igemm_bf16 reads) -> wfrag, same size, on the device
 *   w2c_conv3x3_wreg_supported: 1 when the shape is supported AND the library prefers this kernel for it (a function of the
 *                               layer geometry only, never of M), else 0
 *   w2c_conv3x3_wreg_bf16    : form 0 = the library's choice; 80 / 81 / 83 = 2x2 / 1x4 / 1x2 waves (channel blocks x K groups);
 *                               93 / 94 = 81 with the weight loads 4 / 2 K-steps ahead instead of 8 (93 is the default form).
 *                               95 = 93 with 32 output channels per wave instead of 64 (twice the workgroups; bit-identical): what form 0
 *                               picks for launches of <= 128 workgroups, where 93's workgroups would leave half of the chip empty.
 * Requires H % 8 == 0, W % 16 == 0, Cin % 64 == 0, Cout % 64 == 0 (form 80: % 128), 16-byte aligned rows. */
int w2c_pack_wfrag_bf16(const uint16_t* w, uint16_t* wfrag, int groups, int Cout, int Cin, w2c_stream_t stream);
int w2c_conv3x3_wreg_supported(int H, int W, int Cin, int Cout);
int w2c_conv3x3_wreg_bf16(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                          const uint16_t* wfrag, int Cout, int groups,
                          const float* scale, const float* shift,
                          const uint16_t* residual, int relu,
                          uint16_t* y, int y_cstride, long long y_group_stride, int form, w2c_stream_t stream);
/* w2c_conv3x3_wreg_f32out: the default form of the entry point above writing an f32 NHWC tensor (y_cstride / y_group_stride in f32
 * elements), no residual -- the decoder's first conv `simple_decoder` backbone.py:150 taken through the fusion agent.py:276-284 by
 * linearity (U = conv0 without bias / ReLU of every agent's value map).  Shapes for which w2c_conv3x3_wreg_supported() is 0: W2C_E_ARG. */
int w2c_conv3x3_wreg_f32out(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                            const uint16_t* wfrag, int Cout, int groups,
                            const float* scale, const float* shift, int relu,
                            float* y, int y_cstride, long long y_group_stride, w2c_stream_t stream);

/* Split-K form of K2 for the tail layers (policy_net4 conv3..5 agent.py:128-132, simple_decoder's last conv
 * backbone.py:152): few output tiles under a long weight-streaming K loop.  `ksplit` workgroups share a tile, each
 * summing a contiguous range of K-steps into `workspace` (f32 partial tiles); a second launch adds them in split
 * order (deterministic) and runs the same epilogue.  ksplit = 0 picks the split from the LAYER geometry only (never
 * from M: a shard of the images rounds exactly like the full batch; 1 = falls through to w2c_conv_igemm_bf16).  workspace: device memory, 16-B aligned, >= w2c_conv_splitk_workspace_bytes(...) bytes, contents
 * irrelevant; one workspace may serve consecutive launches on one stream, not concurrent ones. */
int w2c_conv_igemm_bf16_splitk(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                               const uint16_t* w, int Cout, int ksize, int stride, int groups,
                               const float* scale, const float* shift,
                               const uint16_t* residual, int relu,
                               void* y, int y_cstride, int y_is_f32,
                               const void* zero_page, int ksplit,
                               void* workspace, long long workspace_bytes, long long y_group_stride, w2c_stream_t stream);
/* bytes of workspace the call above needs (0: the layer is not split; -1: invalid shape). */
long long w2c_conv_splitk_workspace_bytes(int M, int H, int W, int Cin, int Cout, int ksize, int stride,
                                          int groups, int ksplit);

/* ---- cfg 5 (BASELINE.json configs[4]: "fp8 encoder convs on CDNA4 MFMA"): the same convolution with fp8 operands.
 * x_is_fp8 = 1: x and w are OCP e4m3fn bytes (x NHWC with x_cstride BYTES-per-pixel channels, w
 * [groups][Cout][ksize*ksize][Cin]); the kernels issue v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales --
 * gfx950's only 2x-rate fp8 MFMA -- over K-steps of 128 channels (the same 128-byte rows as the bf16 form, so the same
 * tiles, swizzles and LDS-DMA schedule, with half the K-steps).  Quantisation scales are the caller's: per-output-
 * channel weight scales and the per-tensor input scale are folded into `scale` (y = act(acc*scale + shift + residual)).
 * x_is_fp8 = 0: bf16 operands (as w2c_conv_igemm_bf16); used where a bf16 tensor must produce an fp8 one.
 * Outputs (either or both): y_bf16 (pixel stride y_cstride, groups y_group_stride apart, 0 = Cout) and y_fp8 =
 * e4m3(result / y8_scale), saturated at +-448, groups side by side, pixel stride y8_cstride.  Cin multiple of 128 (fp8) /
 * 64 (bf16).  variant < 0: library's choice.  Results are independent of M and of the variant (one K order). */
int w2c_conv_igemm_fp8(const void* x, int x_is_fp8, int M, int H, int W, int Cin, int x_cstride,
                       const void* w, int Cout, int ksize, int stride, int groups,
                       const float* scale, const float* shift,
                       const uint16_t* residual, int relu,
                       uint16_t* y_bf16, int y_cstride, long long y_group_stride,
                       uint8_t* y_fp8, int y8_cstride, float y8_scale,
                       const void* zero_page, int variant, w2c_stream_t stream);
/* ---- the front of a stride-2 BasicBlock in ONE launch: t = relu(bn1(conv1 3x3/s2/p1 (x))) and
 * idt = downsample.1(downsample.0 1x1/s2 (x)) (third-party resnet18 BasicBlock, backbone.py:66-69).  The 3x3 conv's
 * centre tap reads exactly the 1x1 conv's pixels (input (2oy, 2ox)), so the staged pixel tile is multiplied with a second
 * weight tile at the centre-tap K-steps: one pass over x and one launch less per block; per output the MFMA sequence is
 * that of the separate w2c_conv_igemm_* calls (bit-identical results).
 * x : bf16 or e4m3 (x_is_fp8) NHWC; w3 [groups][Cout][9][Cin], w1 [groups][Cout][Cin] in x's operand type;
 * t : bf16 (t_bf16, nullable) and/or e4m3(t / t8_scale) (t_fp8, nullable), ReLU applied;  idt : bf16, no ReLU.
 * Groups side by side in every output.  variant < 0: library's choice (0 / 3 / 6 = 128x128 / 128x64 / 64x64 tiles). */
int w2c_conv_s2_block(const void* x, int x_is_fp8, int M, int H, int W, int Cin, int x_cstride,
                      const void* w3, const float* scale3, const float* shift3,
                      const void* w1, const float* scale1, const float* shift1,
                      int Cout, int groups,
                      uint16_t* t_bf16, int t_cstride, uint8_t* t_fp8, int t8_cstride, float t8_scale,
                      uint16_t* idt_bf16, int idt_cstride,
                      const void* zero_page, int variant, w2c_stream_t stream);

/* ---- the same block front on the weights-to-registers structure (csrc/conv_s2wreg.inl; round 4): bf16 only, maps whose OUTPUT
 * tiles into 8 x 16 pixels (H, W even, (H/2) % 8 == 0, (W/2) % 16 == 0), Cin in {64, 128, 192, 256}, Cout % 64 == 0 --
 * w2c_conv_s2_block_wreg_supported() says whether the library offers it for a geometry (a function of the geometry only, never
 * of M or the group count, so a sharded batch takes the same kernel as the whole one).
 * w3frag : conv1's weights in w2c_pack_wfrag_bf16 order; w1frag : the 1x1 downsample's weights [groups][Cout][Cin] in the same
 * fragment order with ONE tap: [groups][Cout/32][Cin/64][k slice 0..3][half][channel % 32][8] (ops.pack_w1frag).
 * t = relu(conv1 * scale3 + shift3): sums K in per-K-group partial sums (equal to w2c_conv_s2_block's t to f32 rounding, not bit
 * for bit); idt is bit-identical to w2c_conv_s2_block's.  form 0 = the library's choice (1..4: A/B forms).  The next-call debug
 * hooks of w2c_conv_s2_block (span) apply. */
int w2c_conv_s2_block_wreg_supported(int H, int W, int Cin, int Cout);
int w2c_conv_s2_block_wreg(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                           const uint16_t* w3frag, const float* scale3, const float* shift3,
                           const uint16_t* w1frag, const float* scale1, const float* shift1,
                           int Cout, int groups, uint16_t* t_bf16, int t_cstride, uint16_t* idt_bf16, int idt_cstride,
                           int form, w2c_stream_t stream);

/* ---- the front of the FIRST stride-2 block (layer2.0: Cin = 64 -> Cout = 128 per group) on a persistent weights-stationary kernel
 * (csrc/conv_s2regh.inl; round 4): every wave keeps its 32 output channels' conv1 weights in registers, the 17 x 17-pixel input patch of
 * an 8 x 8 output tile is staged once per workgroup as four phase images, double-buffered across the tiles of a persistent workgroup.
 * H, W even, (H/2) % 8 == 0, (W/2) % 8 == 0; w2c_conv_s2_front_c64_supported() = offered for this geometry (a function of the
 * geometry and the W2C_S2REGH switch only, never of M or the group count).  w3frag / w1frag as for w2c_conv_s2_block_wreg.
 * Outputs: t = relu(bn1(conv1(x))), idt = bn_d(downsample(x)), bf16, group g at element offset g * *_group_stride (128 = side by side
 * inside one [M][H/2][W/2][*_cstride] tensor; M * H/2 * W/2 * 128 = one compact slab per group).  Same K order as
 * w2c_conv_s2_block: both outputs are bit-identical to it. */
int w2c_conv_s2_front_c64_supported(int H, int W, int Cin, int Cout);
int w2c_conv_s2_front_c64(const uint16_t* x, int M, int H, int W, int x_cstride,
                          const uint16_t* w3frag, const float* scale3, const float* shift3,
                          const uint16_t* w1frag, const float* scale1, const float* shift1, int groups,
                          uint16_t* t_bf16, int t_cstride, long long t_group_stride,
                          uint16_t* idt_bf16, int idt_cstride, long long idt_group_stride, w2c_stream_t stream);

/* ---- debug / A-B switches ("W2C_XCD2D", "W2C_NO_S2PATCH", "W2C_STEM_WGS", "W2C_STEM_FORM", "W2C_STEM_BAND", "W2C_STEM_WAVES",
 * "W2C_WGRAD_PATCH", ... "W2C_S2WREG_FORM", "W2C_S2REGH": csrc/w2c_common.h lists them all).  The library reads the environment variables of the same names ONCE, when it is loaded; no launch path calls
 * getenv().  w2c_set_option changes a switch at run time (returns W2C_E_ARG for an unknown name), w2c_get_option reads it (-1 unknown). */
int w2c_set_option(const char* name, int value);
int w2c_get_option(const char* name);

/* ---- SURVEY 8f rank 3 (training backward, first stage): gradients of the path's 3x3 / 1x1 convolutions
 * (nn.Conv2d under loss.backward(), trainer.py:669-673).
 * Weight gradient: dw[g][co][tap][ci] = sum_p dy[p][g*Cout+co] * x[p @ tap][g*Cin+ci], f32, the layout of the packed
 * weights ([groups][Cout][ksize*ksize][Cin]).  x : the conv's bf16 NHWC input; dy : bf16 NHWC gradient of its output.
 * Cin, Cout multiples of 64.  Deterministic (segment partials in `workspace`, summed in order by a second launch).
 * Input gradient: a stride-1 conv's dx is w2c_conv_igemm_bf16 of dy with the flipped, transposed weights
 * ([Cin][2-ky][2-kx][Cout]); a stride-2 conv's is the same applied to w2c_zero_insert2_bf16(dy) (dy on the even
 * positions of the input grid, zeros elsewhere). */
long long w2c_conv_wgrad_workspace_bytes(int M, int H, int W, int Cin, int Cout, int ksize, int stride, int groups);
int w2c_conv_wgrad_bf16(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                        const uint16_t* dy, int Cout, int dy_cstride, int ksize, int stride, int groups,
                        float* dw, void* workspace, long long workspace_bytes, w2c_stream_t stream);
/* the same with groups = 1 and dw written in nn.Conv2d's parameter layout [Cout][Cin][ky][kx] (what autograd hands the optimiser) */
int w2c_conv_wgrad_bf16_oihw(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                             const uint16_t* dy, int Cout, int dy_cstride, int ksize, int stride,
                             float* dw, void* workspace, long long workspace_bytes, w2c_stream_t stream);
int w2c_zero_insert2_bf16(const uint16_t* dy, int M, int Ho, int Wo, int C, uint16_t* u, int H, int W, w2c_stream_t stream);
/* nn.Conv2d's f32 parameter [Cout][Cin][ky][kx] (Cout, Cin multiples of 32; ksize 1 or 3) -> the packed bf16 operand of
 * w2c_conv_igemm_bf16: mode 0 = forward [Cout][tap][Cin]; mode 1 = input-gradient [Cin][taps-1-tap][Cout]. */
int w2c_pack_conv_weights_bf16(const float* w_oihw, int Cout, int Cin, int ksize, int mode, uint16_t* out, w2c_stream_t stream);
/* both operands from one read of the parameter (the training forward packs the input-gradient operand for its backward) */
int w2c_pack_conv_weights_bf16_both(const float* w_oihw, int Cout, int Cin, int ksize, uint16_t* out_fwd, uint16_t* out_dgrad,
                                    w2c_stream_t stream);

/* ---- SURVEY 8f rank 3, stage 2: train-mode BatchNorm2d (batch statistics over all P = M*H*W pixels of the
 * agent-concatenated batch, agent.py:1108-1111) fused with the residual add and ReLU that follow it in
 * conv2DBatchNormRelu (models/utils.py:118-120) and the third-party BasicBlock.  x, y, residual, dy, dx : dense bf16 NHWC
 * [P][C] (C multiple of 8, <= 2048).  Forward: mean/var over P, running stats updated in place with `momentum` (unbiased
 * var, as nn.BatchNorm2d; pass NULL to skip; the int64 step counter num_batches_tracked is incremented when given), y = act(gamma*(x-mean)*rstd + beta (+ residual)); saves mean, rstd [C].
 * Backward: dyr = dy*[y>0] (y_or_null = the forward's output when relu was applied), dbeta = sum dyr, dgamma = sum dyr*xhat,
 * dx = gamma*rstd*(dyr - dbeta/P - xhat*dgamma/P), d_residual = dyr (dres_or_null).  Deterministic two-level reductions;
 * `workspace` >= w2c_bn_workspace_bytes(P, C); ab / k123: [2][C] / [3][C] f32 scratch. */
long long w2c_bn_workspace_bytes(long long P, int C);
int w2c_bn_train_forward(const uint16_t* x, long long P, int C, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, long long* num_batches_tracked_or_null, float momentum, float eps,
                         const uint16_t* residual, int relu, uint16_t* y,
                         float* mean, float* rstd, float* ab, void* workspace, long long workspace_bytes, w2c_stream_t stream);
int w2c_bn_train_backward(const uint16_t* dy, const uint16_t* y_or_null, const uint16_t* x, long long P, int C,
                          const float* gamma, const float* mean, const float* rstd,
                          uint16_t* dx, uint16_t* dres_or_null, float* dgamma, float* dbeta,
                          float* k123, void* workspace, long long workspace_bytes, w2c_stream_t stream);

/* Two-phase forms of the same BatchNorm for AGENT-SHARDED training (round 4): the statistics run over the agent-concatenated batch
 * (agent.py:1108-1111), so with the agents sharded over ranks the per-channel sums are added over the ranks between the phases.
 *   w2c_bn_train_sums(mode, ...)  -> sums f64 [2][C]: mode 0 (sum x, sum x^2), mode 1 (sum dyr, sum dyr * xhat) of THIS rank's P pixels;
 *   the caller all-reduces (SUM) the sums and the pixel count;
 *   w2c_bn_train_forward_sums / _backward_sums finalize from the GLOBAL sums and P_total (dgamma / dbeta = this rank's own sums:
 *   parameter gradients are all-reduced afterwards like every other gradient) and apply on the local pixels. */
int w2c_bn_train_sums(int mode, const uint16_t* x, const uint16_t* dy, const uint16_t* y_or_null, const float* mean,
                      const float* rstd, long long P, int C, double* sums, void* workspace, long long workspace_bytes,
                      w2c_stream_t stream);
int w2c_bn_train_forward_sums(const uint16_t* x, long long P, int C, const double* sums_global, double P_total,
                              const float* gamma, const float* beta, float* running_mean, float* running_var,
                              long long* num_batches_tracked_or_null, float momentum, float eps, const uint16_t* residual, int relu,
                              uint16_t* y, float* mean, float* rstd, float* ab, w2c_stream_t stream);
int w2c_bn_train_backward_sums(const uint16_t* dy, const uint16_t* y_or_null, const uint16_t* x, long long P, int C,
                               const float* gamma, const float* mean, const float* rstd, const double* sums_local,
                               const double* sums_global, double P_total, uint16_t* dx, uint16_t* dres_or_null,
                               float* dgamma, float* dbeta, float* k123, w2c_stream_t stream);

/* maxpool 3x3 s2 p1 for the training path (backbone.py:66): forward also records, per output element, the tap index 3*ky+kx of
 * the first maximum (idx u8 [M,H/2,W/2,C]); backward gathers dy over the <= 4 windows that selected each input element.
 * x, dx : bf16 NHWC [M,H,W,C]; y, dy : [M,H/2,W/2,C]; H, W even; C multiple of 8. */
int w2c_maxpool3x3s2_train_forward(const uint16_t* x, int M, int H, int W, int C, uint16_t* y, uint8_t* idx, w2c_stream_t stream);
int w2c_maxpool3x3s2_train_backward(const uint16_t* dy, const uint8_t* idx, int M, int H, int W, int C, uint16_t* dx,
                                    w2c_stream_t stream);

/* Unit-test probes of the two fp8 primitives: c[32][32] f32 = a[32][64] . b[32][64]^T (e4m3, one MX-scaled MFMA with
 * unit block scales); y[n] = e4m3(x[n]) as the conv epilogues pack it (round to nearest even, saturating). */
int w2c_debug_mx_mfma(const uint8_t* a, const uint8_t* b, float* c, w2c_stream_t stream);
int w2c_debug_fp8_pack(const float* x, uint8_t* y, int n, w2c_stream_t stream);

/* Debug aid (tools/conv_timeline.py): the calling thread's NEXT w2c_conv_igemm_bf16_variant launch records
 * 4 x uint64 wall-clock stamps per workgroup (start, first tile landed, main loop done, end; 100 MHz) into buf
 * (device memory, >= 32 bytes x workgroups). */
int w2c_debug_conv_timeline(void* buf);
/* Measurement: the next conv entry call of this thread (w2c_conv_igemm_bf16[_splitk|_variant], w2c_conv_igemm_fp8, w2c_conv_s2_block)
   records its launch span into slot[0] = min start stamp, slot[1] = max end stamp over its workgroups (u64, 100 MHz wall clock;
   preset {~0, 0}).  The pointer travels as a kernel argument: a launch captured into a HIP graph records on every replay. */
int w2c_debug_conv_span(void* slot);
/* Debug: enqueue a one-thread kernel that writes the 100 MHz wall clock to *slot (u64): a time stamp in stream order. */
int w2c_debug_stamp(void* slot, w2c_stream_t stream);
/* Debug (round 5): install handlers for SIGSEGV / SIGBUS / SIGABRT / SIGFPE / SIGILL that write the NATIVE call stack of the faulting
   thread to stderr (backtrace_symbols_fd: async-signal-safe) and then chain to whatever handler was installed before (Python's
   faulthandler under pytest), so that a crash inside a runtime library is named by its frame, not only by the Python line above it.
   fd: where to write (< 0: stderr; pytest captures fd 2, so tests/conftest.py passes the descriptor the faulthandler plugin kept).
   tests/conftest.py calls it once; the product path never does.  Returns W2C_OK. */
int w2c_debug_install_crash_backtrace(int fd);

/* ---- K5: Linear (+ReLU) for the key/query heads (agent.py:150-159,167-178).
 * x : [M, K] bf16 (x_is_bf16=1, row stride x_stride elements) or f32
 * w : f32 [O, K] row-major; b : f32 [O]; y : f32 [M, O].  K multiple of 4, M <= 64 per call. */
int w2c_linear_f32(const void* x, int x_is_bf16, int x_stride, int M, int K,
                   const float* w, const float* b, int O, int relu, float* y, w2c_stream_t stream);

/* ---- K5b: the two small layers of a head in one launch: out = W2 relu(W1 h0 + b1) + b2
 * (fc.2, ReLU, fc.4 of km_generator / linear, agent.py:152-155).
 * h0 : f32 [M, >=K1] row stride h0_stride (the ReLU'd fc.0 output); w1t : f32 [K1, H1] (K-major,
 * i.e. fc.2.weight transposed), b1 [H1]; w2t : f32 [H1, O] (fc.4.weight transposed), b2 [O];
 * out : f32 [M, O].  H1 <= 256. */
int w2c_head_tail_f32(const float* h0, int h0_stride, int M, int K1, const float* w1t, const float* b1, int H1,
                      const float* w2t, const float* b2, int O, float* out, w2c_stream_t stream);
/* Both heads' tails in one launch (key_net and query_net read disjoint column ranges [col_off, col_off+K1) of the same
 * fc.0 output h0; agent.py:1126-1129): out_x[M, O_x] = W2_x relu(W1_x h0[:, col_off_x : col_off_x+K1] + b1_x) + b2_x. */
int w2c_head_tail2_f32(const float* h0, int h0_stride, int M, int K1, int H1,
                       int col_off_a, const float* w1t_a, const float* b1_a, const float* w2t_a, const float* b2_a, int O_a, float* out_a,
                       int col_off_b, const float* w1t_b, const float* b1_b, const float* w2t_b, const float* b2_b, int O_b, float* out_b,
                       w2c_stream_t stream);


/* ---- K6: communication graph.  MIMOGeneralDotProductAttention scores +
 * softmax over keys (agent.py:256,268,274), the +0.001*I tie-break
 * (agent.py:1164-1167) and the argmax_select / activated_select coefficient
 * transforms (agent.py:1036-1078); with who=1 the diagonal is masked as in
 * MIMOWhoGeneralDotProductAttention (agent.py:299-343) and no tie-break is added.
 * query : f32 [q_n*B, Dq], row (q-q_lo)*B+b, for the produced query agents only
 *         (NULL => all-ones queries, agent.py:1143,1370)
 * key   : f32 [N*B, Dk] agent-major
 * wq, bq: attention_net.linear weight [Dk, Dq] / bias [Dk]
 * mode  : 0 softmax, 1 argmax_test, 2 activated (thres)
 * q_lo, q_n : only query agents [q_lo, q_lo+q_n) are produced (agent-parallel
 *             ranks own a slice of the queries; keys are always all N)
 * workspace : f32 scratch, N*B*(Dq+1) floats (the projected keys Wq^T key, key.bq)
 * prob  : f32 [B, N, q_n]  returned prob_action columns
 * coef  : f32 [B, N, q_n]  fusion coefficients for the returned prediction
 * action: i64 [B, q_n]
 * nnz_offdiag : i32 [B]    off-diagonal non-zeros of coef per sample (num_connect numerator) */
int w2c_comm_graph(const float* query, const float* key, const float* wq, const float* bq,
                   int B, int N, int Dq, int Dk, int who, int mode, float thres, float tie_bias,
                   int q_lo, int q_n, float* workspace,
                   float* prob, float* coef, int64_t* action, int32_t* nnz_offdiag,
                   w2c_stream_t stream);

/* Same graph from PROJECTED keys: tproj f32 [N*B, Dq+1] agent-major with tproj[r][j<Dq] = (Wq^T key_r)[j] and
 * tproj[r][Dq] = key_r . bq.  The engine folds that projection into the key head's last layer at pack time
 * ((Wq^T W_fc4) h1 + Wq^T b_fc4), so the 1024-d key is never formed and agent-parallel ranks exchange Dq+1 floats
 * per agent-sample instead of Dk. */
int w2c_comm_graph_projected(const float* query, const float* tproj, int B, int N, int Dq, int who, int mode,
                             float thres, float tie_bias, int q_lo, int q_n,
                             float* prob, float* coef, int64_t* action, int32_t* nnz_offdiag, w2c_stream_t stream);

/* ---- K7 + a10: attention-weighted fusion (agent.py:276-284) fused with
 * agents2batch (agent.py:1080-1086).
 * v    : bf16 NHWC rows [N*B][hw][v_cstride], agent-major, C channels used
 * coef : f32 [B, N, q_n]
 * out  : bf16 NHWC rows [q_n*B][hw][out_cstride]: row (q*B+b) channels
 *        [0,C) = sum_k coef[b,k,q] * v[k*B+b]; with append_own=1 channels
 *        [C,2C) = v[(q_lo+q)*B+b] (MIMOcomWho cat, agent.py:1382). */
int w2c_fuse_values(const uint16_t* v, int v_cstride, const float* coef, int B, int N, int q_lo, int q_n,
                    int hw, int C, int append_own, uint16_t* out, int out_cstride, w2c_stream_t stream);

/* K6 + K7 in ONE launch (w2c_comm_graph_projected followed by w2c_fuse_values, same arithmetic and outputs): every workgroup
 * of a sample recomputes that sample's graph in LDS and fuses its share of the pixels; one workgroup per sample writes
 * prob / coef / action / nnz_offdiag.  Arguments as in the two calls it replaces. */
int w2c_comm_graph_fuse(const float* query, const float* tproj, int B, int N, int Dq, int who, int mode,
                        float thres, float tie_bias, int q_lo, int q_n,
                        float* prob, float* coef, int64_t* action, int32_t* nnz_offdiag,
                        const uint16_t* v, int v_cstride, int hw, int C, int append_own,
                        uint16_t* out, int out_cstride, w2c_stream_t stream);


/* ---- K5, round 4: fc.0 of the key / query heads (agent.py:150-151) on the f32 matrix pipe (v_mfma_f32_32x32x2_f32: exact f32).
 * x     : bf16 [M][x_stride] policy map rows (K = 4096 features each)
 * wfrag : f32 fc.0 weights of all heads stacked along O ([O][K]), FRAGMENT-PACKED:
 *         wfrag[((o / 32) * (K / 8) + q) * 256 + (half * 32 + o % 32) * 4 + e] = W[o][8 q + 4 half + e]
 * part  : f32 [ksplit][M][O] split-K partial sums (no bias, no ReLU); K % (256 ksplit) == 0, O % 32 == 0
 * w2c_head_tail2p_f32 = w2c_head_tail2_f32 reading its fc.0 output as h0 = relu(sum_p part[p] + b0) (added in p order). */
int w2c_head_fc0_mfma_f32(const uint16_t* x, int x_stride, int M, int K, const float* wfrag, int O, int ksplit,
                          float* part, w2c_stream_t stream);
int w2c_head_tail2p_f32(const float* part, int n_part, long long part_stride, const float* b0, int h0_stride, int M, int K1, int H1,
                        int col_off_a, const float* w1t_a, const float* b1_a, const float* w2t_a, const float* b2_a, int O_a,
                        float* out_a,
                        int col_off_b, const float* w1t_b, const float* b1_b, const float* w2t_b, const float* b2_b, int O_b,
                        float* out_b, w2c_stream_t stream);

/* K6 + K7 with the decoder's first conv taken through the fusion by LINEARITY (round 4; agent.py:276-284, backbone.py:150-152):
 *   u    : f32 NHWC rows [N*B][hw][u_cstride], agent-major: U[k] = conv0 WITHOUT bias of agent k's value map, channels [0, C);
 *   u_own: NULL, or (MIMOcomWho: decoder input cat(fused, V[q]), agent.py:1382) f32 rows [q_n*B][hw][own_cstride] of the LOCAL query
 *          agents: the conv of V[q] with the second half of conv0's filters, channels [0, C).  A separate operand (round 5) so that an
 *          agent-parallel rank all-gathers U alone: U_own never leaves the rank.  (The one-GPU engine passes a view into the same
 *          tensor as u.)
 *   out  : bf16 NHWC rows [q_n*B][hw][out_cstride]: row (q*B+b) = relu(sum_k coef[b,k,q] * u[k*B+b] (+ u_own[q*B+b]) + bias)
 *          = relu(conv0(fused map)), summed in f32 and rounded once; graph outputs as in w2c_comm_graph_fuse.
 *   pack2: optional [indirect-capable] second copy of the graph outputs, packed: prob f32 [B,N,q_n] at byte 0, action i64 [B,q_n] at
 *          act_off, nnz i32 [B] at nnz_off (act_off % 8 == 0, nnz_off % 4 == 0) -- a captured forward's caller-owned outputs. */
int w2c_comm_graph_fuse_u(const float* query, const float* tproj, int B, int N, int Dq, int who, int mode,
                          float thres, float tie_bias, int q_lo, int q_n,
                          float* prob, float* coef, int64_t* action, int32_t* nnz_offdiag,
                          const float* u, int u_cstride, int hw, int C, const float* u_own, int own_cstride, const float* bias,
                          uint16_t* out, int out_cstride, void* pack2, long long act_off, long long nnz_off, w2c_stream_t stream);

/* ---- Indirect operands (round 4).  The module boundary hands the forward a caller-owned input tensor and returns caller-owned
 * output tensors (SURVEY 8b "Ownership"), so the first and the last kernel of a forward touch addresses that change from call to
 * call -- which kept them outside the captured HIP graph (two eager launches + two graph boundaries per forward).  The pointer
 * arguments marked [indirect-capable] below accept, instead of the address itself, the address of a DEVICE-RESIDENT 8-byte POINTER
 * SLOT with bit 0 set:  arg = (uintptr_t)slot | 1.  The kernel reads the slot when it starts.  w2c_set_slots writes up to 8 slots in
 * stream order -- the values travel as kernel arguments, so the host may run any number of forwards ahead -- and the whole forward,
 * stem to upsample, replays from ONE graph.  [indirect-capable]: `x` / `frames` of w2c_stem_conv7x7_bn_relu_maxpool and
 * w2c_stem_u8_conv7x7_bn_relu_maxpool, `out` of w2c_upsample_bilinear32, `labels` / `gt` / `hist` of w2c_upsample32_argmax[_confusion],
 * `dst` of w2c_copy_to_slot.  The target of a slot must satisfy the alignment the direct form requires. */
int w2c_set_slots(void* slots, int n, const void* const* values, w2c_stream_t stream);
/* dst[0 .. nbytes) = src[0 .. nbytes) (nbytes % 4 == 0; dst [indirect-capable]): the packed prob / action / nnz of a captured
 * forward into the caller-owned copy. */
int w2c_copy_to_slot(const void* src, long long nbytes, void* dst, w2c_stream_t stream);

/* ---- K9: bilinear x32 upsample, align_corners=False (backbone.py:160).
 * low : f32 NHWC [M, h, w, low_cstride] (first n_classes channels used)
 * out : f32 NCHW [M, n_classes, 32h, 32w] */
int w2c_upsample_bilinear32(const float* low, int M, int h, int w, int low_cstride, int n_classes,
                            float* out, w2c_stream_t stream);

/* Adjoint of K9 for the training backward: gout f32 NCHW [M, n_classes, 32h, 32w] ->
 * glow f32 NCHW [M, n_classes, h, w] = d loss / d (the low-resolution logits), same source-index rule; deterministic. */
int w2c_upsample_bilinear32_backward(const float* gout, int M, int h, int w, int n_classes, float* glow, w2c_stream_t stream);

/* ---- the 7x7 / stride-2 stem convolution (3 -> 64, backbone.py:65) in TRAINING.  The train path feeds the
 * frames as bf16 NHWC [M, H, W, 3] (torch channels_last of [M,3,H,W]).
 * Forward : y bf16 NHWC [M, H/2, W/2, 64] = the raw convolution (train-mode BatchNorm follows); w = the packed stem weights
 *           [64][7][8][4] bf16 of w2c_stem_conv7x7_bn_relu.  H % 16 == 0, W % 64 == 0.
 * Backward: the frames need no gradient; dw f32 [64][3][7][7] (the parameter's layout) from x and dy (bf16 NHWC
 *           [M, H/2, W/2, dy_cstride >= 64]); workspace >= w2c_stem_wgrad_workspace_bytes(M, H, W); deterministic. */
int w2c_stem_conv7x7_train_bf16(const uint16_t* x_nhwc3, int M, int H, int W, const uint16_t* w, uint16_t* y, w2c_stream_t stream);
long long w2c_stem_wgrad_workspace_bytes(int M, int H, int W);
int w2c_stem_wgrad_bf16(const uint16_t* x_nhwc3, int M, int H, int W, const uint16_t* dy, int dy_cstride, float* dw,
                        void* workspace, long long workspace_bytes, w2c_stream_t stream);

/* ---- cross_entropy2d (ptsemseg/loss/loss.py:5-18 = F.cross_entropy(NCHW logits -> [P, C], target,
 * weight, size_average, ignore_index)) forward and backward.
 * logits : f32 NCHW [N, C, HW]; target : int64 [N, HW]; weight : f32 [C] or NULL.
 * Forward: lse f32 [N*HW] (log-sum-exp per pixel, kept for the backward), loss_px f32 [N*HW] or NULL (w_t * nll per pixel, 0
 * where ignored: the reduce=False form bootstrapped_cross_entropy2d loss.py:51-53 tops), out3 f32 [3] = {loss, denominator
 * (sum of kept w_t), count of targets outside [0,C) that are not ignore_index (dropped)}; loss = sum / denominator when
 * size_average else the sum.  workspace >= w2c_cross_entropy2d_workspace_bytes(N*HW).  Deterministic.
 * Backward: dlogits f32 NCHW = g * w_t * (softmax - onehot) on kept pixels, 0 elsewhere; g = gout[0] (device scalar or NULL = 1)
 * / denom[0] (device scalar = out3 + 1, or NULL) * gpx[pixel] (or NULL). */
long long w2c_cross_entropy2d_workspace_bytes(long long n_pixels);
int w2c_cross_entropy2d_forward(const float* logits, const long long* target, const float* weight, int N, int C, long long HW,
                                int ignore_index, int size_average, float* lse, float* loss_px, float* out3,
                                void* workspace, long long workspace_bytes, w2c_stream_t stream);
int w2c_cross_entropy2d_backward(const float* logits, const long long* target, const float* weight, const float* lse, int N,
                                 int C, long long HW, int ignore_index, const float* denom, const float* gout,
                                 const float* gpx, float* dlogits, w2c_stream_t stream);

/* ---- SURVEY 8f row 4 (output side): K9 fused with the evaluator's class argmax (trainer.py:804
 * `outputs.data.max(1)[1]`): labels u8 [M, 32h, 32w] = argmax over classes of the bilinear x32 upsample of `low`
 * (identical arithmetic to w2c_upsample_bilinear32, lowest index on ties); the f32 logits are never written. */
int w2c_upsample32_argmax(const float* low, int M, int h, int w, int low_cstride, int n_classes,
                          uint8_t* labels, w2c_stream_t stream);

/* ---- SURVEY 8f row 4 (output side, second half): the evaluator's confusion matrix on the device.
 * runningScore._fast_hist (metrics.py:99-108): hist[n*gt + pred] += 1 over the pixels with 0 <= gt < n.
 * w2c_upsample32_argmax_confusion = w2c_upsample32_argmax + that histogram in one launch: neither the logits nor
 * (when labels == NULL) the label map are written; the evaluator reads n*n int64 counters per validation pass.
 * gt   : ground-truth labels [M, 32h, 32w], u8 (gt_is_i64=0; 4-byte aligned) or int64 (gt_is_i64=1) as the loader
 *        yields them; values outside [0, n) are ignored (the reference's mask)
 * hist : int64 [n*n], ACCUMULATED into (zero it once per evaluation); n_classes <= 64.  Exact (integer atomics).
 * ws   : optional (NULL / ws_partials 0: every workgroup adds its counts to hist directly -- ~1e5 atomics on n*n addresses) workspace of
 *        the two-level flush: int64 [ws_partials * S + 16], ws_partials = 32, S = n*n rounded up to 16, zeroed by the caller ONCE and left zeroed by every
 *        launch (the last workgroup to finish moves the partial sums into hist).  One workspace per stream: launches that share one
 *        must be stream-ordered.  Same result either way. */
int w2c_upsample32_argmax_confusion(const float* low, int M, int h, int w, int low_cstride, int n_classes,
                                    const void* gt, int gt_is_i64, uint8_t* labels, long long* hist,
                                    long long* ws, int ws_partials, w2c_stream_t stream);
/* The same histogram for label maps that already exist: pred u8 [n_pixels], gt u8 or int64 [n_pixels]. */
int w2c_confusion_matrix(const void* gt, int gt_is_i64, const uint8_t* pred, long long n_pixels, int n_classes,
                         long long* hist, w2c_stream_t stream);

/* ---- helpers at the boundary ---- */
/* f32 NCHW [M,C,H,W] -> bf16 NHWC [M,H,W,cstride] (channels [0,C)); used by tests and the
 * Single-kernel parity harness, not by the forward path. */
int w2c_nchw_f32_to_nhwc_bf16(const float* x, int M, int C, int H, int W, uint16_t* y, int y_cstride,
                              w2c_stream_t stream);
int w2c_nhwc_bf16_to_nchw_f32(const uint16_t* x, int x_cstride, int M, int C, int H, int W, float* y,
                              w2c_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* W2C_HIP_H */
