/*
 * hawq_mi355.h - C ABI of libhawq_mi355.so: MI355X (gfx950) integer-only kernels for the
 * frozen forward of HAWQ's quantized ResNets.
 *
 * The reference (Zhen-Dong/HAWQ) has no FFI layer: its operator boundary is the Python
 * nn.Module API of utils/quantization_utils/quant_modules.py.  Each entry point below is
 * what a maintainer would bind (ctypes) to replace the arithmetic of
 * one of those modules' frozen forward; the citation on each says which lines it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless named host_*.
 *   - the library never allocates or frees activation memory and keeps no global mutable
 *     state besides a thread-local error string; the device is the pointer's device and
 *     work is enqueued on the hipStream_t passed as `void* stream` (0 = default stream).
 *   - every function returns 0 on success, non-zero on error; hawq_last_error() gives
 *     the message of the calling thread's last failure.
 *   - activations are NHWC.  8-bit tensors are int8; 4-bit tensors are nibble-packed
 *     "hawq4" format: within every group of 8 consecutive channels c0..c7 the 32-bit word is
 *     (c0|c1<<8|c2<<16|c3<<24) | (c4|c5<<8|c6<<16|c7<<24)<<4  (so that two mask ops unpack it
 *     into two int8x4 words in channel order).  16-bit residual tensors are uint16
 *     (post-ReLU values; overflow beyond 65535 sets bit 0 of *flags and saturates) or int32.
 *   - dyadic requantisation tables are (m, ek) pairs, ek = e | k << 8, 0 <= m < 2^31, 1 <= e <= 62:
 *     q = round_half_even(((acc << k) * m) / 2^e)  (quant_utils.py:188-213, 404-408); the
 *     pre-shift k (|acc << k| < 2^31) lets the host lift small e to >= 33 without changing the
 *     rational m * 2^k / 2^e (hawq_amd.quant_utils.requant_table).
 */
#ifndef HAWQ_MI355_H
#define HAWQ_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HAWQ_ABI_VERSION 5

const char *hawq_last_error(void);
int hawq_abi_version(void);
/* number of gfx950 kernels compiled into the library (sanity / "native code loaded" probe) */
int hawq_device_ok(void);

/* ---- epilogue kinds of hawq_conv2d ------------------------------------------------- */
enum {
    HAWQ_EPI_RAW = 0,      /* out_acc[m][c] = acc + bias               (int32 accumulators)  */
    HAWQ_EPI_REQUANT = 1,  /* out_q = clamp(dyadic(relu?(acc+bias)))   conv -> ReLU -> QuantAct */
    HAWQ_EPI_RESIDUAL = 2, /* residual add of two separately requantised branches, ReLU,
                              optional next-unit QuantAct                                     */
    HAWQ_EPI_DEQUANT = 3   /* out_f32[m][c] = float(acc+bias) * fscale[c]   (QuantLinear)     */
};

/*
 * One fused convolution launch.  Replaces, for a frozen model,
 *   QuantBnConv2d.forward   quant_modules.py:489-494   (integer conv + bias)
 *   nn.ReLU                 q_resnet.py:242,246,258
 *   QuantAct.forward/fixedpoint_fn case 0   quant_modules.py:288-293, quant_utils.py:390-413
 *   QuantAct.forward/fixedpoint_fn case 1   quant_modules.py:294-301, quant_utils.py:416-456
 *   (+ the next unit's block-input QuantAct, q_resnet.py:234/239)
 *   QuantLinear.forward     quant_modules.py:125-130   (HAWQ_EPI_DEQUANT, 1x1 "conv" on [B,1,1,K])
 * Implicit GEMM on int8 MFMA; 4-bit operands are unpacked on the way into LDS.
 */
typedef struct hawq_conv_args {
    /* main branch */
    const void *in;       /* [N][H][W][Cin]  int8, or hawq4-packed when in_bits == 4         */
    const void *wgt;      /* [Cout][KH][KW][Cin] int8 / hawq4-packed (see hawq_pack_*)        */
    const int32_t *bias;  /* [Cout] bias_integer                                              */
    int32_t N, H, W, Cin, Cout, KH, KW, stride, pad;
    int32_t in_bits, w_bits; /* 8 or 4 each (any combination)                                  */
    /* optional second branch sharing the output grid: the identity 1x1 conv of a resize unit
       (q_resnet.py:236).  in2 == NULL disables it. */
    const void *in2;
    const void *wgt2;
    const int32_t *bias2;
    int32_t H2, W2, Cin2, stride2, in2_bits, w2_bits;
    int32_t epilogue; /* HAWQ_EPI_* */
    int32_t relu;     /* apply max(.,0) to acc+bias before requantising (REQUANT only)         */
    /* requant tables of the main branch: per output channel */
    const int32_t *m;
    const int32_t *e;
    /* RESIDUAL: identity branch tables - per channel (second branch) or scalar (passthrough) */
    const int32_t *m_id;
    const int32_t *e_id;
    int32_t m_id_scalar, e_id_scalar;
    const void *res_in;   /* [M][Cout] block input carried as residual (uint16 or int32)      */
    int32_t res_in_bits;  /* 16 or 32                                                          */
    void *res_out;        /* [M][Cout] post-ReLU residual for the next unit, or NULL           */
    int32_t res_out_bits; /* 16 or 32                                                          */
    /* quantised output (REQUANT: this conv's QuantAct; RESIDUAL: next unit's QuantAct) */
    void *out_q;          /* [M][Cout] int8 / hawq4, or NULL                                   */
    int32_t out_bits;     /* 8 or 4                                                            */
    int32_t q_lo, q_hi;   /* clamp range                                                       */
    int32_t mq, eq;       /* RESIDUAL: scalar table of the next QuantAct (S_w == 1)            */
    int32_t *out_acc;     /* RAW: [M][Cout] int32                                              */
    float *out_f32;       /* DEQUANT: [M][ldo] fp32, only channels < n_valid are written       */
    const float *fscale;  /* DEQUANT: [Cout]                                                   */
    int32_t ldo, n_valid; /* n_valid > 0 on the general REQUANT / RESIDUAL path (fast_tables == 0): channels >= n_valid are
                             padding (zero weights, bias, tables and identity) and are written as zeros without arithmetic */
    int32_t *flags;       /* device int32: bit0 = uint16 residual overflow                     */
    int32_t tile;         /* 0 = heuristic; else tile config id (see hawq_conv2d_num_tiles)    */
    const int32_t *ctab;    /* fast path only: [Cout][4] = {m, (e-32)|k<<8, lo32(C), hi32(C)}, C = (bias<<k)*m + 2^(e-1) */
    const int32_t *ctab_id; /* fast path, second branch: same for (bias2, m_id, e_id)                   */
    int32_t fast_tables;  /* caller asserts the "fast contract" for EVERY dyadic table of this call:
                             e in [33,62]; |value << k| < 2^31; no exact
                             rounding tie is possible for the value ranges involved (the host proves
                             this from the trailing zeros of m - hawq_amd.quant_utils.tables_are_fast).
                             Enables the 2-instruction requant path and the LDS-staged coalesced
                             epilogue (8/8 and 4/4 operand widths, 16-bit residuals); needs ctab
                             (and ctab_id with a second branch).  0 = exact general path (any e in
                             [1,62], any k, ties handled) driven by bias / m / e.
                             Bit 1 (value 3): hint that every pre-shift k of ctab, ctab_id and
                             (mq, eq) is 0 (checked for eq; currently not used to pick a kernel).
                             Bit 2 (value 5): the tie-freedom proof FAILED for some entry; the kernel
                             then applies the exact round-half-even tie correction to every requant
                             of the call (e in [33,62] and |value << k| < 2^31 still required).
                             Bit 3 (value | 8): every PER-CHANNEL pre-shift k of this conv's ctab (and ctab_id) is 0
                             (the scalar tables may still carry one): hawq_conv_expand_reduce then runs the
                             instantiation without the shift / field extraction (3 VALU instructions per output
                             fewer); other entry points ignore the bit.                                          */
    int32_t in_planar;    /* layout of `in`: 0 = NHWC pixel rows [M][Cin*bits/8];  1 = channel-group planes
                             [Cin/G][M][16 B] with G = 16 (int8) / 32 (hawq4) channels per 16-byte unit.  Planes
                             are what the 3x3 band kernels' LDS-DMA fill wants (64 consecutive pixels of one plane
                             are one contiguous KiB); only those kernels read them.                              */
    int32_t out_planar;   /* same for `out_q` (fast-contract REQUANT / RESIDUAL epilogues only)                  */
    /* ABI 3 - RESIDUAL epilogue of networks whose units end WITHOUT an activation (MobileNetV2's linear bottleneck,
       q_mobilenetv2.py:60-93): exact general path only (fast_tables == 0), 32-bit residual tensors.
       res_no_relu: 1 = the requantised sum is stored as is (signed); 0 = ReLU after it (the ResNets, q_resnet.py:258).
       res_clamp16: 1 = clamp the result to [-32768, 32767]: quant_act_int32 WITHOUT an identity branch is fixedpoint_fn case 0,
                    which clamps to its 16-bit range (quant_utils.py:409-413); with an identity (case 1) nothing clamps (:456).
       With res_in == NULL and in2 == NULL there is no identity branch: o = requant(acc + bias).                          */
    int32_t res_no_relu, res_clamp16;
    /* ABI 4 - narrow tensors stored at their own width (MobileNetV2's 16 / 24 / 32 / 96 / 144 / 160-channel tensors,
       q_mobilenetv2.py:36-58: in the reference they are exactly that wide; padding every one of them to the 64-channel tile
       doubled to quadrupled the bytes of the early layers).
       in_pitch:  bytes from one pixel row of `in` to the next; 0 = dense (Cin * in_bits / 8).  A pitch BELOW the dense row lets a
                  tensor of Cin_valid <= in_pitch channels feed a conv whose K (Cin) is padded to 64: the 64-byte chunk reads run on
                  into the following pixel's bytes, where they meet zero weights (integers: garbage x 0 = 0, exact).  Multiple of
                  16, int8 operands, single branch, not planar, not the band kernels; the buffer must be readable 64 bytes past
                  its last row.
       out_pitch: channels per stored pixel row of out_q / res_out / res_in; 0 = Cout.  Channels >= out_pitch are not written
                  at all; on the general path channels in [n_valid, out_pitch) are written as zeros.  Multiple of 16,
                  REQUANT / RESIDUAL epilogues with NHWC rows (fast REQUANT, or the general path), single branch.           */
    int32_t in_pitch, out_pitch;
    /* ABI 5 - the round-5 3x3 kernels (band_v2.hip; 3x3 / stride 1 / pad 1, int8 operands, Cin >= 128, planar input): the SAME weight
       integers as `wgt` (quant_modules.py:462-466: weight_integer of the BN-folded conv), packed by hawq_pack_w3x3_band into the byte
       stream a workgroup consumes - [Cout/64][Cin/64][kh][kw][64 rows][64 B], the 16-byte slots of a row XOR-swizzled - so that every
       LDS-DMA instruction of the weight ring copies one contiguous KiB.  NULL = not provided: those tile ids refuse the layer. */
    const void *wgt_band;
    /* ABI 5 - the round-5 1x1 kernels (gemm_v2.hip; 1x1 / pad 0 convs whose pixel row is a multiple of 128 BYTES - int8 operands with
       Cin % 128 == 0 or, round 6, hawq4 operands (in_bits == w_bits == 4) with Cin % 256 == 0 -, NHWC rows): the weights of
       `wgt` / `wgt2` packed by hawq_pack_w1x1_k128 - [Cout/64][row bytes/128][64 rows][128 B], 16-byte slot s of row r at s ^ ((r >> 1) & 7) -
       so that the K loop walks full 128-byte lines and every weight piece is a contiguous KiB.  NULL = not provided. */
    const void *wgt_k128;
    const void *wgt2_k128;
} hawq_conv_args;

int hawq_conv2d(const hawq_conv_args *args, void *stream);
int hawq_conv2d_num_tiles(void);
/* The LAST hawq_conv2d_num_band_tiles() tile ids are special-purpose kernels (fast-contract layers only; hawq_conv2d refuses them
 * for any layer they are not built for): the 3x3/stride-1/pad-1 "band" kernels of rounds 1-3, then the weight-stationary persistent
 * kernel for Cin == Cout == 64 (int8 in and out; REQUANT - or, round 5, single-branch RESIDUAL on uint16 residuals; NHWC or planar output)
 * with one / two workgroups per CU, then (ABI 5) the
 * hawq_conv2d_num_band2_tiles() 3x3 kernels and the hawq_conv2d_num_gemm2_tiles() streaming 1x1 kernels of round 5.  All other ids
 * take any layer. */
int hawq_conv2d_num_band_tiles(void);
/* 1-based id of the preferred band tile that takes this layer as described (geometry, widths, epilogue,
 * fast_tables), 0 if none does: lets a caller decide whether the producer should write planar activations. */
int hawq_conv2d_band_tile(const hawq_conv_args *args);
/* ABI 5: hawq_conv2d_num_band2_tiles() of those ids (the ones in front of the 1x1 kernels) are the round-5 3x3 kernels (band_v2.hip; need args->wgt_band and
 * in_planar == 1; int8 operands with Cin >= 128, or hawq4 operands (in_bits == w_bits == 4) with Cin >= 256; REQUANT, or single-branch RESIDUAL on uint16
 * residuals; int8 / hawq4 output, NHWC or planar).  hawq_conv2d_band2_tile: 1-based id of the first of them that takes the layer as described, else 0.
 * hawq_pack_w3x3_band: [Cout][3][3][Cin] int8 (the layout of `wgt`) -> the stream described at hawq_conv_args.wgt_band, on the host
 * (dst and src are host pointers of Cout * 9 * Cin bytes; Cin = BYTES per tap row - the channel count for int8 weights, half of it for hawq4 -; Cin and Cout multiples of 64). */
int hawq_conv2d_num_band2_tiles(void);
int hawq_conv2d_band2_tile(const hawq_conv_args *args);
int hawq_pack_w3x3_band(const int8_t *src, int8_t *dst, int32_t Cout, int32_t Cin);
/* ABI 5: the LAST hawq_conv2d_num_gemm2_tiles() tile ids are the round-5 streaming
 * 1x1 kernels (need args->wgt_k128 - and wgt2_k128 with a second branch -, NHWC input, fast_tables; REQUANT, RESIDUAL on uint16
 * residuals, or RESIDUAL with the identity conv as second branch).  hawq_conv2d_gemm2_first(): the 1-based id of the first of them.
 * hawq_pack_w1x1_k128: [Cout][Cin] int8 -> the stream described at hawq_conv_args.wgt_k128 (host pointers; Cin = BYTES per weight row).
 * Round 6 (no change of the argument block): both round-5 kernel families also take
 *   - HAWQ_EPI_RAW (out_acc: int32 accumulators + bias, dense [M][Cout]; no tables needed) - what the parity tests compare with the
 *     oracle's exact sums (quant_modules.py:489-494), and
 *   - the streaming 1x1 kernels take hawq4 operands (both branches of a dual launch must have the same widths) and write int8 or
 *     hawq4 outputs (out_bits 4: q_lo >= 0, q_hi <= 15), NHWC rows or channel-group planes - the reduce convs of the 4-bit schedules
 *     (bit_config.py:806, 1512). */
int hawq_conv2d_num_gemm2_tiles(void);
int hawq_conv2d_gemm2_first(void);
int hawq_pack_w1x1_k128(const int8_t *src, int8_t *dst, int32_t Cout, int32_t Cin);

/* Fused launch of two consecutive layers of the bottleneck graph (q_resnet.py:231-260): the 1x1 expand conv of unit i
 * with its RESIDUAL epilogue (x + identity -> quant_act_int32 -> ReLU -> quant_act of unit i+1) and the 1x1 reduce
 * conv of unit i+1 with its REQUANT epilogue (ReLU -> quant_act1).  `expand` and `reduce` are filled exactly as for
 * two hawq_conv2d calls, except that expand.out_q and reduce.in are ignored: the 8-bit block input of unit i+1 stays
 * on chip.  Needs: both convs 1x1 / stride 1, int8 operands, fast_tables != 0, uint16 residual in and out (single
 * branch) - or, for expand.Cin == 64, a second branch in2 / wgt2 / ctab_id that is a 1x1 / stride-1 conv over the same pixels with
 * Cin2 == 64 (the first unit of ResNet50's stage 1: its requantised accumulators replace the stored residual) -,
 * reduce.Cin == expand.Cout, reduce.Cout == expand.Cin in {64, 128, 256}; reduce.out_bits 8, or 4 (round 3: the reduce conv's QuantAct
 * is 4-bit and a nibble 3x3 conv follows - the launch packs hawq4 itself, NHWC rows of Cout / 2 bytes or planes [Cout / 32][M][16 B];
 * needs 0 <= q_lo, q_hi <= 15).
 * tile: 0 = default kernel variant for the channel count, 1..hawq_conv_expand_reduce_variants() = a specific one (the variants of
 * fused_er.hip first, then the wave-private ones of fused_wp.hip: same results, different organisation of the launch).
 * reduce.wgt == NULL (round 3): the expand conv ALONE on the wave-private kernel - `expand` exactly as for hawq_conv2d with the
 * RESIDUAL epilogue (single branch, uint16 residual in, 8-bit out_q, res_out optional; expand.Cin in {64, 128, 256, 512}); some
 * variants split the output channels over gridDim.y.  Round 6: expand.out_bits 4 (0 <= q_lo, q_hi <= 15) writes the next unit's block
 * input as hawq4 rows of Cout / 2 bytes - the last expand conv of a stage in a 4-bit schedule, whose successor reads nibbles
 * (bit_config.py:806, 1512).  0 variants = use hawq_conv2d. */
typedef struct hawq_expand_reduce_args {
    hawq_conv_args expand;
    hawq_conv_args reduce;
    int32_t tile;
} hawq_expand_reduce_args;
int hawq_conv_expand_reduce(const hawq_expand_reduce_args *args, void *stream);
/* number of kernel variants that take this pair (0: the pair cannot be fused, launch two hawq_conv2d instead) */
int hawq_conv_expand_reduce_variants(const hawq_expand_reduce_args *args);

/* QuantAct input case (quant_modules.py:271-274; quant_utils.py:73-97, 237-258):
 * q = clamp(rint(inv_scale * x), lo, hi); fp32 NCHW [N][3][H][W] -> int8 NHWC4 with a zero
 * border: out[N][H+2*pad_t..][..][4]; out_h/out_w are the padded extents, (pad_top, pad_left)
 * the offset of pixel (0,0).  The border and channel 3 must have been zeroed once. */
int hawq_quantize_input(const float *x, int8_t *out, int32_t N, int32_t C, int32_t H, int32_t W,
                        int32_t out_h, int32_t out_w, int32_t pad_top, int32_t pad_left,
                        float inv_scale, int32_t lo, int32_t hi, void *stream);

/* Stem: 7x7/2 conv on the padded NHWC4 int8 image + bias, then QuantAct(16b, case 0, clamp)
 * + ReLU fused (requant commutes with the following max-pool because it is monotone):
 * quant_modules.py:489-494 + q_resnet.py:117-122.  wgt is [64][7][8][4] int8 (kw, c zero
 * padded).  out16 [N][Ho][Wo][64] uint16;  out_acc (optional) raw int32. */
int hawq_stem_conv7(const int8_t *in, const int8_t *wgt, const int32_t *bias, const int32_t *m,
                    const int32_t *e, int32_t N, int32_t Hp, int32_t Wp, int32_t Ho, int32_t Wo,
                    int32_t q_lo, int32_t q_hi, uint16_t *out16, int32_t *out_acc, void *stream);

/* Fused stem: QuantAct input case + 7x7/2 conv + bias + MaxPool2d(3,2,1) + QuantAct(16b, clamp) + ReLU
 * + first unit's QuantAct in one launch (q_resnet.py:115-122, 234/239): fp32 NCHW images ->
 * res_out [N][Hp][Wp][64] uint16 and/or out_q int8/hawq4.  The max-pool is taken on the raw accumulators
 * (the per-channel requantisation is monotone), so the 112x112 intermediate never reaches memory.
 * fast_tables: the caller asserts the fast contract for (m, e) and (mq, eq) (see hawq_conv_args). */
int hawq_stem_fused(const float *x, int32_t N, int32_t C, int32_t H, int32_t W, float inv_scale, int32_t in_lo,
                    int32_t in_hi, const int8_t *wgt, const int32_t *bias, const int32_t *m, const int32_t *e,
                    int32_t a_lo, int32_t a_hi, uint16_t *res_out, void *out_q, int32_t out_bits, int32_t mq,
                    int32_t eq, int32_t q_lo, int32_t q_hi, int32_t fast_tables, void *stream);

/* Same kernel fed by uint8 NHWC images [N][H][W][C] (decoder output) and a host-built table
 * lut[c][u] = clamp(rint(fl(1/S) * normalise_c(u / 255))), c < 3, u < 256 (int8, 4-byte aligned): the input
 * QuantAct (quant_modules.py:271-274) of the normalised image becomes a look-up, bit-identical to quantising the
 * fp32 tensor that torchvision's ToTensor + Normalize (quant_train.py:432-440) would have produced.            */
int hawq_stem_fused_u8(const uint8_t *x, const int8_t *lut, int32_t N, int32_t C, int32_t H, int32_t W,
                       const int8_t *wgt, const int32_t *bias, const int32_t *m, const int32_t *e, int32_t a_lo,
                       int32_t a_hi, uint16_t *res_out, void *out_q, int32_t out_bits, int32_t mq, int32_t eq,
                       int32_t q_lo, int32_t q_hi, int32_t fast_tables, void *stream);

/* nn.MaxPool2d(3,2,1) (q_resnet.py:93,119) on the requantised stem output + the first
 * unit's QuantAct: in [N][H][W][C] uint16 -> res_out [N][Ho][Wo][C] uint16 and
 * out_q = clamp(dyadic(res, mq, eq)) int8/hawq4 (either may be NULL). */
int hawq_maxpool3s2_requant(const uint16_t *in, int32_t N, int32_t H, int32_t W, int32_t C,
                            uint16_t *res_out, void *out_q, int32_t out_bits, int32_t mq, int32_t eq,
                            int32_t q_lo, int32_t q_hi, void *stream);

/* Block-input QuantAct on a stored residual (quant_modules.py:288-293 with S_w == 1):
 * in [n] uint16/int32 -> out int8/hawq4.  Used when the producer could not fuse it. */
int hawq_requant_residual(const void *in, int32_t in_bits, int64_t n, void *out_q, int32_t out_bits,
                          int32_t mq, int32_t eq, int32_t q_lo, int32_t q_hi, void *stream);

/* QuantAveragePool2d + quant_act_output (quant_modules.py:596-600, quant_utils.py:334-337,
 * q_resnet.py:129-131): in [N][HW][C] uint16/int32 (>= 0) -> floor(sum/HW) -> dyadic -> clamp
 * -> int8 [N][C].  pooled_out (optional) receives the int32 pooled integers. */
int hawq_avgpool_requant(const void *in, int32_t in_bits, int32_t N, int32_t HW, int32_t C, int8_t *out,
                         int32_t *pooled_out, int32_t mq, int32_t eq, int32_t q_lo, int32_t q_hi,
                         void *stream);

/* QuantLinear on the frozen integer path (quant_modules.py:112-130: F.linear(x_int, weight_integer, bias_integer) * (fc_scaling_factor *
 * prev_act_scaling_factor); the classifier of every Q_ResNet, q_resnet.py:132-134) as its own launch (fc_dequant.hip, round 5):
 *   out_f32[n * ldo + o] = (float)(sum_k q[n][k] * wgt[o][k] + bias[o]) * fscale[o]      for o < n_valid
 * q [N][K] int8, wgt [Nout_p][K] int8 (row-major = the [Cout][1][1][Cin] layout hawq_conv2d takes), bias / fscale [Nout_p].  The same bytes
 * as hawq_conv2d with epilogue DEQUANT on the 1 x 1 map (tests/test_gpu_kernels.py), several times faster for the M = batch, long-K
 * shape.  hawq_fc_dequant_ok: K % 128 == 0 and Nout_p % 32 == 0. */
int hawq_fc_dequant_ok(int32_t N, int32_t K, int32_t Nout_p);
int hawq_fc_dequant(const int8_t *q, const int8_t *wgt, const int32_t *bias, const float *fscale, float *out_f32, int32_t N, int32_t K,
                    int32_t Nout_p, int32_t n_valid, int32_t ldo, void *stream);

/* ---- module-compatible (fp32 tuple convention) adapters ---------------------------------
 * The reference modules exchange fp32 tensors holding integer*scale.  These kernels move
 * between that convention (NCHW fp32) and the integer NHWC tensors the conv kernels use. */

/* x_int = rint(x / S) (quant_modules.py:490, 126): fp32 NCHW [N][C][H][W] -> int8 / hawq4 NHWC
 * with Cpad >= C channels (extra channels zero). */
int hawq_f32_nchw_to_q_nhwc(const float *x, void *out, int32_t N, int32_t C, int32_t H, int32_t W,
                            int32_t Cpad, int32_t bits, float scale, void *stream);
/* y = float(acc) * fscale[c] (quant_modules.py:491-494): int32 NHWC [N][H][W][Cpad] -> fp32 NCHW */
int hawq_acc_nhwc_to_f32_nchw(const int32_t *acc, float *y, int32_t N, int32_t C, int32_t H, int32_t W,
                              int32_t Cpad, const float *fscale, void *stream);
/* QuantAct / fixedpoint_fn on fp32 NCHW tensors (quant_utils.py:363-456).
 * case 0: y = clamp(dyadic(rint(z/S_a/S_w[c]), m[c], e[c])) * S_out
 * case 1: y = (dyadic(rint(ident/S_ida/S_idw[c]), m1, e1) + dyadic(rint((z-ident)/S_a/S_w[c]), m2, e2)) * S_out
 * per_ch = number of entries in sw/m/e (1 or C); per_ch_id likewise for the identity tables. */
int hawq_fixedpoint_f32(const float *z, float *y, int32_t N, int32_t C, int32_t HW, float s_a,
                        const float *s_w, const int32_t *m, const int32_t *e, int32_t per_ch,
                        const float *ident, float s_ida, const float *s_idw, const int32_t *m_id,
                        const int32_t *e_id, int32_t per_ch_id, float s_out, int32_t do_clamp,
                        int32_t q_lo, int32_t q_hi, void *stream);
/* QuantAct input case on fp32 (quant_utils.py:73-97): y = clamp(rint(inv_scale*x)) * scale */
int hawq_fakequant_f32(const float *x, float *y, int64_t n, float inv_scale, float scale, int32_t lo,
                       int32_t hi, void *stream);
/* QuantAveragePool2d on fp32 NCHW (quant_modules.py:596-602): y = trunc(avg(rint(x/S)) + 0.01) * S */
int hawq_avgpool_f32(const float *x, float *y, int32_t NC, int32_t HW, float scale, void *stream);

/* Grouped / depthwise integer conv for the module-compatible path (F.conv2d(..., groups), quant_modules.py:489-494 and
 * 727-736; MobileNetV2's depthwise 3x3): in [N][H][W][Cin] int8, wgt [Cout][KH][KW][Cin/groups] int8, bias [Cout] or NULL
 * -> out_acc [N][Ho][Wo][Cout] int32 (exact). */
int hawq_conv2d_grouped(const int8_t *in, const int8_t *wgt, const int32_t *bias, int32_t N, int32_t H, int32_t W,
                        int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad, int32_t groups,
                        int32_t *out_acc, void *stream);

/* Depthwise 3x3 / pad 1 conv (groups == Cin == Cout; MobileNetV2's conv2, q_mobilenetv2.py:46-48 through QuantBnConv2d.forward,
 * quant_modules.py:489-494): in [N][H][W][C] int8, wgt9c [3][3][C] int8 (tap-major), bias [C] or NULL -> out_acc
 * [N][Ho][Wo][C] int32 (exact).  C % 4 == 0, stride 1 or 2. */
int hawq_depthwise3x3(const int8_t *in, const int8_t *wgt9c, const int32_t *bias, int32_t N, int32_t H, int32_t W, int32_t C,
                      int32_t stride, int32_t *out_acc, void *stream);

/* The same with the conv's activation + QuantAct fused (conv2 -> ReLU6 -> quant_act2 of a MobileNetV2 unit, q_mobilenetv2.py:70-72;
 * ReLU6 == ReLU + the QuantAct's own clamp: its calibrated range never exceeds 6): out_q[N][Ho][Wo][C] int8 =
 * clamp(dyadic_rne(relu ? max(acc + bias, 0) : acc + bias, m[c], e[c]), q_lo, q_hi) with exact (tie-aware) rounding;
 * out_acc (optional) additionally receives the int32 accumulators.  0 < C_valid < C: channels >= C_valid are padding up to the
 * conv kernels' 64-channel tiles (zero weights / bias / m) - they are written as zeros and cost no loads or MACs (0 = all real). */
int hawq_depthwise3x3_requant(const int8_t *in, const int8_t *wgt9c, const int32_t *bias, const int32_t *m, const int32_t *e,
                              int32_t N, int32_t H, int32_t W, int32_t C, int32_t C_valid, int32_t stride, int32_t relu, int32_t q_lo,
                              int32_t q_hi, int8_t *out_q, int32_t *out_acc, void *stream);

/* The same launch with the fast requant contract (round 4): ctab [C][4] = the fused constants of the layer's requant with the bias folded in
 * (hawq_amd.packing.pack_ctab), fast_tables with the meaning it has in hawq_conv_args: 1, or 5 = exact ties (the host proves the contract, hawq_amd.quant_utils);
 * ReLU is the lower clamp (q_lo >= 0).  3-4 instructions per requant instead of the exact form's ~20. */
int hawq_depthwise3x3_requant_fast(const int8_t *in, const int8_t *wgt9c, const int32_t *ctab, int32_t fast_tables, int32_t N, int32_t H, int32_t W,
                                   int32_t C, int32_t C_valid, int32_t stride, int32_t q_lo, int32_t q_hi, int8_t *out_q, void *stream);

/* One launch per linear-bottleneck unit of MobileNetV2 (Q_LinearBottleneck.forward, q_mobilenetv2.py:59-93; round 4):
 *   block-input int8 -> conv1 1x1 (+ReLU6 + quant_act1) -> conv2 depthwise 3x3 / stride 1|2 / pad 1 (+ReLU6 + quant_act2)
 *   -> conv3 1x1 -> quant_act_int32 (with / without the identity branch) -> the next block-input QuantAct,
 * the expanded ("hidden") tensor never leaves the CU: a workgroup owns an 8 x 16 tile of OUTPUT pixels of one image, recomputes the
 * 1x1 expand conv on the tile's halo window, runs the depthwise taps out of LDS and feeds the projection GEMM from LDS, 32 hidden
 * channels at a time.  Same integers as the three launches (hawq_conv2d REQUANT, hawq_depthwise3x3_requant, hawq_conv2d RESIDUAL).
 *   expand:  as for hawq_conv2d with the REQUANT epilogue and fast_tables != 0 (ctab, q_lo / q_hi, relu = 1); 1x1 / stride 1; Cin is the
 *            K of its packed weights (64 or 128; 192 for the wide units below), in_pitch in {16, 32, 64, 96} (0 = Cin); Cout = the hidden
 *            width padded to 64; out_q is
 *            ignored.  fast_tables bit 3 on BOTH expand and dw_fast_tables (no per-channel pre-shift in ctab / dw_ctab) selects the
 *            instantiation without the shift.
 *   dw_*:    wgt9c [9][expand.Cout] int8 tap-major (zero beyond the real channels); dw_ctab [expand.Cout][4] fused constants of its
 *            requant with the bias folded in (hawq_amd.packing.pack_ctab); dw_fast_tables as hawq_conv_args.fast_tables (1, or 5 = exact
 *            ties); clamp [dw_q_lo >= 0 (ReLU), dw_q_hi <= 127].
 *   project: as for hawq_conv2d with the RESIDUAL epilogue on the direct form: fast_tables != 0 (ctab), res_no_relu = 1, res_clamp16,
 *            res_in (int32, optional identity: then H x W, stride 1, out_pitch == expand's in_pitch), res_out (int32, optional), out_q
 *            int8 (optional), mq / eq / q_lo / q_hi; Cin == expand.Cout; Cout = 64 or 128 packed rows, out_pitch in {16, 32, 64, 96}
 *            (0 = Cout); `in` is ignored.
 *   c_mid:   real hidden channels (channels >= c_mid of every hidden-side table / weight are zero padding).
 *   Wide units: on output maps of at most 8 x 8 pixels the launch also takes in_pitch 160 -> out_pitch 160 or 320 at stride 1 and
 *            in_pitch 96 -> out_pitch 160 at stride 2 (c_mid > 96): 8 x 8 tiles, the projection spread over the waves by output blocks.
 * hawq_linear_bottleneck_ok: 1 when this launch takes the unit as described, else 0 (use the three launches). */
typedef struct hawq_bottleneck_args {
    hawq_conv_args expand;
    hawq_conv_args project;
    const int8_t *dw_wgt9c;
    const int32_t *dw_ctab;
    int32_t dw_stride, dw_q_lo, dw_q_hi, dw_fast_tables;
    int32_t c_mid;
    int32_t tile;   /* 0 = default (channel-planar hidden tensor); 1 = the [pixel][channel] organisation (inputs / outputs up to 64 channels) */
} hawq_bottleneck_args;
int hawq_linear_bottleneck(const hawq_bottleneck_args *args, void *stream);
int hawq_linear_bottleneck_ok(const hawq_bottleneck_args *args);

/* Input QuantAct + im2col for a 3x3 / stride 2 / pad 1 first conv on 3 channels (MobileNetV2's init block, q_mobilenetv2.py:182-186
 * after quant_modules.py:271-274): x fp32 [N][3][H][W] -> out int8 [N][Ho][Wo][64], row = the 27 values
 * clamp(rne(inv_scale * x), q_lo, q_hi) of the output pixel's patch in (kh, kw, c) order (zero outside the image), then 37 zeros.
 * The conv then runs as a 1x1 hawq_conv2d with Cin = 64 on weights laid out in the same (kh, kw, c) order.  C must be 3. */
int hawq_quantize_im2col3x3s2(const float *x, int8_t *out, int32_t N, int32_t C, int32_t H, int32_t W, float inv_scale, int32_t q_lo,
                              int32_t q_hi, void *stream);

/* MobileNetV2's init block as ONE launch (round 4; q_mobilenetv2.py:176-186, 60-65 after quant_modules.py:271-274): the input QuantAct,
 * the 3x3 / stride 2 / pad 1 conv on 3 channels, ReLU6 (as ReLU + clamp), quant_act_int32 and the first unit's block-input QuantAct;
 * neither the patch rows of hawq_quantize_im2col3x3s2 nor anything else intermediate reaches memory.
 *   x / x_u8: exactly one of fp32 NCHW [N][3][H][W] (then inv_scale, in_lo, in_hi describe the input QuantAct) and uint8 NHWC [N][H][W][3]
 *             with lut int8 [3][256] (as hawq_quantize_im2col3x3s2_u8);
 *   conv:     the 1x1 hawq_conv2d launch of the im2col path exactly as it would be issued on the patch rows (N, H = Ho, W = Wo, Cin = Cout = 64,
 *             wgt rows = the 27 taps in (kh, kw, c) order then zeros; RESIDUAL epilogue without identity, fast_tables != 0 with ctab,
 *             res_no_relu / res_clamp16, out_q + mq / eq / q_lo / q_hi, optional int32 res_out, out_pitch 16 or 32); `in` is ignored.
 * hawq_stem3x3s2_ok: 1 when the launch takes this description. */
int hawq_stem3x3s2(const float *x, const uint8_t *x_u8, const int8_t *lut, int32_t H, int32_t W, float inv_scale, int32_t in_lo, int32_t in_hi,
                   const hawq_conv_args *conv, void *stream);
int hawq_stem3x3s2_ok(const float *x, const uint8_t *x_u8, const int8_t *lut, int32_t H, int32_t W, const hawq_conv_args *conv);

/* The same patch rows from uint8 NHWC images [N][H][W][3] (decoder output after resize / crop, quant_train.py:428-440): ToTensor +
 * Normalize + the input QuantAct as one table look-up per channel, lut int8 [3][256] built on the host with the reference
 * pipeline's own float operations (hawq_amd.quant_utils.input_quant_lut) - bit-identical to quantising the normalised fp32 tensor. */
int hawq_quantize_im2col3x3s2_u8(const uint8_t *x, const int8_t *lut, int8_t *out, int32_t N, int32_t C, int32_t H, int32_t W, void *stream);

/* One separable pass of Pillow's 8-bit antialiased resampling (what torchvision's Resize(256) does to the decoded PIL image,
 * quant_train.py:428-440): uint8 HWC in / out, int32 coefficients with 22 fractional bits (hawq_amd/image.py builds them as
 * Resample.c's precompute_coeffs + normalize_coeffs_8bpc do).
 *   horizontal = 1: out[l][o][c], l < lines (input rows line0 + l), o < out_n output columns; in rows are in_w pixels wide
 *   horizontal = 0: out[o][l][c], o < out_n output rows, l < lines columns (in_w == lines, line0 == 0)
 * bounds[2o], bounds[2o+1] = first input index and number of taps of output o; coef[o * ksize + k]. */
int hawq_resample_u8(const uint8_t *in, int32_t in_w, int32_t C, const int32_t *bounds, const int32_t *coef, int32_t ksize,
                     int32_t out_n, int32_t horizontal, int32_t lines, int32_t line0, uint8_t *out, void *stream);

/* ---- range statistics of the un-frozen QuantAct (calibration / QAT range tracking) -----------------
 * x.data.min(), x.data.max() (quant_modules.py:233-236) of a fp32 tensor -> out2[0], out2[1] (device floats).
 * scratch: >= 8 bytes of device memory owned by the caller for the duration of the call. */
int hawq_minmax_f32(const float *x, int64_t n, float *out2, void *scratch, void *stream);
/* torch.kthvalue(x.view(-1), k).values (negate = 0) or -torch.kthvalue(-x.view(-1), k).values ... i.e. the k-th
 * smallest (1-based) element of x, or of -x returned with the sign of get_percentile_min_max's use
 * (quant_utils.py:38-70: `lower_bound = -torch.kthvalue(-input, k=lower_index).values`): out[0] = negate ? -kth(-x) : kth(x).
 * Exact (radix select on the order-preserving integer image of the floats).  scratch: >= 1040 bytes of device memory. */
int hawq_kthvalue_f32(const float *x, int64_t n, int64_t k, int32_t negate, float *out, void *scratch, void *stream);

/* ---- hipGraph helpers: capture a sequence of the launches above once, replay per batch */
int hawq_graph_begin(void *stream);
int hawq_graph_end(void *stream, void **graph_exec_out);
int hawq_graph_launch(void *graph_exec, void *stream);
int hawq_graph_destroy(void *graph_exec);

/* ---- timing helper for bench.py: HIP events on the caller's stream ---------------------- */
int hawq_event_create(void **ev);
int hawq_event_record(void *ev, void *stream);
int hawq_event_elapsed_ms(void *ev_start, void *ev_stop, float *ms);
int hawq_event_destroy(void *ev);

#ifdef __cplusplus
}
#endif
#endif /* HAWQ_MI355_H */
