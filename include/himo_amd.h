/*
 * himo_amd.h -- C ABI of libhimo_amd.so: the MI355X (gfx950) motion-compensation hot path
 * of KTH-RPL/HiMo.
 *
 * The reference has no native layer: its per-frame path is numpy code
 * inside two Python loops.  This ABI is what a maintainer of the reference would bind with
 * ctypes to replace that arithmetic (binding shown in INTEGRATION.md).  Each entry point
 * cites the reference lines it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - plain C types only; every `d_*` pointer is DEVICE memory (HBM), every `h_*` pointer is
 *     HOST memory; the caller owns all buffers, the library never allocates or frees;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); every call is
 *     asynchronous on that stream unless stated, re-entrant per stream, and holds no global
 *     state;
 *   - row-major arrays; point rows are `pc_stride` floats apart (3 = xyz, 4 = xyzi as the
 *     reference's `pc0`); 16-byte aligned base pointers take the vectorised path;
 *   - every function returns a himo_status; HIMO_OK == 0.  The Python host layer turns a
 *     non-zero status into the exception the reference would raise (ValueError / KeyError).
 */
#ifndef HIMO_AMD_H
#define HIMO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIMO_ABI_VERSION 1

typedef enum himo_status {
    HIMO_OK = 0,
    HIMO_ERR_INVALID_ARGUMENT = 1, /* NULL where data is required, negative sizes, bad stride */
    HIMO_ERR_EMPTY_FRAME = 2,      /* reference: `max()` of an empty lidar_dt raises ValueError (save_zip.py:120) */
    HIMO_ERR_WORKSPACE = 3,        /* workspace smaller than himo_*_workspace_bytes() */
    HIMO_ERR_SINGULAR_POSE = 4,    /* reference: np.linalg.inv raises LinAlgError (save_zip.py:115) */
    HIMO_ERR_HIP = 5,              /* a HIP runtime call failed; see himo_last_hip_error() */
    HIMO_ERR_UNSUPPORTED = 6
} himo_status;

/* flags for the comp_dis entry points */
#define HIMO_FLAG_F32_CHAIN 0x1u /* do the arithmetic in float32: what numpy does when the poses are
                                    float32 (dataprocess/extract_sca.py:232); default is the float64
                                    chain numpy runs for float64 poses, rounded to f32 at the end
                                    exactly where save_zip.py:70-72 casts */
#define HIMO_FLAG_RAW 0x2u       /* res_name == "raw": est_flow = zeros (save_zip.py:117) */
#define HIMO_FLAG_SCANIA 0x4u    /* eval mask also requires flow_is_valid (eval.py:293-294) */
#define HIMO_FLAG_POSE_IS_EGO 0x8u /* `pose0` already holds ego_pose = inv(pose1) @ pose0 (save_zip.py:115,
                                    computed by the caller, e.g. with numpy); `pose1` is ignored / may be NULL */

int himo_abi_version(void);
/* sizeof(struct) for "himo_conv_desc" | "himo_op" | "himo_sweep" | "himo_instance_record" (0 for an unknown name): lets
 * a binding verify its mirror of the structs that cross the boundary by address */
size_t himo_abi_sizeof(const char* struct_name);
const char* himo_status_string(int status);
/* text of the last HIP error seen by this thread (empty string if none) */
const char* himo_last_hip_error(void);

/* Optional per-kernel timing (used by bench.py for the roofline figure): while enabled, every
 * kernel launch of this library is bracketed by two HIP events on its own stream.
 * himo_prof_summary waits for the recorded launches and writes one line per kernel name:
 * "<name> <count> <total_ms> <min_ms> <max_ms>\n"; returns the bytes needed (incl. NUL). */
void himo_prof_enable(int on);
/* time only kernels whose name contains `substr` (NULL or "" = all): keeps the event records out of the other launches */
void himo_prof_filter(const char* substr);
void himo_prof_reset(void);
size_t himo_prof_summary(char* buf, size_t cap);
/* Diagnostic: the matrix-instruction rate the device sustains on its own -- independent v_mfma chains from registers on every
 * SIMD, no memory traffic, for about `min_seconds` (half of it untimed, to reach the power-managed clock).  kind 0: fp16
 * 32x32x16, 1: bf16 32x32x16, 2: float32 32x32x2; `zero_operands` != 0 feeds all-zero operands (the clock, hence the rate,
 * depends on the operand bits).  Synchronous.  Writes TFLOP/s to *tflops. */
int himo_mfma_sustained_tflops(int kind, int zero_operands, double min_seconds, double* tflops, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a1-a4 (+a5/a6): flow -> per-point de-distortion offsets, batched over ragged frames.
 *
 * Replaces, per frame f with rows [offsets[f], offsets[f+1]):
 *     ego_pose  = inv(pose1) @ pose0                                   save_zip.py:115  eval.py:284
 *     pose_flow = pc0[:, :3] @ ego_pose[:3,:3].T + ego_pose[:3,3] - pc0[:, :3]    save_zip.py:116
 *     est_flow  = flow - pose_flow            (zeros when RAW)          save_zip.py:117
 *     dt0       = max(lidar_dt) - lidar_dt                             save_zip.py:120
 *     comp_dis  = est_flow / sensor_dt * dt0[:, None]    utils/__init__.py:43, save_zip.py:121
 *     refined   = pc0[:, :3] + comp_dis                  utils/__init__.py:46     (optional)
 *     eval_mask = |pc0.xy| <= close_distance & ~gm0 & ego_pts_mask(pc0) [& flow_is_valid]
 *                                                         eval.py:288-296         (optional)
 *
 * d_offsets   int64[n_frames+1], non-decreasing, offsets[0] == 0, offsets[n_frames] == total_points
 * d_pose0/1   double[n_frames][16], row-major 4x4 (`pose0`, `pose1` of the frame dict)
 * d_pc0       float[total_points][pc_stride]; d_flow float[total_points][3] (ignored when RAW);
 * d_lidar_dt  float[total_points]
 * d_comp_dis  float[total_points][3]  (out)
 * d_refined   float[total_points][3]  (out, may be NULL)
 * d_eval_mask uint8[total_points]     (out, may be NULL; then the four arguments below are unused)
 * d_gm0, d_flow_is_valid  uint8/bool[total_points]  (d_flow_is_valid may be NULL unless SCANIA)
 * mask_bounds float[6] HOST: ego box min xyz, max xyz (utils/__init__.py:26; eval.py:296)
 * d_workspace at least himo_compdis_workspace_bytes(n_frames) bytes, 16-byte aligned.
 *
 * Frames with zero points are skipped (the single-frame wrapper reports HIMO_ERR_EMPTY_FRAME).
 * NaN entries of lidar_dt are ignored by the max.  A singular pose1 yields NaN outputs for that
 * frame (the single-frame wrapper reports HIMO_ERR_SINGULAR_POSE instead).
 */
size_t himo_compdis_workspace_bytes(int n_frames);

int himo_compdis_batch(int n_frames, int64_t total_points,
                       const int64_t* d_offsets, const double* d_pose0, const double* d_pose1,
                       const float* d_pc0, int pc_stride, const float* d_flow, const float* d_lidar_dt,
                       double sensor_dt, unsigned flags,
                       float* d_comp_dis, float* d_refined,
                       uint8_t* d_eval_mask, const uint8_t* d_gm0, const uint8_t* d_flow_is_valid,
                       const float* h_mask_bounds, float close_distance,
                       void* d_workspace, size_t workspace_bytes, void* stream);

/* One frame with host-side scalars: the body of the loop at save_zip.py:112-121.  Uploads the two
 * poses into the workspace (a small blocking copy), then runs the batch path with n_frames = 1. */
int himo_compdis_frame(int64_t n_points, const double* h_pose0, const double* h_pose1,
                       const float* d_pc0, int pc_stride, const float* d_flow, const float* d_lidar_dt,
                       double sensor_dt, unsigned flags,
                       float* d_comp_dis, float* d_refined,
                       void* d_workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The three functions of utils/__init__.py as stand-alone element-wise operators.
 * `dtype_is_f64` selects float64 flow/out arrays (numpy's result for float64 poses).
 */
/* utils/__init__.py:36-43  out = flow / sensor_dt * dt0[:, None];  flow,out: [n][3]; dt0: [n]
 * dtype_flags bit 0: flow and out are float64 (else float32); bit 1: dt0 is float64 (needs bit 0,
 * numpy's promotion rule) */
int himo_flow2compdis(int64_t n, const void* d_flow, const void* d_dt0, double sensor_dt,
                      int dtype_flags, void* d_out, void* stream);
/* utils/__init__.py:45-47  out = pc[:, :3] + ds;  pc float[n][pc_stride]; ds,out [n][3] */
int himo_refine_pts(int64_t n, const float* d_pc, int pc_stride, const void* d_ds,
                    int dtype_is_f64, void* d_out, void* stream);
/* utils/__init__.py:26-34  out[i] = 1 when point i is OUTSIDE the open box (min,max) */
int himo_ego_pts_mask(int64_t n, const float* d_pts, int pc_stride, const float* h_bounds,
                      uint8_t* d_out, void* stream);
/* save_zip.py:120 / eval.py:299  dt0 = max(lidar_dt) - lidar_dt  (one frame; d_workspace >= 32 bytes, 16-byte aligned) */
int himo_dt0(int64_t n, const float* d_lidar_dt, float* d_dt0, void* d_workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a8: exact k = 1 nearest neighbour (both halves of the Chamfer distance), brute force through LDS.
 * Replaces the cKDTree build + query pairs of eval.py:56-59 / tools/test/score.py:186-189 and is the
 * correspondence search of the self-supervised Chamfer loss.
 * Queries in rows [q_offsets[s], q_offsets[s+1]) search the references in rows
 * [r_offsets[s], r_offsets[s+1]) (segments = sweeps of a batch, or instances).  d_q, d_r: [n][3]
 * float32, or float64 when dtype_is_f64.  d_dist2: [nq] SQUARED distance in the same dtype (+inf for
 * an empty reference range); d_idx: [nq] int32 global reference row or -1 (may be NULL).
 * float64 mode is bit-comparable with cKDTree ((dx*dx + dy*dy) + dz*dz, then sqrt by the caller). */
int himo_nn_search(int n_segments, const int64_t* d_q_offsets, const int64_t* d_r_offsets,
                   int64_t nq, int64_t nr, const void* d_q, const void* d_r, int dtype_is_f64,
                   void* d_dist2, int32_t* d_idx, void* stream);
/* squared distances -> distances, in place (the sqrt cKDTree applies to the winning squared distance) */
int himo_sqrt_inplace(int64_t n, void* d_values, int dtype_is_f64, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a7 + a8: per-instance refinement metrics for a ragged batch of sweeps.
 * Replaces the per-category / per-instance loops of InstanceMetrics.step_eval (eval.py:64-114) and
 * ScoreMetrics.step (tools/test/score.py:223-321) up to, but not including, the bucket bookkeeping
 * (eval.py:99-147), which needs a few dozen records per sweep and stays on the host.
 * One record per (frame, class group, instance id) among the points with eval_mask != 0 and
 * class_lut[category] != 0, in no particular order (sort on the host). */
typedef struct himo_instance_record {
    int32_t frame;     /* index of the sweep in the batch */
    int32_t group;     /* class_lut value: 1 = CAR, 2 = OTHER_VEHICLES */
    int64_t instance;  /* flow_instance_id */
    int64_t num_pts;   /* eval.py:90 */
    double vel;        /* mean |gt_flow| / sensor_dt                         eval.py:91 */
    double dis;        /* mean |pc0 row| (all columns, float32 norm)         eval.py:94 */
    double mpe;        /* mean |gt_refined - est_refined|                    eval.py:95 */
    double cham;       /* (mean NN(gt->est) + mean NN(est->gt)) / 2          eval.py:50-62, :96 */
} himo_instance_record;

#define HIMO_EVAL_FLOW 0     /* d_est = estimated flow incl. ego motion (EVAL_FLAG 2, eval.py:301-305) */
#define HIMO_EVAL_COMPDIS 1  /* d_est = float32 comp_dis read from a zip (EVAL_FLAG 1, eval.py:306-310) */
#define HIMO_EVAL_RAW 2      /* est_flow = zeros (res_name == "raw") */
#define HIMO_EVAL_SCORE 3    /* leaderboard scorer: d_gt = GT comp_dis, d_est = predicted comp_dis (both float32),
                                d_pc0 = pc0 xyz or NULL, d_lidar_dt = gt_flow_norm or NULL; poses unused
                                (tools/test/score.py:299-306) */

#define HIMO_EVAL_DIRECT 4   /* step_eval's own arguments (eval.py:64): d_pc0 = masked points, d_gt = float64 [T][3]
                                ego-motion-free GT flow, d_lidar_dt = dt0 (already max - dt), d_est = float64 [T][3]
                                ego-motion-free estimated flow, or float32 comp_dis with HIMO_EVAL_DIRECT_EST_IS_DIS;
                                poses unused */
#define HIMO_EVAL_DIRECT_EST_IS_DIS 0x100u

size_t himo_eval_workspace_bytes(int n_frames, int64_t total_points, int64_t max_records);

/* d_gt: GT flow incl. ego motion [T][3] (data['flow']); d_category uint8[T]; d_instance int64[T] (ids must fit
 * 32 bits); d_eval_mask uint8[T] (e.g. from himo_compdis_batch); h_class_lut: HOST uint8[256].
 * d_records: himo_instance_record[max_records]; d_counts: int64[2] = {selected points, records found}.
 * flags: HIMO_FLAG_POSE_IS_EGO.  Always the float64 chain.  Synchronises the stream once internally. */
int himo_eval_instances(int n_frames, int64_t total_points,
                        const int64_t* d_offsets, const double* d_pose0, const double* d_pose1,
                        const float* d_pc0, int pc_stride, const void* d_gt, const void* d_est,
                        const float* d_lidar_dt, const uint8_t* d_category, const int64_t* d_instance,
                        const uint8_t* d_eval_mask, const uint8_t* h_class_lut,
                        double sensor_dt, int mode, unsigned flags,
                        himo_instance_record* d_records, int64_t max_records, int64_t* d_counts,
                        void* d_workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a10: scene-flow network (voxelise -> encoder/decoder -> per-point flow).
 * The reference's implementation is in the absent OpenSceneFlow submodule: these
 * entry points have NO reference lines to cite beyond the call sites README.md:50 (`save.py`) and the
 * hyper-parameters at assets/slurm/ssl-train-av2.sh:32-33; the specification is himo_amd/seflow/spec.py.
 * All feature maps are NHWC float32 addressed as base + n*batch_stride + pixel*pitch + channel.
 */
/* One sweep: rigid transform (h_transform: row-major 4x4 float32), dynamic pillarisation on a
 * grid_w x grid_h grid (h_range = min xyz, h_voxel = cell size, h_centre_offset = voxel/2 + min as
 * float32), pillar feature net Linear(9,32)+BN+ReLU+mean -> 32 floats at d_image[cell * image_pitch] (every cell written;
 * d_image 16-byte aligned, image_pitch a multiple of 4 floats).
 * Also returns the transformed points, each point's cell (-1 = out of range) and its offset to the cell
 * centre. */
size_t himo_pillar_workspace_bytes(int64_t max_points, int grid_w, int grid_h);
int himo_pillarize(int64_t n, const float* d_pts, int pc_stride, const float* h_transform,
                   const float* h_range, const float* h_voxel, const float* h_centre_offset,
                   int grid_w, int grid_h,
                   const float* d_pfn_weight, const float* d_pfn_scale, const float* d_pfn_shift,
                   float* d_xyz_t, int32_t* d_pid, float* d_offsets, float* d_image, int image_pitch,
                   void* d_workspace, size_t workspace_bytes, void* stream);
/* The sweeps of one sample (history, pc0, pc1) -- or of several samples, up to 12 sweeps -- through the same five launches: the stage's kernels are latency
 * chains on small grids, so sharing launches is worth ~2x on the stage.  Every sweep has its own outputs, image channel
 * group and workspace of `workspace_bytes` (himo_pillar_workspace_bytes of the largest sweep). */
typedef struct himo_sweep {
    int64_t n; const float* d_pts; int pc_stride;
    float transform[16];                                   /* row-major 4x4, sweep -> common frame */
    float* d_xyz_t; int32_t* d_pid; float* d_offsets;      /* per-point outputs as in himo_pillarize */
    float* d_image;                                        /* first of this sweep's 32 image channels */
    void* d_workspace;
} himo_sweep;
int himo_pillarize_multi(int n_sweeps, const himo_sweep* h_sweeps, const float* h_range, const float* h_voxel,
                         const float* h_centre_offset, int grid_w, int grid_h, const float* d_pfn_weight,
                         const float* d_pfn_scale, const float* d_pfn_shift, int image_pitch, size_t workspace_bytes,
                         void* stream);
/* image_split bit 0 (HIMO_IMAGE_SPLIT): each sweep's 32 image channels are written in the split activation format
 * (himo_conv_desc.act_layout; image_pitch a multiple of 16 floats, d_image 64-byte aligned); empty cells are zero either way.
 * bit 1 (HIMO_IMAGE_INCREMENTAL): the image persists between calls and only changes are written.  The last bytes of each
 * sweep's workspace (workspace_bytes a multiple of 16, constant per buffer) hold one bit per cell = "non-empty after the
 * previous call"; an empty cell that was empty then is already zero and is skipped -- two thirds of the zero rows of a
 * 120k-point sweep.  Contract: himo_pillar_occupancy_reset() once after allocating the workspace / image (marks every cell
 * dirty), and nobody else writes this sweep's image channels in between. */
#define HIMO_IMAGE_SPLIT 1
#define HIMO_IMAGE_INCREMENTAL 2
int himo_pillar_occupancy_reset(void* d_workspace, size_t workspace_bytes, int grid_w, int grid_h, void* stream);
int himo_pillarize_multi_ex(int n_sweeps, const himo_sweep* h_sweeps, const float* h_range, const float* h_voxel,
                         const float* h_centre_offset, int grid_w, int grid_h, const float* d_pfn_weight,
                         const float* d_pfn_scale, const float* d_pfn_shift, int image_pitch, size_t workspace_bytes, int image_split,
                         void* stream);

/* NHWC float32 convolution (ksize 3 pad 1 stride 1|2, or ksize 1) / row GEMM on v_mfma_f32_32x32x2_f32
 * with a fused epilogue. */
#define HIMO_EPI_BIAS 0          /* y = acc + bias */
#define HIMO_EPI_BIAS_BN_GELU 1  /* y = gelu((acc + bias) * scale + shift) */
#define HIMO_EPI_BIAS_GELU 2     /* y = gelu(acc + bias) */
#define HIMO_EPI_GRU_ZR 3        /* cols [0,C/2): y = sigmoid(.) (z); cols [C/2,C): aux_out = sigmoid(.) * aux_in (r*h) */
#define HIMO_EPI_GRU_Q 4         /* aux_out = (1 - aux_in) * aux_out + aux_in * tanh(acc + bias)   (aux_in = z, aux_out = h) */
#define HIMO_EPI_BIAS_RELU 5     /* y = max(acc + bias, 0) */
#define HIMO_EPI_RELU_MASK 6     /* y = aux_in > 0 ? acc : 0   (bias ignored: pass NULL) */
typedef struct himo_conv_desc {
    const float* x; int64_t x_batch_stride; int x_pitch;   /* input  (device) */
    const float* w;                                        /* [ksize][ksize][cin][cout] (device) */
    const float* bias; const float* scale; const float* shift;   /* [cout] (device; scale/shift for BN) */
    float* y; int64_t y_batch_stride; int y_pitch;         /* output (device) */
    int n, h, w_in, cin, cout, ksize, stride, epilogue;
    const float* aux_in; int aux_in_pitch;                 /* GRU epilogues, [rows][pitch] */
    float* aux_out; int aux_out_pitch;
    const void* w_packed;                                  /* optional: himo_conv_pack_weights[_ex] output; when set the
                                                              split-precision kernels run (stride 1, and 3x3 stride 2):
                                                              float32-class accuracy at a multiple of the float32-MFMA rate */
    int tile_hint;                                         /* 0 = library heuristic; else (channel tile 64|128) << 4 | (1|2):
                                                              pixel tile 64|128 of the LDS-staged-weights kernel, or
                                                              0x1000 | (4|2): image rows per wave of the weights-from-L2
                                                              kernel (3x3, split precision) -- for callers that time
                                                              the variants */
    int packed_format;                                     /* format of w_packed: 0 = three bf16 planes (float32 range,
                                                              6 matrix instructions per product block), 1 = two fp16
                                                              planes, x = h + l with weights packed x 2^6 (|values| < 65504,
                                                              3 matrix instructions): himo_conv_pack_weights_ex */
    int n_outer;                                           /* 0 | 1: `n` images; > 1: n * n_outer images, image i at
                                                              (i % n) * batch_stride + (i / n) * outer_stride -- the n
                                                              frames of a sample (channel groups) x the samples of a batch */
    int64_t x_outer_stride, y_outer_stride;
    int act_layout;                                        /* 0 = float32 activations; HIMO_ACT_SPLIT_IN | HIMO_ACT_SPLIT_OUT:
                                                              x / y in the split activation format (same addressing, every
                                                              16-channel group of a pixel = [16 fp16 high | 16 fp16 low]):
                                                              3x3 layers with packed_format 1, bias or bias+BN+GELU epilogue,
                                                              channel counts and pitches multiples of 16; SPLIT_IN: stride 1 */
    uint32_t* d_range_seen;                                /* HIMO_ACT_SPLIT_OUT only, may be NULL: the launch stores 1 into this
                                                              device word when one of the outputs it samples (one per lane and
                                                              block) has |y| >= 2^-6.  A two-term fp16 split keeps a value to
                                                              max(2^-25 absolute, 2^-23 relative): a layer whose word stays 0
                                                              after a forward pass sits on the absolute floor and the caller
                                                              should redo the pass in the bf16 split (pipeline.HiMoPipeline does) */
} himo_conv_desc;
#define HIMO_ACT_SPLIT_IN 1
#define HIMO_ACT_SPLIT_OUT 2
#define HIMO_ACT_ACCUMULATE 8   /* alone: y += result instead of y = result -- float32 maps, 3x3 stride 1, packed_format HIMO_PACK_BF16X2,
                                   bias epilogue (the training step's stride-2 data gradients add into the decoder's skip gradient in place);
                                   also ksize 1 (row GEMM) with packed weights of either bf16 split and the bias epilogue, output map below
                                   2^30 elements (the decoder's skip gradient added to the pillar-image gradient the head's scatter wrote) */
#define HIMO_ACT_STUFFED_2X 16  /* alone or with HIMO_ACT_ACCUMULATE, same kernel: d_x is a COMPACT [h / 2][w_in / 2] map that the convolution
                                   reads as its zero-stuffed x2 image (x[2 i][2 j] = map[i][j], zeros elsewhere; h, w_in even = the stuffed
                                   size; x_batch_stride / x_pitch describe the compact map) -- the data gradient of a stride-2 convolution
                                   without materialising the stuffed image; matrix work on the all-zero rows is skipped */
/* (split-precision paths: one image -- h * w_in * x_pitch floats -- must stay below 2 GB: 32-bit byte offsets through a buffer resource;
 *  HIMO_ERR_UNSUPPORTED otherwise) */
int himo_conv2d(const himo_conv_desc* h_desc, void* stream);
/* one-time weight preparation for the split-bf16 path: [k][k][cin][cout] float32 -> three bf16 planes */
size_t himo_conv_packed_weight_bytes(int ksize, int cin, int cout);
int himo_conv_pack_weights(const float* d_w, int ksize, int cin, int cout, void* d_packed, void* stream);
#define HIMO_PACK_BF16X3 0
#define HIMO_PACK_F16X2 1
#define HIMO_PACK_BF16X2 2   /* two bf16 planes, x = h + m (16 significant bits, float32 range, three products per block): 3x3 layers and row GEMMs
                                (1x1) with float32 activation maps and the bias epilogue only -- the data gradients of the mixed-precision
                                training step */
int himo_conv_pack_weights_ex(const float* d_w, int ksize, int cin, int cout, int format, void* d_packed, void* stream);

/* bilinear x2 upsampling, align_corners = true; c channels of every pixel, NHWC with pitches */
int himo_upsample2x(const float* d_x, int x_pitch, int h, int w, int c, float* d_y, int y_pitch, void* stream);
int himo_upsample2x_batch(int n, const float* d_x, int64_t x_batch_stride, int x_pitch, int h, int w, int c, float* d_y,
                          int64_t y_batch_stride, int y_pitch, void* stream);
/* out_split != 0: y in the split activation format (himo_conv_desc.act_layout; c, y_pitch multiples of 16, d_y 64-byte aligned) */
int himo_upsample2x_batch_ex(int n, const float* d_x, int64_t x_batch_stride, int x_pitch, int h, int w, int c, float* d_y,
                             int64_t y_batch_stride, int y_pitch, int out_split, void* stream);

/* A prepared operator list (the static part of a forward pass: fixed buffers, shapes, weights) run from one call.
 * With HIMO_OPS_GRAPH the list is captured once into a hipGraph, keyed by the h_ops address and validated by a hash of
 * the list's bytes (a changed list is re-captured), and replayed with one hipGraphLaunch; himo_ops_release(h_ops)
 * frees the cached graph.  The graph path is
 * skipped while the himo_prof_* profiler is on, and falls back to plain launches if capture is not possible. */
#define HIMO_OP_CONV 0
#define HIMO_OP_UPSAMPLE2X 1
#define HIMO_OPS_GRAPH 0x1u
typedef struct himo_op {
    int kind;
    himo_conv_desc conv;                                   /* HIMO_OP_CONV */
    const float* up_x; int up_x_pitch, up_h, up_w, up_c;   /* HIMO_OP_UPSAMPLE2X: the arguments of himo_upsample2x_batch */
    float* up_y; int up_y_pitch;
    int up_n; int64_t up_x_batch_stride, up_y_batch_stride; /* up_n 0 | 1: one image */
    int up_out_split;                                       /* himo_upsample2x_batch_ex's out_split */
} himo_op;
int himo_run_ops(const himo_op* h_ops, int n_ops, unsigned flags, void* stream);
void himo_ops_release(const himo_op* h_ops);

/* The whole per-point head in one kernel (inference, split-bf16): gather -> `iters` GRU iterations -> Linear(192,32)+GELU
 * -> Linear(32,3) -> flow (pose_flow + residual for in-range points, pose_flow otherwise).  Same inputs and result as
 * himo_head_gather + the GRU row-GEMMs of himo_conv2d + himo_head_final; the hidden state never leaves the chip.
 * Fixed widths: hidden 128 (32 + 32 + 64 gathered channels), x 64.  Packed weights =
 * himo_conv_pack_weights_ex(w, 1, 192, cout, packed_format) of zr [192][256] (z | r), q [192][128], dec1 [192][32].  Specification: himo_amd/seflow/spec.py (reference absent). */
int himo_gru_head(int64_t n, const int32_t* d_pid, const float* d_offsets, const float* d_img0, const float* d_img1,
                  int img_pitch, const float* d_dec, int dec_pitch, const float* d_w_off, const float* d_b_off,
                  const void* d_wzr_packed, const float* d_bzr, const void* d_wq_packed, const float* d_bq,
                  const void* d_w1_packed, const float* d_b1, const float* d_w2, const float* d_b2,
                  const float* d_xyz_t, const float* d_pts, int pc_stride, float* d_flow, int iters, int packed_format,
                  void* stream);
/* The same over several samples in ONE launch (a sample's ~1900 blocks are 2.4 rounds on 256 CUs: per-sample launches
 * each end in a half-empty round).  Up to 16 samples; samples with n == 0 are skipped; weights shared.
 * img_split != 0 (packed_format 1 only): d_img0 / d_img1 rows are in the split activation format.  With packed_format 1
 * an image feature enters the head as its two-term fp16 value h + l in either layout, so both give the same bits. */
typedef struct himo_head_sample {
    int64_t n;
    const int32_t* d_pid; const float* d_offsets; const float* d_img0; const float* d_img1; const float* d_dec;
    const float* d_xyz_t; const float* d_pts; int pc_stride; float* d_flow;
} himo_head_sample;
int himo_gru_head_batch(int n_samples, const himo_head_sample* h_samples, int img_pitch, int dec_pitch,
                        const float* d_w_off, const float* d_b_off,
                        const void* d_wzr_packed, const float* d_bzr, const void* d_wq_packed, const float* d_bq,
                        const void* d_w1_packed, const float* d_b1, const float* d_w2, const float* d_b2,
                        int iters, int packed_format, int img_split, void* stream);
/* The same head with the 64 x-columns FOLDED: x = Linear(3,64)(offset) is an affine function of the point's offset o and
 * constant over the GRU iterations, so x W_x = [o, 1] [W_off W_x ; b_off W_x] is a K = 4 product.  The packed weights are
 * himo_conv_pack_weights_ex of [144][cout] matrices -- rows 0..127 the hidden rows of zr | q | dec1, rows 128..130
 * W_off W_x, row 131 b_off W_x, rows 132..143 zero -- and every GEMM of the head runs 9 slabs of 16 instead of 12.  Results
 * equal himo_gru_head_batch's up to float32 rounding of the folded rows. */
int himo_gru_head_batch_folded(int n_samples, const himo_head_sample* h_samples, int img_pitch, int dec_pitch,
                               const void* d_wzr_packed, const float* d_bzr, const void* d_wq_packed, const float* d_bq,
                               const void* d_w1_packed, const float* d_b1, const float* d_w2, const float* d_b2,
                               int iters, int packed_format, int img_split, void* stream);
/* himo_gru_head_batch (d_w_off / d_b_off given) or himo_gru_head_batch_folded (both NULL: folded [144][cout] weights) with the
 * finite-flow guard of the fp16-split arithmetic built into the output stage: *d_nonfinite (device uint32, may be NULL) is
 * OR-ed with 1 when any flow value this launch writes is NaN or infinite.  The caller clears the word (himo_clear_u32) before
 * the launches it wants covered.  Replaces the host framework's isfinite reduction over the flow buffer. */
int himo_gru_head_batch_guarded(int n_samples, const himo_head_sample* h_samples, int img_pitch, int dec_pitch,
                                const float* d_w_off, const float* d_b_off,
                                const void* d_wzr_packed, const float* d_bzr, const void* d_wq_packed, const float* d_bq,
                                const void* d_w1_packed, const float* d_b1, const float* d_w2, const float* d_b2,
                                int iters, int packed_format, int img_split, uint32_t* d_nonfinite, void* stream);
/* The head's TRAINING forward in one launch (BASELINE config 5; csrc/gruhead.hip with its saves enabled): gather -> `iters` (<= 4) GRU
 * iterations -> Linear(192,32) + GELU -> Linear(32,3), the explicit-x form of himo_gru_head_batch, writing the network's RESIDUAL flow
 * d_res [rows][4] (zeros for dropped points and in column 3) and every tensor the backward pass reads.  All saved buffers have
 * h_saved->rows rows per iteration, a multiple of 64 >= ceil(n / 64) * 64 (a caller may keep capacity beyond the current sweep).  d_w2 is [32][w2_pitch], w2_pitch 3 or 4.  Replaces, in the trainer, himo_head_gather + 2 * iters row
 * products (himo_conv2d) + 2 * iters gate kernels (himo_gru_gates_fwd) + the decoder.  Spec: himo_amd/seflow/spec.py steps 5-6
 * (reference model source absent: PARITY UNPINNED). */
typedef struct himo_head_saved {
    int64_t rows;              /* the row count of every buffer below = the stride between the stacked iterations: a multiple of 64,
                                  >= ceil(n / 64) * 64 */
    float* d_hx;               /* [iters + 1][rows][192]  [h_t | x], t = 0 .. iters */
    float* d_rhx;              /* [iters][rows][192]      [r_t h_t | x] */
    float* d_z; float* d_r; float* d_q;               /* [iters][rows][128] */
    float* d_pre1; float* d_y1;                       /* [rows][32] */
    float* d_res;                                     /* [rows][4] */
} himo_head_saved;
int himo_gru_head_train(int64_t n, const int32_t* d_pid, const float* d_offsets, const float* d_img0, const float* d_img1,
                        int img_pitch, const float* d_dec, int dec_pitch, const float* d_w_off, const float* d_b_off,
                        const void* d_wzr_packed, const float* d_bzr, const void* d_wq_packed, const float* d_bq,
                        const void* d_w1_packed, const float* d_b1, const float* d_w2, int w2_pitch, const float* d_b2,
                        int iters, int packed_format, const himo_head_saved* h_saved, uint32_t* d_nonfinite, void* stream);
/* Backpropagation through the head's GRU iterations in one launch (csrc/gruheadbwd.hip): d_dhx_last [rows][192] = d loss / d [h_T | x]
 * (from the decoder's backward) and the states himo_gru_head_train saved -> d_daq [iters][rows][128], d_dazr [iters][rows][256] (the gate
 * pre-activation gradients: the dz operands of the q / z|r weight-gradient products, zero in the padding rows) and d_dhx0 [rows][192] =
 * d loss / d [h_0 | x].  Every buffer has h_saved->rows rows (per iteration); the sweep writes whole 64-row blocks of the n points.  d_wq_t_packed / d_wzr_t_packed =
 * himo_weight_prepare_batch's flipped copies (ksize 1: the transposes) of q [192][128] and zr [192][256], packed_format
 * HIMO_PACK_BF16X3 or HIMO_PACK_BF16X2.  Replaces gru_bwd1/2/3 + two himo_conv2d row products per iteration. */
int himo_gru_head_backward(int64_t n, int iters, const float* d_dhx_last, const himo_head_saved* h_saved,
                           const void* d_wq_t_packed, const void* d_wzr_t_packed, int packed_format, float* d_daq,
                           float* d_dazr, float* d_dhx0, void* stream);
int himo_clear_u32(uint32_t* d_words, int n, void* stream);

/* per-point head glue: hx[i] = [img0[cell], img1[cell], dec[cell], Linear(3,64)(offset)] (192 floats; zeros for
 * dropped points), rhx[i][128:192] = the same Linear output */
int himo_head_gather(int64_t n, const int32_t* d_pid, const float* d_offsets, const float* d_img0,
                     const float* d_img1, int img_pitch, const float* d_dec, int dec_pitch,
                     const float* d_w_off, const float* d_b_off, float* d_hx, float* d_rhx, int pitch, void* stream);
/* flow[i] = (xyz_t[i] - pts[i]) + (cell >= 0 ? y1[i] @ w2 + b2 : 0): the (N,3) flow INCLUDING ego motion that
 * save_zip.py:117 consumes */
int himo_head_final(int64_t n, const float* d_y1, int y1_pitch, const float* d_w2, const float* d_b2,
                    const int32_t* d_pid, const float* d_xyz_t, const float* d_pts, int pc_stride,
                    float* d_flow, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a11: KNN/Chamfer correspondence + the self-supervised loss terms.  The reference's `seflowppLoss` is in the
 * absent OpenSceneFlow submodule; the only in-tree facts are the four term names and unit weights at
 * assets/slurm/ssl-train-av2.sh:33.  Definitions: himo_amd/csrc/sslloss.hip header (this build's own spec).
 */
/* exact k=1 NN between two sweeps through a uniform BEV grid (cell metres, grid_w x grid_h cells from
 * (x0, y0), at most 2^20 cells; points outside are binned into border cells).  float32 [n][3] in, squared distances + int32
 * reference rows out (idx may be NULL; -1 / +inf when nr == 0).  Ties keep the lowest reference row.  BOTH sets are
 * binned (the query side is searched in cell order, 64 neighbouring queries per block): the workspace is sized for
 * n_max = max(nq, nr). */
size_t himo_nn_grid_workspace_bytes(int64_t n_max, int grid_w, int grid_h);
int himo_nn_grid(int64_t nq, const float* d_q, int64_t nr, const float* d_r, float x0, float y0, float cell,
                 int grid_w, int grid_h, float* d_dist2, int32_t* d_idx, void* d_workspace,
                 size_t workspace_bytes, void* stream);
/* loss[0..3] = chamfer_dis, static_flow_loss, dynamic_chamfer_dis, cluster_based_pc0pc1; loss[4] = their sum
 * (float64, device); d_grad_flow [n0][3] = d loss[4] / d flow.  pc0 must already be in pc1's frame; labels:
 * 0 = static, > 0 = dynamic cluster id (< n_labels).  Synchronises the stream once internally. */
size_t himo_ssl_loss_workspace_bytes(int n0, int n1, int n_labels, int grid_w, int grid_h);
int himo_ssl_loss(int n0, int n1, const float* d_pc0, const float* d_pc1, const float* d_flow,
                  const int32_t* d_label0, const int32_t* d_label1, int n_labels,
                  float grid_x0, float grid_y0, float grid_cell, int grid_w, int grid_h,
                  double* d_loss, float* d_grad_flow, void* d_workspace, size_t workspace_bytes, void* stream);
/* ... with the raw correspondences pc0 -> pc1 supplied: d_raw_dist2 [n0] / d_raw_idx [n0] = himo_nn_grid(n0, d_pc0, n1, d_pc1) on the same
 * grid (they depend on the inputs only: a training step computes them beside its forward pass); both NULL = himo_ssl_loss. */
int himo_ssl_loss_ex(int n0, int n1, const float* d_pc0, const float* d_pc1, const float* d_flow,
                     const int32_t* d_label0, const int32_t* d_label1, int n_labels,
                     float grid_x0, float grid_y0, float grid_cell, int grid_w, int grid_h,
                     const float* d_raw_dist2, const int32_t* d_raw_idx,
                     double* d_loss, float* d_grad_flow, void* d_workspace, size_t workspace_bytes, void* stream);
/* ... and without the internal synchronisation: n_dyn0 / n_dyn1 = the number of points with label > 0 in d_label0 / d_label1, which
 * size the dynamic-subset searches (himo_ssl_loss_ex copies them back in the middle of the call -- in a training step that drains the
 * forward pass out of the queue before the backward pass can be enqueued).  himo_ssl_dyn_sizes counts them into d_counts [2] ahead of
 * time (they depend on the labels only; the caller copies them to pinned host memory beside the forward pass).  Sizes that do not
 * match the labels return every loss term as NaN.  (Reference: the loss of assets/slurm/ssl-train-av2.sh:33, source absent.) */
int himo_ssl_dyn_sizes(int n0, const int32_t* d_label0, int n1, const int32_t* d_label1, int32_t* d_counts, void* stream);
int himo_ssl_loss_presized(int n0, int n1, const float* d_pc0, const float* d_pc1, const float* d_flow,
                           const int32_t* d_label0, const int32_t* d_label1, int n_labels,
                           float grid_x0, float grid_y0, float grid_cell, int grid_w, int grid_h,
                           const float* d_raw_dist2, const int32_t* d_raw_idx, int n_dyn0, int n_dyn1,
                           double* d_loss, float* d_grad_flow, void* d_workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a12: optimisation-based scene flow ("fastnsf", README.md:53).  Reference implementation absent (OpenSceneFlow
 * submodule); specification in himo_amd/fastnsf.py.  The MLP forward / input-gradient products use himo_conv2d
 * (ksize 1) with HIMO_EPI_BIAS_RELU / HIMO_EPI_RELU_MASK; these are the fitting-specific pieces.
 */
/* dW[cin][cout] = X^T dZ and (optionally) db[cout] = column sums of dZ over n rows; cin, cout <= 128 */
size_t himo_wgrad_workspace_bytes(int64_t n_rows);
int himo_linear_wgrad(int64_t n, const float* d_x, int x_pitch, int cin, const float* d_dz, int z_pitch, int cout,
                      float* d_dw, float* d_db, void* d_workspace, size_t workspace_bytes, void* stream);
/* the same for any cin / cout (tiled 128 x 128); flags bit 0: accumulate into d_dw / d_db (BPTT over GRU iterations); bit 1 (2):
 * multiply split-bf16 operands (x = h + m, 16 significant bits, float32 sums) on the 16-bit matrix instructions */
size_t himo_wgrad_workspace_bytes_ex(int64_t n_rows, int cin, int cout);
int himo_linear_wgrad_ex(int64_t n, const float* d_x, int x_pitch, int cin, const float* d_dz, int z_pitch, int cout,
                         float* d_dw, float* d_db, unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream);
int himo_transpose(const float* d_w, int rows, int cols, float* d_wt, void* stream);
/* torch.optim.Adam semantics (no weight decay, no amsgrad); step counts from 1 */
int himo_adam_step(int64_t n, float* d_param, const float* d_grad, float* d_m, float* d_v, float lr, float beta1,
                   float beta2, float eps, int step, void* stream);
/* loss = mean_i [d_a_i <= trunc^2] d_a_i + mean_j [d_b_j <= trunc^2] d_b_j on squared NN distances (from himo_nn_grid),
 * and d loss / d moved (correspondences constant) */
size_t himo_chamfer_trunc_workspace_bytes(int n0, int n1);
int himo_chamfer_trunc(int n0, int n1, const float* d_moved, const float* d_pc1, const float* d_dist_a,
                       const int32_t* d_idx_a, const float* d_dist_b, const int32_t* d_idx_b, float trunc_dist,
                       double* d_loss, float* d_grad_moved, void* d_workspace, size_t workspace_bytes, void* stream);
/* y[i][c] = a[i][c] + b_scale * b[i][c] for c < cols (b may be NULL); zero_tail clears y's columns cols..y_pitch-1 */
int himo_rows_add(int64_t n, int cols, const float* d_a, int a_pitch, const float* d_b, int b_pitch, float b_scale,
                  float* d_y, int y_pitch, int zero_tail, void* stream);
/* out[i][0:3] = R p_i + t with d_transform a DEVICE row-major 4x4 float32 (rounding as seflow/spec.py step 0); the rest
 * of each out row (out_pitch floats) is zeroed */
int himo_rigid_transform(int64_t n, const float* d_pts, int pc_stride, const float* d_transform, float* d_out,
                         int out_pitch, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a11, training side: element-wise pieces of the network's backward pass (csrc/train.hip).  The reference trains through
 * the absent OpenSceneFlow/train.py (assets/slurm/ssl-train-av2.sh:31); conventions in himo_amd/seflow/train.py.
 * Matrix products use himo_conv2d (row GEMM) and himo_linear_wgrad_ex.
 */
/* GRU cell, training forward.  which = 1: pre = [az | ar] (n x 256), hx = [h | x] (n x 192) -> z, r (n x 128),
 * out = [r * h | x].  which = 2: pre = aq (n x 128), z_in, hx -> q (n x 128), out = [(1 - z) h + z q | x]. */
int himo_gru_gates_fwd(int64_t n, int which, const float* d_pre, const float* d_z_in, const float* d_hx,
                       float* d_z, float* d_r, float* d_q, float* d_out, void* stream);
/* GRU cell, backward (three element-wise stages around the two transposed GEMMs; see csrc/train.hip) */
int himo_gru_bwd1(int64_t n, const float* d_dh_next, const float* d_z, const float* d_q, const float* d_hx,
                  float* d_daq, float* d_dz, float* d_dhp, void* stream);
int himo_gru_bwd2(int64_t n, const float* d_d_rhx, const float* d_hx, const float* d_z, const float* d_r,
                  const float* d_dz, float* d_dhp, float* d_dazr, float* d_dx, void* stream);
int himo_gru_bwd3(int64_t n, const float* d_d_hx, const float* d_dhp, float* d_dh, float* d_dx, void* stream);
/* pre = x * scale[c] + shift[c] (scale/shift NULL: identity), y = gelu(pre); and dx = dy * gelu'(pre) * scale[c] */
int himo_affine_gelu_fwd(int64_t rows, int ch, const float* d_x, int x_pitch, const float* d_scale, const float* d_shift,
                         float* d_pre, int pre_pitch, float* d_y, int y_pitch, void* stream);
int himo_affine_gelu_bwd(int64_t rows, int ch, const float* d_dy, int dy_pitch, const float* d_pre, int pre_pitch,
                         const float* d_scale, float* d_dx, int dx_pitch, void* stream);
/* zero the rows of v whose cell id is negative */
int himo_mask_rows(int64_t n, int cols, const int32_t* d_pid, float* d_v, int pitch, void* stream);
/* convolution backward: wf[k][k][cout][cin] = w[k][k][cin][cout] with mirrored taps (the data gradient of a stride-1 "same"
 * convolution is himo_conv2d of dY with wf); zero-stuffing of dY for the stride-2 layers; the adjoint of himo_upsample2x;
 * the 3x3 weight gradient of ONE image as a split-K MFMA product (flags bit 0: accumulate) */
int himo_weight_flip(const float* d_w, int ksize, int cin, int cout, float* d_wf, void* stream);
int himo_zero_stuff2x(int n_img, int h, int w, int c, const float* d_dy, int64_t dy_batch_stride, int dy_pitch,
                      float* d_z, int64_t z_batch_stride, int z_pitch, void* stream);
int himo_upsample2x_bwd(const float* d_dy, int dy_pitch, int h, int w, int c, float* d_dx, int dx_pitch, void* stream);
int himo_add2d(int64_t rows, int cols, const float* d_b, int b_pitch, float* d_y, int y_pitch, void* stream);   /* y += b */
/* column sums of a pitched [n][cout] matrix (bias gradients); workspace >= ceil(cout/128)*ceil(n/256)*512 bytes */
int himo_colsum(int64_t n, const float* d_z, int z_pitch, int cout, float* d_out, unsigned flags, void* d_workspace,
                size_t workspace_bytes, void* stream);
/* backward of the pillar stage; d_pillar_workspace = the workspace himo_pillarize(n, ...) of the same sweep left behind.
 * himo_pfn_backward: d loss / d pfn.weight [9][32] from the gradient of the sweep's 32 image channels (flags bit 0:
 * accumulate).  himo_head_scatter: adjoint of himo_head_gather -- per-point rows [d img0 | d img1 | d dec] (128 columns)
 * summed per pillar into channel groups group0 / group1 of d_db0 (other groups and empty cells zeroed) and d_ddec. */
size_t himo_pfn_backward_workspace_bytes(void);
int himo_pfn_backward(int64_t n, const float* h_voxel, const float* h_centre_offset, int grid_w, int grid_h,
                      const float* d_pfn_weight, const float* d_pfn_scale, const float* d_pfn_shift,
                      const float* d_xyz_t, const void* d_pillar_workspace, const float* d_dimage, int image_pitch,
                      float* d_dweight, unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream);
int himo_head_scatter(int64_t n, int grid_w, int grid_h, const void* d_pillar_workspace, const float* d_dhx, int dhx_pitch,
                      float* d_db0, int b0_pitch, int group0, int group1, int n_groups, float* d_ddec, int dec_pitch,
                      void* stream);

/* Every packed weight copy of a training step in ONE launch.  A job = himo_conv_pack_weights_ex(w, ksize, cin, cout, format) into
 * `packed`; with flip != 0 the copy is of the layer's DATA-GRADIENT weights instead -- taps mirrored and the channel roles swapped,
 * i.e. pack(wf, ksize, cout, cin) with wf[t][co][ci] = w[k*k-1-t][ci][co] (for ksize 1: the transpose).  The job table lives in DEVICE
 * memory (the caller uploads it once; the tensors it names are updated in place by the optimiser); first_block = the running sum
 * of himo_weight_job_blocks over the jobs before it, total_blocks the sum over all.  Spec: himo_amd/seflow/train.py (reference
 * absent: its training loop packs nothing). */
typedef struct himo_weight_job {
    const float* w;           /* [k][k][cin][cout] float32 */
    void* packed;             /* himo_conv_packed_weight_bytes(ksize, cin, cout) bytes (the same for the flipped form) */
    int ksize, cin, cout;
    int format;               /* HIMO_PACK_* */
    int flip;
    int first_block;
} himo_weight_job;
int himo_weight_job_blocks(int ksize, int cin, int cout, int flip);
int himo_weight_prepare_batch(const himo_weight_job* d_jobs, int n_jobs, int total_blocks, void* stream);

/* FastNSF's MLP after an optimiser step, ONE launch for all layers: every W [cin][cout] -> its fp16-split copy (format 1, forward
 * products) and the two-term bf16 copy of its transpose (format 2, input-gradient products); either destination may be NULL. */
int himo_mlp_repack(int n_layers, const float* const* h_w, const int* h_cin, const int* h_cout, void* const* h_fwd_packed,
                    void* const* h_bwd_packed, void* stream);

/* Self-supervised cluster labels (`+ssl_label=seflow_auto`, assets/slurm/ssl-train-av2.sh:32; the reference's generator is in the
 * absent OpenSceneFlow submodule: PARITY UNPINNED, specification himo_amd/seflow/ssl_label.py, oracle sklearn.cluster.DBSCAN):
 * DBSCAN(eps, min_pts) over 3-D points on the GPU (csrc/dbscan.hip).  d_xyz [n][pitch >= 3] float32; d_skip [n] bytes or NULL
 * (non-zero: the point takes no part); BEV cell grid of `cell` >= eps metres from (x0, y0); d_labels [n] int32: 0 = noise / skipped,
 * 1 .. K = clusters in the order of their lowest point index (a border point joins the neighbouring cluster of lowest such index):
 * a pure function of the input.  d_n_clusters: K (or NULL). */
size_t himo_dbscan_workspace_bytes(int n, int grid_w, int grid_h);
int himo_dbscan(int n, const float* d_xyz, int pitch, const unsigned char* d_skip, float eps, int min_pts, float x0, float y0, float cell,
                int grid_w, int grid_h, int32_t* d_labels, int32_t* d_n_clusters, void* d_workspace, size_t workspace_bytes, void* stream);

/* FastNSF (README.md:53 `model=fastnsf`; specification himo_amd/fastnsf.py, PARITY UNPINNED) -- one optimiser iteration of the
 * coordinate MLP as THREE launches (csrc/nsffused.hip):
 *   himo_nsf_forward   the MLP over all points (activations spilled as two-term bf16 matrix fragments + ReLU mask bits for the backward
 *                      pass) and, when d_dout is given, the distance-transform objective of the moved points: d loss / d out
 *                      WITHOUT the 1 / (points in the volume) factor, per-tile loss and count sums, and the last layer's own
 *                      gradients per tile (into d_spill's tail);
 *   himo_nsf_backward  the whole chain of input gradients AND the gradient of every parameter, summed over each block of 256
 *                      points: d_partial [himo_nsf_backward_blocks(n)][partial_stride], each row laid out like the flat parameter
 *                      vector (h_off_w[i] / h_off_b[i]: float offsets of layer i's W [cin][cout] / b; i = 0 first (W [4][128]),
 *                      1 .. n_hidden - 1 hidden, n_hidden last (W [128][4]));
 *   himo_nsf_update    fixed-order sum of the block rows / (points in the volume) -> d_grad, one Adam step on d_param / d_m / d_v,
 *                      the packed copies of every hidden W (himo_mlp_repack's two formats), the loss and the count.
 * EVERY [n][.] buffer (d_x0, d_out, d_dout) holds himo_nsf_padded_rows(n) rows (whole blocks of 4 x 64 points, no bounds checks);
 * d_x0's padding rows must be zero.  d_spill: himo_nsf_spill_bytes(n, n_hidden).  2 <= n_hidden <= 12. */
int64_t himo_nsf_padded_rows(int64_t n);
size_t himo_nsf_spill_bytes(int64_t n, int n_hidden);
int himo_nsf_backward_blocks(int64_t n);
int himo_nsf_forward(int64_t n, const float* d_x0, int n_hidden, const float* d_w_first, const float* d_b_first,
                     const void* const* h_w_hidden_packed, const float* const* h_b_hidden, const float* d_w_last, const float* d_b_last,
                     void* d_spill, float* d_out, const float* h_origin, float cell, const int* h_dims, int window, const void* d_volume,
                     float trunc_dist, float* d_dout, double* d_loss_partial, int* d_count_partial, void* stream);
int himo_nsf_backward(int64_t n, const float* d_x0, const float* d_dout, int n_hidden, const void* const* h_wT_hidden_packed,
                      const float* d_w_last, const void* d_spill, const int* h_off_w, const int* h_off_b, int64_t partial_stride,
                      float* d_partial, void* stream);
int himo_nsf_update(int total, int n_partials, int64_t partial_stride, const float* d_partial, int n_fwd_blocks,
                    const double* d_loss_partial, const int* d_count_partial, float* d_param, float* d_grad, float* d_m, float* d_v,
                    float lr, float beta1, float beta2, float eps, int step, int n_hidden, const int* h_off_w, void* const* h_fwd_packed,
                    void* const* h_bwd_packed, double* d_loss, int* d_count, void* stream);

/* FastNSF's coordinate MLP (3 -> 128 x n_hidden, ReLU -> 3; himo_amd/fastnsf.py) over all n points as ONE kernel per direction
 * (csrc/mlpfused.hip): the forward pass writes the post-ReLU activations h_H[k] [n][128] and d_out [n][4]; the backward pass turns
 * d_dout [n][4] into the masked gradients h_dZ[k] [n][128] at every hidden layer's output.  Hidden layer k = 1 .. n_hidden - 1 is
 * given by himo_mlp_repack's copies of W_k (forward: fp16 split) / W_k^T (backward: two-term bf16); entry 0 of those arrays is
 * ignored.  First layer W [4][128] and last layer W [128][4] are float32 (padded with zeros).  Replaces 17 row-GEMM launches.
 * EVERY [n][.] buffer must hold ceil(n / 64) * 64 rows (whole 64-row blocks, no bounds checks; pad d_x0 / d_dout with zeros). */
int himo_mlp_forward_fused(int64_t n, const float* d_x0, int n_hidden, const float* d_w_first, const float* d_b_first,
                           const void* const* h_w_hidden_packed, const float* const* h_b_hidden, const float* d_w_last,
                           const float* d_b_last, float* const* h_H, float* d_out, void* stream);
int himo_mlp_backward_fused(int64_t n, const float* d_dout, int n_hidden, const void* const* h_wT_hidden_packed,
                            const float* d_w_last, float* const* h_H, float* const* h_dZ, void* stream);
/* the same, also yielding the hidden layers' bias gradients h_db[k] [128] (column sums of dZ_k, taken while dZ_k passes through the kernel:
 * one reduction launch for all layers instead of a column-sum pass per layer); the padding rows of d_dout must be zero */
size_t himo_mlp_bias_workspace_bytes(int64_t n, int n_hidden);
int himo_mlp_backward_fused_bias(int64_t n, const float* d_dout, int n_hidden, const void* const* h_wT_hidden_packed,
                                 const float* d_w_last, float* const* h_H, float* const* h_dZ, float* const* h_db,
                                 void* d_workspace, size_t workspace_bytes, void* stream);

/* ---- FastNSF's distance-transform objective (BASELINE config 4, `model=fastnsf`, README.md:53; implementation absent from the
 * reference tree: PARITY UNPINNED, specification in csrc/dtloss.hip / himo_amd/fastnsf.py).  himo_dt_build: ONCE per sweep pair, the
 * target sweep d_pc1 [n1][3] -> a volume of squared cell distances to the nearest occupied cell (uint16, x fastest, exact up to
 * `window` cells; h_dims = {nx, ny, nz}); d_volume holds himo_dt_volume_bytes() (two volumes: the passes ping-pong, the result is
 * the first).  himo_dt_loss: per optimiser iteration, loss = (1/n) sum [D <= trunc] D(moved_i) by trilinear lookup and its gradient
 * with respect to the moved points [n][3] -- what replaces the two exact NN searches + himo_chamfer_trunc of the NN objective. */
size_t himo_dt_volume_bytes(const int* h_dims);
int himo_dt_build(int n1, const float* d_pc1, const float* h_origin, float cell, const int* h_dims, int window,
                  void* d_volume, size_t volume_bytes, void* stream);
size_t himo_dt_loss_workspace_bytes(int n);
int himo_dt_loss(int n, const float* d_moved, const float* h_origin, float cell, const int* h_dims, int window,
                 const void* d_volume, float trunc_dist, double* d_loss, float* d_grad_moved, void* d_workspace,
                 size_t workspace_bytes, void* stream);

/* ---- BatchNorm in TRAINING mode (BASELINE config 5).  Replaces: the torch.nn.BatchNorm layers of the model the reference's
 * training job builds from scratch (assets/slurm/ssl-train-av2.sh:31-34: no checkpoint=, 12 epochs, batch_size=8; the model
 * source, OpenSceneFlow/, is absent -- PARITY UNPINNED, semantics = torch's: biased variance normalises, the unbiased one
 * feeds the running estimate, momentum 0.1).  Maps are NHWC float32 views [n_img][rows][ch]: element (i, r, c) at
 * p + i * img_stride + r * pitch + c (16-byte aligned bases, strides multiples of 4 floats, ch % 4 == 0).
 * himo_bn_train_fwd: batch mean / invstd over all n_img * rows rows -> d_mean, d_invstd [ch] (kept for the backward pass);
 *   d_xhat = (x - mean) * invstd (may alias d_x; NULL: not written, see himo_bn_train_bwd_x); d_y = gelu(gamma * xhat + beta); running statistics updated in place when
 *   given.  himo_bn_train_bwd: d_dy = d loss / d y -> d_dx = d loss / d x (the three-term BatchNorm gradient through the
 *   exact-erf GELU; may alias d_dy), d_dgamma / d_dbeta [ch] (flags bit 0: accumulate).  Every reduction is a fixed tree.
 * himo_bn_fold: eval-mode constants scale = gamma / sqrt(var + eps), shift = beta - mean * scale from the running statistics. */
size_t himo_bn_workspace_bytes(int64_t total_rows, int ch);
int himo_bn_train_fwd(int n_img, int64_t rows, int ch, const float* d_x, int64_t x_img_stride, int x_pitch,
                      const float* d_gamma, const float* d_beta, float eps, float momentum, float* d_running_mean,
                      float* d_running_var, float* d_mean, float* d_invstd, float* d_xhat, int64_t xhat_img_stride, int xhat_pitch,
                      float* d_y, int64_t y_img_stride, int y_pitch, void* d_workspace, size_t workspace_bytes, void* stream);
int himo_bn_train_bwd(int n_img, int64_t rows, int ch, const float* d_dy, int64_t dy_img_stride, int dy_pitch,
                      const float* d_xhat, int64_t xhat_img_stride, int xhat_pitch, const float* d_gamma, const float* d_beta,
                      const float* d_invstd, float* d_dx, int64_t dx_img_stride, int dx_pitch, float* d_dgamma, float* d_dbeta,
                      unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream);
/* himo_bn_train_bwd from the layer's INPUT x (+ d_mean) instead of xhat: himo_bn_train_fwd may then be called with d_xhat = NULL and
 * writes 4 B per activation less; xhat is re-formed as the forward pass formed it (same results bit for bit). */
int himo_bn_train_bwd_x(int n_img, int64_t rows, int ch, const float* d_dy, int64_t dy_img_stride, int dy_pitch,
                        const float* d_x, int64_t x_img_stride, int x_pitch, const float* d_gamma, const float* d_beta,
                        const float* d_mean, const float* d_invstd, float* d_dx, int64_t dx_img_stride, int dx_pitch,
                        float* d_dgamma, float* d_dbeta, unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream);
int himo_bn_fold(int ch, const float* d_gamma, const float* d_beta, const float* d_mean, const float* d_var, float eps,
                 float* d_scale, float* d_shift, void* stream);
/* The same for the pillar feature net's BatchNorm1d (statistics over the in-range points of ONE sweep; y = feats W):
 * himo_pfn_bn_stats reads the cell lists himo_pillarize* left in d_pillar_workspace and yields the sweep's constants
 *   d_scale = gamma invstd, d_shift = beta - mean gamma invstd [32] (+ d_mean, d_invstd for the backward pass; running
 *   statistics updated in place when given);
 * himo_pillar_features_multi re-runs ONLY the feature kernel of himo_pillarize_multi_ex with per-sweep constants
 *   (d_scale / d_shift: [n_sweeps][32]);
 * himo_pfn_backward_bn = himo_pfn_backward through the batch statistics, also yielding d gamma / d beta [32]. */
size_t himo_pfn_bn_workspace_bytes(void);
/* the same for the n_sweeps (<= 12) sweeps of a sample in one call: host arrays of per-sweep point counts, d_xyz_t and pillar
 * workspaces (and, backward, image-gradient bases); d_scale / d_shift / d_mean / d_invstd are [n_sweeps][32]; the running statistics
 * (forward) and d_dweight / d_dgamma / d_dbeta (backward) are updated sweep after sweep, so the results have the bits of n_sweeps single
 * calls.  Workspace: n_sweeps * himo_pfn_bn_workspace_bytes(). */
/* ... and for the sweeps of a per-process BATCH (the reference launcher's batch_size=8 on one process, assets/slurm/ssl-train-av2.sh:32-34):
 * sweep i belongs to GROUP i % n_groups, and a group shares ONE set of statistics -- with the sweeps ordered sample-major, frame-minor
 * and n_groups = the frames per sample, group f is frame slot f of every sample: what one call of the pillar net on a batch of sweeps
 * normalises over (torch.nn.BatchNorm1d on the concatenated points).  d_scale / d_shift / d_mean / d_invstd are [n_groups][32]; up to 16
 * sweeps per group; the _multi forms are these with every sweep its own group.  Workspace: n_sweeps * himo_pfn_bn_workspace_bytes(). */
int himo_pfn_bn_stats_groups(int n_sweeps, int n_groups, const int64_t* h_n, const float* const* h_xyz_t,
                             const void* const* h_pillar_workspace, const float* h_voxel, const float* h_centre_offset, int grid_w,
                             int grid_h, const float* d_pfn_weight, const float* d_gamma, const float* d_beta, float eps, float momentum,
                             float* d_running_mean, float* d_running_var, float* d_scale, float* d_shift, float* d_mean, float* d_invstd,
                             void* d_workspace, size_t workspace_bytes, void* stream);
int himo_pfn_backward_bn_groups(int n_sweeps, int n_groups, const int64_t* h_n, const float* const* h_xyz_t,
                                const void* const* h_pillar_workspace, const float* const* h_dimage, int image_pitch, const float* h_voxel,
                                const float* h_centre_offset, int grid_w, int grid_h, const float* d_pfn_weight, const float* d_scale,
                                const float* d_shift, const float* d_mean, const float* d_invstd, float* d_dweight, float* d_dgamma,
                                float* d_dbeta, unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream);
int himo_pfn_bn_stats_multi(int n_sweeps, const int64_t* h_n, const float* const* h_xyz_t, const void* const* h_pillar_workspace,
                            const float* h_voxel, const float* h_centre_offset, int grid_w, int grid_h,
                            const float* d_pfn_weight, const float* d_gamma, const float* d_beta, float eps, float momentum,
                            float* d_running_mean, float* d_running_var, float* d_scale, float* d_shift, float* d_mean,
                            float* d_invstd, void* d_workspace, size_t workspace_bytes, void* stream);
int himo_pfn_backward_bn_multi(int n_sweeps, const int64_t* h_n, const float* const* h_xyz_t, const void* const* h_pillar_workspace,
                               const float* const* h_dimage, int image_pitch, const float* h_voxel, const float* h_centre_offset,
                               int grid_w, int grid_h, const float* d_pfn_weight, const float* d_scale, const float* d_shift,
                               const float* d_mean, const float* d_invstd, float* d_dweight, float* d_dgamma, float* d_dbeta,
                               unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream);
int himo_pfn_bn_stats(int64_t n, const float* h_voxel, const float* h_centre_offset, int grid_w, int grid_h,
                      const float* d_pfn_weight, const float* d_xyz_t, const void* d_pillar_workspace, const float* d_gamma,
                      const float* d_beta, float eps, float momentum, float* d_running_mean, float* d_running_var,
                      float* d_scale, float* d_shift, float* d_mean, float* d_invstd, void* d_workspace, size_t workspace_bytes,
                      void* stream);
int himo_pillar_features_multi(int n_sweeps, const himo_sweep* h_sweeps, const float* h_range, const float* h_voxel,
                               const float* h_centre_offset, int grid_w, int grid_h, const float* d_pfn_weight,
                               const float* d_scale, const float* d_shift, int image_pitch, size_t workspace_bytes,
                               int image_split, void* stream);
int himo_pfn_backward_bn(int64_t n, const float* h_voxel, const float* h_centre_offset, int grid_w, int grid_h,
                         const float* d_pfn_weight, const float* d_scale, const float* d_shift, const float* d_mean,
                         const float* d_invstd, const float* d_xyz_t, const void* d_pillar_workspace, const float* d_dimage,
                         int image_pitch, float* d_dweight, float* d_dgamma, float* d_dbeta, unsigned flags, void* d_workspace,
                         size_t workspace_bytes, void* stream);
/* the 3x3 weight gradient over a batch of images, LDS-tiled (dY tile + X halo staged once, all 9 taps read them);
 * h, w = INPUT image size.  stride 1: h even, w % 32 == 0, cin % 64 == 0; stride 2: h even, w % 64 == 0, cin == 32 or
 * cin % 64 == 0; cout % 64 == 0 -- HIMO_ERR_UNSUPPORTED otherwise (workspace_bytes returns 0).
 * flags bit 0: accumulate into d_dw; bit 1 (2): stride-1 layers multiply split-bf16 operands (x = h + m, 16 significant bits,
 * float32 accumulation) on the 16-bit matrix instructions instead of float32 ones; bit 2 (4): the launch runs BESIDE another stream's
 * kernels -- one block per CU instead of two, so that the other stream's blocks find registers and LDS on every CU (results differ
 * from the two-block launch only in the summation order of the pixel chunks) */
size_t himo_conv_wgrad_batch_workspace_bytes(int n_img, int h, int w, int cin, int cout, int stride);
int himo_conv3x3_wgrad_batch(int n_img, const float* d_x, int64_t x_batch_stride, int x_pitch, int h, int w, int cin,
                             const float* d_dy, int64_t dy_batch_stride, int dy_pitch, int cout, int stride, float* d_dw,
                             unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream);
/* the same with the layer's BIAS gradient d_db [cout] (column sums of dY) from the same pass over dY: stride 1 with flags bit 1 only
 * (HIMO_ERR_UNSUPPORTED otherwise; himo_colsum is the stand-alone form); flags bit 0 accumulates into both.  Same workspace. */
int himo_conv3x3_wgrad_batch_bias(int n_img, const float* d_x, int64_t x_batch_stride, int x_pitch, int h, int w, int cin,
                                  const float* d_dy, int64_t dy_batch_stride, int dy_pitch, int cout, int stride, float* d_dw,
                                  float* d_db, unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream);
size_t himo_conv_wgrad_workspace_bytes(int ho, int wo, int cin, int cout);
int himo_conv3x3_wgrad(const float* d_x, int x_pitch, int h, int w, int cin, const float* d_dy, int dy_pitch, int cout,
                       int stride, float* d_dw, unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a9 helper (host memory, no GPU): LZ4-frame decoder for the compressed buffers of Feather V2 files as pandas /
 * pyarrow write them (save_zip.py:81 `to_feather`).  Returns the decompressed size, or -1. */
int64_t himo_lz4_frame_decompress(const void* h_src, int64_t n, void* h_dst, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* HIMO_AMD_H */
