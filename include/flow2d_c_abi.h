/*
 * flow2d_c_abi.h -- the C boundary of the MI355X (gfx950) optical-flow hot path.
 *
 * What it replaces.  axruff/cuda-flow2d binds its host code to device code purely by C symbol
 * name: every operator class does cuModuleLoad("<exe>/kernels/<op>.ptx") + cuModuleGetFunction
 * and launches through cuLaunchKernel with a void* argument array (e.g.
 * src/cuda_operations/2d/cuda_operation_add_2d.cpp:50-55,96-103), and moves memory through the
 * CUDA driver API (src/utils/cuda_utils.cpp:26-105, src/optical_flow/optical_flow_2d.cpp:84-140).
 * This header is the drop-in for exactly that surface: device/context set-up, pitched plane
 * memory, one launcher per reference kernel, the solver's fixed-point loop, events.  Plain
 * pointers and sizes only; every function returns a flow2d_status (0 = ok) instead of printing.
 *
 * Conventions
 *  - A "plane" is a row-major fp32 image inside a pitched container: element (x, y) lives at
 *    base + y * pitch_bytes + 4 * x.  pitch_bytes is what DataSize3::pitch holds in the reference
 *    (src/data_types/data_structs.h:31-35); it must be a multiple of 16.  A pyramid level of
 *    w x h pixels occupies the top-left corner of a full-resolution container, as in the
 *    reference (`container_size` module constant, IND(X,Y) macro of every .cu file).
 *  - All launches go to the context's stream and return without synchronising.
 *  - Device pointers are raw HIP device addresses (void* / float*), caller-owned.
 *  - Not thread-safe per context; use one context per host thread / per GPU.
 *  - Arithmetic is IEEE fp32 in the reference's operation order with no fused multiply-add
 *    (kernels are built -ffp-contract=off), so results are bit-identical to the CPU oracle.
 */
#ifndef FLOW2D_C_ABI_H_
#define FLOW2D_C_ABI_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLOW2D_API __attribute__((visibility("default")))

#define FLOW2D_ABI_VERSION 1

typedef enum flow2d_status {
    FLOW2D_OK = 0,
    FLOW2D_ERR_INVALID_ARGUMENT = 1, /* null pointer, bad size, in == out, misaligned pitch */
    FLOW2D_ERR_NO_DEVICE = 2,        /* no usable HIP device / bad ordinal */
    FLOW2D_ERR_DEVICE = 3,           /* a HIP runtime call failed; see flow2d_last_error() */
    FLOW2D_ERR_OUT_OF_MEMORY = 4,
    FLOW2D_ERR_UNSUPPORTED = 5       /* e.g. median window not in {3,5,7}, Gaussian longer than 51 taps */
} flow2d_status;

/* Data term.  GREY, GRADIENT and LOG_DERIVATIVES are `enum class DataConstancy` Grey, Gradient and LogDerivatives
 * of the reference (src/data_types/data_structs.h:27), which selects solve_2d / solve_2d_grad / solve_2d_log with
 * them (src/cuda_operations/2d/cuda_operation_solve_2d.cpp:65-82); the numeric values here are the C-ABI's own
 * (the host layer maps the enum).  GRADIENT_UNTILED is an extra, opt-in mode with no counterpart in the reference: the gradient-constancy
 * tensor of solve_2d_grad with its second derivatives taken over the true neighbours (reflected at the image
 * border) instead of being cut at the reference's 16x8 launch tiles.  Results differ from
 * GRADIENT at tile edges by design. */
typedef enum flow2d_constancy {
    FLOW2D_CONSTANCY_GREY = 0,
    FLOW2D_CONSTANCY_GRADIENT = 1,
    FLOW2D_CONSTANCY_GRADIENT_UNTILED = 2,
    FLOW2D_CONSTANCY_LOG_DERIVATIVES = 3
} flow2d_constancy;

typedef struct flow2d_context flow2d_context; /* opaque: device ordinal + stream + scratch */

/* ---- library / errors ------------------------------------------------------------------ */
FLOW2D_API int flow2d_abi_version(void);
FLOW2D_API const char* flow2d_status_string(int status);
/* Text of the last failing HIP call on this thread ("" if none). */
FLOW2D_API const char* flow2d_last_error(void);

/* ---- device / context  (replaces InitCudaContextWithFirstAvailableDevice,
 *      src/utils/cuda_utils.cpp:26-62, and cuCtxDestroy at src/main.cpp:226) ----------------- */
FLOW2D_API int flow2d_device_count(int* count);
/* Hardware queues the HIP runtime deals this process's streams onto (GPU_MAX_HW_QUEUES; the runtime reads it once, at
 * its first call): the batched host path runs four lanes (streams) side by side and a lane that shares a queue waits
 * behind its neighbour.  flow2d_hw_queues returns the value the environment holds now (4 = the runtime's default when
 * unset or unparsable).  flow2d_request_hw_queues(n) asks for n queues: it sets the variable when the caller has not
 * and the process has not started the HIP runtime yet (FLOW2D_OK; also when the environment already grants n), and
 * reports FLOW2D_ERR_UNSUPPORTED -- with the reason in flow2d_last_error() -- when it came too late or the caller's own
 * value is smaller; nothing is changed then.  The host layer calls it before its first HIP call (InitDeviceContext).
 * No reference counterpart (one context, the NULL stream: src/utils/cuda_utils.cpp:43). */
FLOW2D_API int flow2d_hw_queues(void);
FLOW2D_API int flow2d_request_hw_queues(int queues);
FLOW2D_API int flow2d_context_create(int device_ordinal, flow2d_context** out_ctx);
/* Same, but launches on an existing hipStream_t owned by the caller (e.g. a PyTorch stream). */
FLOW2D_API int flow2d_context_create_on_stream(int device_ordinal, void* hip_stream, flow2d_context** out_ctx);
FLOW2D_API int flow2d_context_destroy(flow2d_context* ctx);
FLOW2D_API int flow2d_context_device(const flow2d_context* ctx, int* device_ordinal);
FLOW2D_API int flow2d_context_stream(const flow2d_context* ctx, void** hip_stream);
FLOW2D_API int flow2d_synchronize(flow2d_context* ctx); /* cuStreamSynchronize(NULL), cuda_operation_solve_2d.cpp:291 */
/* Lock-step batches of independent pairs (no reference counterpart): after flow2d_context_set_batch(ctx, count,
 * stride_bytes) every launcher, memset and device copy of this context acts on `count` instances of its planes,
 * instance b at plane pointer + b * stride_bytes (pairs stored one below the other in tall containers; stride a
 * multiple of 16).  One launch then holds the work of all instances (grid.z), so a level of a mid-size frame fills the
 * chip and the launch-bound coarse levels cost one launch per batch instead of one per pair.  count = 1 switches it
 * off (the default).  Results per instance are those of the unbatched call.  At most FLOW2D_BATCH_MAX instances (the
 * two-plane launches put 2 x count into grid.z). */
#define FLOW2D_BATCH_MAX 32767
FLOW2D_API int flow2d_context_set_batch(flow2d_context* ctx, size_t count, size_t stride_bytes);
/* A hint, not a mode: lone != 0 says that the caller runs this context's launches alone on the device -- one pair after the
 * other, no other lane of the same job beside them (OpticalFlow2D::lone sets it; no reference counterpart, the reference has one
 * context and blocks after every launch, cuda_operation_solve_2d.cpp:291).  Strip launches of the solver that leave half the
 * device's wave slots empty then take the build of the strip kernel with packed arithmetic (a wave alone on its SIMD has nothing
 * to share issue turns with; the same IEEE operations, the same bits).  Default 0. */
FLOW2D_API int flow2d_context_set_lone(flow2d_context* ctx, int lone);
/* cuMemGetInfo, optical_flow_2d.cpp:91 */
FLOW2D_API int flow2d_mem_info(flow2d_context* ctx, size_t* free_bytes, size_t* total_bytes);
FLOW2D_API int flow2d_device_name(flow2d_context* ctx, char* buf, size_t buf_len);

/* ---- pitched plane memory  (replaces cuMemAllocPitch / cuMemFree / cuMemsetD2D8 / cuMemcpy2D /
 *      cuMemcpyDtoD: optical_flow_2d.cpp:114-133,309-312,574-577; cuda_utils.cpp:66-105;
 *      cuda_operation_median_2d.cpp:100-104) -------------------------------------------------- */
/* Row pitch this library uses for a plane of `width` floats (multiple of 256 bytes). */
FLOW2D_API size_t flow2d_plane_pitch_bytes(size_t width);
FLOW2D_API int flow2d_plane_alloc(flow2d_context* ctx, size_t width, size_t height, void** out_dev_ptr,
                                  size_t* out_pitch_bytes);
FLOW2D_API int flow2d_plane_free(flow2d_context* ctx, void* dev_ptr);
FLOW2D_API int flow2d_memset_2d(flow2d_context* ctx, void* dev_ptr, size_t pitch_bytes, int byte_value,
                                size_t width_bytes, size_t height);
FLOW2D_API int flow2d_copy_h2d_2d(flow2d_context* ctx, void* dst_dev, size_t dst_pitch_bytes, const void* src_host,
                                  size_t src_pitch_bytes, size_t width_bytes, size_t height);
FLOW2D_API int flow2d_copy_d2h_2d(flow2d_context* ctx, void* dst_host, size_t dst_pitch_bytes, const void* src_dev,
                                  size_t src_pitch_bytes, size_t width_bytes, size_t height);
FLOW2D_API int flow2d_copy_d2d(flow2d_context* ctx, void* dst_dev, const void* src_dev, size_t bytes);

/* `count` (at most FLOW2D_COPY_PLANES_MAX) independent planes of one geometry copied by ONE launch on the context's stream:
 * plane i from src_planes[i] to dst_planes[i] (width floats x height rows, both of pitch_bytes).  This is how a lock-step
 * group gathers pairs that live in containers of their own into its tall staging containers and hands the flows back
 * (OpticalFlow2D::ComputeFlowGroupDevice): two launches per group instead of 4 x count cuMemcpyDtoD calls.  Not subject
 * to flow2d_context_set_batch (the tables name every plane). */
#define FLOW2D_COPY_PLANES_MAX 64
FLOW2D_API int flow2d_copy_planes(flow2d_context* ctx, size_t count, const void* const* src_planes,
                                  void* const* dst_planes, size_t pitch_bytes, size_t width, size_t height);

/* ---- page-locked host memory  (replaces cuMemAllocHost / cuMemFreeHost, the reference's ALLOCATE_PINNED_MEMORY
 *      option: src/data_types/data2d.cpp:34,60-61,80-82) --------------------------------------------------------
 * flow2d_copy_h2d_2d / flow2d_copy_d2h_2d are asynchronous on the context's stream; from and to pageable memory the
 * runtime stages them through its own bounce buffers and the call returns when the bytes have left the caller's
 * buffer (about 20 GB/s).  From and to memory allocated here the copy is one DMA transfer at the PCIe rate, returns at
 * once and overlaps kernels of other streams: the host buffer must stay untouched until the stream has passed the
 * copy (flow2d_synchronize, or an event recorded behind it).  ctx may be NULL (the calling thread's current device). */
FLOW2D_API int flow2d_host_alloc(flow2d_context* ctx, size_t bytes, void** out_host_ptr);
FLOW2D_API int flow2d_host_free(flow2d_context* ctx, void* host_ptr);

/* ---- events  (replaces cuEventCreate/Record/Synchronize/ElapsedTime/Destroy,
 *      optical_flow_2d.cpp:173-179,548-557; cuda_operation_solve_2d.cpp:214-220,302-313) ------- */
FLOW2D_API int flow2d_event_create(flow2d_context* ctx, void** out_event);
FLOW2D_API int flow2d_event_record(flow2d_context* ctx, void* event);
FLOW2D_API int flow2d_event_synchronize(flow2d_context* ctx, void* event);
FLOW2D_API int flow2d_event_elapsed_ms(flow2d_context* ctx, void* start_event, void* stop_event, float* out_ms);
FLOW2D_API int flow2d_event_destroy(flow2d_context* ctx, void* event);
/* Everything queued on `ctx` after this call waits until the work recorded into `event` (by flow2d_event_record on any
 * context of the same device) has finished; the host does not wait.  An event never recorded counts as finished.  This
 * is what chains work on several contexts -- copy streams, compute streams -- into a pipeline (no reference
 * counterpart: one NULL stream, host-synchronous copies, src/utils/cuda_utils.cpp:66-105). */
FLOW2D_API int flow2d_stream_wait_event(flow2d_context* ctx, void* event);

/* ---- stream capture (no counterpart in the reference, which launches eagerly and blocks after every
 *      sweep).  Everything queued on the context between begin and end is recorded into a HIP graph
 *      instead of being executed; flow2d_graph_launch replays it with one host call.  Used by the host
 *      layer to replay a whole pyramid (several hundred launches) for repeated pairs of one size.
 *      Only stream-ordered calls are legal while capturing (no alloc/free/synchronise/event query). --- */
FLOW2D_API int flow2d_capture_begin(flow2d_context* ctx);
FLOW2D_API int flow2d_capture_end(flow2d_context* ctx, void** out_graph_exec);
FLOW2D_API int flow2d_graph_launch(flow2d_context* ctx, void* graph_exec);
FLOW2D_API int flow2d_graph_destroy(flow2d_context* ctx, void* graph_exec);

/* ---- kernel launchers: one per `extern "C" __global__` symbol of src/kernels/ -------------- */

/* add_2d (src/kernels/add_2d.cu:33-46): operand_0 += operand_1 on w x h. */
FLOW2D_API int flow2d_add_2d(flow2d_context* ctx, float* operand_0, const float* operand_1, size_t width,
                             size_t height, size_t pitch_bytes);

/* Host-side Gaussian taps, CudaOperationConvolution2D::ComputeGaussianKernel(sigma, 3, 1.0)
 * (src/cuda_operations/2d/cuda_operation_convolution_2d.cpp:83-112).  taps must hold 51 floats
 * (MAX_KERNEL_LENGTH, convolution_2d.cu:49); writes 2*radius+1 of them. */
FLOW2D_API int flow2d_gaussian_kernel(float sigma, float* taps, int* out_radius);

/* convolutionRowsKernel / convolutionColumnsKernel (src/kernels/convolution_2d.cu:74-168,181-261).
 * `taps` is a HOST array of 2*radius+1 floats (the reference uploads it to the module constant
 * c_Kernel, cuda_operation_convolution_2d.cpp:163-164).  Zero padding outside the image. */
FLOW2D_API int flow2d_convolution_rows(flow2d_context* ctx, float* dst, const float* src, size_t width, size_t height,
                                       size_t pitch_bytes, const float* taps, int radius);
FLOW2D_API int flow2d_convolution_columns(flow2d_context* ctx, float* dst, const float* src, size_t width,
                                          size_t height, size_t pitch_bytes, const float* taps, int radius);

/* Both passes of the separable Gaussian in one launch (rows pass into LDS, columns pass out of it): the
 * result equals flow2d_convolution_rows into a temp followed by flow2d_convolution_columns bit for bit,
 * with half the DRAM traffic.  Replaces the pair of launches of CudaOperationConvolution2D::Execute
 * (cuda_operation_convolution_2d.cpp:169-175); no temp plane needed. */
FLOW2D_API int flow2d_gaussian_blur(flow2d_context* ctx, float* dst, const float* src, size_t width, size_t height,
                                    size_t pitch_bytes, const float* taps, int radius);

/* median_2d (src/kernels/median_2d.cu:87-299): `window` is the window width (3, 5 or 7; the
 * reference calls it "radius"), mirror borders. */
FLOW2D_API int flow2d_median_2d(flow2d_context* ctx, const float* input, size_t width, size_t height,
                                size_t pitch_bytes, size_t window, float* output);

/* registration_2d (src/kernels/registration_2d.cu:34-73): backward bilinear warp of frame_1. */
FLOW2D_API int flow2d_registration_2d(flow2d_context* ctx, const float* frame_0, const float* frame_1,
                                      const float* flow_u, const float* flow_v, size_t width, size_t height,
                                      size_t pitch_bytes, float hx, float hy, float* output);

/* The flow of the previous pyramid level resampled to this level's size (the bits of flow2d_resample_xy_pair into out_u / out_v)
 * and frame_1 warped by it (the bits of flow2d_registration_2d into `output`) in one launch: replaces the
 * CudaOperationResample2D::Execute + CudaOperationRegistration2D::Execute pair of optical_flow_2d.cpp:314-362 at every
 * level; at the coarsest one -- flow_u = flow_v = NULL, in_width = in_height = 0 -- out_u = out_v = 0 over width x height (the two
 * whole-plane memsets of optical_flow_2d.cpp:308-313) and frame_1 warped by that.  Written planes must be distinct from each other
 * and from every plane read. */
FLOW2D_API int flow2d_upsample_registration_2d(flow2d_context* ctx, const float* flow_u, const float* flow_v, size_t in_width,
                                               size_t in_height, float* out_u, float* out_v, const float* frame_0,
                                               const float* frame_1, size_t width, size_t height, size_t pitch_bytes, float hx,
                                               float hy, float* output);

/* resample_x / resample_y (src/kernels/resample_2d.cu:34-75,77-118): area-weighted 1-D resample. */
FLOW2D_API int flow2d_resample_x(flow2d_context* ctx, const float* input, float* output, size_t out_width,
                                 size_t out_height, size_t in_width, size_t pitch_bytes);
FLOW2D_API int flow2d_resample_y(flow2d_context* ctx, const float* input, float* output, size_t out_width,
                                 size_t out_height, size_t in_height, size_t pitch_bytes);

/* Two-plane forms of the three launchers the pyramid calls once for u and once for v (or once per frame) with
 * identical geometry: one launch does what two calls of the single-plane entry do, plane set `a` and plane set
 * `b` independently and with the same results (the second set rides in grid.z).  They replace the back-to-back
 * launch pairs of optical_flow_2d.cpp:284-303 (frames), :314-338 (flow resample), :480-500 (add), :505-530 (median);
 * on the coarse levels a launch costs more than its work.  The four output planes must be distinct. */
FLOW2D_API int flow2d_add_2d_pair(flow2d_context* ctx, float* operand_0_a, const float* operand_1_a,
                                  float* operand_0_b, const float* operand_1_b, size_t width, size_t height,
                                  size_t pitch_bytes);
FLOW2D_API int flow2d_median_2d_pair(flow2d_context* ctx, const float* input_a, const float* input_b, size_t width,
                                     size_t height, size_t pitch_bytes, size_t window, float* output_a,
                                     float* output_b);
/* add_2d followed by median_2d (optical_flow_2d.cpp:480-530: u += du, then the median of u) in one launch: the filter runs
 * over input + addend, formed per pixel as it is read -- add_2d's sum is one rounded addition (add_2d.cu:33-46), so the
 * result is the same -- and the plane of sums is neither written nor read back.  input is NOT modified.  input_b,
 * addend_b, output_b: optional second plane set (all three or none). */
FLOW2D_API int flow2d_add_median_2d_pair(flow2d_context* ctx, const float* input_a, const float* addend_a,
                                         const float* input_b, const float* addend_b, size_t width, size_t height,
                                         size_t pitch_bytes, size_t window, float* output_a, float* output_b);
FLOW2D_API int flow2d_resample_x_pair(flow2d_context* ctx, const float* input_a, float* output_a,
                                      const float* input_b, float* output_b, size_t out_width, size_t out_height,
                                      size_t in_width, size_t pitch_bytes);
FLOW2D_API int flow2d_resample_y_pair(flow2d_context* ctx, const float* input_a, float* output_a,
                                      const float* input_b, float* output_b, size_t out_width, size_t out_height,
                                      size_t in_height, size_t pitch_bytes);

/* resample_x into a temp plane followed by resample_y (CudaOperationResample2D::Execute,
 * src/cuda_operations/2d/cuda_operation_resample_2d.cpp:99-152) as ONE launch without the temp plane: every output evaluates
 * the x pass for the input rows of its y cells (same cell sums, rounded to float like the temp) and then the y pass.
 * Bit-identical to the two calls; meant for up-sampling (the flow of the previous pyramid level: one or two cells per
 * direction), correct for any ratio.  input_b / output_b: optional second plane (both or neither). */
FLOW2D_API int flow2d_resample_xy_pair(flow2d_context* ctx, const float* input_a, float* output_a, const float* input_b,
                                       float* output_b, size_t in_width, size_t in_height, size_t out_width,
                                       size_t out_height, size_t pitch_bytes);

/* The x pass of resample_2d.cu:34-75 for SEVERAL output widths in one trip over the input.  The reference resamples
 * both frames from full resolution at every pyramid level (optical_flow_2d.cpp:284-303), i.e. reads each frame once per
 * level; here every input row is read once, kept in LDS, and the x-resampled rows of all `level_count` widths are written
 * side by side into one "packed" plane: level l occupies columns [column_offsets[l], column_offsets[l] + out_widths[l])
 * of every row (offsets in floats, multiples of 4, segments disjoint, all within the pitch).  Each output is the same
 * left-to-right cell sum as flow2d_resample_x produces for that width (bit-identical).  The y pass of a level is then
 * flow2d_resample_y on `packed + column_offsets[l]`.  input_b / packed_b: optional second plane (both or neither).
 * Needs in_width <= 15360 (a row lives in LDS); FLOW2D_ERR_UNSUPPORTED beyond, or for more than 16 levels. */
#define FLOW2D_RESAMPLE_MAX_LEVELS 16
FLOW2D_API int flow2d_resample_x_levels(flow2d_context* ctx, const float* input_a, float* packed_a, const float* input_b,
                                        float* packed_b, size_t in_width, size_t height, size_t pitch_bytes,
                                        size_t level_count, const size_t* out_widths, const size_t* column_offsets);

/* The y passes of resample_2d.cu:77-118 for SEVERAL levels in one launch, the counterpart of flow2d_resample_x_levels: level l
 * reads columns [column_offsets[l], column_offsets[l] + out_widths[l]) of the packed plane (in_height rows) and writes an
 * out_widths[l] x out_heights[l] plane region whose first row is row output_rows[l] of the output plane (regions disjoint;
 * the region of a level is a plane of its own: pointer = output + output_rows[l] * pitch).  Each output is the same top-to-bottom
 * cell sum as flow2d_resample_y produces for that level (bit-identical).  packed_b / output_b: optional second plane. */
FLOW2D_API int flow2d_resample_y_levels(flow2d_context* ctx, const float* packed_a, float* output_a, const float* packed_b,
                                        float* output_b, size_t in_height, size_t pitch_bytes, size_t level_count,
                                        const size_t* out_widths, const size_t* out_heights, const size_t* column_offsets,
                                        const size_t* output_rows);

/* compute_phi_ksi (src/kernels/solve_2d.cu:43-198). */
FLOW2D_API int flow2d_compute_phi_ksi(flow2d_context* ctx, const float* frame_0, const float* frame_1,
                                      const float* flow_u, const float* flow_v, const float* flow_du,
                                      const float* flow_dv, size_t width, size_t height, size_t pitch_bytes, float hx,
                                      float hy, float equation_smoothness, float equation_data, float* phi,
                                      float* ksi);

/* solve_2d (src/kernels/solve_2d.cu:200-377): one Jacobi sweep, brightness constancy. */
FLOW2D_API int flow2d_solve_2d(flow2d_context* ctx, const float* frame_0, const float* frame_1, const float* flow_u,
                               const float* flow_v, const float* flow_du, const float* flow_dv, const float* phi,
                               const float* ksi, size_t width, size_t height, size_t pitch_bytes, float hx, float hy,
                               float equation_alpha, float* temp_du, float* temp_dv);

/* One Jacobi sweep with the opt-in FLOW2D_CONSTANCY_GRADIENT_UNTILED data term (no reference counterpart;
 * same arguments as flow2d_solve_2d_grad). */
FLOW2D_API int flow2d_solve_2d_grad_untiled(flow2d_context* ctx, const float* frame_0, const float* frame_1,
                                            const float* flow_u, const float* flow_v, const float* flow_du,
                                            const float* flow_dv, const float* phi, const float* ksi, size_t width,
                                            size_t height, size_t pitch_bytes, float hx, float hy,
                                            float equation_alpha, float* temp_du, float* temp_dv);

/* solve_2d_grad (src/kernels/solve_2d.cu:683-952): one Jacobi sweep, gradient constancy, including
 * the reference's 16x8 block rule for the second derivatives. */
FLOW2D_API int flow2d_solve_2d_grad(flow2d_context* ctx, const float* frame_0, const float* frame_1,
                                    const float* flow_u, const float* flow_v, const float* flow_du,
                                    const float* flow_dv, const float* phi, const float* ksi, size_t width,
                                    size_t height, size_t pitch_bytes, float hx, float hy, float equation_alpha,
                                    float* temp_du, float* temp_dv);

/* solve_2d_log (src/kernels/solve_2d.cu:391-669): one Jacobi sweep on the logarithmic derivatives
 * (gradient constancy of log(I + 1)), including that kernel's block rule: every 16x8 block's halo -- of the
 * frames, u, v, du, dv, phi and ksi alike -- holds the block's own edge pixel (:448,462,476,490). */
FLOW2D_API int flow2d_solve_2d_log(flow2d_context* ctx, const float* frame_0, const float* frame_1,
                                   const float* flow_u, const float* flow_v, const float* flow_du,
                                   const float* flow_dv, const float* phi, const float* ksi, size_t width,
                                   size_t height, size_t pitch_bytes, float hx, float hy, float equation_alpha,
                                   float* temp_du, float* temp_dv);

/* Opt-in red-black successive over-relaxation: ONE iteration = the pixels with even (x + y), then the odd
 * ones, relaxed in place on flow_du / flow_dv with factor omega in (0, 2).  NOT a reference kernel: the
 * reference relaxes with Jacobi sweeps, so this mode has no parity with it at equal iteration
 * counts; it exists because BASELINE.json names the scheme, and is checked against its own oracle restatement. */
FLOW2D_API int flow2d_solve_2d_sor(flow2d_context* ctx, const float* frame_0, const float* frame_1,
                                   const float* flow_u, const float* flow_v, float* flow_du, float* flow_dv,
                                   const float* phi, const float* ksi, size_t width, size_t height, size_t pitch_bytes,
                                   float hx, float hy, float equation_alpha, float omega, int data_constancy);

/* ---- the solver's fixed-point loop of one level -------------------------------------------
 * Replaces the launch loop of CudaOperationSolve2D::Execute
 * (src/cuda_operations/2d/cuda_operation_solve_2d.cpp:229-300): zero du/dv (level width x
 * container_height rows), then outer x [compute_phi_ksi, inner x (sweep, swap)].  The library
 * owns the ping-pong; *result_in_temp tells the caller which pair holds the result (0: flow_du /
 * flow_dv, 1: temp_du / temp_dv), so a host wrapper can swap its own pointers like the reference
 * does (:288-289).  No host synchronisation inside.  `algorithm`: see flow2d_solver_algorithm. */
typedef enum flow2d_solver_algorithm {
    FLOW2D_SOLVER_AUTO = 0,      /* library picks the fastest bit-exact path for the level size */
    FLOW2D_SOLVER_PER_SWEEP = 1, /* one launch per reference kernel launch (K6, K7/K9/K11) */
    FLOW2D_SOLVER_FUSED = 2,     /* phi/ksi + the inner sweeps of an outer iteration fused into ceil(inner / 5)
                                  * launches (one for inner <= 5); needs inner >= 1 */
    FLOW2D_SOLVER_SINGLE_WORKGROUP = 3, /* the whole level (all outer x inner iterations) in one launch on one
                                          CU; levels up to 64 x 64 pixels */
    FLOW2D_SOLVER_TILED = 4      /* one launch per outer iteration like FUSED, on 16x16 / 32x16 LDS tiles with a halo:
                                  * spreads small and mid-size levels over the whole chip; 1 <= inner <= 5; Grey,
                                  * Gradient and Gradient-untiled data terms */
} flow2d_solver_algorithm;

typedef struct flow2d_solve_params {
    size_t width, height;        /* level size */
    size_t pitch_bytes;          /* container pitch */
    size_t container_height;     /* rows of the full-resolution container (memset extent) */
    float hx, hy;                /* grid spacing of the level */
    float equation_alpha;
    float equation_smoothness;
    float equation_data;
    size_t outer_iterations_count;
    size_t inner_iterations_count;
    int data_constancy;          /* flow2d_constancy */
    int algorithm;               /* flow2d_solver_algorithm */
    float sor_omega;             /* 0 (default): Jacobi sweeps as in the reference.  In (0, 2): every inner iteration
                                    is one red-black SOR iteration instead (opt-in, no reference parity).  AUTO runs it
                                    temporally blocked -- LDS tiles or strips by level size, two iterations per launch --,
                                    FUSED / TILED (at most two iterations per outer iteration) ask for one of the two,
                                    PER_SWEEP for two half-sweep launches per iteration, in place; SINGLE_WORKGROUP and
                                    the LogDerivatives term have no red-black form (FLOW2D_ERR_UNSUPPORTED). */
} flow2d_solve_params;

/* Which algorithm flow2d_solve_level runs for a request: `requested` resolved (AUTO -> one of the four), or -1 when the
 * requested algorithm cannot run the level (SINGLE_WORKGROUP above 64 x 64; FUSED without sweeps, or on planes of
 * 4 GiB and more, which its 32-bit buffer offsets cannot address -- AUTO takes the per-sweep kernels there; TILED with
 * more than 5 sweeps or the LogDerivatives term).  Host logic only, needs no device. */
FLOW2D_API int flow2d_solver_algorithm_for(int requested, size_t width, size_t height, size_t pitch_bytes,
                                           size_t outer_iterations_count, size_t inner_iterations_count,
                                           int data_constancy);

FLOW2D_API int flow2d_solve_level(flow2d_context* ctx, const float* frame_0, const float* frame_1,
                                  const float* flow_u, const float* flow_v, float* flow_du, float* flow_dv,
                                  float* phi, float* ksi, float* temp_du, float* temp_dv,
                                  const flow2d_solve_params* params, int* result_in_temp);

/* ---- launch timing of the solver (measurement only; bench.py's roofline leg) ----------------
 * mode 1: every flow2d_solve_level call is bracketed by a pair of events on the context's stream
 * (the reference's own per-level timer, cuda_operation_solve_2d.cpp:220,302).  mode 2: additionally
 * every launch of the level's dominant kernel (the Jacobi sweep, or the fused outer-iteration kernel)
 * is bracketed by its own event pair.  Nothing synchronises; read the records after
 * flow2d_synchronize.  mode 0 switches the collection off. */
typedef struct flow2d_timing_record {
    size_t width, height;
    size_t outer, inner;
    int data_constancy;
    int algorithm;          /* the algorithm actually used (flow2d_solver_algorithm, never AUTO) */
    int kernel_launches;    /* launches of the dominant solver kernel inside the bracket */
    float elapsed_ms;       /* event time of the whole solve call */
    float kernel_ms;        /* mode 2: sum of the dominant kernel's launch durations; -1 otherwise */
    double algorithmic_bytes_per_launch; /* W*H*40 per sweep launch, W*H*(32+40*inner)/ceil(inner/5) per fused launch */
} flow2d_timing_record;

/* Diagnostics of the fused kernel (FLOW2D_SOLVER_FUSED): its sweeps divide through a reciprocal prepared once per pixel
 * and outer iteration (three instructions instead of the eleven of a correctly rounded division; bit-identical for all
 * operands inside the normal range, proven by exhaustion over all significand pairs), and a wavefront that meets
 * operands outside that range -- a denominator outside [2^-30, 2^40], a non-zero numerator below 2^-80, an infinite or
 * NaN result -- repeats its strip with the plain division.  Number of such repeats on this context since it was
 * created (synchronises the stream). */
FLOW2D_API int flow2d_fused_fallbacks(flow2d_context* ctx, unsigned long long* waves);
/* Waves of launches that never tried the short forms because the level's grid spacing lies outside the range they are
 * proven for (2h or 4h outside [2^-30, 2^40], a NaN spacing): such a launch runs the plain expressions throughout.  Counted
 * apart from the guard trips above (synchronises the stream). */
FLOW2D_API int flow2d_fused_plain_waves(flow2d_context* ctx, unsigned long long* waves);
/* The shader clock the device holds while other work runs.  flow2d_clock_probe_start queues, on this context's stream, one
 * sleeping wave per XCD that brackets `duration_us` microseconds with the constant 100 MHz clock and the shader clock (it takes
 * no issue slots from the kernels it runs beside: queue it on a context of its own while the work of interest runs on others);
 * flow2d_clock_probe_read waits for it and returns cycles / time per XCD in GHz (0 for an XCD no wave landed on).  Why it matters:
 * the chip's power management moves the clock between about 1.6 and 2.4 GHz with the power the running kernels draw -- the strip
 * kernel's instruction stream takes 13-16 % longer beside its own HBM traffic than on cache-resident rows AT THE SAME CYCLE COUNT,
 * and the eight XCDs of one chip differ by 3-5 % -- so a launch duration is only comparable at a known clock. */
FLOW2D_API int flow2d_clock_probe_start(flow2d_context* ctx, double duration_us);
FLOW2D_API int flow2d_clock_probe_read(flow2d_context* ctx, double* ghz_per_xcd);
/* The blocks a strip launch of a width x height level with `inner` sweeps and `instances` lock-step instances runs on this
 * device, in launch order: out[4 i .. 4 i + 3] = block column, strip, first row, end row of launch block i (all -1 for an id the
 * plan leaves empty); *grid_blocks = the launch's grid (a multiple of eight: workgroups are dealt to the eight XCDs in turn and
 * every XCD gets a contiguous, equally heavy run of the plan).  FLOW2D_ERR_INVALID_ARGUMENT when capacity_blocks is too small
 * (*grid_blocks is set).  A test hook: the order must be a permutation of the plan, the strips a partition of the level. */
FLOW2D_API int flow2d_fused_block_order(flow2d_context* ctx, size_t width, size_t height, size_t inner, size_t instances,
                                        int* out, size_t capacity_blocks, size_t* grid_blocks);

FLOW2D_API int flow2d_timing_enable(flow2d_context* ctx, int mode);
/* mode 2 brackets individual launches only for levels of at least min_width x min_height pixels
 * (default 0 x 0 = every level); smaller levels still get their mode-1 record. */
FLOW2D_API int flow2d_timing_launch_filter(flow2d_context* ctx, size_t min_width, size_t min_height);
FLOW2D_API int flow2d_timing_count(flow2d_context* ctx, size_t* count);
FLOW2D_API int flow2d_timing_get(flow2d_context* ctx, size_t index, flow2d_timing_record* out);
FLOW2D_API int flow2d_timing_reset(flow2d_context* ctx);

#ifdef __cplusplus
}
#endif

#endif /* FLOW2D_C_ABI_H_ */
