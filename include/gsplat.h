/* gsplat.h — C-ABI of libgsplat_mi355.so: the Gaussian-splat training step of
 * osreboot/Gaussian-Splatterer (reference v1.1.0) as hand-written HIP for MI355X (gfx950).
 *
 * Boundary.  The reference exposes this path as three C++ classes
 *     ModelSplatsHost    src/ModelSplatsHost.h:8-39
 *     ModelSplatsDevice  src/ModelSplatsDevice.h:5-30
 *     Trainer            src/Trainer.cuh:10-75
 * and consumes, underneath, CudaRasterizer::Rasterizer::forward / ::backward (call sites
 * src/Trainer.cu:175-201, :334-360, :378-412).  Every entry point below names the reference
 * interface it replaces.  include/gsplat_shim.hpp re-creates the three classes on top of this
 * header; INTEGRATION.md shows the binding a maintainer adds.
 *
 * Conventions
 *   - every function returns a gs_status (0 = OK, negative = error); gs_last_error() gives the
 *     thread-local message.  No C++ exception crosses this ABI.  The reference's explicit throws
 *     map to GS_ERR_NO_TRUTH (src/Trainer.cu:253), GS_ERR_DIMENSIONS (src/ModelSplatsHost.cpp:39-42),
 *     GS_ERR_CAPACITY (:66), GS_ERR_BOUNDS (:80-82).
 *   - host splat arrays use the reference layout (src/ModelSplatsHost.h:16-20): flat fp32,
 *     locations[3P], shs[3*M*P] (sh[(i*M + s)*3 + c]), scales[3P], opacities[P], rotations[4P]
 *     (element 0 = real part).  On the device the library keeps them SoA.
 *   - matrices are glm column-major float[16] (glm::value_ptr, src/Trainer.cu:323-324).
 *   - handles are not thread-safe; one caller thread per trainer (the reference's wx UI thread).
 *   - all device work of a trainer is enqueued on one HIP stream; calls return after enqueue unless
 *     stated otherwise.
 */
#ifndef GSPLAT_H
#define GSPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_VERSION_MAJOR 0
#define GS_VERSION_MINOR 1
#define GS_VERSION_PATCH 0

typedef enum gs_status {
    GS_OK = 0,
    GS_ERR_INVALID_ARGUMENT = -1,
    GS_ERR_HIP = -2,          /* a HIP runtime call failed; message carries hipGetErrorString */
    GS_ERR_NO_TRUTH = -3,     /* "Can't run training iteration, no truth data available!" src/Trainer.cu:253 */
    GS_ERR_CAPACITY = -4,     /* "Model ran out of capacity!" src/ModelSplatsHost.cpp:66 */
    GS_ERR_BOUNDS = -5,       /* "Can't copy splat in model, incorrect bounds..." src/ModelSplatsHost.cpp:80-82 */
    GS_ERR_DIMENSIONS = -6,   /* "Inconsistent feature dimensions..." src/ModelSplatsHost.cpp:39-42 */
    GS_ERR_OUT_OF_MEMORY = -7,
    GS_ERR_NO_MODEL = -8,
    GS_ERR_NO_DEVICE = -9,    /* no gfx950 device / HIP runtime unusable: there is no CPU fallback */
    GS_ERR_INTERNAL = -10,
    GS_ERR_COLLECTIVE = -11   /* a data-parallel collective hook failed on this rank: the caller must end the process (peers are
                                 blocked in the same collective); the library's own communicator has been aborted */
} gs_status;

const char* gs_last_error(void);
const char* gs_status_string(int status);
int gs_version(int* major, int* minor, int* patch);
/* Number of visible HIP devices (0 when none); never fails. */
int gs_device_count(void);
/* Switches.  gs_set_option edits the process-wide DEFAULTS: a trainer copies them when it is created and keeps its own
 * set afterwards (gs_trainer_set_option, below); the rasterizer seam reads "cull" from the defaults.
 * "cull" (default 1): the render kernels skip (splat, 8x8 pixel
 * block) pairs whose alpha >= 1/255 box misses the block; 0 evaluates every staged pair.  Results
 * are bit-identical either way (tests/test_gpu_raster.py checks exactly that).
 * "share_camera_passes" (default 1; read by gs_trainer_set_views): passes whose camera parameters are bit-identical
 * (the reference's white/black pair per camera, src/Trainer.cu:311-318) share projection, tile lists and the
 * forward blend; 0 recomputes them per pass like the reference.  Results are bit-identical either way.
 * "fuse_camera_passes" (default 1): gs_trainer_step WITHOUT densify runs ONE backward per camera on the sum of its
 * passes' residual images instead of one per pass (the backward is linear in dL/dpixel for a fixed camera).  All
 * averaged gradients are the same sums re-associated (measured distance from the per-pass form: <= 3e-6 of the splat's
 * sum|term|, 99.9 % of the entries <= 4e-7; tests/test_gpu_trainer.py::test_step_sgd_matches_oracle); only `var`, which needs every pass's own
 * location gradient (src/Trainer.cu:52) and is read by the densify block alone (:444,453), is then not produced (its
 * plane of the gradient buffer is zero).  gs_trainer_accumulate and densify steps always take the per-pass form.
 * "arena_entries" (default 0 = max(2^20, 16*P)): initial capacity, in (splat, tile) entries per camera, of the binning
 * arena of trainers created afterwards; when a step needs more the arena grows and the step is replayed.
 * "debug_sync" (default 0): wait and check for device errors after every stage of a step, as the reference's
 * debug=true rasterizer calls do (src/Trainer.cu:201,360,412); a failing step then names its stage.
 * "scan_single_max" (default 65536): the per-view scans of super-tile counters and tile counts run as one workgroup
 * per view up to this many items and as a three-phase scan beyond; results are identical either way. */
int gs_set_option(const char* name, int value);
/* Diagnostic: runs the backward kernel's 9-value reduce-scatter over the four 16-lane rows of one wave64 (a row = one
 * parked hit in the kernel's contraction).  in_host[q*64 + lane] (q = 0..8), out_host[lane]: lane 2q of a row holds that
 * row's total of value q (q < 8), lane 1 its total of value 8. */
int gs_debug_wave_reduce9(const float* in_host, float* out_host);
/* Diagnostic: counters of the render kernels, all zero unless the library was built with -DGS_DIAG_COUNT_ACTIVE (a tuning
 * build, tools/build_variant.sh): [0] forward hits — evaluated (tile entry, 8x8 pixel block) pairs —, [1] lanes of those hits that
 * blend (pixel alive, alpha >= 1/255), [2] / [3] the same for the backward, [4] / [5] forward / backward pairs staged in front of
 * the block test, [6] / [7] backward iterations if hits were packed by 8x4 half / 4x4 quadrant.  active / (64 * hits) is the
 * useful-lane fraction of a hit.  reset != 0 zeroes them. */
int gs_debug_counters(unsigned long long out[8], int reset);
/* Diagnostic: what this device's HBM sustains for a plain streaming copy — the achievable ceiling SURVEY 8(d) asks to be stated
 * next to the 8 TB/s specification.  Copies `bytes` (a multiple of 16; two buffers of that size are allocated and freed) with a
 * float4-per-lane kernel in sixteen forms (grid size, one or four loads in flight per lane, plain or non-temporal accesses), each
 * `repeats` times after a warm-up launch, and returns the FASTEST launch's (read + written bytes) / time in GB/s, from HIP events on a
 * stream of its own.  bench.py reports it as roofline.peak_measured. */
int gs_debug_hbm_copy_rate(size_t bytes, int repeats, double* gbytes_per_s);
/* Which of the sixteen forms won the last gs_debug_hbm_copy_rate (-1: none yet): bits 0-1 grid = 256 x (8 << g) workgroups, bit 2 four
 * loads in flight per lane, bit 3 non-temporal accesses. */
int gs_debug_hbm_copy_form(void);

/* ------------------------------------------------------------------------------------------
 * Raw device-memory helpers (current HIP device).  They exist so that C / ctypes callers and the
 * parity tests can stage buffers for the device-pointer entry points without another runtime.
 * ------------------------------------------------------------------------------------------ */
int gs_device_malloc(void** ptr, size_t bytes);
int gs_device_free(void* ptr);
int gs_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int gs_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);
int gs_memset_d(void* dst_dev, int value, size_t bytes);
int gs_device_synchronize(void);

/* ------------------------------------------------------------------------------------------
 * Splat model  (replaces class ModelSplatsDevice, src/ModelSplatsDevice.h:5-30)
 * ------------------------------------------------------------------------------------------ */
typedef struct gs_model gs_model;

/* ModelSplatsDevice(const ModelSplatsHost&), src/ModelSplatsDevice.cpp:24-40.  `count` splats are
 * uploaded; capacity / sh_degree / sh_coeffs are carried as metadata exactly like the reference.
 * count == 0 is legal (Trainer's placeholder model, src/Trainer.cu:112). */
int gs_model_create(int capacity, int sh_degree, int sh_coeffs, int count, const float* locations,
                    const float* shs, const float* scales, const float* opacities, const float* rotations,
                    gs_model** out);
/* ModelSplatsDevice(const ModelSplatsDevice&), src/ModelSplatsDevice.cpp:6-22 (device-to-device clone). */
int gs_model_clone(const gs_model* src, gs_model** out);
/* ModelSplatsHost(const ModelSplatsDevice&), src/ModelSplatsHost.cpp:16-26: copies `count` splats to
 * host arrays in the reference layout.  Synchronous. */
int gs_model_download(const gs_model* model, float* locations, float* shs, float* scales, float* opacities,
                      float* rotations);
/* public fields capacity / shDegree / shCoeffs / count, src/ModelSplatsDevice.h:8-12. */
int gs_model_info(const gs_model* model, int* capacity, int* sh_degree, int* sh_coeffs, int* count);
/* ~ModelSplatsDevice, src/ModelSplatsDevice.cpp:42-48. */
int gs_model_destroy(gs_model* model);

/* ------------------------------------------------------------------------------------------
 * Trainer  (replaces class Trainer, src/Trainer.cuh:10-75)
 * ------------------------------------------------------------------------------------------ */
typedef struct gs_trainer gs_trainer;

/* One (camera, background) pass.  The values the reference builds per pass at
 * src/Trainer.cu:317-326 (bg, view = -lookAt, projview = perspective * view, camera position) and
 * :355-356 (tan_fovx = tan_fovy = tan(radians(fovDegY)/2)).  40 floats, no padding. */
typedef struct gs_view {
    float view[16];
    float projview[16];
    float campos[3];
    float tan_fovx;
    float tan_fovy;
    float bg[3];
} gs_view;

enum { GS_UPDATE_SGD_CLAMP = 0, /* applyGradients, src/Trainer.cu:81-101 — the reference rule */
       GS_UPDATE_ADAM = 1       /* BASELINE.json extension: Adam ascent, then the same clamps */ };
enum { GS_QUAT_XYZW = 1, /* glm default member order: densify split permutes the stored quaternion
                            (src/Trainer.cu:463,493-494) */
       GS_QUAT_WXYZ = 0 };

/* Run-time hyper-parameters read on every call like Trainer::train reads Project
 * (src/Project.h:26-41; used at src/Trainer.cu:430-431,451-454,478-481,511). */
typedef struct gs_hyper {
    float lr_location, lr_sh, lr_scale, lr_opacity, lr_rotation; /* Project.h:26-30 */
    float scale_max;                                             /* paramScaleMax, Project.h:32 */
    float cull_opacity, cull_size, densify_variance;             /* Project.h:34-36 */
    float split_size, split_distance, split_scale, clone_distance; /* Project.h:38-41 */
    int update_rule;                                             /* GS_UPDATE_* */
    float adam_beta1, adam_beta2, adam_eps;
    int quat_layout;                                             /* GS_QUAT_* (densify only) */
} gs_hyper;
/* Project's defaults (src/Project.h:26-41), SGD_CLAMP, beta 0.9/0.999, eps 1e-15, GS_QUAT_XYZW. */
int gs_hyper_defaults(gs_hyper* out);

typedef struct gs_step_stats {
    int count_before, count_after; /* model->count around the step */
    int views;                     /* local passes executed */
    long long num_rendered;        /* sum over local views of the rasterizer's num_rendered (R) */
    int max_tile_list;             /* longest per-tile list seen this step */
    int arena_regrows;             /* times the binning arena had to grow and the step was replayed */
    float loss;                    /* sum over local views and pixels of residual^2 (diagnostic) */
} gs_step_stats;

/* Trainer::Trainer(), src/Trainer.cu:103-113.  width/height replace the compile-time
 * RENDER_RESOLUTION_X/Y (src/Config.h:13-14).  Binds to the current HIP device. */
int gs_trainer_create(int width, int height, gs_trainer** out);
/* Trainer::~Trainer(), src/Trainer.cu:115-146 (also destroys the owned model). */
int gs_trainer_destroy(gs_trainer* trainer);
/* The `delete trainer->model; trainer->model = new ModelSplatsDevice(host);` idiom
 * (src/ui/UiFrame.cpp:157-158,173-174,261-262,448-449).  The trainer takes ownership of `model`
 * and destroys the previous one.  Optimizer state (Adam) is reset. */
int gs_trainer_set_model(gs_trainer* trainer, gs_model* model);
/* Memory note: after its first densify step a trainer keeps three spare plane sets (parameters and the two Adam moments, the
 * sets the previous densify replaced, each with 50 % headroom: about 1 GB at 1M splats with M = 16) so that later densify
 * steps neither allocate nor free; growing a spare does synchronise the device once.  gs_trainer_set_model returns the spares
 * to the allocator when the new model is less than half their size. */
/* public member Trainer::model, src/Trainer.cuh:50 (borrowed pointer; owned by the trainer). */
gs_model* gs_trainer_get_model(gs_trainer* trainer);
/* Replaces Trainer::captureTruths (src/Trainer.cu:218-250), whose OptiX renderer is out of scope:
 * the caller supplies the passes this process owns and their truth images (RGBA8, R in the low
 * byte, width*height each; src/rtx/RtxDevice.cu:100).  `truth[i]` is a host pointer, or a device
 * pointer when truth_on_device != 0; images are copied.  `total_samples` is S, the number of
 * passes of the whole iteration over ALL processes (2 * #cameras, src/Trainer.cu:419): equal to
 * n_views on one GPU, larger when views are sharded.  n_views == 0 with total_samples > 0 is a
 * data-parallel rank that owns no pass (more ranks than passes): its steps contribute a zero
 * gradient, still run the collective hook and apply the common update.  n_views == 0 with
 * total_samples == 0 is the reference's "no truth data" state (gs_trainer_step fails with
 * GS_ERR_NO_TRUTH, src/Trainer.cu:253). */
int gs_trainer_set_views(gs_trainer* trainer, int n_views, const gs_view* views, const uint32_t* const* truth,
                         int truth_on_device, int total_samples);
/* Trainer::train(Project&, bool densify), src/Trainer.cu:252-543: all local passes (forward, loss,
 * backward, gradient averaging), the collective hook if one is installed, the parameter update,
 * and densify/prune when `densify` != 0.  The caller increments its own Project::iterations
 * (src/Trainer.cu:255).  Synchronises with the device once (arena check) and fully when
 * densify != 0 or stats != NULL.  With densify == 0 the `var` plane of the gradient buffer is not produced
 * (option "fuse_camera_passes"); the model after the step is what the reference's step leaves. */
int gs_trainer_step(gs_trainer* trainer, const gs_hyper* hyper, int densify, gs_step_stats* stats);

/* The same step split at the point where data-parallel ranks exchange gradients:
 *   gs_trainer_accumulate : src/Trainer.cu:303-425 for the local passes
 *   gs_trainer_grad_buffer: the averaged-gradient buffer [loc 3 | sh 3M | scale 3 | opacity 1 |
 *                           rot 4 | var 1] planes x plane stride floats, contiguous, fp32 (device).
 *                           The `var` plane holds accumulateGradients' var (src/Trainer.cu:52) after
 *                           gs_trainer_accumulate and after a gs_trainer_step WITH densify; a
 *                           gs_trainer_step without densify leaves it ZERO (option "fuse_camera_passes":
 *                           the reference itself reads var only inside the densify block, :444,453)
 *   gs_trainer_apply      : src/Trainer.cu:427-542 (update, optional densify) */
int gs_trainer_accumulate(gs_trainer* trainer, gs_step_stats* stats);
int gs_trainer_grad_buffer(gs_trainer* trainer, float** device_ptr, size_t* n_floats);
int gs_trainer_apply(gs_trainer* trainer, const gs_hyper* hyper, int densify, gs_step_stats* stats);

/* Per-trainer switches: the names gs_set_option documents (except "scan_single_max"), plus
 * "sh_fp16" (default 0; BASELINE config 5's "fp16 SH coeffs"): the projection reads the SH coefficients from an IEEE-half
 * READ COPY of the SH planes.  The fp32 planes stay the model (what gs_model_download returns and the optimiser updates:
 * a learning rate of 1e-4 would vanish below half an fp16 ulp) and all gradients stay fp32; the update kernel refreshes
 * the copy of every element it writes.  Geometry, tile lists and ranges do not depend on SH and are bit-identical with
 * the switch on or off; colours and gradients are those of the fp32 path run on the half-rounded coefficients, bit for
 * bit (tests/test_gpu_trainer.py::test_sh_fp16_*), i.e. within 2^-11 relative per coefficient of the fp32 result.
 * "long_list_sort_launch" (default -1): the sorts of tile lists of 512 and of 2048 entries and more are launches of their own
 * that many scenes leave empty; -1 skips each while the longest list of two steps ago stayed a quarter below its limit (a
 * list that outgrows the hint is sorted in global scratch by the sorter one class down: slower, same result) and sizes the
 * launches' grids from the tile order of two steps ago; 0 never launches them; 1 always does, with grids that need no hint.
 * "debug_sort_grids" (default -1; a test hook): a value >= 0 replaces that hint by small_first | mid_grid << 16 (tiles): any
 * grids must produce the same lists.
 * "row_marks" (default -1): the backward leaves one gradient row per (tile entry, camera) for the per-splat kernel.  With marks
 * (1) a row is written, marked and later read only for an entry some 8x8 pixel block evaluated; all others — everything behind a
 * tile's last contributor: 84 % of the rows of a dense scene — are implicit zeros.  Without (0) every entry owns a row, which is
 * cheaper where lists are short and nearly every row exists.  -1 decides per camera and step by the camera's longest tile list
 * (marks from 1024 entries on; the kernels read it from the tile scan of the same step).  Bit-identical gradients either way.
 * "reuse_hit_masks" (default 1): the backward reuses the block ballots the forward of the same camera stored (which entries of
 * a tile list can reach which 8x8 pixel block) instead of running the block test again; 0 makes it test itself.  Bit-identical.
 * "exchange_overlap" (default 1): see gs_trainer_set_compact_exchange.
 * "fuse_update" (default 1): gs_trainer_step applies the update (applyGradients, src/Trainer.cu:81-101, or Adam) inside the kernel that
 * averages the gradients (the per-splat reduction; under the compact exchange the SH rebuild) whenever no collective sits between the
 * two, instead of launching an update over the gradient buffer; the gradient buffer is written either way.  0: always the launch.
 * Bit-identical parameters, moments and gradient buffer.
 * "list_cut" (default 1), "list_cut_min_avg", "list_cut_margin": the depth cut of the tile lists, see gs_trainer_list_cut_stats.
 * "roctx" (default 0, or 1 when the environment holds GS_ROCTX=1 at gs_trainer_create): a roctx range named like
 * gs_stage_name() is pushed around the launches of every stage of a step (librocprofiler-sdk-roctx is loaded on first use), so
 * that `rocprofv3 --marker-trace --kernel-trace` attributes the kernels to the nine stages.
 * Changing "share_camera_passes" regroups the passes already set. */
int gs_trainer_set_option(gs_trainer* trainer, const char* name, int value);

/* Optimizer state of GS_UPDATE_ADAM (build-side extension; the reference has none): device pointers to the first and
 * second moment, each [11+3M planes][plane stride] fp32 like the parameters, and the number of Adam steps taken.
 * NULL / 0 before the first Adam step.  The moments follow their splats through densify/prune (the two halves of a
 * split and a clone's twin start from the parent's moments) and the step counter keeps running; replacing the model
 * (gs_trainer_set_model) resets both.  Synchronises the trainer's stream. */
int gs_trainer_adam_state(gs_trainer* trainer, float** moment1, float** moment2, size_t* n_floats, int* steps);
/* The other direction — resume of an Adam run from a checkpoint (the reference's .gobj / settings.json hold
 * no optimizer state because its update rule has none; src/ui/UiFrame.cpp:323-358 save, :373-450 load): installs the two
 * moments (n_floats each = (11 + 3M) x the plane stride of the trainer's CURRENT model, the layout gs_trainer_adam_state
 * reports; host pointers, or device pointers when on_device != 0; copied before return) and the number of Adam steps
 * already taken.  Call it after gs_trainer_set_model (which resets the state).  Model + moments + step counter restored,
 * the next steps equal the uninterrupted run's bit for bit (tests/test_gpu_trainer.py::test_adam_state_restore_resumes_bit_exact).
 * moment1 == moment2 == NULL with steps == 0 clears the state.  Not available under the sharded update (each rank holds
 * its own chunk's moments only). */
int gs_trainer_set_adam_state(gs_trainer* trainer, const float* moment1, const float* moment2, size_t n_floats, int steps,
                              int on_device);

/* Collective hook called by gs_trainer_step between accumulate and apply: must sum `n_floats`
 * fp32 values at `device_buf` in place over all ranks, enqueued on `hip_stream`.  Return 0 on success. */
typedef int (*gs_allreduce_fn)(float* device_buf, size_t n_floats, void* hip_stream, void* user);
int gs_trainer_set_allreduce(gs_trainer* trainer, gs_allreduce_fn fn, void* user);
/* Sharded update, the other data-parallel form of gs_trainer_step: instead of all-reducing the gradient buffer and
 * repeating the whole update on every rank, the step
 *   1. reduce-scatters the gradient buffer: rank r ends with the sums of chunk r only,
 *   2. updates chunk r of the parameter planes (Adam moments exist and advance for that chunk only),
 *   3. all-gathers the parameter planes.
 * Both hooks work IN PLACE on `n_floats` fp32 at `device_buf`, enqueued on `hip_stream`; n_floats is a multiple of
 * `world` (the library pads its plane-major buffers), chunk r = [r, r + 1) * n_floats / world.  reduce_scatter: on
 * return chunk `rank` holds the sum over ranks of that chunk (the rest is unspecified); all_gather: every rank's
 * chunk `rank` is copied to all ranks.  A densify step additionally all-gathers the gradient buffer (var, location
 * gradient) and the Adam moments so that every rank densifies identically.  Replicas stay bit-identical because each
 * element is reduced and updated on exactly one rank.  Pass NULL hooks to return to the unsharded step. */
typedef int (*gs_collective_fn)(float* device_buf, size_t n_floats, void* hip_stream, void* user);
int gs_trainer_set_sharded_update(gs_trainer* trainer, gs_collective_fn reduce_scatter, gs_collective_fn all_gather, void* user,
                                  int rank, int world);
/* Compact exchange, the third data-parallel form of gs_trainer_step — the one that moves the fewest bytes when a rank holds few
 * cameras.  3M of the 12 + 3M gradient planes are SH gradients, and one record's SH gradient of a splat is rank one:
 * basis(view direction)[M] x dL_dRGB[3] (what accumulateGradients sums, src/Trainer.cu:60-64, built per camera).  So the step
 *   1. leaves, per rank, the sums of the twelve other planes (loc 3 | scale 3 | opacity | rot 4 | var) over its cameras and the
 *      dL_dRGB record (3 floats per splat) of each of its cameras (each of its passes on a densify step),
 *   2. all-reduces the twelve planes (hook all_reduce, on the trainer's second stream) WHILE it all-gathers the records (hook
 *      all_gather, in place: chunk r of the buffer is rank r's), 
 *   3. rebuilds the SH planes on every rank from ALL records in the order a single GPU accumulates them, and applies the update.
 * Received per rank at BASELINE cfg3 on 8 GPUs (8 cameras, M = 16, 100k splats): 8.4 MB + 8.4 MB instead of the 42 MB a ring
 * all-reduce of the 24 MB buffer moves; the form pays while cameras < 2 M (bench.py --collective auto chooses by that).
 * The SH gradients are those of the single-GPU step bit for bit (no collective sums them); replicas stay bit-identical as
 * long as the all-reduce leaves identical sums on all ranks (RCCL's does).
 * Layout contract: the iteration has n_cameras cameras in the reference's order (src/Trainer.cu:311-314: pass c = camera c on
 * white, pass n_cameras + c = camera c on black); camera c belongs to rank c % world with BOTH passes (at least one camera
 * per rank), gs_trainer_set_views on each rank holds exactly those passes and total_samples = 2 * n_cameras.  The camera
 * positions the SH basis needs travel WITH the records: every rank puts the positions of its cameras, as set by its latest
 * gs_trainer_set_views, in a small header of its chunk (3 floats per camera, padded to 64), so re-captured / re-rotated cameras
 * (the reference re-captures every intervalCapture iterations, src/ui/UiFrame.cpp:266-298) need no second call here.  `campos`
 * is kept in the signature for ABI stability and is NOT read (may be NULL).  The all_gather hook is called with
 * world * (header + records) floats.  Pass NULL hooks to leave the form.
 * Trainer option "exchange_overlap" (default 1): 0 issues the two collectives one after the other on the trainer's stream. */
int gs_trainer_set_compact_exchange(gs_trainer* trainer, gs_collective_fn all_gather, gs_allreduce_fn all_reduce, void* user, int rank,
                                    int world, int n_cameras, const float* campos);
/* Depth cut of the tile lists (trainer option "list_cut", default 1): in a dense scene a training step lists, per tile,
 * only the entries in front of the depth at which the PREVIOUS step's forward stopped reading that tile (+ a margin); the forward checks
 * that every pixel of a shortened list still finishes inside it, and a step whose cut was wrong is replayed uncut before anything of it is
 * applied — a cut step that stands equals the uncut step bit for bit.  steps_cut: accumulate attempts that ran with cut lists; replays:
 * those of them the forward found wrong.  Related options (test hooks): "list_cut_min_avg" (entries per tile from which the cut is used,
 * default 384), "list_cut_margin" (entries kept behind the last one read, default 64). */
int gs_trainer_list_cut_stats(gs_trainer* trainer, long long* steps_cut, long long* replays);
/* Diagnostic (synchronises): what the newest step's binning listed, summed over the cameras — out[0] (splat, super-tile) candidates the coarse
 * scatter emitted, out[1] tile-list entries, out[2] tiles that carry a finite depth bound for the next step, out[3] tiles. */
int gs_trainer_debug_list_totals(gs_trainer* trainer, long long out[4]);
/* The HIP stream (hipStream_t) all of this trainer's work is enqueued on. */
int gs_trainer_get_stream(gs_trainer* trainer, void** hip_stream);
int gs_trainer_synchronize(gs_trainer* trainer);

/* Trainer::render(uint32_t* frameBuffer, int sizeX, int sizeY, float splatScale, const Camera&),
 * src/Trainer.cu:148-216: forward only, RGBA8 out via imageFloatToInt (:19-29).  `view->bg` is
 * honoured (the reference passes black, :151); `view->tan_fovx` carries the reference's
 * tan(radians(sizeX*fovY/sizeY)/2) (:196).  Synchronous like the reference (:215). */
int gs_trainer_render(gs_trainer* trainer, uint32_t* framebuffer, int fb_on_device, int size_x, int size_y,
                      float splat_scale, const gs_view* view);
/* Per-stage device timing with HIP events recorded on the trainer's stream (evidence for the
 * roofline report; off by default).  Stage i is named gs_stage_name(i): preprocess, scan, scatter,
 * tile_sort, render_forward, render_backward, splat_backward, update, collective.
 * enable: 0 off, 1 every stage, otherwise a stage mask shifted left by one (bit i+1 times stage i only:
 * an event costs ~3 us of stream time, so timing one kernel over a long run should not pay for all nine).
 * gs_trainer_stage_times returns the sums (ms) and launch counts since profiling was switched on. */
#define GS_STAGE_COUNT 9
int gs_trainer_set_profiling(gs_trainer* trainer, int enable);
int gs_trainer_stage_times(gs_trainer* trainer, double ms_sum[GS_STAGE_COUNT], long long launches[GS_STAGE_COUNT]);
const char* gs_stage_name(int stage);
/* Diagnostics for tests: copy pass `view_index`'s last rendered float image [3][H][W] to host. */
int gs_trainer_read_image(gs_trainer* trainer, int view_index, float* host_chw);

/* ------------------------------------------------------------------------------------------
 * Native RCCL communicator for the hook above (one process per GPU, xGMI within a node).
 * ------------------------------------------------------------------------------------------ */
typedef struct gs_comm gs_comm;
#define GS_COMM_ID_BYTES 128
int gs_comm_unique_id(char id[GS_COMM_ID_BYTES]);                 /* ncclGetUniqueId on one rank */
int gs_comm_create(const char id[GS_COMM_ID_BYTES], int rank, int n_ranks, gs_comm** out); /* ncclCommInitRank */
int gs_comm_destroy(gs_comm* comm);
/* Installs an RCCL sum all-reduce of the gradient buffer as the trainer's collective hook. */
int gs_trainer_attach_comm(gs_trainer* trainer, gs_comm* comm);
/* Installs ncclReduceScatter / ncclAllGather (in place) as the hooks of gs_trainer_set_sharded_update. */
int gs_trainer_attach_comm_sharded(gs_trainer* trainer, gs_comm* comm);
/* Installs ncclAllGather on `gather` and ncclAllReduce on `reduce` as the hooks of gs_trainer_set_compact_exchange.  Two
 * communicators so that the two collectives can run side by side (operations of one communicator execute in issue order);
 * reduce == NULL or == gather serialises them. */
int gs_trainer_attach_comm_compact(gs_trainer* trainer, gs_comm* gather, gs_comm* reduce, int n_cameras, const float* campos);

/* ------------------------------------------------------------------------------------------
 * Inner seam: the rasterizer pair the reference calls.  Same argument order as
 * CudaRasterizer::Rasterizer::forward (26 args, src/Trainer.cu:175-201 / :334-360) and ::backward
 * (34 args, src/Trainer.cu:378-412); std::function allocators become (fn, user) pairs.  All data
 * pointers are DEVICE pointers in the reference (AoS) layout.  colors_precomp, cov3D_precomp and
 * radii must be NULL and prefiltered 0, as at every reference call site.  The three chunks are
 * opaque; gs_raster_chunk_field() locates arrays inside them for inspection.  Synchronous.
 * ------------------------------------------------------------------------------------------ */
typedef char* (*gs_alloc_fn)(size_t bytes, void* user);

int gs_rasterize_forward(gs_alloc_fn geometry_alloc, void* geometry_user, gs_alloc_fn binning_alloc,
                         void* binning_user, gs_alloc_fn image_alloc, void* image_user, int P, int D, int M,
                         const float* background, int width, int height, const float* means3D, const float* shs,
                         const float* colors_precomp, const float* opacities, const float* scales,
                         float scale_modifier, const float* rotations, const float* cov3D_precomp,
                         const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                         float tan_fovy, int prefiltered, float* out_color, int* radii, int debug,
                         int* num_rendered);

int gs_rasterize_backward(int P, int D, int M, int R, const float* background, int width, int height,
                          const float* means3D, const float* shs, const float* colors_precomp,
                          const float* scales, float scale_modifier, const float* rotations,
                          const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                          const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                          char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix,
                          float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                          float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                          int debug);

/* Byte offset and size of a named array inside one of the three chunks.  chunk: "geometry",
 * "binning" or "image".  Geometry fields: "record" (64-byte per-splat records: means2D, conic,
 * opacity, rgb, cull box, depth, radius, clamp flags, tile rect), "tiles_touched", "point_offsets".
 * Binning fields: "point_list" (u32[R], the sorted splat ids), "point_list_slots".
 * Image fields: "ranges" (u32x2 per tile), "final_T", "n_contrib", "tile_order" (u32[T]: the order
 * the per-tile workgroups take the tiles in — speed only). */
int gs_raster_chunk_field(const char* chunk, const char* field, int P, int width, int height, int R,
                          size_t* offset, size_t* bytes);

/* The reference's two image kernels as device entry points (device pointers; synchronous):
 * imageFloatToInt src/Trainer.cu:19-29 and imageIntToLoss src/Trainer.cu:33-44. */
int gs_image_float_to_int(const float* source_chw, uint32_t* framebuffer, int w, int h);
int gs_image_int_to_loss(const uint32_t* truth, const float* rasterized_chw, float* loss_chw, int w, int h);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_H */
