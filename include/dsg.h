/* dsg.h -- C ABI of libdsg_hip.so: the MI355X-native DDPM/DDIM sampling path of DiffuseStyleGesture.
 *
 * The reference (YoungSeng/DiffuseStyleGesture) is pure Python on PyTorch; it has no FFI for this path.  The
 * plugin surface it exposes is two Python call signatures, and every entry point below names the reference
 * interface it stands in for (paths relative to /root/reference):
 *
 *   dsg_create / dsg_load_tensor / dsg_finalize_weights
 *        MDM.__init__ + load_model_wo_clip(model, state_dict)       main/model/mdm.py:10-151, main/utils/model_util.py:8-12
 *        (tensor names = the checkpoint's state_dict keys)
 *   dsg_set_schedule
 *        SpacedDiffusion(use_timesteps, betas=...)                   main/diffusion/respace.py:73-87,
 *        GaussianDiffusion.__init__ tables                           main/diffusion/gaussian_diffusion.py:161-198
 *   dsg_set_window_cond
 *        the `y` dict of model_kwargs (style, seed, audio, mask_local) main/mydiffusion_zeggs/sample.py:227-251
 *   dsg_set_seed_last
 *        y['seed_last'] of DiffuseStyleGesture++ (cross_local_attention5)   BEAT-TWH-main/model/mdm.py:226-230,
 *                                                                    BEAT-TWH-main/mydiffusion_beat_twh/sample.py:85-93
 *   dsg_forward
 *        MDM.forward(x, timesteps, y)                                main/model/mdm.py:166-358
 *                                                                    BEAT-TWH-main/model/mdm.py:134-267
 *   dsg_sample
 *        GaussianDiffusion.p_sample_loop / ddim_sample_loop          main/diffusion/gaussian_diffusion.py:608-671, :889-936
 *   dsg_set_window_cond_cfg
 *        ClassifierFreeSampleModel.forward (y['scale'], y['uncond'])   main/model/cfg_sampler.py:8-31
 *   dsg_clone / dsg_sample_multi / dsg_set_kernel_set / dsg_get_kernel_set / dsg_recommend_kernel_set / dsg_last_kernel_set
 *        (no reference counterpart: the reference samples one clip at a time, sample.py:418 batch_size = 1; these run
 *         several clips of one GPU concurrently over one copy of the weights -- BASELINE config[3] "one clip per stream")
 *   dsg_noise
 *        th.randn(*shape) / th.randn_like(x)                         main/diffusion/gaussian_diffusion.py:704, :542
 *   dsg_pose2bvh
 *        pose2bvh(poses, outpath, length, smoothing)                 main/process/process_zeggs_bvh.py:219-275
 *   dsg_q_sample / dsg_predict_xstart_from_eps / dsg_posterior_step / dsg_ddim_step
 *        q_sample :236-254, _predict_xstart_from_eps :400-405, q_posterior_mean_variance + p_sample :256-278/:542-557,
 *        ddim_sample :773-792   (same file)
 *
 * Conventions: every function returns 0 on success or a negative DSG_E_* code; the message is available from
 * dsg_last_error() (thread local).  No C++ exception crosses this boundary.  Pointers may be host or device
 * pointers (detected with hipPointerGetAttributes); tensors are contiguous fp32 in the reference's layouts
 * ([B, J, 1, T] for poses/noise, frames fastest).  A handle is bound to one device and is not thread safe;
 * distinct handles are independent.  All work is enqueued on the handle's own stream and ordered after/before
 * the optional caller stream (`stream`, a hipStream_t) with events.
 */
#ifndef DSG_H_
#define DSG_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSG_VERSION 330

enum {
    DSG_OK = 0,
    DSG_E_INVALID = -1,       /* ValueError in the shim */
    DSG_E_RUNTIME = -2,       /* HIP runtime failure -> RuntimeError */
    DSG_E_UNEXPECTED_KEY = -3,/* load_model_wo_clip: unexpected state_dict key */
    DSG_E_MISSING_KEY = -4,   /* load_model_wo_clip: missing key at finalize */
    DSG_E_NOT_IMPLEMENTED = -5,
    DSG_E_STATE = -6          /* call order (e.g. sample before finalize / set_window_cond) */
};

/* DSG_PREC_BF16W2 (ABI 320): bf16 activations, every weight as hi + lo bf16 (16 mantissa bits), two MFMAs per weight fragment --
 * the precision mode between bf16 and fp32 (the bf16 drift of a 1000-step chain is the weights' 8-bit mantissa).  Kernel sets
 * LATENCY and TILE (every batch size) and, at latent_dim 128 / 256 without fused guidance, ROWS (round 6: k_clip_attn + the feed-forward kernel on
 * two-register fragments; DSG_KSET_AUTO picks it from 800 token rows -- 16 clips in lock step 453 -> 331 us per step); BLOCK / STREAM are
 * DSG_E_NOT_IMPLEMENTED.  The reference computes in fp32
 * (main/train/training_loop.py:39: no autocast): DSG_PREC_FP32 is its arithmetic, the other two trade accuracy for speed. */
enum { DSG_PREC_FP32 = 0, DSG_PREC_BF16 = 1, DSG_PREC_BF16W2 = 2 };
/* kernel sets (dsg_set_kernel_set): which hand-written kernels one denoising step is made of.  Same arithmetic, different
 * grouping / tiling, i.e. last-bit differences between sets in bf16 -- which is why the set is an explicit, sticky property of
 * a handle and never depends on how a call is issued. */
enum {
    DSG_KSET_AUTO = 0,      /* by batch: LATENCY for batch <= 2 (latent_dim <= 256; the wider DSG+ models run TILE there), TILE below 1000
                               token rows, BLOCK from there, STREAM from 2000 */
    DSG_KSET_LATENCY = 1,   /* fused redundant-compute kernels, 2 + 3L dispatches: one clip in flight */
    DSG_KSET_TILE = 2,      /* one 16 x 16 MFMA tile per wave: small batches */
    DSG_KSET_BLOCK = 3,     /* 32-row block GEMMs + fused attention/out_proj/LayerNorm: large batches, several lanes.  ABI 320 (bf16, ZEGGS / tiny
                               widths): per layer k_clip_attn (per (clip, head): Q / K / V slices in LDS + attention) + the feed-forward half split
                               over ff with out_proj + LayerNorm1 as its prologue + the slab sum / LayerNorm2 pass: 3 + 3L dispatches; the DSG+
                               widths and fp32 get the ff-split behind k_attn_op_w */
    DSG_KSET_STREAM = 4,    /* weight-stationary persistent GEMMs (32x32x16 MFMA, global->LDS staging, 64-row blocks) for the pose
                               embedding and the pose head; per layer k_clip_attn + ONE feed-forward kernel (out_proj + LayerNorm1 as its
                               prologue, linear1 + GELU + linear2 + residual + LayerNorm2; ABI 320: 3 + 2L dispatches per step): >= 2000
                               token rows (23 ZEGGS clips) in one lane, >= 850 rows (10 clips) per lane with several lanes.  bf16, latent_dim 128 / 256, 4 heads -- the ZEGGS model;
                               DSG_E_NOT_IMPLEMENTED elsewhere */
    DSG_KSET_ROWS = 5       /* ABI 330: BLOCK's pose embedding / local attention / pose head around STREAM's per-layer pair, the feed-forward kernel on
                               ONE 16-row tile per workgroup (no ff-split, no partial slabs, no slab-sum pass: 3 + 2L dispatches).  Every workgroup
                               streams a layer's W_o + W1 + W2 for its 16 rows: it pays while the row tiles of all lanes fit the 256 CUs in one
                               round -- 1000 .. 4000 token rows in one lane (12 .. 45 ZEGGS clips), fewer per lane with several lanes.  Same shapes
                               as STREAM, in bf16 and (without fused guidance) bf16w2.  At the DSG+ widths (bf16, latent_dim 384 / 512, 4 heads, ff 1024;
                               no fused guidance): streamed pose embedding, then per layer the attention half per (clip, head) (k_clip_attn_w: the rows pass
                               through the LDS in chunks) + the same feed-forward kernel -- on 32-row blocks when >= 3 lanes together exceed one round of the
                               CUs -- = 3 + 2L dispatches; from
                               9 BEAT / 13 TWH clips in one lane, 4 per lane and 16 in all with several lanes.  DSG_E_NOT_IMPLEMENTED elsewhere */
};
enum { DSG_MODE_DDPM = 0, DSG_MODE_DDIM = 1 };

typedef struct dsg_config {
    int32_t variant;        /* 3 = cross_local_attention3_style1 (ZEGGS), 4 = cross_local_attention4 (BEAT/TWH) */
    int32_t njoints;        /* J */
    int32_t n_poses;        /* T, frames per window (multiple of `window`) */
    int32_t n_seed;         /* S */
    int32_t latent_dim;     /* D (multiple of 64, <= 512) */
    int32_t audio_src_dim;  /* A_src */
    int32_t audio_dim;      /* A */
    int32_t style_dim_in;
    int32_t window;         /* local attention window (<= 16) */
    int32_t num_layers;
    int32_t num_heads;      /* self-attention heads; head dim in {32, 64, 96, 128} */
    int32_t ff_size;
    int32_t local_heads;    /* 8 in the reference; head dim <= 64 */
    int32_t pe_max_len;     /* rows of sequence_pos_encoder.pe (5000) */
    int32_t train_steps;    /* rows of the time-embedding table = original diffusion steps (1000) */
    int32_t max_batch;
    int32_t precision;      /* DSG_PREC_* */
    int32_t device;         /* HIP device ordinal */
    int32_t steps_per_graph;/* > 0: denoising steps captured per hipGraph replay; 0 = default (eager: measured faster), -1 = eager */
    int32_t latency_mode;   /* DSG_KSET_AUTO only: 0 = by batch, 1 = never the LATENCY set, 2 = always (where AUTO can pick it at all:
                               latent_dim <= 256; dsg_recommend_kernel_set and the handle itself apply the same rule) */
    int32_t reserved[4];
} dsg_config;

typedef struct dsg_handle dsg_handle;

int dsg_version(void);
const char* dsg_last_error(void);

int dsg_create(const dsg_config* cfg, dsg_handle** out);
/* a further sampling lane over the SAME weights (reference counted; call after dsg_finalize_weights): own stream / HSA queue,
 * state, conditioning and schedule.  max_batch <= 0: the source's.  One lane per concurrently sampled clip ("one clip per
 * stream", BASELINE config[3]); see dsg_sample_multi.  Reloading weights into the source does not update existing clones. */
int dsg_clone(dsg_handle* src, int max_batch, dsg_handle** out);
int dsg_destroy(dsg_handle* h);

/* dtype: 0 = float32.  shape/ndim are checked against the model dims. */
int dsg_load_tensor(dsg_handle* h, const char* name, const void* data, const int64_t* shape, int ndim, int dtype);
/* repack weights into MFMA fragment order (bf16 or fp32), fold input_process2 . poseEmbedding, build the
 * [train_steps, D] time-embedding tables and rotary tables */
int dsg_finalize_weights(dsg_handle* h);
/* betas: the (respaced) beta schedule, float64[n]; timestep_map: original timestep of each kept step, int64[n] */
int dsg_set_schedule(dsg_handle* h, const double* betas, const int64_t* timestep_map, int n);
/* host-only helper (no device needed): the 11 float64[n] tables of GaussianDiffusion.__init__, written to
 * out[11*n] in the order betas, alphas_cumprod, alphas_cumprod_prev, sqrt_alphas_cumprod,
 * sqrt_one_minus_alphas_cumprod, sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod, posterior_variance,
 * posterior_log_variance_clipped, posterior_mean_coef1, posterior_mean_coef2 */
int dsg_schedule_tables(const double* betas, int n, double* out);

/* variant 5 only: seed_last [B, J, 1, S]; kept until replaced; must precede dsg_set_window_cond with the same B */
int dsg_set_seed_last(dsg_handle* h, const float* seed_last, int B, void* stream);

/* style [B, style_dim_in]; seed [B, J, 1, S]; audio [B, T_a, A_src] (T_a = T for variant 3, T-S for variant 4, T-2S for 5);
 * mask_local uint8 [mask_batch, T] (1 = keep), mask_batch in {1, B}; NULL = the reference's `mask=None` (nothing masked but
 * the causal future: the look-back pad keys of window 0 attend with value -1, local_attention.py:196);
 * uncond != 0 -> uncond_info / y['uncond'].  The caller's buffers may be reused once `stream` has passed this call. */
int dsg_set_window_cond(dsg_handle* h, const float* style, const float* seed, const float* audio,
                        const uint8_t* mask_local, int mask_batch, int B, int uncond, void* stream);
/* classifier-free guidance, ClassifierFreeSampleModel.forward (main/model/cfg_sampler.py:8-31) fused into the path: the
 * B elements and their unconditional twins (y['uncond'] = True) run as ONE batch of 2B rows (max_batch >= 2B) and the
 * pose-head epilogue forms out_uncond + scale[b] * (out - out_uncond) before the sampler update.  scale: float[B]
 * (y['scale']).  dsg_forward / dsg_sample are then called with the user batch B as usual. */
int dsg_set_window_cond_cfg(dsg_handle* h, const float* style, const float* seed, const float* audio,
                            const uint8_t* mask_local, int mask_batch, int B, const float* scale, void* stream);

/* x, out: [B, J, 1, T] fp32; t: model timesteps int64[B] (each < train_steps) */
int dsg_forward(dsg_handle* h, const float* x, const int64_t* t, float* out, int B, void* stream);

typedef struct dsg_sample_args {
    int32_t mode;             /* DSG_MODE_DDPM / DSG_MODE_DDIM */
    int32_t skip_timesteps;
    float eta;                /* DDIM only */
    int32_t const_noise;      /* p_sample(const_noise=True): batch element 0's noise for everyone */
    const float* init_noise;  /* nullable [B,J,1,T]: the reference's `noise=` argument (x_T) */
    const float* step_noise;  /* nullable [n_run,B,J,1,T]: replayed per-step noise; else the Philox stream */
    const float* init_image;  /* nullable [B,J,1,T] */
    uint64_t seed;            /* Philox key */
    uint64_t stream_id;       /* Philox stream (e.g. clip index) */
    uint32_t draw_base;       /* draw index of x_T; step i uses draw_base + 1 + i */
    int32_t n_dump;           /* dump_steps support: number of entries in dump_steps */
    const int32_t* dump_steps;/* host int32[n_dump], ascending loop indices */
    float* dump_out;          /* [n_dump,B,J,1,T] */
    int32_t clip_denoised;    /* != 0: x0 clamped to [-1, 1] before the update (clip_denoised=True, gaussian_diffusion.py:377-379) */
    int32_t first_step;       /* ABI 310: run the chain in pieces (the lazy p_sample_loop_progressive, gaussian_diffusion.py:673-740): loop */
    int32_t max_steps;        /* index this call starts at (> 0: init_noise is x_t of that step, taken as it is) and how many steps it runs */
    int32_t reserved[1];      /* (0 = to the end); draw indices, dump_steps and step_noise stay those of the whole chain */
} dsg_sample_args;

/* runs num_timesteps - skip_timesteps denoising steps for the conditioning set by dsg_set_window_cond;
 * out [B,J,1,T] receives the final sample.  With the default AQL submission of the step loop (csrc/dsg_aql.h) the call
 * returns when the steps have run; with HIP launches (DSG_AQL=0, or under a profiler) it only enqueues and is asynchronous
 * w.r.t. the host when `out` is device memory. */
int dsg_sample(dsg_handle* h, const dsg_sample_args* args, float* out, int B, void* stream);
/* n lanes (a handle and its dsg_clone()s: one device, shared weights; n <= 16), one independent sampling call each, run
 * concurrently from this one host thread -- "one clip per stream": every lane owns an HSA queue, the dependent packet chains
 * of the lanes overlap on the GPU.  args[n], outs[n]; every lane samples a batch of B.  Every lane runs the kernel set of ITS
 * handle, so lane i's result is bit-identical to dsg_sample(lanes[i], &args[i], outs[i], B, stream) issued on its own. */
int dsg_sample_multi(dsg_handle** lanes, int n, const dsg_sample_args* args, float** outs, int B, void* stream);
/* Kernel set of a handle (DSG_KSET_*; sticky; clones inherit the source's at dsg_clone).  dsg_recommend_kernel_set: the set
 * measured fastest for `lanes` lanes of batch B advanced together (lanes = 1: what DSG_KSET_AUTO picks) -- several lanes share
 * the CUs and prefer the throughput-shaped sets earlier; the caller applies it to each lane.  dsg_last_kernel_set: the set the
 * last dsg_forward / dsg_sample of the handle ran. */
int dsg_set_kernel_set(dsg_handle* h, int set);
int dsg_get_kernel_set(dsg_handle* h, int* set);      /* the set in force (DSG_KSET_*), incl. a DSG_KSET environment pin */
int dsg_recommend_kernel_set(dsg_handle* h, int B, int lanes, int* set);
int dsg_last_kernel_set(dsg_handle* h, int* set);
int dsg_sync(dsg_handle* h);
/* time of the step loop of the last dsg_sample (HIP events on the handle's stream; AQL path: first doorbell to the completion
 * signal of the last packet), and its step count */
int dsg_last_sample_ms(dsg_handle* h, float* ms, int* n_steps);
/* how the step loop of the last dsg_sample was submitted: 0 = HIP launches, 1 = hand-written AQL packets, 2 = hipGraph replay */
int dsg_last_sample_path(dsg_handle* h, int* path);
/* 1 when the AQL packets of that loop carried no acquire / release fences: the buffers the loop writes live in uncached device
 * memory (default for handles of max_batch <= 16; DSG_UC=0 selects cached buffers + agent-scope fences) and the one-time
 * hand-off self-check of the device passed */
int dsg_last_sample_fence_free(dsg_handle* h, int* fence_free);
/* ABI 320.  The loop buffers of fence-free handles are sub-allocated from per-device arenas of uncached memory that the library
 * keeps across handles (a range that changed its caching attribute between two lives proved incoherent on MI355X / ROCm 7.2, so a
 * destroyed handle's blocks go back to the arena, not to hipFree).  dsg_trim hands every arena of `device` (< 0: all devices)
 * without a live block back to the HIP allocator -- for a long-lived service between bursts of work -- and reports the bytes
 * released / still held (either pointer may be null).  DSG_UC_POOL_CAP_MB (default 16384) bounds the arenas of a device; a handle
 * created past the cap gets cached loop buffers + fenced packets (dsg_last_sample_fence_free reports 0).  No reference counterpart
 * (torch's caching allocator: torch.cuda.empty_cache()).
 * HAZARD (stated, not solved): a trimmed range returns to the HIP allocator, i.e. the very recycling the pool exists to avoid can happen to
 * whoever allocates next -- a later uncached arena is fill / read-back checked before its first use, a CACHED allocation (the caller's tensors,
 * this library's weights) that lands on the range is not.  Call it between bursts, when no handle of the device is live or about to be
 * created, and prefer leaving the pool alone.  The caller's current HIP device is left as it was. */
int dsg_trim(int device, long long* bytes_released, long long* bytes_held);
/* the framework's noise stream as a tensor: out [B, J, 1, T] (device) = draw `draw` of (seed, stream_id), i.e. exactly the
 * noise the fused sampler uses for that draw index (x_T is draw_base, step i is draw_base + 1 + i).  Stands in for
 * th.randn / th.randn_like of gaussian_diffusion.py:704, :542 in the generic loop. */
int dsg_noise(float* out, int B, int J, int T, uint64_t seed, uint64_t stream_id, uint32_t draw, void* stream);

/* ZEGGS pose vectors -> BVH (pose2bvh of main/process/process_zeggs_bvh.py:219-275 and what it calls; host code, no GPU).
 * poses: host [frames, 1141], dtype 0 = float32, 1 = float64.  mean / std (float64[1141], both or neither): the sampler's
 * normalised output is de-normalised first, `poses * clip(std, 0.01) + mean` (sample.py:320-326); NULL: poses are taken as
 * they are.  smoothing != 0: Savitzky-Golay (15, 2) per feature (frames >= 15).  Output: `length` = frames, 3 * frames BVH
 * frames at 60 fps, 75 joints, text identical in layout to the reference writer's.
 * _channels: the numbers only -- offsets [75 * 3] (frame-0 positions, may be NULL) and motion [3 * frames, 228] in file order.
 * _batch: n_clips clips of `frames` frames, one file each, formatted on several host threads. */
int dsg_pose2bvh(const void* poses, int dtype, int frames, const double* mean, const double* std, int smoothing, const char* outpath);
int dsg_pose2bvh_channels(const void* poses, int dtype, int frames, const double* mean, const double* std, int smoothing,
                          double* offsets, double* motion);
int dsg_pose2bvh_batch(const void* poses, int dtype, int n_clips, int frames, const double* mean, const double* std, int smoothing,
                       const char* const* outpaths);

/* fused sampler arithmetic on caller tensors (flat fp32 arrays of B*per_batch elements, per-batch scalars on host) */
int dsg_q_sample(float* out, const float* x_start, const float* noise, const float* sqrt_ac, const float* sqrt_1mac,
                 int B, int64_t per_batch, void* stream);
int dsg_predict_xstart_from_eps(float* out, const float* x_t, const float* eps, const float* sqrt_recip,
                                const float* sqrt_recipm1, int B, int64_t per_batch, void* stream);
int dsg_posterior_step(float* out, const float* x_start, const float* x_t, const float* noise, const float* coef1,
                       const float* coef2, const float* sigma_nz, int B, int64_t per_batch, void* stream);
/* coef: host float[B*5] = {sqrt_recip, sqrt_recipm1, sqrt(abar_prev), sqrt(1-abar_prev-sigma^2), nonzero*sigma} */
int dsg_ddim_step(float* out, const float* x_start, const float* x_t, const float* noise, const float* coef, int B,
                  int64_t per_batch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSG_H_ */
