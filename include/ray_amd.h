/* ray_amd.h -- C ABI of the MI355X-native ray-intersection + shading path.
 *
 * This is the drop-in boundary for the hot path of joblobob/ray. The
 * reference has no FFI layer of its own: the path sits behind plain C++ free functions and
 * header-only classes (calc.h, hitPosition.h, shapes.h, camera.h) consumed by RayView. Every entry
 * point below therefore cites the reference interface it stands in for; the C++ mirror headers
 * under ray_amd/host/ (same names as the reference's) forward to these functions.
 *
 * Conventions
 *  - plain C, no C++/torch types; every function returns 0 on success, non-zero on error, and
 *    never throws. rt_last_error() returns the message for the calling thread's last failure.
 *  - pointers named d_* are DEVICE pointers (HBM) owned by the caller (rt_malloc, hipMalloc or a
 *    torch tensor's data_ptr()); everything else is host memory owned by the caller and may be
 *    freed as soon as the call returns.
 *  - one context per GPU, any number of GPUs per process: rt_init(device) creates the context of that HIP
 *    device (scene, camera, stream) on first use and makes it the CURRENT context of the calling host
 *    thread; rt_set_device(device) switches between initialised contexts. Every other entry point acts on
 *    the calling thread's current context. Calls are blocking (they return when their kernel has finished)
 *    unless RT_RENDER_NOWAIT is given. A context must not be used by two threads at the same time.
 *  - stream ordering: kernels run on the context's stream. By default that is a stream of the library's own
 *    (hipStreamNonBlocking), which does NOT wait for work the caller queued on other streams: a caller that
 *    fills or clears d_* buffers asynchronously (torch: tensor.zero_(), a collective) must either
 *    synchronise its stream before calling, or hand that stream to the library with rt_set_stream(), after
 *    which the library's kernels are ordered with the caller's work like any other launch on that stream.
 *  - the library never falls back to the CPU: without a usable HIP device every compute entry
 *    point fails with an error.
 */
#ifndef RAY_AMD_H
#define RAY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- primitive / material / light kinds ------------------------------------------------------ */
enum {
    RT_PRIM_SPHERE = 0,   /* shapes.h:67  sphere        data = {cx, cy, cz, r}                     */
    RT_PRIM_XY_RECT = 1,  /* shapes.h:161 rectangle     data = {x0, x1, y0, y1, k}   (plane z = k)  */
    RT_PRIM_XZ_RECT = 2,  /* shapes.h:226 xz_rect       data = {x0, x1, z0, z1, k}   (plane y = k)  */
    RT_PRIM_YZ_RECT = 3,  /* shapes.h:292 yz_rect       data = {y0, y1, z0, z1, k}   (plane x = k)  */
    RT_PRIM_TRIANGLE = 4, /* not in the reference       data = {v0xyz, v1xyz, v2xyz}               */
    /* shapes.h:374 constant_medium, by the shape of its boundary; d = density, data holds -1/d as the
     * reference stores it (neg_inv_density); the material must be RT_MAT_ISOTROPIC                    */
    RT_PRIM_MEDIUM_SPHERE = 5, /* boundary = sphere     data = {cx, cy, cz, r, neg_inv_density}    */
    RT_PRIM_MEDIUM_BOX = 6     /* boundary = box        data = {p0xyz, p1xyz, neg_inv_density}     */
};
enum {
    RT_MAT_LAMBERTIAN = 0,    /* hitPosition.h:197  rgb = albedo                                   */
    RT_MAT_METAL = 1,         /* hitPosition.h:222  rgb = albedo, fuzz (already clamped to <= 1)    */
    RT_MAT_DIELECTRIC = 2,    /* hitPosition.h:246  ir                                             */
    RT_MAT_DIFFUSE_LIGHT = 3, /* hitPosition.h:290  rgb = emitted colour                            */
    RT_MAT_ISOTROPIC = 4      /* hitPosition.h:316  rgb = albedo; the phase function of a constant_medium */
};
enum {
    RT_LIGHT_BASE = -1 /* a shape without pdf_value/random (shapes.h:56-62): value 0, dir (1,0,0)   */
    /* otherwise one of RT_PRIM_SPHERE / RT_PRIM_XY_RECT / RT_PRIM_XZ_RECT                          */
};
#define RT_PRIM_STRIDE 9 /* doubles per primitive in RtScene.prim_data                              */

/* Flattened scene: `worldObjects` (std::vector<std::shared_ptr<shape>>, rayview.h:62) in list
 * order, a `box` (shapes.h:339) expanded in place into its six sides in constructor order
 * (shapes.h:347-354). Closest-hit semantics are those of hitFromList (hitPosition.cpp:6-22):
 * the smallest t in [t_min, t_max] wins and among equal t the LATER list entry wins.
 * float-typed reference fields (sphere centre/radius, triangle vertices) are stored in the double
 * array exactly (every float is a double); rect bounds are doubles in the reference too.        */
typedef struct RtScene {
    int32_t n_prims;
    const int32_t* prim_kind;  /* [n_prims] RT_PRIM_*                                               */
    const int32_t* prim_mat;   /* [n_prims] index into the material arrays                          */
    const int32_t* prim_outer; /* [n_prims] index in the caller's worldObjects list                 */
    const int32_t* prim_side;  /* [n_prims] 0..5 for a box side, -1 otherwise                       */
    const double* prim_data;   /* [n_prims][RT_PRIM_STRIDE]                                         */
    int32_t n_mats;
    const int32_t* mat_kind;   /* [n_mats] RT_MAT_*                                                 */
    const float* mat_rgb;      /* [n_mats][3]                                                       */
    const float* mat_fuzz;     /* [n_mats]                                                          */
    const double* mat_ir;      /* [n_mats]                                                          */
    int32_t light_kind;        /* the `lights` shape of calc::ray_color (calc.h:92)                 */
    double light_data[RT_PRIM_STRIDE];
    float background[3];       /* calc::ray_color's `background` argument (calc.h:90)               */
} RtScene;

/* The eight values camera::resetCam derives (camera.h:57-63).                                     */
typedef struct RtCamera {
    float origin[3], lower_left_corner[3], horizontal[3], vertical[3];
    float u[3], v[3], w[3];
    double lens_radius;
} RtCamera;

/* flags for rt_render                                                                              */
#define RT_RENDER_ACCUMULATE 1u /* add to d_accum instead of overwriting it (progressive passes)    */
#define RT_RENDER_NORMALS 2u    /* drawOnlyColors mode of calc::ray_color (calc.cpp:105-109)        */
#define RT_RENDER_VERIFY 4u     /* self-check: every closest-hit scan is repeated as the reference's  */
                                /* list-order loop and compared (slow; see rt_verify_mismatches)      */
#define RT_RENDER_NOWAIT 8u     /* rt_render / rt_render_bands / rt_render_noalloc return as soon as  */
                                /* the kernel is queued; rt_sync() (or the next call on the context)  */
                                /* waits for it. Lets ONE host thread keep several GPUs busy.          */
#define RT_RENDER_CONTINUE 16u  /* rt_render / rt_render_bands / rt_render_noalloc: every pixel's sum goes on   */
                                /* from d_accum's contents IN SAMPLE ORDER, ((acc + c_s) + c_s+1) + ..., .w += */
                                /* spp_count: calls over consecutive sample ranges leave bit for bit what ONE  */
                                /* call over the whole range leaves, i.e. renderOneRay's in-order sum          */
                                /* (rayview.cpp:79-84). (ACCUMULATE adds the pass's own sum: acc + (c_s + ...).) */

/* ---- context ------------------------------------------------------------------------------------ */
/* Bind this process to HIP device `device` and create the library's stream. Fails (no CPU
 * fallback) when there is no such device.                                                          */
int rt_init(int device);
/* Number of HIP devices rt_init accepts (0 and an error when there is none).                               */
int rt_device_count(int* count);
/* Make the (already initialised) context of `device` the calling thread's current one. No counterpart in the
 * reference (one thread, no device); it is how one C++ caller drives several GPUs (ray_amd/host/render_multi.cpp). */
int rt_set_device(int device);
/* Run the current context's kernels on the caller's HIP stream (a hipStream_t, e.g. torch's
 * torch.cuda.current_stream().cuda_stream; NULL is HIP's default stream, which is torch's default stream)
 * instead of the library's own; rt_reset_stream() goes back to the library's. See "stream ordering" above.  */
int rt_set_stream(void* hip_stream);
int rt_reset_stream(void);
/* How rt_render / rt_render_bands / rt_render_noalloc cut a pixel's samples into the units of work their kernel hands out
 * (no counterpart in the reference, whose unit is a whole pixel on one thread, rayview.cpp:67-87). A pixel's samples
 * are rendered as consecutive SLICES whose sizes halve down to min_samples (default 8): large units while the frame is
 * in full swing, small ones at its end so that no lane waits long for another's last pixel; a slice continues the
 * pixel's sum in sample order from the accumulator, so every setting gives the same bits as whole pixels do.
 * min_samples = 0: whole pixels only. max_slices: at most that many slices per call (1..16, default 16).
 * Device memory this needs: 4 bytes per 8x8 tile of the frame (grown on demand, freed by rt_shutdown).                 */
int rt_set_sample_slices(int min_samples, int max_slices);
/* Upper limit (MiB; default 1400, 0 = none) of the colour buffer that rt_render / rt_render_bands keep for the END of a
 * frame (no counterpart in the reference). Up to half of a frame's tiles (all of them on small frames) are rendered last, in units of 8 samples per pixel whose colours (12 bytes each: 768 bytes per tile and
 * sample) go to this buffer and are added in sample order by a second kernel: the same bits as a pixel summed by one
 * thread, but parallel over a pixel's samples too, which is what keeps every lane fed while the frame runs out. The buffer
 * is allocated by the first frame that wants it, grown on demand up to the limit, and released by this call (when it
 * holds more than the new limit), by rt_upload_scene and by rt_shutdown; a device that cannot spare it gets a smaller
 * one or none. Where the limit leaves less than 4 / 9 of the buffer a frame would like (768 B x tiles x samples: 4K frames of
 * hundreds of samples), the other tiles are rendered in sample slices (rt_set_sample_slices) instead of as whole pixels.
 * Frames are bit-identical whatever the limit; only their last part is slower with less (4K / 256 spp on MI355X, tiles in order
 * of cost (rt_set_tile_order, the default): 378.5 - 379.5 ms with 1400 MiB .. 2 GiB, 389 with 1344 MiB and less; 1080p / 64 spp frames want 0.8 GiB). */
int rt_set_tail_limit(int64_t megabytes);
/* Tiles in order of cost (no counterpart in the reference). mode 1 (the default): rt_render / rt_render_bands calls of 64
 * samples and more whose frame has little filler for its end (under five tiles per resident wave in the colour buffer: 1080p frames, 4K
 * frames whose buffer the limit caps) render the first sample of every pixel as a pass of its own that also counts every tile's rays,
 * sort the tiles by that count on the device, and render the other samples with the heaviest tiles first (continuing the sums in sample
 * order: the bits of the plain call; no sample is rendered twice, none is skipped). mode 2: every call of 2 samples and more (tests). mode 0:
 * tiles in image order. Device memory: 20 bytes per 8x8 tile and a sort workspace, kept in the context. rt_last_kernel_ms covers all
 * launches of the call.                                                                                      */
int rt_set_tile_order(int mode);
/* Block-sorted shading, an experiment kept for the record: mode 1 shades the bounces of a block's 256 rays in
 * order of material, so that a wave executes one material's code instead of all of them (plain frames of scenes without triangles,
 * constant_medium or a rect flag pass; other calls ignore the mode). Frames are bit-identical either way. Measured on MI355X
 * it does not pay -- the reference's built-in scene at 1080p / 64 spp: lane utilisation 37 -> 48 %, 16 % fewer instructions, but the
 * two barriers per trip leave issue slots empty: 14.0 ms either way (1 - 2 % slower before the kernels lost the triangle scan); the
 * 485-sphere frame +8 % -- hence mode 0 is the default.
 * No counterpart in the reference.                                                                       */
int rt_set_shading_sort(int mode);
/* Destroys the calling thread's current context.                                                           */
void rt_shutdown(void);
const char* rt_last_error(void);
/* "name|gcnArch|CUs|clockMHz" of the bound device.                                                 */
int rt_device_info(char* buf, int buflen);

/* device memory helpers for hosts that have no allocator of their own                               */
int rt_malloc(void** d_ptr, uint64_t bytes);
int rt_free(void* d_ptr);
/* Page-locked HOST memory that the device reads and writes in place (over the host link): the pointer is valid on both sides and
 * may be passed wherever a d_* pointer is expected. For callers that trace a few rays per call (calc::ray_color with the reference's
 * signature, ray_amd/host/ray_amd_host.h): rays in, results out, no copy calls -- the host sees what a kernel wrote once the call that
 * launched it has returned (or rt_sync() after RT_RENDER_NOWAIT). Not for frames: every access crosses the link. No counterpart in
 * the reference.                                                                                     */
int rt_malloc_host(void** ptr, uint64_t bytes);
int rt_free_host(void* ptr);
int rt_memcpy_d2h(void* dst, const void* d_src, uint64_t bytes);
int rt_memcpy_h2d(void* d_dst, const void* src, uint64_t bytes);
int rt_memset(void* d_ptr, int value, uint64_t bytes);
int rt_sync(void);

/* ---- scene + camera ------------------------------------------------------------------------------ */
/* Replaces holding `m_worldObjects` / `worldLights` (rayview.h:62-63) by pointer: validates the
 * flattened scene, converts it to per-kind structure-of-arrays device buffers and uploads them.
 * Call again after any mutation (the sphere-Y slider, rayview.cpp:130-134).                        */
int rt_upload_scene(const RtScene* scene);

/* camera::resetCam (camera.h:32-54) with identical host arithmetic.                                */
int rt_camera_from_lookat(const float lookfrom[3], const float lookat[3], const float vup[3],
                          double focus_dist, double vfov, double aspect_ratio, double aperture,
                          RtCamera* out);
/* Replaces the by-value `camera cam` member (rayview.h:87).                                        */
int rt_set_camera(const RtCamera* cam);

/* ---- the hot path -------------------------------------------------------------------------------- */
/* Primary-ray closest hit for every pixel: camera::get_ray (camera.h:21-30) at the pixel centre,
 * u = (x + 0.5f) / w, v = (y + 0.5f) / h, then hitFromList(world, ray, 0.001f, inf)
 * (calc.cpp:99, hitPosition.cpp:6-22). Lens samples (aperture != 0) come from the counter RNG
 * stream (seed, pixel, sample 0). Outputs, row-major [h][w], any may be NULL:
 *   d_prim  flattened primitive index of the closest hit, -1 on a miss
 *   d_t     hit_record::t (double, bit pattern as the reference computes it), 0 on a miss        */
int rt_primary_hits(int w, int h, uint64_t seed, int32_t* d_prim, double* d_t);

/* hitFromList for caller-supplied rays (hitPosition.h:187-191): n rays, origin/direction as
 * [n][3] floats. Outputs [n]: prim (-1 = std::nullopt), t, and optionally p[n][3], normal[n][3],
 * front_face[n] of the hit_record (hitPosition.h:95-109). All pointers are device pointers.      */
int rt_hit_rays(int64_t n, const float* d_origin, const float* d_dir, double t_min, double t_max,
                int32_t* d_prim, double* d_t, float* d_p, float* d_normal, uint8_t* d_front);

/* RayView::renderAll (declared rayview.h:32) with the per-pixel contract of renderOneRay
 * (rayview.cpp:79-84): for every pixel (x, y) and every sample s in [spp_begin, spp_begin +
 * spp_count):  u = (x + xi) / w, v = (y + xi) / h, sum += calc::ray_color(cam.get_ray(u, v),
 * background, world, lights, depth). Random numbers come from the counter RNG stream keyed by
 * (seed, pixel y*w + x, s), so any split of the sample range over calls or GPUs draws the same
 * numbers. d_accum is [h][w][4] floats (r, g, b sums in sample order; .w = number of samples
 * accumulated), indexed by the un-mirrored (x, y) of renderOneRay. d_rays (may be NULL) receives
 * the number of closest-hit scans executed (primary + bounce rays), added atomically.
 * Pixels outside [y_begin, y_begin + y_count) are not touched (image-tile split).
 * Device memory the call keeps in the context (grown on demand, freed by rt_shutdown): the path stack beyond
 * six bounces (16 bytes x (depth - 6) per resident thread), 4 bytes per 8x8 tile of the frame for the order
 * between a pixel's sample slices (rt_set_sample_slices), and for spp_count >= 16 the colour buffer of the frame's
 * end, at most 1400 MiB unless rt_set_tail_limit says otherwise; 20 bytes per tile and a sort workspace for rt_set_tile_order. */
int rt_render(int w, int h, int y_begin, int y_count, int spp_begin, int spp_count, int depth,
              uint64_t seed, uint32_t flags, float* d_accum, uint64_t* d_rays);

/* rt_render for callers that cannot allocate device memory. NO COUNTERPART IN THE REFERENCE: it exists for
 * tight serving loops, harnesses that forbid hipMalloc and other code that must not allocate (the call itself still
 * synchronises its stream and reads event times, so it is NOT capturable into a HIP graph as it stands). Same arguments, same
 * kernels and bit-identical results as rt_render, but the kernel's scratch (the tile counter and the HBM part
 * of the per-thread path stack, which rt_render keeps in buffers it allocates and grows on demand) lives in a
 * file-scope device array of the library: 44 MiB + 256 B on every GPU the library is bound to, held until the
 * process ends. The call makes no hipMalloc / hipFree and calls no allocating routine (so it renders without
 * rt_render's colour buffer for the end of the frame: same result, a slower last part). The array also holds the
 * per-tile words of the sample slices for frames of up to 2^20 tiles (67 Mpixel; larger frames: whole pixels).
 * Limits: depth <= 16 (RT_NOALLOC_MAX_DEPTH; deeper paths are rejected with an error, use rt_render);
 * RT_RENDER_VERIFY is rejected. Every call on one GPU shares that GPU's array: calls that overlap in time on
 * one GPU race (the library issues them on one stream, so calls from one host thread never do).             */
#define RT_NOALLOC_MAX_DEPTH 16
int rt_render_noalloc(int w, int h, int y_begin, int y_count, int spp_begin, int spp_count, int depth,
                      uint64_t seed, uint32_t flags, float* d_accum, uint64_t* d_rays);

/* rt_render for an interleaved share of the image: the frame is cut into bands of 8 rows (band b =
 * rows [8b, 8b + 8)), and the call renders bands band_first, band_first + band_stride, ... Other rows
 * are not touched. With band_first = rank and band_stride = number of GPUs every GPU gets the same mix
 * of sky and geometry rows (the balanced form of the image-tile split; the reference renders random
 * pixels on one thread, rayview.cpp:72-76, so it has no counterpart). Same arguments otherwise.     */
int rt_render_bands(int w, int h, int band_first, int band_stride, int spp_begin, int spp_count, int depth,
                    uint64_t seed, uint32_t flags, float* d_accum, uint64_t* d_rays);

/* RayView::renderOneRay (rayview.cpp:67-87) for a LIST of pixels: the reference's progressive view draws
 * random pixels (randX, randY, rayview.cpp:69-70), renders numSamples samples of each and overwrites that
 * pixel. d_pixels holds n pixel indices y * w + x (each in [0, w*h); the caller draws them, e.g. with
 * calc::random_doubleW / H); every listed pixel gets samples [spp_begin, spp_begin + spp_count) exactly as
 * rt_render computes them, its accumulator entry is overwritten (added to with RT_RENDER_ACCUMULATE), and
 * no other pixel is touched. A pixel listed twice is rendered twice: overwriting, both writes carry the same
 * values; with RT_RENDER_ACCUMULATE each listing adds its sums and its sample count once (atomic adds of
 * identical values, so the result does not depend on their order). An index outside [0, w*h) is skipped.
 * The ORDER of the list does not matter for the result and, from 2048 entries on, not for the speed either: the
 * library renders a copy sorted by 8x8 tile (device-side radix sort on the same stream, in a workspace that it
 * allocates once and grows when a longer list comes); d_pixels itself is only read.                         */
int rt_render_pixels(int w, int h, int64_t n, const int32_t* d_pixels, int spp_begin, int spp_count, int depth,
                     uint64_t seed, uint32_t flags, float* d_accum, uint64_t* d_rays);

/* Number of closest-hit scans of the last RT_RENDER_VERIFY render whose fast result (primitive or t
 * bits) differed from hitFromList's list-order loop (hitPosition.cpp:6-22) run on the same ray. No
 * counterpart in the reference: it checks the sphere filter of rt_scan.hpp at full frame sizes.     */
int rt_verify_mismatches(uint64_t* count);

/* calc::ray_color (calc.h:89-94, calc.cpp:90-131) for caller-supplied rays: n rays, origin/direction
 * as [n][3] floats, against the uploaded world/lights/background; ray i draws from the counter RNG
 * stream (seed, i, 0). d_rgb receives [n][3] linear radiance. flags: RT_RENDER_NORMALS.            */
int rt_trace_rays(int64_t n, const float* d_origin, const float* d_dir, int depth, uint64_t seed, uint32_t flags,
                  float* d_rgb);

/* calc::rgbPerSamples + RayView::writeToImg (calc.cpp:133-154, rayview.cpp:57-65,86): gamma-2
 * tonemap of sum / n_samples (n from accum.w when n_samples <= 0), clamp, pack as
 * QImage::Format_RGB32 (0xffRRGGBB) at the mirrored position (w-1-x, h-1-y).                      */
int rt_tonemap(int w, int h, const float* d_accum, int n_samples, uint32_t* d_rgb32);

/* Time (ms) the last rt_render / rt_primary_hits / rt_hit_rays kernel took on the library's
 * stream, measured with HIP events around the launch; and the number of launches it made.         */
int rt_last_kernel_ms(float* ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* RAY_AMD_H */
