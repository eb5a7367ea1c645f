/*
 * mipt.h -- C ABI of the MI355X path-tracing hot path (libmipt.so).
 *
 * The library replaces, behind the reference raytracer's own scene model, the
 * three seams of its unity build:
 *     closestIntersection / existsIntersection   CW/raytracer/Source/intersections.h:7-13
 *     trace (one path per pixel sample)           CW/raytracer/Source/pathtracer.h:4
 *     Draw  (pixel loop + tonemap to the screen)  CW/raytracer/Source/definitions.h:61,
 *                                                 body CW/raytracer/Source/main.cpp:110-146
 * Everything crossing the boundary is plain C: pointers, sizes, POD structs.
 * All entry points return MIPT_OK (0) or a negative MiptError; nothing throws.
 * mipt_last_error() gives a thread-local message for the last failure.
 *
 * Threading: one MiptScene handle is externally synchronised (one host thread at
 * a time); different handles may be used concurrently. A scene owns scratch memory
 * on its GPU (work-queue heads, per-item radiance sums, counters) that every render
 * and tree-batch launch uses: launches of one scene are ordered on the device in
 * call order, whichever streams they were given (each waits for an event its
 * predecessor recorded), so frames or streams sharing a scene never overlap there.
 * The entry points that take HOST pointers stage their arguments in device memory the
 * scene keeps and grows on demand (no allocation per call once it is large enough) and
 * return after the device has finished; the *_dev variants neither copy nor wait.
 *
 * Arithmetic contract: IEEE-754 binary32, every product and sum separately
 * rounded (no FMA contraction), correctly rounded division and square root,
 * subnormals kept -- the reference source evaluated under strict IEEE rules.
 */
#ifndef MIPT_H
#define MIPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum MiptError {
  MIPT_OK = 0,
  MIPT_E_INVALID = -1,     /* bad argument (null pointer, size out of range, unsupported mode) */
  MIPT_E_HIP = -2,         /* a HIP runtime call failed; mipt_last_error() names it */
  MIPT_E_NOMEM = -3,       /* host or device allocation failed */
  MIPT_E_UNSUPPORTED = -4, /* valid request this build cannot serve (e.g. oracle-only RNG mode) */
  MIPT_E_RCCL = -5         /* librccl.so could not be loaded, or a collective failed */
} MiptError;

/* Material of a primitive. The reference stores a `const char* type`
 * (CW/raytracer/Source/material.h:5-11) and dispatches with strcmp
 * (pathtracer.cpp:88,103-110); the host adapter maps the four strings to this
 * enum. Any other string behaves as MIPT_MAT_NONE: the path ends with zero
 * radiance at a non-terminal vertex and is shaded like diffuse at a terminal one. */
typedef enum MiptMaterial {
  MIPT_MAT_DIFFUSE = 0,
  MIPT_MAT_MIRROR = 1,
  MIPT_MAT_GLASS = 2,
  MIPT_MAT_SPECULAR = 3,
  MIPT_MAT_NONE = 4
} MiptMaterial;

/* Field-for-field `Triangle` (CW/raytracer/Source/TestModelH.h:11-49), 80 bytes. */
typedef struct MiptTriangle {
  float v0[4];
  float v1[4];
  float v2[4];
  float color[3];
  float normal[4];
  uint32_t material; /* MiptMaterial */
} MiptTriangle;

/* `Sphere` (CW/raytracer/Source/sphere.h:9-16), 36 bytes. */
typedef struct MiptSphere {
  float radius;
  float center[4];
  float color[3];
  uint32_t material; /* MiptMaterial */
} MiptSphere;

/* `Camera` (CW/raytracer/Source/definitions.h:16-23) without yaw/pitch, which the
 * hot path never reads. Matrices are column-major like glm::mat4: rot[4*col + row]. */
typedef struct MiptCamera {
  float focal;
  float pos[4];
  float rot_x[16];
  float rot_y[16];
} MiptCamera;

/* `Light` (CW/raytracer/Source/definitions.h:34-37). */
typedef struct MiptLight {
  float pos[4];
  float color[3];
} MiptLight;

typedef enum MiptRngMode {
  /* Per path: libstdc++ minstd_rand0 + generate_canonical<float,24>
   * (the reference's `std::default_random_engine generator` /
   * `uniform_real_distribution<float>(0,1)`, definitions.h:40-41), re-seeded for
   * every (pixel, sample) from mipt_path_key(). Draw order inside a path is the
   * reference's (g++ right-to-left argument evaluation at pathtracer.cpp:8). */
  MIPT_RNG_MINSTD_PATH = 0,
  /* Fully counter-based: draw k of path (pixel, sample) is a hash of
   * (seed, pixel, sample, k). No sequential state at all. */
  MIPT_RNG_COUNTER = 1,
  /* One engine shared by all pixels in x-major loop order, default seed 1: the
   * reference's single-threaded behaviour. Sequential by construction, so only
   * the CPU oracle implements it; the GPU library returns MIPT_E_UNSUPPORTED. */
  MIPT_RNG_MINSTD_GLOBAL = 2
} MiptRngMode;

enum {
  MIPT_FLAG_STATS = 1u /* count rays / triangle tests / gate passes (slower kernel variant) */
};

#define MIPT_MAX_DEPTH_LIMIT 16

/* Compile-time constants of the reference (definitions.h:43-55) as run-time
 * parameters; mipt_params_default() fills in the reference's values. */
typedef struct MiptParams {
  int32_t width;         /* SCREEN_WIDTH  */
  int32_t height;        /* SCREEN_HEIGHT */
  int32_t max_depth;     /* MAX_DEPTH 10; 0..MIPT_MAX_DEPTH_LIMIT */
  float rr_prob;         /* RUSSIAN_ROULETTE_FACTOR 0.15f */
  float diffuse_weight;  /* DIFFUSE_MAGIC_NUMBER 0.5f */
  float glass_weight;    /* GLASS_MAGIC_NUMBER 1.15f */
  float ior_glass;       /* GLASS_INDEX_OF_REFRACTION 1.512f */
  float ior_air;         /* AIR_INDEX_OF_REFRACTION 1.000f */
  uint32_t rng_mode;     /* MiptRngMode */
  uint32_t seed;
  /* Homogeneous fog. Not in the reference source (its README lists it as a
   * feature only); sigma_t == 0 disables it and leaves the RNG stream untouched. */
  float fog_sigma_t;
  float fog_albedo;
  uint32_t flags;        /* MIPT_FLAG_* */
} MiptParams;

typedef struct MiptStats {
  uint64_t paths;          /* pixel samples traced */
  uint64_t closest_rays;   /* closestIntersection calls */
  uint64_t shadow_rays;    /* existsIntersection calls */
  uint64_t tri_tests;      /* determinant-ratio gate evaluations */
  uint64_t gate_passes;    /* tests that went on to the inverse path */
  uint64_t sphere_tests;
  uint64_t fog_events;     /* scattering events in the medium */
  uint64_t diffuse_bounces; /* bounceDiffuse calls */
  uint64_t mirror_bounces;  /* bounceMirror calls */
  uint64_t glass_bounces;   /* bounceGlass calls */
  uint64_t bvh_redo;        /* LBVH scenes: rays re-run through the exhaustive traversal */
  uint64_t bvh_overflow;    /* LBVH scenes: rays whose stack / hit list overflowed (must be 0) */
  uint64_t bvh_nodes;       /* LBVH scenes: four-wide nodes fetched (tri_tests then counts leaves visited) */
  double kernel_ms;        /* device time of the last render call's path kernel */
  uint64_t specular_bounces; /* bounceSpecular calls */
} MiptStats;

typedef struct MiptScene MiptScene; /* opaque */

void mipt_params_default(MiptParams* p, int width, int height);

/* The reference's built-in scenes: which = 0 LoadTestModel, 1 LoadTestModelColorBleeding,
 * 2 LoadTestModelGlassCube, 3 LoadTestModelNone (TestModelH.h:55-201, 203-349, 351-482,
 * 484-580), bit-for-bit the triangles those functions produce. Writes at most `capacity`
 * triangles (30 suffice) and returns the count (30 / 30 / 22 / 10) or a MiptError. */
int mipt_load_test_model(int which, MiptTriangle* out, size_t capacity);
/* reset_camera (camera.cpp:4-13) with focalLength = width. */
void mipt_default_camera(MiptCamera* cam, int width);
/* reset_light (light.cpp:6-9); glass_scene != 0 gives the light main.cpp:68-69 sets for
 * the glassCube / glassCubeSphere / glassWolf scenes. */
void mipt_default_light(MiptLight* light, int glass_scene);

/* LoadObject (parser.cpp:8-46) over the objl loader it drives (OBJ_Loader.h:430-992): the triangles
 * of a Wavefront OBJ file in file order, bit for bit what the reference builds -- positions through
 * strtof, w = 1, colour (1,1,1), material diffuse, normal = (x, y of the first vertex' normal, z of
 * the SECOND vertex' normal, 1), a face without normals carrying the unnormalised (P0-P1) x (P2-P1),
 * polygons cut the way objl cuts them. Returns the number of triangles in the file (call with
 * out = NULL to size the buffer; at most `capacity` are written) or a MiptError. */
int mipt_load_obj(const char* path, MiptTriangle* out, size_t capacity);
/* normaliseTriangles (utils.cpp:52-140), in place: divide all four components of every vertex by
 * scale * L (L = largest |coordinate| of the set), add the displacement, rotate by Rx Ry Rz; recompute
 * a normal whose x is exactly 0, scale normals by adjust_normal, rotate them, set the material.
 * main.cpp:84-88 places the wolf with (1, {-0.2,-1,-0.5}, {3.15,-2.5,0}, {-1,1,1}, glass). */
int mipt_normalise_triangles(MiptTriangle* tris, size_t n, float scale, const float displacement[3],
                             const float rotate[3], const float adjust_normal[3], uint32_t material);

/* SDL_SaveImage (SDLauxiliary.h:25-54): the ARGB8888 screen buffer as the 32-bit BI_BITFIELDS,
 * BITMAPV4HEADER, bottom-up .bmp that SDL_SaveBMP writes for the reference's surface. */
int mipt_save_bmp(const char* path, const uint32_t* screen_buffer, int width, int height);

const char* mipt_last_error(void);
/* "mipt <version> gfx950 hip" */
const char* mipt_version(void);

/* Copies the host scene (main.cpp:46-47 `vector<Triangle> triangles`,
 * `vector<Sphere> spheres`, immutable after load) to `device` and prepares the
 * per-triangle edge/cofactor table the kernels read. The caller keeps its arrays. */
int mipt_scene_create(const MiptTriangle* tris, size_t n_tris,
                      const MiptSphere* spheres, size_t n_spheres,
                      int device, MiptScene** out);
/* Scenes of MIPT_BVH_MIN_TRIS triangles or more get an LBVH built on the GPU at creation
 * (the reference has none: main.cpp:19 lists "KD trees?" as a TODO and always scans linearly);
 * results are those of the linear scan, with one known exception: a triangle whose box the ray
 * never enters (or enters only beyond the nearest hit) but whose reference test is so ill
 * conditioned for that ray that rounding noise alone makes it accept; the tree does not look at
 * such a triangle. mipt_debug_selfcheck counts rays that differ: none among 10^9 camera, interior
 * and on-surface rays per tested scene; 3 to 4 per million among rays constructed to graze a
 * triangle's plane just outside its box.
 * mipt_scene_has_bvh returns 1 / 0, the count of four-wide 64-byte nodes in *n_nodes (nullable)
 * and the build time in *build_ms (nullable). */
#define MIPT_BVH_MIN_TRIS 256
/* mipt_scene_create with options. MIPT_SCENE_LINEAR_SCAN: never build the tree -- every query scans
 * all triangles in index order like intersections.cpp:11-37 does, whatever the scene's size (slow
 * beyond a few thousand triangles; it is the on-device cross-check of the tree). */
enum { MIPT_SCENE_LINEAR_SCAN = 1u };
int mipt_scene_create_opts(const MiptTriangle* tris, size_t n_tris,
                           const MiptSphere* spheres, size_t n_spheres,
                           int device, uint32_t options, MiptScene** out);
int mipt_scene_has_bvh(MiptScene* scene, uint64_t* n_nodes, double* build_ms);
/* How the tree was built (no counterpart in the reference). Two binary trees are built over the
 * Morton-sorted triangles -- one by locally-ordered clustering (neighbours whose joint box has the
 * smallest surface merge, round by round), one from the bits of the keys (the radix tree) -- and the
 * clustered one is collapsed into the four-wide nodes if its surface-area cost is below 0.9 of the
 * radix tree's, else the radix tree. *clustered = 1 if that
 * was the clustered tree, *depth = the clustered tree's depth in binary levels (it is also dropped if
 * deeper than the walk's stack is sized for: 47; 0 = the clustering was given up after 512 rounds, which
 * only a scene that clusters into a chain needs), *cost_ratio = its cost / the radix tree's (0 if the
 * clustered tree was dropped before the comparison). All
 * nullable; returns 1 / 0 like mipt_scene_has_bvh. Query results do not depend on the choice. */
int mipt_scene_tree_info(MiptScene* scene, int* clustered, int* depth, float* cost_ratio);
/* Linear-scan scenes: how many pairs of consecutive coplanar triangles sharing an edge (the two
 * halves of a wall or box face, TestModelH.h:87-160) the candidate filter treats as one unit.
 * Purely an optimisation of the filter; every candidate still goes through the reference's test. */
int mipt_scene_pair_units(MiptScene* scene);
/* ... and how many of those pairs lie in a plane normal to a coordinate axis (walls, floors, box
 * tops): the filter solves their ray/plane system without cross products. No counterpart in the
 * reference either. */
int mipt_scene_axis_pair_units(MiptScene* scene);
/* ... and how many groups of at least three such pairs are the faces of one parallelepiped (the
 * room's five walls, the five faces of a block, TestModelH.h:87-106, 108-139, 141-172): the filter maps the ray
 * into the parallelepiped's frame once and tests all its faces there. The pairs inside such a
 * group still count in mipt_scene_pair_units. No counterpart in the reference. */
int mipt_scene_box_units(MiptScene* scene);
/* The same three counts for a list of triangles scanned linearly, computed on the host alone (no
 * GPU, no scene): how the filter would group them. Outputs are nullable. No counterpart in the
 * reference. */
int mipt_plan_scan_units(const MiptTriangle* tris, size_t n_tris, int* pair_units, int* axis_pair_units,
                         int* box_units);
void mipt_scene_destroy(MiptScene* scene);

/* Batch closestIntersection (intersections.cpp:6-55). Host pointers.
 * tri_idx[i] = -1 on a miss, the triangle index, or n_tris + s for sphere s;
 * t[i] = Intersection::distance (|x.x| of the inverse path; untouched on a miss);
 * position4 (nullable) = Intersection::position. Bit-exact contract. */
int mipt_closest_hit(MiptScene* scene, const float* origin4, const float* dir4,
                     size_t n_rays, int32_t* tri_idx, float* t, float* position4);

/* Batch existsIntersection (intersections.cpp:57-85): triangles only, gate
 * t <= max_t[i], barycentric bound <= 1. occluded[i] is 0 or 1. */
int mipt_any_hit(MiptScene* scene, const float* origin4, const float* dir4,
                 const float* max_t, size_t n_rays, uint8_t* occluded);

/* Replaces the pixel loop of Draw (main.cpp:121-139) for samples
 * [sample_begin, sample_begin + sample_count) of every pixel, and ADDS the
 * radiance sums into the caller's accumulation buffer: accum_rgb is a host
 * float[width*height*3], pixel (x, y) at ((x*height)+y)*3 -- x-major, like the
 * reference's `vec3** pixels` columns (main.cpp:55-57).
 * Association of the sums: a pixel's samples are added 16 at a time (in sample order, from zero), and
 * these sums join the buffer's running value one by one in sample order. Adding [a, b) and then
 * [b, c) therefore leaves the very bits that adding [a, c) in one call leaves whenever b - a is a
 * multiple of 16; a split elsewhere agrees to rounding. */
int mipt_render_accumulate(MiptScene* scene, const MiptCamera* cam, const MiptLight* light,
                           const MiptParams* params, uint32_t sample_begin,
                           uint32_t sample_count, float* accum_rgb);

/* The tonemap loop of Draw (main.cpp:141-145) with PutPixelSDL
 * (SDLauxiliary.h:149-161): c = accum/(N*(1/(2*3.1415f))), clamp(255c, 0, 255),
 * screen_buffer[y*width + x] = (128<<24) + (r<<16) + (g<<8) + b. Host pointers. */
int mipt_resolve_argb(MiptScene* scene, const float* accum_rgb, int width, int height,
                      uint32_t n_samples, uint32_t* screen_buffer);

/* Radiance of individual paths, for matched-stream parity checks: path i is
 * pixel (px[i], py[i]), sample index sample[i]; rgb gets 3 floats per path,
 * ray_counts (nullable) gets {closest calls, shadow calls} per path. Host pointers. */
int mipt_trace_paths(MiptScene* scene, const MiptCamera* cam, const MiptLight* light,
                     const MiptParams* params, const int32_t* px, const int32_t* py,
                     const uint32_t* sample, size_t n_paths, float* rgb, uint32_t* ray_counts);

/* Device-pointer variants: every buffer already lives in the memory of the scene's
 * GPU, `stream` is a hipStream_t (NULL = default stream). Launch is asynchronous;
 * the caller synchronises the stream. (Batches of 2^20 rays or more against a tree
 * scene are probed for coherence in passing; once a scene's batches have shown
 * scattered origins, later ones are walked in the order of their origins' cells on a
 * 16^3 grid. Results land at the rays' own indices and do not depend on it.) Used by the one-process-per-GPU driver so
 * that the accumulation buffer can be handed to RCCL without a host round trip. */
int mipt_closest_hit_dev(MiptScene* scene, const float* d_origin4, const float* d_dir4,
                         size_t n_rays, int32_t* d_tri_idx, float* d_t, float* d_position4,
                         void* stream);
int mipt_any_hit_dev(MiptScene* scene, const float* d_origin4, const float* d_dir4,
                     const float* d_max_t, size_t n_rays, uint8_t* d_occluded, void* stream);
int mipt_render_accumulate_dev(MiptScene* scene, const MiptCamera* cam, const MiptLight* light,
                               const MiptParams* params, uint32_t sample_begin,
                               uint32_t sample_count, float* d_accum_rgb, void* stream);
int mipt_resolve_argb_dev(MiptScene* scene, const float* d_accum_rgb, int width, int height,
                          uint32_t n_samples, uint32_t* d_screen_buffer, void* stream);

/* The two halves of mipt_render_accumulate_dev; no counterpart in the reference (its OpenMP loop,
 * main.cpp:121, shares one image). mipt_render_items_dev traces the samples and leaves the sums of
 * the work items (16 samples each) in the scene's scratch; mipt_fold_items_dev joins them to
 * d_accum_rgb's running values one by one in sample order, on any stream (it waits for the render).
 * Together they are mipt_render_accumulate_dev, bit for bit. Apart, they let several GPUs produce
 * the image ONE GPU produces, bit for bit: every GPU renders the items of its share of the sample
 * range at once, then the image travels from GPU to GPU in the order of the shares and each folds
 * its items into it (computer-graphics_amd/parallel.py: render_chained). Shares must be whole
 * work items (multiples of 16 samples; the last may be ragged). A scene holds the items of one
 * render at a time: a second mipt_render_*_dev before the fold returns MIPT_E_INVALID; a fold with
 * nothing pending leaves the buffer as it is. */
int mipt_render_items_dev(MiptScene* scene, const MiptCamera* cam, const MiptLight* light,
                          const MiptParams* params, uint32_t sample_begin, uint32_t sample_count,
                          void* stream);
int mipt_fold_items_dev(MiptScene* scene, float* d_accum_rgb, void* stream);

/* mipt_render_accumulate_dev for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc). It has no counterpart in the
 * reference. Same arguments, same results bit for bit; the path kernel's work-queue head and
 * per-item radiance sums live in a fixed array inside the library's code object instead of the
 * scene's growable scratch, and the call makes no hipMalloc / hipFree / event call: one memset, the
 * path kernel and the ordered sum, all on `stream`.
 *   Limit: width * height * 12 bytes * ceil(sample_count / 16) <= MIPT_NOALLOC_MAX_BYTES
 *   (1024 x 1024 at up to 32 samples per call, 512 x 512 at up to 160); larger calls return
 *   MIPT_E_INVALID -- split the sample range over several calls of a multiple of 16 samples: the
 *   ranges compose bit for bit (see mipt_render_accumulate).
 *   The array is shared by every scene and stream on a GPU (each GPU has its own copy, resident
 *   until the process ends): two such calls that overlap on one GPU race; issue them on one stream
 *   or order them yourself. The same holds for such a call and ANY other render call of the same
 *   tree scene (a scene of MIPT_BVH_MIN_TRIS triangles or more keeps its paths' vertex stacks in
 *   device memory it allocated at creation). mipt_get_stats is not updated by this call, and
 *   MIPT_FLAG_STATS is refused (the counters are scene scratch ordered by events this call does not
 *   use). */
#define MIPT_NOALLOC_MAX_BYTES (32u << 20)
int mipt_render_accumulate_noalloc_dev(MiptScene* scene, const MiptCamera* cam, const MiptLight* light,
                                       const MiptParams* params, uint32_t sample_begin,
                                       uint32_t sample_count, float* d_accum_rgb, void* stream);

/* Progressive rendering with the image resident on the device: `vec3** pixels` + `samplesSeenSoFar`
 * (main.cpp:50-56) as one object, so that a frame of the interactive loop (main.cpp:98-103) moves
 * only the ARGB screen over PCIe.
 *   mipt_frame_reset        reset_evolutionModel (utils.cpp:24-32): what handleKeyboard calls when
 *                           the camera or the light moved
 *   mipt_frame_add_samples  Draw's pixel loop (main.cpp:121-139) for n more samples of every pixel,
 *                           continuing the sample index; asynchronous (the frame's own stream)
 *                           (added 16, or a multiple of 16, at a time the image is at every moment the
 *                           one a single call for all its samples would have left, bit for bit)
 *   mipt_frame_present      Draw's tonemap loop (main.cpp:141-145) into the host screen buffer; waits
 *                           for the samples added so far
 *   mipt_frame_read_pixels  the accumulated sums, x-major W*H*3 floats (the reference's `pixels`)
 * The frame uses its scene's scratch memory: one host thread per scene, as for the other calls;
 * several frames of one scene may be driven in turn, their launches run one after the other.
 * mipt_frame_destroy may be called after the scene was destroyed. */
typedef struct MiptFrame MiptFrame;
int mipt_frame_create(MiptScene* scene, int width, int height, MiptFrame** out);
void mipt_frame_destroy(MiptFrame* frame);
int mipt_frame_reset(MiptFrame* frame);
int mipt_frame_add_samples(MiptFrame* frame, const MiptCamera* cam, const MiptLight* light,
                           const MiptParams* params, uint32_t n_samples);
int mipt_frame_present(MiptFrame* frame, uint32_t* screen_buffer);
int mipt_frame_read_pixels(MiptFrame* frame, float* accum_rgb);
int mipt_frame_samples(const MiptFrame* frame); /* samplesSeenSoFar */
/* Non-blocking: 1 when every sample added so far has been rendered (mipt_frame_present would not
 * wait), 0 while the GPU is still at it, or a MiptError. Lets the event loop of main.cpp:98-103 keep
 * handling input (Update / handleKeyboard) while a long add_samples call runs. */
int mipt_frame_poll(MiptFrame* frame);

/* Several GPUs of one node behind one render call, from ONE host process -- what replaces the
 * reference's only parallel construct, `#pragma omp parallel for` over image columns (main.cpp:121).
 * mipt_multi_create puts a replica of the scene on every listed device. mipt_multi_render_accumulate
 * is mipt_render_accumulate (host accumulation buffer, samples ADDED): the sample range is split into
 * contiguous parts, one per device (mipt_multi_sample_range; a path's random numbers depend on the
 * global sample index only, so the parts are exactly the paths one GPU would trace), each device
 * renders its part on its own stream, and ONE ncclReduce(sum, float32, 3 W H) over RCCL adds the
 * per-device images on the first device -- which has rendered its own part straight onto the caller's
 * running sums, so the reduce (in place at the root) leaves the new total and the host adds nothing.
 * The image equals the single-GPU image up to the order of the float additions. librccl.so is loaded
 * on first use; MIPT_E_RCCL if that or a collective fails. A call that fails half-way drains every
 * device and leaves the replicas ready for the next call. */
typedef struct MiptMulti MiptMulti;
int mipt_multi_create(const MiptTriangle* tris, size_t n_tris, const MiptSphere* spheres, size_t n_spheres,
                      const int* devices, int n_devices, MiptMulti** out);
void mipt_multi_destroy(MiptMulti* multi);
int mipt_multi_device_count(const MiptMulti* multi);
/* mipt_multi_create with options. MIPT_MULTI_ALLOW_SHARED_DEVICE: a device may be listed more than once,
 * each entry with a scene replica and a stream of its own -- for rehearsing the several-device code
 * (sample split in work items, event chain, peer copies) on a machine with one GPU. RCCL takes one
 * rank per device, so such a scene has no communicator: the relay calls
 * (mipt_multi_render_accumulate_chained, MIPT_MULTI_FRAME_EXACT frames) work, the sum-reduce ones
 * return MIPT_E_UNSUPPORTED. No counterpart in the reference. */
enum { MIPT_MULTI_ALLOW_SHARED_DEVICE = 1u };
int mipt_multi_create_opts(const MiptTriangle* tris, size_t n_tris, const MiptSphere* spheres, size_t n_spheres,
                           const int* devices, int n_devices, uint32_t options, MiptMulti** out);
/* the replica on the rank-th listed device (owned by `multi`), e.g. for mipt_resolve_argb; NULL if out of range */
MiptScene* mipt_multi_scene(MiptMulti* multi, int rank);
int mipt_multi_render_accumulate(MiptMulti* multi, const MiptCamera* cam, const MiptLight* light,
                                 const MiptParams* params, uint32_t sample_begin, uint32_t sample_count,
                                 float* accum_rgb);
/* share of device `rank` out of `n`: contiguous, the first count % n ranks get one sample more */
void mipt_multi_sample_range(uint32_t begin, uint32_t count, int rank, int n, uint32_t* part_begin,
                             uint32_t* part_count);
/* The same call without the collective and without a rounding difference: the sample range is split in
 * whole work items (mipt_multi_item_range: multiples of 16 samples, the last share in use takes
 * the ragged rest), all devices trace at once (mipt_render_items_dev), and the caller's image goes to
 * the first device and on from device to device by peer copies, each folding its items in
 * (mipt_fold_items_dev). The additions happen in the order one device performs them: the result is
 * mipt_render_accumulate's on ONE GPU, bit for bit, whatever the number of devices. */
int mipt_multi_render_accumulate_chained(MiptMulti* multi, const MiptCamera* cam, const MiptLight* light,
                                         const MiptParams* params, uint32_t sample_begin, uint32_t sample_count,
                                         float* accum_rgb);
void mipt_multi_item_range(uint32_t begin, uint32_t count, int rank, int n, uint32_t* part_begin,
                           uint32_t* part_count);

/* mipt_frame_* on several GPUs: `vec3** pixels` + `samplesSeenSoFar` (main.cpp:50-60) resident on the
 * devices of a MiptMulti, so that a frame of the interactive loop (main.cpp:96-101) launches work on
 * every GPU and moves only the ARGB screen to the host -- no image crosses PCIe, nothing is added on
 * the host.
 *   mode MIPT_MULTI_FRAME_REDUCE  every add_samples splits its samples over the devices
 *                                 (mipt_multi_sample_range); the first device renders straight into
 *                                 the frame's image and ONE ncclReduce(sum) brings the others' in.
 *   mode MIPT_MULTI_FRAME_EXACT   the relay of mipt_multi_render_accumulate_chained: whole work items
 *                                 per device, the image handed on by peer copies; it stays on the
 *                                 device where its round ended and the next round starts there. The
 *                                 image is the ONE-GPU frame's (mipt_frame_*), bit for bit, for
 *                                 additions of a multiple of 16 samples.
 *   mipt_multi_frame_add_samples  asynchronous: returns once the work is queued on the devices' streams
 *   mipt_multi_frame_present      Draw's tonemap loop (main.cpp:141-145) on the device holding the
 *                                 image; waits for the samples added so far
 *   mipt_multi_frame_reset        reset_evolutionModel (utils.cpp:24-32)
 *   mipt_multi_frame_poll         1 when every device has finished what was queued, 0 while busy
 *   mipt_multi_frame_device       index (into the multi's device list) of the device holding the image
 * A call that fails half-way leaves the image undefined: the frame refuses further samples until it
 * is reset. Destroy a frame before its MiptMulti. One host thread per MiptMulti, as for scenes. */
typedef struct MiptMultiFrame MiptMultiFrame;
enum { MIPT_MULTI_FRAME_REDUCE = 0u, MIPT_MULTI_FRAME_EXACT = 1u };
int mipt_multi_frame_create(MiptMulti* multi, int width, int height, uint32_t mode, MiptMultiFrame** out);
void mipt_multi_frame_destroy(MiptMultiFrame* frame);
int mipt_multi_frame_reset(MiptMultiFrame* frame);
int mipt_multi_frame_add_samples(MiptMultiFrame* frame, const MiptCamera* cam, const MiptLight* light,
                                 const MiptParams* params, uint32_t n_samples);
int mipt_multi_frame_present(MiptMultiFrame* frame, uint32_t* screen_buffer);
int mipt_multi_frame_read_pixels(MiptMultiFrame* frame, float* accum_rgb);
int mipt_multi_frame_samples(const MiptMultiFrame* frame);
int mipt_multi_frame_device(const MiptMultiFrame* frame);
int mipt_multi_frame_poll(MiptMultiFrame* frame);

/* Counters of the most recent render / trace call made with MIPT_FLAG_STATS, and
 * the device time of its path kernel (measured with HIP events on the launch
 * stream; synchronises that stream). */
int mipt_get_stats(MiptScene* scene, MiptStats* out);

/* Self-check with no counterpart in the reference: runs on the GPU (1) the library's fast
 * correctly-rounded reciprocal against IEEE 1.0f/b for all 2^32 floats and (2) its fast triangle
 * scan against the literal reference arithmetic on n_rays generated rays (camera rays, incoherent
 * interior rays, origins on triangle surfaces and edges, rays within 0.01 degrees of a triangle's
 * plane), closest-hit and any-hit rule.
 * For a scene with an LBVH, (2) compares the tree traversal with the literal linear scan.
 * out[0] = reciprocal mismatches, out[1] = worst error of the hardware reciprocal estimate in
 * 2^-10 ulp, out[2] / out[3] = rays whose closest-hit / any-hit result differs, out[4] = rays
 * re-run through the exhaustive tree traversal, out[5] = traversal overflows, out[6] = floats in
 * [0, 2 pi] whose fused-arithmetic sine / cosine differs from the plain evaluation, out[7] = tree
 * scenes only: plane-grazing rays (a fifth of n_rays) whose tree result differs from the linear scan,
 * counted apart from out[2] / out[3] -- the known exception described at MIPT_BVH_MIN_TRIS.
 * All of out[0], out[2], out[3], out[5], out[6] must be 0. Any n_rays up to 2^36 (launches of 2^30
 * rays, each with a seed derived from `seed`). */
int mipt_debug_selfcheck(MiptScene* scene, uint64_t n_rays, uint32_t seed, uint64_t out[8]);

/* The (seed, pixel, sample) -> 32-bit key shared by both per-path RNG modes.
 * pixel = x*height + y. Exposed so that hosts and tests can reproduce streams. */
uint32_t mipt_path_key(uint32_t seed, uint32_t pixel, uint32_t sample);

#ifdef __cplusplus
}
#endif
#endif /* MIPT_H */
