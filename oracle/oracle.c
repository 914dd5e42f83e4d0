/*
 * oracle/oracle.c -- CPU restatement of the stijnherfst/BrickMap path-trace hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under brickmap_amd/ may include, link, import or
 * execute this file or the library built from it.  Allowed callers: tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg (as the checker / the timed
 * CPU baseline, never as the product).
 *
 * What it restates (all file:line citations are into /root/reference/src):
 *   - SimplexNoise::noise(x,y) / fractal(oct,x,y)      SimplexNoise.cpp:47-50,73-87,101-103,144-149,216-292,435-450
 *   - Scene::generate_supercell / generate             Scene.cpp:44-116,118-175
 *   - Scene::process_load_queue + upload kernel        Scene.cpp:200-252, kernel.cu:141-151,407-414
 *   - RNG + samplers                                   kernel.cu:19-61,76-103
 *   - primary_rays / extend / shade / connect          kernel.cu:154-346  (per-ray functions)
 *   - set_wavefront_globals, launch_kernels sequencing kernel.cu:122-139,366-439 (mode A)
 *   - intersect_aabb_branchless2/byte/brick/voxel      voxel.cuh:13-261
 *   - sun / sky / sunsky / getConeSample               sunsky.cu:10-183, sunsky.cuh:25-42
 *
 * PARITY PINNING STATUS (see DESIGN.md "Oracle"):
 *   - noise / fractal: PINNED bit-exactly against the real reference SimplexNoise.cpp
 *     compiled from /root/reference by oracle/Makefile into oracle/_ref/ (tests compare
 *     when that library is present) and against tests/golden/noise_ref.npz generated
 *     from it.
 *   - everything that depends on GLM/CUDA (voxel.cuh, kernel.cu, sunsky.cu, Scene.cpp):
 *     the reference cannot be built in this image without writing stand-ins for GLM and
 *     the CUDA runtime, which is not allowed, and the reference ships no tests or golden
 *     vectors.  Those parts are pinned only against the known answers recorded in
 *     SURVEY.md section 8 (obtained by the surveyor from the reference's own functions):
 *     see tests/golden/survey_probes.json.  Beyond those: "parity unpinned".
 *
 * Numeric contract: IEEE fp32 + - * / sqrt, no FMA contraction (-ffp-contract=off),
 * float->int by truncation.  sin/cos used for *direction sampling* go through
 * orc_sincos() (fp32 throughout: three-constant Cody-Waite reduction + fixed-order degree-7/8
 * polynomials, the specification csrc/detmath.h implements independently on the device) so
 * that CPU and GPU agree bit-for-bit on ray geometry; the sky model uses the platform libm
 * (expf/powf/acosf), compared with a 1e-4 relative tolerance.
 *
 * Two schedules are provided:
 *   mode A  orc_wavefront_*   : the reference's wavefront loop run sequentially
 *   mode B  orc_render        : canonical one-path-per-pixel loop (what the HIP kernel does)
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORC_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------- constants (variables.h:3-35, kernel.cu:12-13) */
static const float k_pi = 3.1415926535897932f;
#define BRICK_SIZE 8
#define SUPERCELL 16
#define CELL_MEMBERS 16
static const float k_epsilon = 0.001f;
#define BRICK_INDEX_BITS 0xFFFu
#define BRICK_LOD_BITS 0xFF000u
#define BRICK_LOADED_BIT 0x80000000u
#define BRICK_UNLOADED_BIT 0x40000000u
#define BRICK_REQUESTED_BIT 0x20000000u
static const float VERY_FAR = 1e20f;

typedef struct { float x, y, z; } v3;
typedef struct { int x, y, z; } i3;

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 div3(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 muls(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 divs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
/* GLM: min(a,b) = (b<a)?b:a ; max(a,b) = (a<b)?b:a ; sign = (0<x)-(x<0) */
static inline float gmin(float a, float b) { return (b < a) ? b : a; }
static inline float gmax(float a, float b) { return (a < b) ? b : a; }
static inline float gsign(float x) { return (float)((0.f < x) - (x < 0.f)); }
/* GLM compute_dot<vec3>: tmp = a*b; tmp.x + tmp.y + tmp.z */
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross3(v3 x, v3 y) {
	return V3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* GLM normalize: v * inversesqrt(dot(v,v)), inversesqrt = 1/sqrt */
static inline v3 normalize3(v3 v) { return muls(v, 1.0f / sqrtf(dot3(v, v))); }
static inline float comp(v3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
static inline int compi(i3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

/* ---------------------------------------------------------------- deterministic sin/cos
 * Spec (shared, by specification not by code, with brickmap_amd/csrc/detmath.h), everything in fp32:
 *   k = (int)(x*2/pi + (x>=0 ? .5 : -.5)); r = ((x - k*C1) - k*C2) - k*C3 with C1+C2+C3 = pi/2 (k*C1, k*C2 exact)
 *   z = r*r; sin r = (((S3 z + S2) z + S1) z r) + r; cos r = (((K3 z + K2) z + K1) z z - z/2) + 1; quadrant by k&3.
 * <= 1.5 ulp for |x| <= 2 pi.  The reference uses CUDA's sin()/cos() here (kernel.cu:102,296; sunsky.cu:183),
 * which are specified to a couple of ulp only, so any sincos of that quality restates it. */
static void orc_sincos_impl(float x, float* s_out, float* c_out) {
	static const float TWO_OVER_PI = 0.6366197723675814f;
	static const float PIO2_1 = 1.5703125f, PIO2_2 = 4.837512969970703125e-4f, PIO2_3 = 7.54978995489188e-8f;
	static const float S1 = -1.6666654611e-1f, S2 = 8.3321608736e-3f, S3 = -1.9515295891e-4f;
	static const float K1 = 4.166664568298827e-2f, K2 = -1.388731625493765e-3f, K3 = 2.443315711809948e-5f;
	float half = x >= 0.0f ? 0.5f : -0.5f;
	int k = (int)(x * TWO_OVER_PI + half);
	float kf = (float)k;
	float r = x - kf * PIO2_1;
	r = r - kf * PIO2_2;
	r = r - kf * PIO2_3;
	float z = r * r;
	float ps = S3 * z + S2;
	ps = ps * z + S1;
	float sr = (ps * z) * r + r;
	float pc = K3 * z + K2;
	pc = pc * z + K1;
	float cr = ((pc * z) * z - 0.5f * z) + 1.0f;
	float s, c;
	switch (k & 3) {
	case 0: s = sr; c = cr; break;
	case 1: s = cr; c = -sr; break;
	case 2: s = -sr; c = -cr; break;
	default: s = -cr; c = sr; break;
	}
	*s_out = s;
	*c_out = c;
}
ORC_API void orc_sincos(int n, const float* x, float* s, float* c) {
	for (int i = 0; i < n; i++) orc_sincos_impl(x[i], &s[i], &c[i]);
}
static inline float det_sin(float x) { float s, c; orc_sincos_impl(x, &s, &c); return s; }
static inline float det_cos(float x) { float s, c; orc_sincos_impl(x, &s, &c); return c; }

/* ---------------------------------------------------------------- simplex noise (SimplexNoise.cpp) */
static inline int32_t fastfloor(float fp) { /* :47-50 */
	int32_t i = (int32_t)fp;
	return (fp < i) ? (i - 1) : i;
}
static const uint8_t perm[256] = { /* :73-87 */
	151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142,
	8, 99, 37, 240, 21, 10, 23, 190, 6, 148, 247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203,
	117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136, 171, 168, 68, 175, 74,
	165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122, 60, 211, 133, 230, 220,
	105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54, 65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132,
	187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3,
	64, 52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227,
	47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213, 119, 248, 152, 2, 44, 154, 163, 70, 221,
	153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232, 178, 185,
	112, 104, 218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241, 81, 51,
	145, 235, 249, 14, 239, 107, 49, 192, 214, 31, 181, 199, 106, 157, 184, 84, 204, 176, 115, 121,
	50, 45, 127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78,
	66, 215, 61, 156, 180
};
static inline uint8_t nhash(int32_t i) { return perm[(uint8_t)i]; } /* :101-103 */
static float grad2(int32_t hash, float x, float y) { /* :144-149 (the h<4 test on hash&0x3F is an upstream quirk) */
	const int32_t h = hash & 0x3F;
	const float u = h < 4 ? x : y;
	const float v = h < 4 ? y : x;
	return ((h & 1) ? -u : u) + ((h & 2) ? -2.0f * v : 2.0f * v);
}
static float noise2(float x, float y) { /* :216-292 */
	float n0, n1, n2;
	const float F2 = 0.366025403f, G2 = 0.211324865f;
	const float s = (x + y) * F2;
	const float xs = x + s, ys = y + s;
	const int32_t i = fastfloor(xs), j = fastfloor(ys);
	const float t = (float)(i + j) * G2;
	const float X0 = i - t, Y0 = j - t;
	const float x0 = x - X0, y0 = y - Y0;
	int32_t i1, j1;
	if (x0 > y0) { i1 = 1; j1 = 0; } else { i1 = 0; j1 = 1; }
	const float x1 = x0 - i1 + G2, y1 = y0 - j1 + G2;
	const float x2 = x0 - 1.0f + 2.0f * G2, y2 = y0 - 1.0f + 2.0f * G2;
	const int gi0 = nhash(i + nhash(j));
	const int gi1 = nhash(i + i1 + nhash(j + j1));
	const int gi2 = nhash(i + 1 + nhash(j + 1));
	float t0 = 0.5f - x0 * x0 - y0 * y0;
	if (t0 < 0.0f) n0 = 0.0f; else { t0 *= t0; n0 = t0 * t0 * grad2(gi0, x0, y0); }
	float t1 = 0.5f - x1 * x1 - y1 * y1;
	if (t1 < 0.0f) n1 = 0.0f; else { t1 *= t1; n1 = t1 * t1 * grad2(gi1, x1, y1); }
	float t2 = 0.5f - x2 * x2 - y2 * y2;
	if (t2 < 0.0f) n2 = 0.0f; else { t2 *= t2; n2 = t2 * t2 * grad2(gi2, x2, y2); }
	return 45.23065f * (n0 + n1 + n2);
}
/* fractal(octaves,x,y) with SimplexNoise(1,1,2,0.5) (SimplexNoise.cpp:435-450, Scene.cpp:45) */
static float fractal2(size_t octaves, float x, float y) {
	float output = 0.f, denom = 0.f, frequency = 1.0f, amplitude = 1.0f;
	for (size_t i = 0; i < octaves; i++) {
		output += (amplitude * noise2(x * frequency, y * frequency));
		denom += amplitude;
		frequency *= 2.0f;
		amplitude *= 0.5f;
	}
	return output / denom;
}
ORC_API float orc_noise2(float x, float y) { return noise2(x, y); }
ORC_API float orc_fractal2(int octaves, float x, float y) { return fractal2((size_t)octaves, x, y); }
ORC_API void orc_fractal2_grid(int octaves, int n, const float* xs, const float* ys, float* out) {
	for (int i = 0; i < n; i++) out[i] = fractal2((size_t)octaves, xs[i], ys[i]);
}

/* ---------------------------------------------------------------- world (Scene.h:3-44) */
typedef struct {
	uint32_t nbricks;      /* host bricks.size()                                   */
	uint32_t* bricks;      /* host bricks, 16 words each, generation order          */
	uint32_t* indices;     /* host index words [4096]: slot | loaded | lod<<12      */
	uint32_t* dev_indices; /* emulated device index block [4096]                    */
	uint32_t* dev_bricks;  /* emulated device pool, gpu_count*16 words              */
	int gpu_count;         /* pool capacity, starts at supergrid_starting_size = 16 */
	int gpu_index_highest;
} orc_supercell;

typedef struct {
	int grid_size, grid_height;    /* voxels; variables.h:7-8 (runtime here)        */
	int cells, cells_height;       /* bricks                                        */
	int sg_xy, sg_z, nsc;          /* supercells                                    */
	int lod_distance_8x8x8, lod_distance_2x2x2; /* variables.h:24-27                 */
	int queue_cap;                 /* brick_load_queue_size, variables.h:35 = 1024  */
	orc_supercell* sc;
	i3* load_queue;
	uint32_t load_queue_count;
	uint32_t* bricks_queue;  /* queue_cap * 16 */
	uint32_t* indices_queue; /* queue_cap      */
	uint64_t total_uploaded;
	/* overlapped servicing (the product's two-ring mode, see orc_process_load_queue_overlapped) */
	int overlapped;
	i3* pending;            /* the ring copied out by the previous call, queue_cap entries */
	uint32_t pending_count;
} orc_world;

typedef struct {
	uint64_t index_loads; /* brick-grid index words read (one per outer DDA iteration) */
	uint64_t brick_tests; /* intersect_brick calls                                     */
	uint64_t byte_tests;  /* intersect_byte calls                                      */
	uint64_t voxel_steps; /* iterations of the 8^3 / 2^3 inner loops                   */
	uint64_t extend_rays;
	uint64_t shadow_rays;
	uint64_t requests;    /* brick requests queued                                     */
	uint64_t paths;
} orc_counters;

ORC_API orc_world* orc_world_create(int grid_size, int grid_height) {
	if (grid_size % 128 || grid_height % 128 || grid_size <= 0 || grid_height <= 0) return NULL;
	orc_world* w = (orc_world*)calloc(1, sizeof(orc_world));
	w->grid_size = grid_size;
	w->grid_height = grid_height;
	w->cells = grid_size / BRICK_SIZE;
	w->cells_height = grid_height / BRICK_SIZE;
	w->sg_xy = grid_size / BRICK_SIZE / SUPERCELL;
	w->sg_z = grid_height / BRICK_SIZE / SUPERCELL;
	w->nsc = w->sg_xy * w->sg_xy * w->sg_z;
	w->lod_distance_8x8x8 = 600000;
	w->lod_distance_2x2x2 = 100000;
	w->queue_cap = 1024;
	w->sc = (orc_supercell*)calloc((size_t)w->nsc, sizeof(orc_supercell));
	w->load_queue = (i3*)calloc((size_t)w->queue_cap, sizeof(i3));
	w->bricks_queue = (uint32_t*)calloc((size_t)w->queue_cap * 16, 4);
	w->indices_queue = (uint32_t*)calloc((size_t)w->queue_cap, 4);
	w->pending = (i3*)calloc((size_t)w->queue_cap, sizeof(i3));
	return w;
}
ORC_API void orc_world_destroy(orc_world* w) {
	if (!w) return;
	for (int i = 0; i < w->nsc; i++) {
		free(w->sc[i].bricks); free(w->sc[i].indices); free(w->sc[i].dev_indices); free(w->sc[i].dev_bricks);
	}
	free(w->sc); free(w->load_queue); free(w->bricks_queue); free(w->indices_queue); free(w->pending); free(w);
}
ORC_API void orc_world_set_lod(orc_world* w, int lod8, int lod2) { w->lod_distance_8x8x8 = lod8; w->lod_distance_2x2x2 = lod2; }
ORC_API void orc_world_set_queue_cap(orc_world* w, int cap) {
	w->queue_cap = cap;
	w->load_queue = (i3*)realloc(w->load_queue, (size_t)cap * sizeof(i3));
	w->bricks_queue = (uint32_t*)realloc(w->bricks_queue, (size_t)cap * 64);
	w->indices_queue = (uint32_t*)realloc(w->indices_queue, (size_t)cap * 4);
	w->pending = (i3*)realloc(w->pending, (size_t)cap * sizeof(i3));
	w->pending_count = 0;
	w->load_queue_count = 0;
}
ORC_API void orc_world_set_overlapped(orc_world* w, int overlapped) { w->overlapped = overlapped; }

/* heights of one supercell column (Scene.cpp:47-58).  The reference recomputes this for every
 * z-layer supercell; it is a pure function of (sx, sy), so it is computed once per column here. */
static void column_heights(const orc_world* w, int sx, int sy, float* heights) {
	const int n = SUPERCELL * BRICK_SIZE;
	for (int y = 0; y < n; y++)
		for (int x = 0; x < n; x++) {
			float h = fractal2(8, (sx * SUPERCELL * BRICK_SIZE + x) / 2048.f, (sy * SUPERCELL * BRICK_SIZE + y) / 2048.f);
			h *= w->grid_height / 2.f;
			h += w->grid_height / 2.f;
			heights[x + y * n] = h;
		}
}
ORC_API void orc_column_heights(const orc_world* w, int sx, int sy, float* heights) { column_heights(w, sx, sy, heights); }

/* Scene::generate_supercell (Scene.cpp:44-116) */
static void generate_supercell(orc_world* w, int sx, int sy, int sz, const float* heights) {
	orc_supercell* sc = &w->sc[sx + sy * w->sg_xy + sz * w->sg_xy * w->sg_xy];
	sc->indices = (uint32_t*)calloc(4096, 4);
	uint32_t cap = 64, n = 0;
	uint32_t* bricks = (uint32_t*)malloc((size_t)cap * 64);
	for (int z = 0; z < SUPERCELL; z++)
		for (int y = 0; y < SUPERCELL; y++)
			for (int x = 0; x < SUPERCELL; x++) {
				uint32_t brick[16];
				memset(brick, 0, sizeof brick);
				int empty = 1;
				uint32_t lod = 0;
				for (int cx = 0; cx < BRICK_SIZE; cx++)
					for (int cy = 0; cy < BRICK_SIZE; cy++) {
						float height = heights[cx + x * BRICK_SIZE + (cy + y * BRICK_SIZE) * BRICK_SIZE * SUPERCELL];
						for (int cz = 0; cz < BRICK_SIZE; cz++) {
							if ((sz * SUPERCELL + z) * BRICK_SIZE + cz < height) {
								uint32_t b = (uint32_t)(cx + cy * BRICK_SIZE + cz * BRICK_SIZE * BRICK_SIZE);
								brick[b / 32] |= (1u << (b % 32));
								empty = 0;
								lod |= 1u << (((cx & 4) >> 2) + ((cy & 4) >> 1) + (cz & 4));
							}
						}
					}
				if (!empty) {
					if (n == cap) { cap *= 2; bricks = (uint32_t*)realloc(bricks, (size_t)cap * 64); }
					memcpy(bricks + (size_t)n * 16, brick, 64);
					n++;
					sc->indices[x + y * SUPERCELL + z * SUPERCELL * SUPERCELL] = (n - 1) | BRICK_LOADED_BIT | (lod << 12);
				}
			}
	sc->bricks = bricks;
	sc->nbricks = n;
}

typedef struct { orc_world* w; volatile int next; } gen_job;
static void* gen_worker(void* arg) {
	gen_job* job = (gen_job*)arg;
	orc_world* w = job->w;
	const int ncol = w->sg_xy * w->sg_xy;
	float* heights = (float*)malloc(128 * 128 * sizeof(float));
	for (;;) {
		int c = __atomic_fetch_add(&job->next, 1, __ATOMIC_RELAXED);
		if (c >= ncol) break;
		int sx = c % w->sg_xy, sy = c / w->sg_xy;
		column_heights(w, sx, sy, heights);
		for (int sz = 0; sz < w->sg_z; sz++) generate_supercell(w, sx, sy, sz, heights);
	}
	free(heights);
	return NULL;
}

/* mode: 0 = reference initial state (every non-empty brick "unloaded | lod", Scene.cpp:157-175)
 *       1 = all bricks resident (device words = host words, pools = host brick vectors)       */
ORC_API void orc_world_reset_device(orc_world* w, int preload_all) {
	for (int i = 0; i < w->nsc; i++) {
		orc_supercell* sc = &w->sc[i];
		if (!sc->dev_indices) sc->dev_indices = (uint32_t*)malloc(4096 * 4);
		free(sc->dev_bricks);
		if (preload_all) {
			memcpy(sc->dev_indices, sc->indices, 4096 * 4);
			sc->gpu_count = sc->nbricks > 16 ? (int)sc->nbricks : 16;
			sc->dev_bricks = (uint32_t*)calloc((size_t)sc->gpu_count, 64);
			memcpy(sc->dev_bricks, sc->bricks, (size_t)sc->nbricks * 64);
			sc->gpu_index_highest = (int)sc->nbricks;
		} else {
			for (int j = 0; j < 4096; j++)
				sc->dev_indices[j] = (sc->indices[j] & BRICK_LOADED_BIT) ? (BRICK_UNLOADED_BIT | (sc->indices[j] & BRICK_LOD_BITS)) : 0u;
			sc->gpu_count = 16;
			sc->dev_bricks = (uint32_t*)calloc(16, 64);
			sc->gpu_index_highest = 0;
		}
	}
	w->load_queue_count = 0;
	w->pending_count = 0;
	w->total_uploaded = 0;
}

/* Scene::generate (Scene.cpp:118-194): CPU build on a thread pool, then the device-side initial state. */
ORC_API void orc_world_generate(orc_world* w, int threads) {
	gen_job job = { w, 0 };
	if (threads < 1) threads = 1;
	if (threads > 64) threads = 64;
	pthread_t th[64];
	for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, gen_worker, &job);
	for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
	orc_world_reset_device(w, 0);
}

ORC_API int orc_world_nsc(const orc_world* w) { return w->nsc; }
ORC_API uint32_t orc_world_sc_nbricks(const orc_world* w, int sc) { return w->sc[sc].nbricks; }
ORC_API const uint32_t* orc_world_sc_indices(const orc_world* w, int sc) { return w->sc[sc].indices; }
ORC_API const uint32_t* orc_world_sc_bricks(const orc_world* w, int sc) { return w->sc[sc].bricks; }
ORC_API const uint32_t* orc_world_sc_dev_indices(const orc_world* w, int sc) { return w->sc[sc].dev_indices; }
ORC_API int orc_world_sc_gpu_count(const orc_world* w, int sc) { return w->sc[sc].gpu_count; }
ORC_API int orc_world_sc_gpu_index_highest(const orc_world* w, int sc) { return w->sc[sc].gpu_index_highest; }
ORC_API uint64_t orc_world_total_bricks(const orc_world* w) {
	uint64_t t = 0;
	for (int i = 0; i < w->nsc; i++) t += w->sc[i].nbricks;
	return t;
}
ORC_API uint32_t orc_world_queue_count(const orc_world* w) { return w->load_queue_count; }
ORC_API uint64_t orc_world_total_uploaded(const orc_world* w) { return w->total_uploaded; }

/* FNV-1a 64 over (indices words, then brick words) of every supercell in supercell order. */
static uint64_t fnv64(uint64_t h, const void* data, size_t n) {
	const uint8_t* p = (const uint8_t*)data;
	for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
	return h;
}
ORC_API uint64_t orc_world_hash(const orc_world* w) {
	uint64_t h = 14695981039346656037ull;
	for (int i = 0; i < w->nsc; i++) {
		h = fnv64(h, w->sc[i].indices, 4096 * 4);
		h = fnv64(h, w->sc[i].bricks, (size_t)w->sc[i].nbricks * 64);
	}
	return h;
}

/* ---------------------------------------------------------------- streaming (Scene.cpp:200-252, kernel.cu:141-151,407-414) */
static int supergrid_index(const orc_world* w, i3 p) { /* Scene.cpp:196-198 */
	return p.x / SUPERCELL + p.y / SUPERCELL * w->sg_xy + p.z / SUPERCELL * w->sg_xy * w->sg_xy;
}
/* upload kernel + count reset, as launch_kernels does at the start of a frame (kernel.cu:407-414). */
ORC_API uint32_t orc_upload(orc_world* w) {
	uint32_t count = w->load_queue_count;
	if (count > (uint32_t)w->queue_cap) count = (uint32_t)w->queue_cap;
	if (count == 0) return 0;
	for (uint32_t i = 0; i < count; i++) {
		i3 pos = w->load_queue[i];
		int sci = supergrid_index(w, pos);
		uint32_t local = (uint32_t)((pos.x % SUPERCELL) + (pos.y % SUPERCELL) * SUPERCELL + (pos.z % SUPERCELL) * SUPERCELL * SUPERCELL);
		orc_supercell* sc = &w->sc[sci];
		memcpy(sc->dev_bricks + (size_t)(w->indices_queue[i] & BRICK_INDEX_BITS) * 16, w->bricks_queue + (size_t)i * 16, 64);
		sc->dev_indices[local] = w->indices_queue[i];
	}
	w->load_queue_count = 0;
	w->total_uploaded += count;
	return count;
}
/* Scene::process_load_queue: stage requested bricks and grow pools; the count is NOT reset here. */
ORC_API uint32_t orc_process_load_queue(orc_world* w) {
	uint32_t count = w->load_queue_count;
	if (count > (uint32_t)w->queue_cap) count = (uint32_t)w->queue_cap;
	if (count == 0) return 0;
	for (uint32_t i = 0; i < count; i++) {
		i3 pos = w->load_queue[i];
		orc_supercell* sc = &w->sc[supergrid_index(w, pos)];
		uint32_t local = (uint32_t)((pos.x % SUPERCELL) + (pos.y % SUPERCELL) * SUPERCELL + (pos.z % SUPERCELL) * SUPERCELL * SUPERCELL);
		uint32_t index = sc->indices[local];
		memcpy(w->bricks_queue + (size_t)i * 16, sc->bricks + (size_t)(index & BRICK_INDEX_BITS) * 16, 64);
		w->indices_queue[i] = ((uint32_t)sc->gpu_index_highest | BRICK_LOADED_BIT | (index & BRICK_LOD_BITS));
		sc->gpu_index_highest++;
	}
	for (uint32_t i = 0; i < count; i++) {
		orc_supercell* sc = &w->sc[supergrid_index(w, w->load_queue[i])];
		if (sc->gpu_index_highest >= sc->gpu_count) {
			int new_size = (int)pow(2.0, ceil(log2((double)(sc->gpu_index_highest + 1))));
			uint32_t* nb = (uint32_t*)calloc((size_t)new_size, 64);
			memcpy(nb, sc->dev_bricks, (size_t)sc->gpu_count * 64);
			free(sc->dev_bricks);
			sc->dev_bricks = nb;
			sc->gpu_count = new_size;
		}
	}
	return count;
}

/* Overlapped servicing -- the product's two-ring mode (brickmap_amd/csrc/scene.cpp Scene::process_load_queue): the host
 * never waits for the frame it has just launched.  One call, made after frame k,
 *   (1) stages and uploads the ring that the PREVIOUS call copied out (the requests of frame k-1): they are resident
 *       from frame k+1 on -- one frame later than in the reference's order, where process_load_queue after frame k-1
 *       stages them and the upload kernel at the start of frame k scatters them (kernel.cu:407-414, main.cpp:142-144);
 *   (2) copies out the ring frame k wrote and hands frame k+1 the other, empty ring.
 * Bricks whose request sits in a copied-out ring keep their requested bit, so a later frame does not ask again.
 * Returns the number of bricks uploaded by (1). */
ORC_API uint32_t orc_process_load_queue_overlapped(orc_world* w) {
	const uint32_t ring_count = w->load_queue_count > (uint32_t)w->queue_cap ? (uint32_t)w->queue_cap : w->load_queue_count;
	/* keep the ring frame k wrote aside */
	i3* ring_copy = (i3*)malloc((size_t)(ring_count ? ring_count : 1) * sizeof(i3));
	memcpy(ring_copy, w->load_queue, (size_t)ring_count * sizeof(i3));
	/* (1): run the reference's process_load_queue + upload on the pending ring */
	uint32_t serviced = 0;
	if (w->pending_count > 0) {
		memcpy(w->load_queue, w->pending, (size_t)w->pending_count * sizeof(i3));
		w->load_queue_count = w->pending_count;
		orc_process_load_queue(w);
		serviced = orc_upload(w);
	}
	/* (2) */
	memcpy(w->pending, ring_copy, (size_t)ring_count * sizeof(i3));
	w->pending_count = ring_count;
	w->load_queue_count = 0;
	free(ring_copy);
	return serviced;
}

/* ---------------------------------------------------------------- traversal (voxel.cuh:13-261) */
typedef struct {
	int hit;       /* 0 miss, 1 hit                                                          */
	int level;     /* 0: solid brick by LoD (>lod8)  1: 2^3 LoD byte  2: 8^3 voxel  3: unloaded brick treated as solid */
	int brick_id;  /* pos.x + pos.y*cells + pos.z*cells*cells of the brick that was hit      */
	int sub_id;    /* voxel x+8y+64z (level 2) or x+2y+4z (level 1), else 0                   */
} orc_hit;

static int intersect_aabb(const orc_world* w, v3 origin, v3 direction, float* tmin) { /* :13-24 */
	const v3 box_min = V3(0, 0, 0);
	const v3 box_max = V3((float)w->grid_size, (float)w->grid_size, (float)w->grid_height);
	const v3 t1 = div3(sub3(box_min, origin), direction);
	const v3 t2 = div3(sub3(box_max, origin), direction);
	const v3 tMin = V3(gmin(t1.x, t2.x), gmin(t1.y, t2.y), gmin(t1.z, t2.z));
	const v3 tMax = V3(gmax(t1.x, t2.x), gmax(t1.y, t2.y), gmax(t1.z, t2.z));
	*tmin = gmax(gmax(tMin.x, 0.f), gmax(tMin.y, tMin.z));
	return gmin(tMax.x, gmin(tMax.y, tMax.z)) > *tmin;
}

#define SET_AXIS(v, a, val) do { if ((a) == 0) (v).x = (val); else if ((a) == 1) (v).y = (val); else (v).z = (val); } while (0)

/* Shared body of intersect_byte (:26-77, N=2) and intersect_brick (:79-133, N=8). */
static int intersect_grid(v3 origin, v3 direction, v3* normal, float* distance, int N, const uint32_t* words, int* sub_id, orc_counters* cnt) {
	i3 pos = { (int)origin.x, (int)origin.y, (int)origin.z };
	v3 cb;
	cb.x = direction.x > 0.f ? (float)(pos.x + 1) : (float)pos.x;
	cb.y = direction.y > 0.f ? (float)(pos.y + 1) : (float)pos.y;
	cb.z = direction.z > 0.f ? (float)(pos.z + 1) : (float)pos.z;
	i3 out;
	out.x = direction.x > 0.f ? N : -1;
	out.y = direction.y > 0.f ? N : -1;
	out.z = direction.z > 0.f ? N : -1;
	v3 step = V3(gsign(direction.x), gsign(direction.y), gsign(direction.z));
	v3 rdinv = V3(1.f / direction.x, 1.f / direction.y, 1.f / direction.z);
	rdinv.x = direction.x == 0.0f ? 0.0f : rdinv.x;
	rdinv.y = direction.y == 0.0f ? 0.0f : rdinv.y;
	rdinv.z = direction.z == 0.0f ? 0.0f : rdinv.z;
	v3 tmax;
	tmax.x = direction.x != 0.f ? (cb.x - origin.x) * rdinv.x : 1000000.f;
	tmax.y = direction.y != 0.f ? (cb.y - origin.y) * rdinv.y : 1000000.f;
	tmax.z = direction.z != 0.f ? (cb.z - origin.z) * rdinv.z : 1000000.f;
	v3 tdelta = mul3(step, rdinv);
	pos.x %= N; pos.y %= N; pos.z %= N;
	*distance = 0.f;
	int step_axis = -1;
	for (;;) {
		cnt->voxel_steps++;
		int b = pos.x + pos.y * N + pos.z * N * N;
		/* "& 15" / "& 31" only define what the reference leaves undefined (negative b); no effect otherwise */
		if (words[(b / 32) & 15] & (1u << (b & 31))) {
			if (step_axis > -1) {
				*normal = V3(0, 0, 0);
				SET_AXIS(*normal, step_axis, -comp(step, step_axis));
				*distance = comp(tmax, step_axis) - comp(tdelta, step_axis);
			}
			*sub_id = b;
			return 1;
		}
		step_axis = (tmax.x < tmax.y) ? ((tmax.x < tmax.z) ? 0 : 2) : ((tmax.y < tmax.z) ? 1 : 2);
		v3 mask;
		mask.x = (float)(tmax.x < tmax.y && tmax.x < tmax.z);
		mask.y = (float)(tmax.y <= tmax.x && tmax.y < tmax.z);
		mask.z = (float)(tmax.z <= tmax.x && tmax.z <= tmax.y);
		pos.x += (int)(mask.x * step.x);
		pos.y += (int)(mask.y * step.y);
		pos.z += (int)(mask.z * step.z);
		if (compi(pos, step_axis) == compi(out, step_axis)) break;
		tmax = add3(tmax, mul3(mask, tdelta));
	}
	return 0;
}

typedef void (*orc_brick_probe_t)(const float* origin_in_brick, const float* direction, uint32_t index_word, const uint32_t* brick16, int hit, unsigned steps);
static orc_brick_probe_t orc_brick_probe_fn = 0;
ORC_API void orc_set_brick_probe(orc_brick_probe_t fn) { orc_brick_probe_fn = fn; }
/* analysis door (tools/sim): one call per ray of the canonical per-pixel render, in path order, before it is traced
 * (kind 0 = extend, 1 = shadow); single-threaded renders only */
typedef void (*orc_ray_probe_t)(unsigned pixel, int sample, int kind, const float* origin, const float* direction);
static orc_ray_probe_t orc_ray_probe_fn = 0;
ORC_API void orc_set_ray_probe(orc_ray_probe_t fn) { orc_ray_probe_fn = fn; }
static int intersect_voxel(orc_world* w, v3 origin, const v3 direction, v3* normal, float* distance, i3 camera_position, orc_hit* hit, orc_counters* cnt, int atomic_requests) { /* :135-261 */
	float tminn;
	hit->hit = 0; hit->level = 0; hit->brick_id = -1; hit->sub_id = 0;
	if (!intersect_aabb(w, origin, direction, &tminn)) return 0;
	if (tminn > 0) {
		origin = add3(origin, muls(direction, tminn));
		const float gs = (float)w->grid_size, gh = (float)w->grid_height;
		const v3 scale = V3(1.f / (gs / gh), 1.f / (gs / gh), 1.f / (gh / gh));
		const v3 grid_center = V3(gs / 2.f, gs / 2.f, gh / 2.f);
		v3 d = sub3(grid_center, origin);
		v3 to_center = mul3(V3(fabsf(d.x), fabsf(d.y), fabsf(d.z)), scale);
		v3 e = sub3(origin, grid_center);
		v3 signs = V3(gsign(e.x), gsign(e.y), gsign(e.z));
		to_center = divs(to_center, gmax(to_center.x, gmax(to_center.y, to_center.z)));
		*normal = mul3(signs, V3(truncf(to_center.x + 0.000001f), truncf(to_center.y + 0.000001f), truncf(to_center.z + 0.000001f)));
		origin = sub3(origin, muls(*normal, k_epsilon));
	}
	origin = divs(origin, 8.f);
	i3 pos = { (int)origin.x, (int)origin.y, (int)origin.z };
	const int cells = w->cells, cells_height = w->cells_height;
	if (pos.x < 0 || pos.x >= cells || pos.y < 0 || pos.y >= cells || pos.z < 0 || pos.z >= cells_height) return 0;
	v3 cb;
	cb.x = direction.x > 0.f ? (float)(pos.x + 1) : (float)pos.x;
	cb.y = direction.y > 0.f ? (float)(pos.y + 1) : (float)pos.y;
	cb.z = direction.z > 0.f ? (float)(pos.z + 1) : (float)pos.z;
	i3 out;
	out.x = direction.x > 0.f ? cells : -1;
	out.y = direction.y > 0.f ? cells : -1;
	out.z = direction.z > 0.f ? cells_height : -1;
	v3 step = V3(gsign(direction.x), gsign(direction.y), gsign(direction.z));
	v3 rdinv = V3(1.f / direction.x, 1.f / direction.y, 1.f / direction.z);
	rdinv.x = direction.x == 0.0f ? 0.0f : rdinv.x;
	rdinv.y = direction.y == 0.0f ? 0.0f : rdinv.y;
	rdinv.z = direction.z == 0.0f ? 0.0f : rdinv.z;
	v3 tmax;
	tmax.x = direction.x != 0.f ? (cb.x - origin.x) * rdinv.x : 1000000.f;
	tmax.y = direction.y != 0.f ? (cb.y - origin.y) * rdinv.y : 1000000.f;
	tmax.z = direction.z != 0.f ? (cb.z - origin.z) * rdinv.z : 1000000.f;
	v3 tdelta = mul3(step, rdinv);
	int step_axis = -1;
	/* hang guard only: a well-formed ray makes at most 2*cells + cells_height steps */
	long guard = 4L * (2L * cells + cells_height) + 16;
	while (guard-- > 0) {
		int supercell_index = pos.x / SUPERCELL + (pos.y / SUPERCELL) * w->sg_xy + (pos.z / SUPERCELL) * w->sg_xy * w->sg_xy;
		orc_supercell* sc = &w->sc[supercell_index];
		uint32_t* pindex = &sc->dev_indices[(pos.x % SUPERCELL) + (pos.y % SUPERCELL) * SUPERCELL + (pos.z % SUPERCELL) * SUPERCELL * SUPERCELL];
		uint32_t index = atomic_requests ? __atomic_load_n(pindex, __ATOMIC_RELAXED) : *pindex;
		cnt->index_loads++;
		if (index) {
			float new_distance = 0.f;
			if (step_axis != -1) {
				*normal = V3(0, 0, 0);
				SET_AXIS(*normal, step_axis, -comp(step, step_axis));
				new_distance = comp(tmax, step_axis) - comp(tdelta, step_axis);
			}
			i3 diff = { camera_position.x - pos.x, camera_position.y - pos.y, camera_position.z - pos.z };
			int lod_distance_squared = diff.x * diff.x + diff.y * diff.y + diff.z * diff.z;
			float sub_distance = 0.f;
			int brick_id = pos.x + pos.y * cells + pos.z * cells * cells;
			if (lod_distance_squared > w->lod_distance_8x8x8) {
				*distance = new_distance * 8.f + tminn;
				hit->hit = 1; hit->level = 0; hit->brick_id = brick_id; hit->sub_id = 0;
				return 1;
			} else if (lod_distance_squared > w->lod_distance_2x2x2) {
				uint32_t byte = (index & BRICK_LOD_BITS) >> 12;
				uint32_t words[16];
				memset(words, 0, sizeof words);
				words[0] = byte;
				cnt->byte_tests++;
				v3 o2 = sub3(muls(add3(origin, muls(direction, new_distance)), 2.f), muls(muls(*normal, 0.2f), k_epsilon));
				int sub = 0;
				if (intersect_grid(o2, direction, normal, &sub_distance, 2, words, &sub, cnt)) {
					*distance = new_distance * 8.f + sub_distance * 4.f + tminn;
					hit->hit = 1; hit->level = 1; hit->brick_id = brick_id; hit->sub_id = sub;
					return 1;
				}
			} else {
				if (index & BRICK_LOADED_BIT) {
					const uint32_t* brick = sc->dev_bricks + (size_t)(index & BRICK_INDEX_BITS) * 16;
					cnt->brick_tests++;
					v3 o8 = sub3(muls(add3(origin, muls(direction, new_distance)), 8.f), muls(*normal, k_epsilon));
					int sub = 0;
					const uint64_t steps_before = cnt->voxel_steps;
					const int brick_hit = intersect_grid(o8, direction, normal, &sub_distance, 8, brick, &sub, cnt);
					if (orc_brick_probe_fn) { /* analysis door (tools/sim): one call per 8^3 test */
						const float po[3] = { o8.x - 8.f * (float)pos.x, o8.y - 8.f * (float)pos.y, o8.z - 8.f * (float)pos.z };
						const float pd[3] = { direction.x, direction.y, direction.z };
						orc_brick_probe_fn(po, pd, index, brick, brick_hit, (unsigned)(cnt->voxel_steps - steps_before));
					}
					if (brick_hit) {
						*distance = new_distance * 8.f + sub_distance + tminn;
						hit->hit = 1; hit->level = 2; hit->brick_id = brick_id; hit->sub_id = sub;
						return 1;
					}
				} else if (index & BRICK_UNLOADED_BIT) { /* :228-245 request protocol */
					uint32_t old;
					if (atomic_requests) old = __atomic_fetch_or(pindex, BRICK_REQUESTED_BIT, __ATOMIC_RELAXED);
					else { old = *pindex; *pindex = old | BRICK_REQUESTED_BIT; }
					if (!(old & BRICK_REQUESTED_BIT)) {
						uint32_t load_index;
						if (atomic_requests) load_index = __atomic_fetch_add(&w->load_queue_count, 1, __ATOMIC_RELAXED);
						else load_index = w->load_queue_count++;
						if (load_index < (uint32_t)w->queue_cap) {
							w->load_queue[load_index] = pos;
							cnt->requests++;
						} else {
							if (atomic_requests) __atomic_fetch_and(pindex, ~BRICK_REQUESTED_BIT, __ATOMIC_RELAXED);
							else *pindex &= ~BRICK_REQUESTED_BIT;
						}
					}
					*distance = new_distance * 8.f + tminn;
					hit->hit = 1; hit->level = 3; hit->brick_id = brick_id; hit->sub_id = 0;
					return 1;
				}
			}
		}
		step_axis = (tmax.x < tmax.y) ? ((tmax.x < tmax.z) ? 0 : 2) : ((tmax.y < tmax.z) ? 1 : 2);
		v3 mask;
		mask.x = (float)(tmax.x < tmax.y && tmax.x < tmax.z);
		mask.y = (float)(tmax.y <= tmax.x && tmax.y < tmax.z);
		mask.z = (float)(tmax.z <= tmax.x && tmax.z <= tmax.y);
		pos.x += (int)(mask.x * step.x);
		pos.y += (int)(mask.y * step.y);
		pos.z += (int)(mask.z * step.z);
		if (compi(pos, step_axis) == compi(out, step_axis)) break;
		tmax = add3(tmax, mul3(mask, tdelta));
	}
	return 0;
}

/* standalone entry points for unit tests on hand-built data */
ORC_API int orc_intersect_brick(const float* origin, const float* direction, float* normal_io, float* distance_out, const uint32_t* brick16, int* sub_id) {
	orc_counters c; memset(&c, 0, sizeof c);
	v3 n = V3(normal_io[0], normal_io[1], normal_io[2]);
	int r = intersect_grid(V3(origin[0], origin[1], origin[2]), V3(direction[0], direction[1], direction[2]), &n, distance_out, 8, brick16, sub_id, &c);
	normal_io[0] = n.x; normal_io[1] = n.y; normal_io[2] = n.z;
	return r;
}
ORC_API int orc_intersect_byte(const float* origin, const float* direction, float* normal_io, float* distance_out, uint32_t byte, int* sub_id) {
	orc_counters c; memset(&c, 0, sizeof c);
	uint32_t words[16]; memset(words, 0, sizeof words); words[0] = byte & 0xFFu;
	v3 n = V3(normal_io[0], normal_io[1], normal_io[2]);
	int r = intersect_grid(V3(origin[0], origin[1], origin[2]), V3(direction[0], direction[1], direction[2]), &n, distance_out, 2, words, sub_id, &c);
	normal_io[0] = n.x; normal_io[1] = n.y; normal_io[2] = n.z;
	return r;
}
/* out4 = {hit, level, brick_id, sub_id} */
ORC_API int orc_intersect_voxel(orc_world* w, const float* origin, const float* direction, float* normal_io, float* distance_io, const int* campos, int* out4, uint64_t* index_loads) {
	orc_counters c; memset(&c, 0, sizeof c);
	orc_hit h;
	v3 n = V3(normal_io[0], normal_io[1], normal_io[2]);
	i3 cp = { campos[0], campos[1], campos[2] };
	int r = intersect_voxel(w, V3(origin[0], origin[1], origin[2]), V3(direction[0], direction[1], direction[2]), &n, distance_io, cp, &h, &c, 0);
	normal_io[0] = n.x; normal_io[1] = n.y; normal_io[2] = n.z;
	out4[0] = h.hit; out4[1] = h.level; out4[2] = h.brick_id; out4[3] = h.sub_id;
	if (index_loads) *index_loads = c.index_loads;
	return r;
}

/* ---------------------------------------------------------------- RNG + samplers (kernel.cu:19-103) */
static inline unsigned RandomInt(unsigned* seed) { /* :19-24 */
	*seed ^= *seed << 13;
	*seed ^= *seed >> 17;
	*seed ^= *seed << 5;
	return *seed;
}
static inline float RandomFloat(unsigned* seed) { return RandomInt(seed) * 2.3283064365387e-10f; }   /* :27-29 */
static inline float RandomFloat2(unsigned* seed) { return (RandomInt(seed) >> 16) / 65535.0f; }       /* :31-33 */
static inline int RandomIntBetween0AndMax(unsigned* seed, int max) { return (int)(RandomFloat(seed) * (max + 0.99999f)); } /* :35-37 */
static void Random2DStratifiedSample(unsigned* seed, float* sx, float* sy) { /* :40-61 */
	const int width2D = 4, height2D = 4;
	const float pixelWidth = 1.0f / width2D, pixelHeight = 1.0f / height2D;
	const int chosenStratum = RandomIntBetween0AndMax(seed, width2D * height2D);
	const int stratumX = chosenStratum % width2D;
	const int stratumY = (chosenStratum / width2D) % height2D;
	const float stratumXStart = pixelWidth * stratumX;
	const float stratumYStart = pixelHeight * stratumY;
	*sx = stratumXStart + (RandomFloat(seed) * pixelWidth);
	*sy = stratumYStart + (RandomFloat(seed) * pixelHeight);
}
static void computeOrthonormalBasisNaive(v3 w, v3* u, v3* v) { /* :76-84 */
	if (fabs((double)w.x) > .9) *u = V3(0.0f, 1.0f, 0.0f); else *u = V3(1.0f, 0.0f, 0.0f);
	*u = normalize3(cross3(*u, w));
	*v = cross3(w, *u);
}
static void ConcentricSampleDisk(float ux, float uy, float* ox, float* oy) { /* :85-103 */
	float offx = 2.f * ux - 1.f, offy = 2.f * uy - 1.f;
	if (offx == 0 && offy == 0) { *ox = 0; *oy = 0; return; }
	float theta, r;
	if (fabsf(offx) > fabsf(offy)) { r = offx; theta = k_pi / 4 * (offy / offx); }
	else { r = offy; theta = k_pi / 2 - k_pi / 4 * (offx / offy); }
	*ox = r * det_cos(theta);
	*oy = r * det_sin(theta);
}
ORC_API void orc_rng_stream(unsigned seed, int n, unsigned* out_int) { for (int i = 0; i < n; i++) out_int[i] = RandomInt(&seed); }
ORC_API void orc_rng_floats(unsigned seed, int n, float* f1, float* f2) {
	unsigned a = seed, b = seed;
	for (int i = 0; i < n; i++) { f1[i] = RandomFloat(&a); f2[i] = RandomFloat2(&b); }
}
ORC_API void orc_stratified(unsigned seed, float* out2, unsigned* seed_after) { Random2DStratifiedSample(&seed, &out2[0], &out2[1]); *seed_after = seed; }

/* ---------------------------------------------------------------- sun / sky (sunsky.cu, sunsky.cuh:25-42) */
static const float sunSize = 1.5f;
static const float cutoffAngle = 3.1415926535897932f / 1.95f;
static const float steepness = 1.5f;
static const float SkyFactor = 1.f;
static const float turbidity = 1.f;
static const float mieCoefficient = 0.005f;
static const float mieDirectionalG = 0.80f;
static const float sky_v = 4.0f;
static const float rayleighZenithLength = 8.4E3f;
static const float mieZenithLength = 1.25E3f;
static const float sunIntensity = 1000.0f;

typedef struct {
	v3 sunDirection;
	float sunAngularDiameterCos;
} orc_sky_state;

static float RayleighPhase(float c) { return (float)((3.0 / (16.0 * (double)k_pi)) * (1.0 + (double)powf(c, 2.0f))); } /* :10-12 */
static v3 totalMie(v3 lambda, v3 K, float T) { /* :14-18 */
	float c = (float)((0.2 * (double)T) * 10E-18);
	float s = 0.434f * c * k_pi;
	float e = (float)((double)sky_v - 2.0);
	v3 p = V3(powf((2.0f * k_pi) / lambda.x, e), powf((2.0f * k_pi) / lambda.y, e), powf((2.0f * k_pi) / lambda.z, e));
	return mul3(muls(p, s), K);
}
static float hgPhase(float c, float g) { /* :20-22 */
	return (float)((1.0 / (4.0 * (double)k_pi)) * ((1.0 - (double)powf(g, 2.0f)) / pow(1.0 - 2.0 * (double)g * (double)c + (double)powf(g, 2.0f), 1.5)));
}
static float SunIntensity(float zenithAngleCos) { /* :24-26 */
	double e = 1.0 - (double)expf(-((cutoffAngle - acosf(zenithAngleCos)) / steepness));
	return (float)((double)sunIntensity * (0.0 < e ? e : 0.0)); /* glm::max(0.0, e) = (0.0<e)?e:0.0 */
}
static v3 fromSpherical(float px, float py) { /* :28-30 (host; float overloads of cos/sin) */
	return V3(cosf(px) * sinf(py), sinf(px) * sinf(py), cosf(py));
}
typedef struct { float sunE; v3 rayleighAtX, mieAtX, Fex, somethingElse; float cosViewSun, mixf; } sky_common;
static const v3 sky_up = { 0.0f, 0.0f, 1.0f };
static void sky_eval_common(const orc_sky_state* st, v3 viewDir, int rayleigh_float_literals, sky_common* o) {
	const v3 K = V3(0.686f, 0.678f, 0.666f);
	const v3 primaryWavelengths = V3(680E-9f, 550E-9f, 450E-9f);
	float cosViewSunAngle = dot3(viewDir, st->sunDirection);
	float cosSunUpAngle = dot3(st->sunDirection, sky_up);
	float cosUpViewAngle = dot3(sky_up, viewDir);
	(void)rayleigh_float_literals; /* double literals narrowed to float give the same floats as the f-suffixed ones */
	o->sunE = SunIntensity(cosSunUpAngle);
	o->rayleighAtX = V3(5.176821E-6f, 1.2785348E-5f, 2.8530756E-5f);
	o->mieAtX = muls(totalMie(primaryWavelengths, K, turbidity), mieCoefficient);
	float zenithAngle = gmax(0.0f, cosUpViewAngle);
	float rayleighOpticalLength = rayleighZenithLength / zenithAngle;
	float mieOpticalLength = mieZenithLength / zenithAngle;
	v3 a = add3(muls(o->rayleighAtX, rayleighOpticalLength), muls(o->mieAtX, mieOpticalLength));
	o->Fex = V3(expf(-a.x), expf(-a.y), expf(-a.z));
	v3 rayleighXtoEye = muls(o->rayleighAtX, RayleighPhase(cosViewSunAngle));
	v3 mieXtoEye = muls(o->mieAtX, hgPhase(cosViewSunAngle, mieDirectionalG));
	v3 totalLightAtX = add3(o->rayleighAtX, o->mieAtX);
	v3 lightFromXtoEye = add3(rayleighXtoEye, mieXtoEye);
	o->somethingElse = muls(div3(lightFromXtoEye, totalLightAtX), o->sunE); /* sunE * (a/b) */
	o->cosViewSun = cosViewSunAngle;
	float m = powf(1.0f - dot3(sky_up, st->sunDirection), 5.0f);
	o->mixf = gmin(gmax(m, 0.0f), 1.0f); /* glm::clamp = min(max(x,lo),hi) */
}
static v3 sky_term(const sky_common* c) { /* sky = sE*(1-Fex); sky *= mix(1, pow(sE*Fex, .5), mixf) */
	v3 sky = mul3(c->somethingElse, V3(1.0f - c->Fex.x, 1.0f - c->Fex.y, 1.0f - c->Fex.z));
	v3 q = mul3(c->somethingElse, c->Fex);
	v3 p = V3(powf(q.x, 0.5f), powf(q.y, 0.5f), powf(q.z, 0.5f));
	float a = c->mixf;
	v3 mx = V3(1.0f * (1.0f - a) + p.x * a, 1.0f * (1.0f - a) + p.y * a, 1.0f * (1.0f - a) + p.z * a);
	return mul3(sky, mx);
}
static v3 sky_sun(const orc_sky_state* st, v3 viewDir) { /* sun(): sunsky.cu:32-74 */
	sky_common c;
	sky_eval_common(st, viewDir, 0, &c);
	/* quirk kept: (cosViewSunAngle ? 1.0 : 0.0) tests non-zero */
	float sundisk = (float)((double)st->sunAngularDiameterCos < (c.cosViewSun != 0.0f ? 1.0 : 0.0));
	v3 sun = muls(muls(c.Fex, c.sunE * 19000.0f), sundisk);
	return muls(sun, 0.01f);
}
static v3 sky_sky(const orc_sky_state* st, v3 viewDir) { /* sky(): :76-114 */
	sky_common c;
	sky_eval_common(st, viewDir, 0, &c);
	return muls(sky_term(&c), SkyFactor * 0.01f);
}
static v3 sky_sunsky(const orc_sky_state* st, v3 viewDir) { /* sunsky(): :116-161 */
	if (st->sunAngularDiameterCos == 1.0f) return V3(1.0f, 0.0f, 0.0f);
	sky_common c;
	sky_eval_common(st, viewDir, 1, &c);
	v3 sky = sky_term(&c);
	float e0 = st->sunAngularDiameterCos, e1 = st->sunAngularDiameterCos + 0.00002f;
	float t = gmin(gmax((c.cosViewSun - e0) / (e1 - e0), 0.0f), 1.0f);
	float sundisk = t * t * (3.0f - 2.0f * t);
	v3 sun = muls(muls(muls(c.Fex, c.sunE * 19000.0f), sundisk), 1E-5f);
	return muls(add3(sun, sky), 0.01f);
}
static v3 ortho(v3 v) { return fabsf(v.x) > fabsf(v.z) ? V3(-v.y, v.x, 0.0f) : V3(0.0f, -v.z, v.y); } /* :163-166 */
static v3 getConeSample(v3 dir, float extent, unsigned* seed) { /* :170-183 */
	dir = normalize3(dir);
	v3 o1 = normalize3(ortho(dir));
	v3 o2 = normalize3(cross3(dir, o1));
	float rx = RandomFloat2(seed);
	float ry = RandomFloat2(seed);
	rx = rx * 2.f * k_pi;
	ry = 1.0f - ry * extent;
	float oneminus = sqrtf(1.0f - ry * ry);
	float s, c;
	orc_sincos_impl(rx, &s, &c);
	return add3(add3(muls(o1, c * oneminus), muls(o2, s * oneminus)), muls(dir, ry));
}
/* launch_kernels:374,393 */
static void sky_state_init(orc_sky_state* st, float sun_x, float sun_y) {
	st->sunAngularDiameterCos = cosf(sunSize * k_pi / 180.f);
	/* (sun_position - glm::vec2(0.0, 0.5)) * glm::vec2(6.28f, 3.14f), all in float */
	float px = (sun_x - 0.0f) * 6.28f;
	float py = (sun_y - 0.5f) * 3.14f;
	st->sunDirection = normalize3(fromSpherical(px, py));
}
ORC_API void orc_sky_probe(float sun_x, float sun_y, const float* viewdir, float* sun_dir3, float* sun3, float* sky3, float* sunsky3) {
	orc_sky_state st;
	sky_state_init(&st, sun_x, sun_y);
	v3 d = V3(viewdir[0], viewdir[1], viewdir[2]);
	v3 a = sky_sun(&st, d), b = sky_sky(&st, d), c = sky_sunsky(&st, d);
	sun_dir3[0] = st.sunDirection.x; sun_dir3[1] = st.sunDirection.y; sun_dir3[2] = st.sunDirection.z;
	sun3[0] = a.x; sun3[1] = a.y; sun3[2] = a.z;
	sky3[0] = b.x; sky3[1] = b.y; sky3[2] = b.z;
	sunsky3[0] = c.x; sunsky3[1] = c.y; sunsky3[2] = c.z;
}
ORC_API void orc_cone_sample(float sun_x, float sun_y, unsigned seed, float* out3, unsigned* seed_after) {
	orc_sky_state st;
	sky_state_init(&st, sun_x, sun_y);
	v3 r = getConeSample(st.sunDirection, 1.0f - st.sunAngularDiameterCos, &seed);
	out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
	*seed_after = seed;
}

/* ---------------------------------------------------------------- camera + frame description */
typedef struct {
	float position[3];
	float direction[3];
	float up[3];
	float focal_distance; /* camera.h:8  default 1 */
	float lens_radius;    /* camera.h:9  default 0 */
} orc_camera;

typedef struct {
	int width, height;
	int spp;           /* samples per pixel rendered by this call                          */
	int sample_base;   /* absolute index of the first sample (progressive accumulation)    */
	int max_bounces;   /* kernel.cu:13 MAX_BOUNCES (3 => up to 4 segments per path)         */
	unsigned base_frame; /* kernel.cu:369 `frame`, starts at 1                              */
	int primary_only;  /* bit 0: config 1 -- extend the primary ray only, no shadow ray, no bounce;
	                      bit 1: dbg holds the order-independent RAY DIGEST (render_pixel)      */
	int band_rows, shard_rank, shard_count; /* interleaved row bands; (height,0,1) = whole image */
	float sun_x, sun_y; /* variables.cpp:3 default (0.05, 0.1)                             */
} orc_frame;

typedef struct { v3 right, up, dir, O; i3 campos; } cam_basis;
static void camera_basis(const orc_camera* cam, int W, int H, cam_basis* b) { /* launch_kernels:384-385, :416-418 */
	v3 d = V3(cam->direction[0], cam->direction[1], cam->direction[2]);
	v3 u = V3(cam->up[0], cam->up[1], cam->up[2]);
	float aspect = (float)W / (float)H;
	b->right = muls(muls(normalize3(cross3(d, u)), 1.5f), aspect);
	b->up = muls(normalize3(cross3(b->right, d)), 1.5f);
	b->dir = d;
	b->O = V3(cam->position[0], cam->position[1], cam->position[2]);
	v3 c8 = divs(b->O, 8.f);
	b->campos.x = (int)c8.x; b->campos.y = (int)c8.y; b->campos.z = (int)c8.z;
}
/* Camera::update (camera.cpp:48-54): direction from angles, computed in double then narrowed, then glm::normalize */
ORC_API void orc_camera_direction(double horizontal_angle, double vertical_angle, float* out3) {
	v3 d = V3((float)(cos(vertical_angle) * sin(horizontal_angle)), (float)(cos(vertical_angle) * cos(horizontal_angle)), (float)sin(vertical_angle));
	d = normalize3(d);
	out3[0] = d.x; out3[1] = d.y; out3[2] = d.z;
}

typedef struct { v3 origin, direction; unsigned pixel_index; } primary_ray;
/* body of primary_rays (kernel.cu:157-200) for queue slot `index`, with start_position given */
static void make_primary(const cam_basis* b, const orc_camera* cam, unsigned frame, unsigned index, unsigned start_position, unsigned W, unsigned H, primary_ray* out) {
	unsigned seed = (frame * 147565741u) * 720898027u * index;
	const unsigned x = (start_position + index) % W;
	const unsigned y = ((start_position + index) / W) % H;
	float sx, sy;
	Random2DStratifiedSample(&seed, &sx, &sy);
	const float rand_point_pixelX = x - sx;
	const float rand_point_pixelY = y - sy;
	const float normalized_i = (rand_point_pixelX / (float)W) - 0.5f;
	const float normalized_j = ((H - rand_point_pixelY) / (float)H) - 0.5f;
	v3 directionToFocalPlane = add3(add3(b->dir, muls(b->right, normalized_i)), muls(b->up, normalized_j));
	directionToFocalPlane = normalize3(directionToFocalPlane);
	const int ImGui_slider_hack = 3;
	v3 convergencePoint = add3(b->O, muls(directionToFocalPlane, cam->focal_distance * ImGui_slider_hack));
	/* canonical argument order: left to right */
	float l0 = RandomFloat(&seed);
	float l1 = RandomFloat(&seed);
	float dx, dy;
	ConcentricSampleDisk(l0, l1, &dx, &dy);
	float plx = cam->lens_radius * dx, ply = cam->lens_radius * dy;
	v3 newOrigin = add3(add3(b->O, muls(b->right, plx)), muls(b->up, ply));
	out->origin = newOrigin;
	out->direction = normalize3(sub3(convergencePoint, newOrigin));
	out->pixel_index = y * W + x;
}

/* hashing of per-path events for bit-exact comparison of hit indices */
static inline uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint32_t hmix(uint32_t h, uint32_t v) { h ^= v; h *= 16777619u; h ^= h >> 15; return h; }
static inline uint32_t pack_normal(v3 n) {
	uint32_t r = 0;
	float c[3] = { n.x, n.y, n.z };
	for (int i = 0; i < 3; i++) {
		uint32_t code = c[i] == 0.0f ? 0u : (c[i] == 1.0f ? 1u : (c[i] == -1.0f ? 2u : 3u));
		r |= code << (2 * i);
	}
	return r;
}

/* ---------------------------------------------------------------- mode B: canonical per-pixel render
 * Canonical schedule (DESIGN.md "Canonical path"): sample s of pixel p uses queue slot
 * slot = p + s*W*H and frame = base_frame + bounce; its events are accumulated in path order
 * (bounce 0 shade, bounce 0 connect, bounce 1 shade, ...), samples in increasing s.
 * dbg (optional) holds 8 uint32 per pixel:
 *  [0] primary distance bits  [1] packed normal | hit<<8 | level<<12  [2] brick id  [3] sub id
 *  [4] hash over all extend segments  [5] hash over all shadow rays  [6] extend rays | shadow rays<<16
 *  [7] index loads (outer DDA iterations) of this pixel's rays
 * With primary_only bit 1 (the product's BM_FLAG_RAY_DIGEST: frames whose rays are traced by whichever lane is free, in any order)
 * words 4-7 are SUMS (mod 2^32) over the pixel's rays instead of chains over them:
 *  [4] sum over its extend rays of  E = hmix(... hmix(2166136261, key), is_hit) [, distance bits, normal | level<<12, brick id, sub id]
 *  [5] sum over its shadow rays of  S = hmix(hmix(2166136261, key), occluded) [, brick id, sub id | level<<12]
 *      key = sample << 8 | segment: sample counted from 0 inside the call, segment = number of the path's extend ray (the one that
 *      produced the shadow ray, for S) -- the keyed hash makes "the same rays" mean "the same ray at the same place of the same path"
 *  [6] extend rays | shadow rays<<16   [7] index loads, both as before (sums anyway)
 */
typedef struct {
	orc_world* w;
	const orc_camera* cam;
	const orc_frame* f;
	float* accum;   /* full frame W*H*4 */
	uint32_t* dbg;  /* full frame W*H*8 or NULL */
	orc_counters cnt;
	volatile int* next_row;
	int atomic_requests;
	char pad[128]; /* one job per thread in an array: keep the per-thread counters of neighbours off each other's cache lines */
} __attribute__((aligned(128))) render_job;

/* Optional second output of orc_render (test infrastructure): per pixel of the full frame the two ray-digest sums (words 4 and 5 of a
 * primary_only-bit-1 record), so that ONE render yields both the hash chains and the sums.  NULL = off. */
static uint32_t* orc_ray_digest_out = NULL;
ORC_API void orc_set_ray_digest_buffer(uint32_t* two_words_per_pixel) { orc_ray_digest_out = two_words_per_pixel; }

static int row_in_shard(const orc_frame* f, int y) {
	int band = f->band_rows > 0 ? f->band_rows : f->height;
	int count = f->shard_count > 0 ? f->shard_count : 1;
	return (y / band) % count == f->shard_rank;
}

static void render_pixel(render_job* job, const cam_basis* cb, const orc_sky_state* sky, unsigned x, unsigned y) {
	const orc_frame* f = job->f;
	orc_world* w = job->w;
	const unsigned W = (unsigned)f->width, H = (unsigned)f->height;
	const unsigned p = y * W + x;
	float* acc = job->accum + (size_t)p * 4;
	float r = acc[0], g = acc[1], b = acc[2], a = acc[3];
	uint32_t d0 = 0, d1 = 0, d2 = 0xFFFFFFFFu, d3 = 0, hseg = 2166136261u, hsh = 2166136261u, next = 0, nsh = 0;
	uint64_t loads_before = job->cnt.index_loads;
	const int ray_digest = (f->primary_only & 2) != 0, primary_only = (f->primary_only & 1) != 0;
	uint32_t sum_e = 0, sum_s = 0;
	for (int s = 0; s < f->spp; s++) {
		int seg = -1; /* number of the path's current extend ray */
		const unsigned slot = p + (unsigned)(f->sample_base + s) * W * H;
		primary_ray pr;
		make_primary(cb, job->cam, f->base_frame, slot, 0u, W, H, &pr);
		v3 origin = pr.origin, direction = pr.direction, throughput = V3(1.f, 1.f, 1.f), normal = V3(0.f, 0.f, 0.f);
		int bounces = 0;
		job->cnt.paths++;
		for (;;) {
			/* extend (kernel.cu:226-238) */
			float distance = VERY_FAR;
			orc_hit hit;
			if (orc_ray_probe_fn) orc_ray_probe_fn(p, s, 0, &origin.x, &direction.x);
			intersect_voxel(w, origin, direction, &normal, &distance, cb->campos, &hit, &job->cnt, job->atomic_requests);
			job->cnt.extend_rays++;
			next++;
			int is_hit = distance < VERY_FAR;
			if (s == 0 && bounces == 0) {
				d0 = is_hit ? fbits(distance) : 0u;
				d1 = is_hit ? (pack_normal(normal) | (1u << 8) | ((uint32_t)hit.level << 12)) : 0u;
				d2 = is_hit ? (uint32_t)hit.brick_id : 0xFFFFFFFFu;
				d3 = is_hit ? (uint32_t)hit.sub_id : 0u;
			}
			hseg = hmix(hseg, (uint32_t)is_hit);
			if (is_hit) {
				hseg = hmix(hseg, fbits(distance));
				hseg = hmix(hseg, pack_normal(normal) | ((uint32_t)hit.level << 12));
				hseg = hmix(hseg, (uint32_t)hit.brick_id);
				hseg = hmix(hseg, (uint32_t)hit.sub_id);
			}
			seg++;
			{
				uint32_t e = hmix(hmix(2166136261u, ((uint32_t)s << 8) | (uint32_t)seg), (uint32_t)is_hit);
				if (is_hit) {
					e = hmix(e, fbits(distance));
					e = hmix(e, pack_normal(normal) | ((uint32_t)hit.level << 12));
					e = hmix(e, (uint32_t)hit.brick_id);
					e = hmix(e, (uint32_t)hit.sub_id);
				}
				sum_e += e;
			}
			if (primary_only) {
				if (!is_hit) { v3 c = mul3(throughput, sky_sunsky(sky, direction)); r += c.x; g += c.y; b += c.z; }
				a += 1.f;
				break;
			}
			/* shade (kernel.cu:242-325) */
			const unsigned frame = f->base_frame + (unsigned)bounces;
			unsigned seed = (frame * pr.pixel_index * 147565741u) * 720898027u * slot;
			if (is_hit) {
				origin = add3(origin, muls(direction, distance));
				origin = add3(origin, muls(muls(normal, 2.f), k_epsilon));
				throughput = mul3(throughput, V3(1.f, 1.f, 1.f));
				v3 sunSampleDir = getConeSample(sky->sunDirection, 1.0f - sky->sunAngularDiameterCos, &seed);
				float sunLight = dot3(normal, sunSampleDir);
				int cast = sunLight > 0.f;
				v3 scolor = V3(0, 0, 0);
				if (cast) scolor = muls(muls(mul3(throughput, sky_sun(sky, sunSampleDir)), sunLight), 1E-5f);
				v3 shadow_origin = origin;
				int terminated = 0;
				if (bounces < f->max_bounces) {
					float r1 = 2.f * k_pi * RandomFloat(&seed);
					float r2 = RandomFloat(&seed);
					float r2s = sqrtf(r2);
					v3 u, v;
					computeOrthonormalBasisNaive(normal, &u, &v);
					float sn, cs;
					orc_sincos_impl(r1, &sn, &cs);
					direction = normalize3(add3(add3(muls(muls(u, cs), r2s), muls(muls(v, sn), r2s)), muls(normal, sqrtf(1 - r2))));
					bounces++;
				} else {
					a += 1.f; /* :301 */
					terminated = 1;
				}
				/* connect (kernel.cu:328-346) runs after shade within the same frame */
				if (cast) {
					v3 yn = V3(0, 0, 0);
					float t = 0.f;
					orc_hit sh;
					if (orc_ray_probe_fn) orc_ray_probe_fn(p, s, 1, &shadow_origin.x, &sunSampleDir.x);
					int occluded = intersect_voxel(w, shadow_origin, sunSampleDir, &yn, &t, cb->campos, &sh, &job->cnt, job->atomic_requests);
					job->cnt.shadow_rays++;
					nsh++;
					hsh = hmix(hsh, (uint32_t)occluded);
					if (occluded) { hsh = hmix(hsh, (uint32_t)sh.brick_id); hsh = hmix(hsh, (uint32_t)sh.sub_id | ((uint32_t)sh.level << 12)); }
					{
						uint32_t e = hmix(hmix(2166136261u, ((uint32_t)s << 8) | (uint32_t)seg), (uint32_t)occluded);
						if (occluded) { e = hmix(e, (uint32_t)sh.brick_id); e = hmix(e, (uint32_t)sh.sub_id | ((uint32_t)sh.level << 12)); }
						sum_s += e;
					}
					if (!occluded) { r += scolor.x; g += scolor.y; b += scolor.z; }
				}
				if (terminated) break;
			} else {
				v3 c = mul3(throughput, bounces == 0 ? sky_sunsky(sky, direction) : sky_sky(sky, direction));
				r += c.x; g += c.y; b += c.z; a += 1.f;
				break;
			}
		}
	}
	acc[0] = r; acc[1] = g; acc[2] = b; acc[3] = a;
	if (orc_ray_digest_out) { orc_ray_digest_out[(size_t)p * 2] = sum_e; orc_ray_digest_out[(size_t)p * 2 + 1] = sum_s; }
	if (job->dbg) {
		uint32_t* d = job->dbg + (size_t)p * 8;
		d[0] = d0; d[1] = d1; d[2] = d2; d[3] = d3; d[4] = ray_digest ? sum_e : hseg; d[5] = ray_digest ? sum_s : hsh; d[6] = next | (nsh << 16);
		d[7] = (uint32_t)(job->cnt.index_loads - loads_before);
	}
}

/* Work is handed out in tiles of 64 x 8 pixels (a row-granular queue leaves most of a many-core host idle at the end of
 * the frame: 1080 rows over 256 threads, and a terrain row costs several times a sky row). */
#define ORC_TILE_W 64
#define ORC_TILE_H 8
static void* render_worker(void* arg) {
	render_job* job = (render_job*)arg;
	const orc_frame* f = job->f;
	cam_basis cb;
	camera_basis(job->cam, f->width, f->height, &cb);
	orc_sky_state sky;
	sky_state_init(&sky, f->sun_x, f->sun_y);
	const int tiles_x = (f->width + ORC_TILE_W - 1) / ORC_TILE_W, tiles_y = (f->height + ORC_TILE_H - 1) / ORC_TILE_H;
	for (;;) {
		int t = __atomic_fetch_add(job->next_row, 1, __ATOMIC_RELAXED);
		if (t >= tiles_x * tiles_y) break;
		const int x0 = (t % tiles_x) * ORC_TILE_W, y0 = (t / tiles_x) * ORC_TILE_H;
		for (int y = y0; y < y0 + ORC_TILE_H && y < f->height; y++) {
			if (!row_in_shard(f, y)) continue;
			for (int x = x0; x < x0 + ORC_TILE_W && x < f->width; x++) render_pixel(job, &cb, &sky, (unsigned)x, (unsigned)y);
		}
	}
	return NULL;
}

/* Renders the shard's rows of `accum` (full-frame float4 buffer, accumulated into).  Returns seconds. */
ORC_API double orc_render(orc_world* w, const orc_camera* cam, const orc_frame* f, float* accum, uint32_t* dbg, orc_counters* counters_out, int threads) {
	if (threads < 1) threads = 1;
	if (threads > 256) threads = 256;
	volatile int next_row = 0;
	render_job* jobs = (render_job*)aligned_alloc(128, (size_t)threads * sizeof(render_job));
	memset(jobs, 0, (size_t)threads * sizeof(render_job));
	pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (int i = 0; i < threads; i++) {
		jobs[i].w = w; jobs[i].cam = cam; jobs[i].f = f; jobs[i].accum = accum; jobs[i].dbg = dbg;
		jobs[i].next_row = &next_row; jobs[i].atomic_requests = threads > 1;
		if (threads > 1) pthread_create(&th[i], NULL, render_worker, &jobs[i]);
	}
	if (threads == 1) render_worker(&jobs[0]);
	else for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	if (counters_out) {
		uint64_t* dst = (uint64_t*)counters_out;
		for (int i = 0; i < threads; i++) {
			const uint64_t* src = (const uint64_t*)&jobs[i].cnt;
			for (size_t k = 0; k < sizeof(orc_counters) / 8; k++) dst[k] += src[k];
		}
	}
	free(jobs); free(th);
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---------------------------------------------------------------- mode A: reference wavefront, sequential schedule */
typedef struct { /* variables.h:43-52, 64 bytes */
	v3 origin, direction, throughput, normal;
	float distance;
	int identifier;
	int bounces;
	unsigned pixel_index;
} RayQueue;
typedef struct { v3 origin, direction, color; unsigned pixel_index; } ShadowQueue; /* variables.h:54-59 */

typedef struct {
	unsigned queue_size; /* ray_queue_buffer_size, variables.h:61 = 2*1048576 */
	RayQueue* work;
	RayQueue* next;
	ShadowQueue* shadow;
	unsigned primary_ray_cnt, start_position, shadow_ray_cnt; /* kernel.cu:106-119 */
	unsigned frame;                                            /* kernel.cu:369    */
	int max_bounces;
	orc_counters cnt;
	/* stats of the last frame */
	unsigned last_survivors, last_shadow, last_generated;
} orc_wavefront;

ORC_API orc_wavefront* orc_wavefront_create(unsigned queue_size, int max_bounces) {
	orc_wavefront* s = (orc_wavefront*)calloc(1, sizeof(orc_wavefront));
	s->queue_size = queue_size;
	s->work = (RayQueue*)calloc(queue_size, sizeof(RayQueue));
	s->next = (RayQueue*)calloc(queue_size, sizeof(RayQueue));
	s->shadow = (ShadowQueue*)calloc(queue_size, sizeof(ShadowQueue));
	s->frame = 1;
	s->max_bounces = max_bounces;
	return s;
}
ORC_API void orc_wavefront_destroy(orc_wavefront* s) { if (s) { free(s->work); free(s->next); free(s->shadow); free(s); } }
/* reset_buffer branch of launch_kernels (:397-403): caller zeroes accum; primary_ray_cnt = 0 */
ORC_API void orc_wavefront_reset(orc_wavefront* s) { s->primary_ray_cnt = 0; }
ORC_API void orc_wavefront_stats(const orc_wavefront* s, unsigned* out6) {
	out6[0] = s->last_survivors; out6[1] = s->last_shadow; out6[2] = s->start_position; out6[3] = s->frame; out6[4] = s->last_generated; out6[5] = s->primary_ray_cnt;
}
/* test door: copy records out of the work queue (which = 0; after a frame its first `survivors` slots hold the
 * continuing paths) or the shadow queue (which = 1) */
ORC_API int orc_wavefront_read_queue(const orc_wavefront* s, int which, unsigned first, unsigned count, void* out) {
	if (first > s->queue_size || count > s->queue_size - first) return -1;
	if (which == 0) memcpy(out, s->work + first, (size_t)count * sizeof(RayQueue));
	else if (which == 2) memcpy(out, s->next + first, (size_t)count * sizeof(RayQueue)); /* the rays extend traced in the last frame (queue before the swap) */
	else memcpy(out, s->shadow + first, (size_t)count * sizeof(ShadowQueue));
	return 0;
}
ORC_API void orc_wavefront_counters(const orc_wavefront* s, orc_counters* out) { *out = s->cnt; }

/* One call of launch_kernels (kernel.cu:366-439) followed by process_load_queue + swap (main.cpp:142-146). */
ORC_API void orc_wavefront_frame(orc_wavefront* s, orc_world* w, const orc_camera* cam, int W, int H, float sun_x, float sun_y, float* accum) {
	cam_basis cb;
	camera_basis(cam, W, H, &cb);
	orc_sky_state sky;
	sky_state_init(&sky, sun_x, sun_y);
	const unsigned Q = s->queue_size;
	if (!w->overlapped) orc_upload(w); /* :407-414 */
	/* primary_rays :154-223 */
	unsigned generated = 0;
	for (unsigned index = 0;; index++) {
		const unsigned ray_index_buffer = index + s->primary_ray_cnt;
		if (ray_index_buffer > Q - 1) break;
		primary_ray pr;
		make_primary(&cb, cam, s->frame, index, s->start_position, (unsigned)W, (unsigned)H, &pr);
		RayQueue* r = &s->work[ray_index_buffer];
		r->origin = pr.origin; r->direction = pr.direction; r->throughput = V3(1.f, 1.f, 1.f); r->normal = V3(0.f, 0.f, 0.f);
		r->distance = 0.f; r->identifier = 0; r->bounces = 0; r->pixel_index = pr.pixel_index;
		generated++;
	}
	s->last_generated = generated;
	/* set_wavefront_globals :122-139 */
	{
		const unsigned progress_last_frame = Q - s->primary_ray_cnt;
		s->start_position += progress_last_frame;
		s->start_position = s->start_position % ((unsigned)W * (unsigned)H);
		s->shadow_ray_cnt = 0;
		s->primary_ray_cnt = 0;
	}
	/* extend :226-238 */
	for (unsigned index = 0; index < Q; index++) {
		RayQueue* ray = &s->work[index];
		ray->distance = VERY_FAR;
		orc_hit h;
		intersect_voxel(w, ray->origin, ray->direction, &ray->normal, &ray->distance, cb.campos, &h, &s->cnt, 0);
		s->cnt.extend_rays++;
	}
	/* shade :242-325 */
	for (unsigned index = 0; index < Q; index++) {
		RayQueue ray = s->work[index];
		unsigned seed = (s->frame * ray.pixel_index * 147565741u) * 720898027u * index;
		float* px = accum + (size_t)ray.pixel_index * 4;
		if (ray.distance < VERY_FAR) {
			ray.origin = add3(ray.origin, muls(ray.direction, ray.distance));
			ray.origin = add3(ray.origin, muls(muls(ray.normal, 2.f), k_epsilon));
			ray.throughput = mul3(ray.throughput, V3(1.f, 1.f, 1.f));
			v3 sunSampleDir = getConeSample(sky.sunDirection, 1.0f - sky.sunAngularDiameterCos, &seed);
			float sunLight = dot3(ray.normal, sunSampleDir);
			if (sunLight > 0.f) {
				unsigned shadow_index = s->shadow_ray_cnt++;
				ShadowQueue* sq = &s->shadow[shadow_index];
				sq->origin = ray.origin; sq->direction = sunSampleDir;
				sq->color = muls(muls(mul3(ray.throughput, sky_sun(&sky, sunSampleDir)), sunLight), 1E-5f);
				sq->pixel_index = ray.pixel_index;
			}
			if (ray.bounces < s->max_bounces) {
				float r1 = 2.f * k_pi * RandomFloat(&seed);
				float r2 = RandomFloat(&seed);
				float r2s = sqrtf(r2);
				v3 u, v;
				computeOrthonormalBasisNaive(ray.normal, &u, &v);
				float sn, cs;
				orc_sincos_impl(r1, &sn, &cs);
				ray.direction = normalize3(add3(add3(muls(muls(u, cs), r2s), muls(muls(v, sn), r2s)), muls(ray.normal, sqrtf(1 - r2))));
				ray.bounces++;
				unsigned primary_index = s->primary_ray_cnt++;
				s->next[primary_index] = ray;
			} else {
				px[3] += 1.f;
			}
		} else {
			v3 color = mul3(ray.throughput, ray.bounces == 0 ? sky_sunsky(&sky, ray.direction) : sky_sky(&sky, ray.direction));
			px[0] += color.x; px[1] += color.y; px[2] += color.z; px[3] += 1.f;
		}
	}
	s->last_survivors = s->primary_ray_cnt;
	s->last_shadow = s->shadow_ray_cnt;
	/* connect :328-346 */
	for (unsigned index = 0; index < s->shadow_ray_cnt; index++) {
		ShadowQueue ray = s->shadow[index];
		v3 yn = V3(0, 0, 0);
		float t = 0.f;
		orc_hit h;
		s->cnt.shadow_rays++;
		if (!intersect_voxel(w, ray.origin, ray.direction, &yn, &t, cb.campos, &h, &s->cnt, 0)) {
			float* px = accum + (size_t)ray.pixel_index * 4;
			px[0] += ray.color.x; px[1] += ray.color.y; px[2] += ray.color.z;
		}
	}
	s->frame++;
	/* main.cpp:144-146 */
	if (w->overlapped) orc_process_load_queue_overlapped(w);
	else orc_process_load_queue(w);
	RayQueue* tmp = s->work; s->work = s->next; s->next = tmp;
}

/* blit_onto_framebuffer (kernel.cu:348-364): rgb/a, alpha=1, pow(1/2.2) */
ORC_API void orc_resolve(const float* accum, float* out, int n_pixels) {
	for (int i = 0; i < n_pixels; i++) {
		float a = accum[i * 4 + 3];
		out[i * 4 + 0] = powf(accum[i * 4 + 0] / a, 1.f / 2.2f);
		out[i * 4 + 1] = powf(accum[i * 4 + 1] / a, 1.f / 2.2f);
		out[i * 4 + 2] = powf(accum[i * 4 + 2] / a, 1.f / 2.2f);
		out[i * 4 + 3] = powf(1.f, 1.f / 2.2f);
	}
}

ORC_API int orc_sizeof_counters(void) { return (int)sizeof(orc_counters); }
ORC_API int orc_sizeof_rayqueue(void) { return (int)sizeof(RayQueue); }
ORC_API int orc_sizeof_shadowqueue(void) { return (int)sizeof(ShadowQueue); }
