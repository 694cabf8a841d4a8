/*
 * mmf_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (nvblox_mindmap_amd/ + libmmfusion.so) never links, imports
 * or calls anything in oracle/.
 *
 * PARITY STATUS: **parity unpinned** for everything in this file.
 * The algorithm restated here lives in a third-party dependency that is ABSENT from
 * /root/reference: `nvblox` (github.com/nvidia-isaac/nvblox, branch `public`, commit not
 * recorded -- submodules/nvblox is an empty un-vendored submodule, .gitmodules:1-3; built
 * by docker/install_nvblox.sh:24-26 with NVBLOX_FEATURE_ARRAY_NUM_ELEMENTS=768).  All the
 * reference's golden data for this path are un-fetched git-LFS pointers
 * (mindmap/tests/baseline_data/, .gitattributes:1-4).  So this file restates nvblox's
 * PUBLISHED algorithm (projective TSDF fusion over an 8x8x8 voxel-block hash, raycast
 * block selection, sphere-traced occlusion test for appearance fusion, weight decay,
 * marching-cubes surface vertices) as a precise float32 spec, anchored on the reference's
 * call sites:
 *   add_depth_frame   mindmap/mapping/helpers/nvblox_mapping_helpers.py:207-209
 *   add_color_frame   mindmap/mapping/helpers/nvblox_mapping_helpers.py:212-218
 *   add_feature_frame mindmap/mapping/helpers/nvblox_mapping_helpers.py:255-261
 *   decay / clear     mindmap/mapping/isaaclab_nvblox_mapper.py:252-258
 *   update/get_feature_mesh  mindmap/mapping/helpers/nvblox_output_helpers.py:49-52
 *   parameters        mindmap/mapping/helpers/nvblox_mapping_helpers.py:40-70
 * Every arithmetic step is written with an explicit operation order; the HIP kernels use
 * the same order with FMA contraction disabled, so results are compared BIT-EXACT
 * (tests additionally state the north-star tolerance of 1e-5 abs).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, optional -fopenmp).
 *
 * ---------------------------------------------------------------------------------
 * NORMATIVE SPEC (all arithmetic IEEE-754 binary32 unless noted, no fused multiply-add)
 * ---------------------------------------------------------------------------------
 * v = voxel_size, bs = 8*v, inv_bs = 1/bs, inv_v = 1/v, trunc = truncation_vox * v.
 * Block index of point p:  b = floor(p * inv_bs)   (per axis).
 * Voxel (vx,vy,vz) of block b has centre  c = (float)b*bs + ((float)vi + 0.5f)*v.
 * Voxel linear id inside a block: lin = (vx*8 + vy)*8 + vz   (z fastest).
 * Pose T_L_C row-major 4x4 (camera -> layer/world); T_C_L is its rigid inverse
 *   Rinv = R^T,  tinv_i = -((Rinv_i0*t0 + Rinv_i1*t1) + Rinv_i2*t2).
 * Transform  q_i = ((R_i0*p0 + R_i1*p1) + R_i2*p2) + t_i.
 * Projection of p_C: reject z <= 1e-6; iz = 1/z; u = fx*(x*iz) + cx; v = fy*(y*iz) + cy;
 *   reject u < 0 | v < 0 | u > W | v > H.  Image-plane coordinates are corner referenced:
 *   pixel (col,row) covers [col,col+1) x [row,row+1), its centre is (col+.5,row+.5).
 * Bilinear sample at (u,v): uc = u-.5, vc = v-.5; x0 = floor(uc), y0 = floor(vc);
 *   need x0>=0, y0>=0, x0+1<=W-1, y0+1<=H-1; wx = uc-x0, wy = vc-y0;
 *   val = (1-wy)*((1-wx)*a00 + wx*a10) + wy*((1-wx)*a01 + wx*a11)   (aXY: x0+X, y0+Y).
 */
#include "../include/mmf_mc_table.h" /* generated marching-cubes table (tools/gen_mc_table.py), shared with the HIP kernels */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define VPS 8   /* voxels per side */
#define VPB 512 /* voxels per block */

/* ------------------------------------------------------------------------------- */
/* parameters                                                                      */
/* ------------------------------------------------------------------------------- */
typedef struct {
  float voxel_size;
  float max_integration_distance_m;    /* nvblox default 7.0; reference sets 5.0            */
  float truncation_distance_vox;       /* 4.0                                               */
  float max_weight;                    /* 5.0                                               */
  int weighting_mode;                  /* upstream's WeightingFunctionType family, see tsdf_weight(): 0 constant, 1 inverse-square,
                                          2 constant-dropoff, 3 inverse-square-dropoff, 4 inverse-square-TSDF-distance-penalty,
                                          5 linear-with-max                                                                  */
  float lin_interp_max_diff_vox;       /* 2.0: bilinear depth only if taps agree, else NN   */
  float appearance_measurement_weight; /* 1.0  (nvblox_mapper_constants.py:41)              */
  float appearance_max_weight;         /* 5.0                                               */
  int raycast_subsampling;             /* reference sets 1 (nvblox_mapping_helpers.py:52)   */
  int workspace_bounds_type;           /* 0 unbounded, 1 height (z) bounds, 2 bounding box  */
  float ws_min[3];
  float ws_max[3];
  float tsdf_decay_factor;             /* reference: 0.98 / 0.999                           */
  float decayed_weight_threshold;      /* 1e-3                                              */
  int deallocate_decayed_blocks;       /* 1                                                 */
  float mesh_min_weight;               /* 1e-4                                              */
  int st_subsampling;                  /* sphere-tracing ray subsampling, 4                 */
  int st_max_steps;                    /* 100                                               */
  float st_max_ray_length_m;           /* 15.0                                              */
  float st_surface_eps_vox;            /* 0.1                                               */
  int feature_channels;                /* C (compile-time 768 in the reference build)       */
  int raycast_to_truncation;           /* 1: blocks in view are marked up to depth + truncation; 0: up to the depth    */
  int decay_appearance_layers;         /* 0: decay() leaves colour / feature weights alone; 1: multiplies them too      */
  /* the two places where this spec was arranged for the GPU, kept switchable so that a pin against CUDA
   * nvblox can tell which form upstream has (tests/pin_report.py flips them):                                              */
  int raycast_walk_from_camera;        /* 0: the block walk of a ray starts where it enters the workspace bounds; 1: at the camera
                                          (the same block sets by construction: tests/test_cpu_raycast_walk.py)             */
  int appearance_blend_division;       /* 0: A' = (A W + a w) * (1 / (W + w)), one reciprocal per voxel; 1: (A W + a w) / (W + w)
                                          per channel (<= 1 ulp of the stored type apart)                                   */
  /* This file is compiled with -ffp-contract=off: a*b + c is two rounded operations.  nvcc contracts by default (-fmad=true), so
   * CUDA nvblox's voxels almost certainly hold fused multiply-adds; WHICH ones cannot be known here (its source is absent).
   * 1: the contraction a LLVM-family compiler makes of the expressions of this spec as written,
   *      a*b + c -> fma(a, b, c);   a*x + b*y -> fma(a, x, b*y);   (X + e*f) + t -> fma(e, f, X) + t
   * at the projection of a voxel centre (voxel_centre, xform, project), every bilinear sample (depth, synthetic depth in the
   * appearance gate, colour / feature taps), the TSDF update's numerator and the appearance blend's numerator -- MADD / MADD2
   * below.  The raycast walk, the sphere tracer and the mesh stay as they are (they decide block sets and gates, not values). */
  int fma_contraction;
  /* Three more places where this restatement may have mis-recollected upstream; each switchable in the oracle
   * and in the HIP library alike (frames of a mapper with any of them set take the stand-alone launches), so that a pin that
   * differs only there is attributable:
   * block_index_by_division      0: block of p = floor(p * (1 / block_size)), voxel = floor((p - b * block_size) * (1 / voxel_size));
   *                              1: floor(p / block_size), floor((p - b * block_size) / voxel_size) -- differs exactly for points within
   *                                 an ulp of a block / voxel face (the block SET can differ: workspace bounds, ray end points, walk
   *                                 coordinates, the sphere tracer's and the mesh's voxel look-ups all use it).
   * view_truncation_band_marking 0: a pixel's ray marks the blocks it traverses (camera -> depth + truncation);
   *                              1: additionally every block that intersects the axis-aligned cube of half-width `truncation` around
   *                                 the SURFACE point of the pixel (depth, clamped like the ray's) -- SURVEY.md App. A.2's "second kernel".
   * bilinear_four_weight_sum     0: nested lerps  (1-wy)((1-wx) a00 + wx a10) + wy((1-wx) a01 + wx a11);
   *                              1: four weighted taps, in upstream's order of terms
   *                                 (((1-wx)(1-wy)) a00 + ((1-wx) wy) a01) + (wx (1-wy)) a10 + (wx wy) a11   (left to right),
   *                                 contracted under fma_contraction as fma(w11, a11, fma(w10, a10, fma(w00, a00, w01 a01))).
   *                                 Applies to every bilinear sample of the integrator (depth, synthetic depth, colour, feature taps). */
  int block_index_by_division;
  int view_truncation_band_marking;
  int bilinear_four_weight_sum;
} orc_params;

void orc_default_params(orc_params* p) {
  memset(p, 0, sizeof(*p));
  p->voxel_size = 0.05f;
  p->max_integration_distance_m = 7.0f;
  p->truncation_distance_vox = 4.0f;
  p->max_weight = 5.0f;
  p->weighting_mode = 1;
  p->lin_interp_max_diff_vox = 2.0f;
  p->appearance_measurement_weight = 1.0f;
  p->appearance_max_weight = 5.0f;
  p->raycast_subsampling = 4;
  p->workspace_bounds_type = 0;
  p->tsdf_decay_factor = 0.95f;
  p->decayed_weight_threshold = 1e-3f;
  p->deallocate_decayed_blocks = 1;
  p->mesh_min_weight = 1e-4f;
  p->raycast_to_truncation = 1;
  p->decay_appearance_layers = 0;
  p->raycast_walk_from_camera = 0;
  p->appearance_blend_division = 0;
  p->fma_contraction = 0;
  p->block_index_by_division = 0;
  p->view_truncation_band_marking = 0;
  p->bilinear_four_weight_sum = 0;
  p->st_subsampling = 4;
  p->st_max_steps = 100;
  p->st_max_ray_length_m = 15.0f;
  p->st_surface_eps_vox = 0.1f;
  p->feature_channels = 768;
}

int orc_params_size(void) { return (int)sizeof(orc_params); }
static inline int bilin_mode(const orc_params* P) { return (P->fma_contraction ? 1 : 0) | (P->bilinear_four_weight_sum ? 2 : 0); }

/* ------------------------------------------------------------------------------- */
/* half <-> float (round-to-nearest-even, denormals kept)                          */
/* ------------------------------------------------------------------------------- */
static inline float h2f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal: normalise */
      int e = -1;
      do {
        man <<= 1;
        e++;
      } while (!(man & 0x400u));
      man &= 0x3ffu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static inline uint16_t f2h(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) { /* inf / nan */
    return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0u));
  }
  if (ax >= 0x477ff000u) { /* rounds to >= 65520 -> inf */
    return (uint16_t)(sign | 0x7c00u);
  }
  if (ax < 0x33000001u) { /* < 2^-25 (or == 2^-25 tie -> even 0) */
    return (uint16_t)sign;
  }
  int e = (int)(ax >> 23) - 127;
  uint32_t man = (ax & 0x7fffffu) | 0x800000u; /* 24-bit significand */
  int shift;
  uint32_t hexp;
  if (e < -14) { /* result is subnormal */
    shift = 13 + (-14 - e);
    hexp = 0;
  } else {
    shift = 13;
    hexp = (uint32_t)(e + 15);
  }
  uint32_t q = man >> shift;
  uint32_t rem = man & ((1u << shift) - 1u);
  uint32_t half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) q++;
  uint32_t out;
  if (hexp == 0) {
    out = q; /* may carry into exponent 1: correct by construction */
  } else {
    out = ((hexp - 1) << 10) + q; /* q has the implicit bit at 0x400 */
  }
  return (uint16_t)(sign | out);
}

/* exported for tests */
uint16_t orc_f2h(float f) { return f2h(f); }
float orc_h2f(uint16_t h) { return h2f(h); }

/* ------------------------------------------------------------------------------- */
/* small geometry helpers                                                          */
/* ------------------------------------------------------------------------------- */
typedef struct {
  float R[9];
  float t[3];
} rigid_t;

typedef struct {
  float fx, fy, cx, cy;
  int W, H;
} cam_t;

static inline int ifloor(float x) { return (int)floorf(x); }

/* the two contraction forms of orc_params.fma_contraction (fmaf: one rounding; the build flags forbid any other contraction) */
static inline float MADD(int fma, float a, float b, float c) { return fma ? fmaf(a, b, c) : a * b + c; }
static inline float MADD2(int fma, float a, float x, float b, float y) { return fma ? fmaf(a, x, b * y) : a * x + b * y; }

static void rigid_from_T(const float* T, rigid_t* o) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) o->R[i * 3 + j] = T[i * 4 + j];
    o->t[i] = T[i * 4 + 3];
  }
}

static void rigid_inverse(const rigid_t* a, rigid_t* o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o->R[i * 3 + j] = a->R[j * 3 + i];
  for (int i = 0; i < 3; ++i)
    o->t[i] = -((o->R[i * 3 + 0] * a->t[0] + o->R[i * 3 + 1] * a->t[1]) + o->R[i * 3 + 2] * a->t[2]);
}

static inline void xform(const rigid_t* T, const float* p, float* q) {
  for (int i = 0; i < 3; ++i)
    q[i] = ((T->R[i * 3 + 0] * p[0] + T->R[i * 3 + 1] * p[1]) + T->R[i * 3 + 2] * p[2]) + T->t[i];
}
/* the spec sites' form (voxel centre -> camera): (R0 p0 + R1 p1) + R2 p2, contracted when asked, then + t */
static inline void xform_c(const rigid_t* T, const float* p, float* q, int fma) {
  for (int i = 0; i < 3; ++i)
    q[i] = MADD(fma, T->R[i * 3 + 2], p[2], MADD2(fma, T->R[i * 3 + 0], p[0], T->R[i * 3 + 1], p[1])) + T->t[i];
}

static inline void rotate(const rigid_t* T, const float* p, float* q) {
  for (int i = 0; i < 3; ++i)
    q[i] = (T->R[i * 3 + 0] * p[0] + T->R[i * 3 + 1] * p[1]) + T->R[i * 3 + 2] * p[2];
}

static void cam_from_K(const float* K, int W, int H, cam_t* c) {
  c->fx = K[0];
  c->fy = K[4];
  c->cx = K[2];
  c->cy = K[5];
  c->W = W;
  c->H = H;
}

/* returns 1 if p_C projects into the image; u,v corner-referenced image-plane coords */
static inline int project_c(const cam_t* c, const float* p, float* u, float* v, int fma) {
  if (p[2] <= 1e-6f) return 0;
  float iz = 1.0f / p[2];
  float uu = MADD(fma, c->fx, p[0] * iz, c->cx);
  float vv = MADD(fma, c->fy, p[1] * iz, c->cy);
  if (uu < 0.0f || vv < 0.0f || uu > (float)c->W || vv > (float)c->H) return 0;
  *u = uu;
  *v = vv;
  return 1;
}

/* bilinear footprint; returns 0 if the 2x2 footprint leaves the image */
static inline int bilin_setup(float u, float v, int W, int H, int* x0, int* y0, float* wx, float* wy) {
  float uc = u - 0.5f, vc = v - 0.5f;
  float fx0 = floorf(uc), fy0 = floorf(vc);
  int ix = (int)fx0, iy = (int)fy0;
  if (ix < 0 || iy < 0 || ix + 1 > W - 1 || iy + 1 > H - 1) return 0;
  *x0 = ix;
  *y0 = iy;
  *wx = uc - fx0;
  *wy = vc - fy0;
  return 1;
}

/* fma: bit 0 = orc_params.fma_contraction, bit 1 = orc_params.bilinear_four_weight_sum (bilin_mode() below packs them) */
static inline float bilin_c(float a00, float a10, float a01, float a11, float wx, float wy, int mode) {
  const int fma = mode & 1;
  if (mode & 2) {
    float w00 = (1.0f - wx) * (1.0f - wy), w01 = (1.0f - wx) * wy, w10 = wx * (1.0f - wy), w11 = wx * wy;
    float t = MADD2(fma, w00, a00, w01, a01);
    t = MADD(fma, w10, a10, t);
    return MADD(fma, w11, a11, t);
  }
  float top = MADD2(fma, 1.0f - wx, a00, wx, a10);
  float bot = MADD2(fma, 1.0f - wx, a01, wx, a11);
  return MADD2(fma, 1.0f - wy, top, wy, bot);
}

/* ------------------------------------------------------------------------------- */
/* block layer: live list (allocation order) + open-addressing hash                */
/* ------------------------------------------------------------------------------- */
typedef struct {
  int bx, by, bz;
  void* data;
} block_t;

typedef struct {
  block_t* blocks; /* live list, allocation order; deallocation preserves order */
  int n, cap;
  int* table; /* hash: -1 empty, else position in blocks[] */
  int tcap;   /* power of two */
  size_t block_bytes;
} layer_t;

static inline uint32_t hash3(int x, int y, int z) {
  return ((uint32_t)x * 73856093u) ^ ((uint32_t)y * 19349663u) ^ ((uint32_t)z * 83492791u);
}

static void layer_init(layer_t* L, size_t block_bytes) {
  memset(L, 0, sizeof(*L));
  L->block_bytes = block_bytes;
  L->tcap = 1024;
  L->table = (int*)malloc(sizeof(int) * L->tcap);
  for (int i = 0; i < L->tcap; ++i) L->table[i] = -1;
}

static void layer_rehash(layer_t* L, int tcap) {
  free(L->table);
  L->tcap = tcap;
  L->table = (int*)malloc(sizeof(int) * tcap);
  for (int i = 0; i < tcap; ++i) L->table[i] = -1;
  for (int i = 0; i < L->n; ++i) {
    uint32_t h = hash3(L->blocks[i].bx, L->blocks[i].by, L->blocks[i].bz) & (uint32_t)(tcap - 1);
    while (L->table[h] != -1) h = (h + 1) & (uint32_t)(tcap - 1);
    L->table[h] = i;
  }
}

static int layer_find(const layer_t* L, int x, int y, int z) {
  uint32_t h = hash3(x, y, z) & (uint32_t)(L->tcap - 1);
  while (L->table[h] != -1) {
    const block_t* b = &L->blocks[L->table[h]];
    if (b->bx == x && b->by == y && b->bz == z) return L->table[h];
    h = (h + 1) & (uint32_t)(L->tcap - 1);
  }
  return -1;
}

/* returns position; *is_new set when freshly allocated (zero-filled) */
static int layer_alloc(layer_t* L, int x, int y, int z, int* is_new) {
  int pos = layer_find(L, x, y, z);
  if (pos >= 0) {
    if (is_new) *is_new = 0;
    return pos;
  }
  if (L->n == L->cap) {
    L->cap = L->cap ? L->cap * 2 : 256;
    L->blocks = (block_t*)realloc(L->blocks, sizeof(block_t) * L->cap);
  }
  pos = L->n++;
  L->blocks[pos].bx = x;
  L->blocks[pos].by = y;
  L->blocks[pos].bz = z;
  L->blocks[pos].data = calloc(1, L->block_bytes);
  if (L->n * 2 > L->tcap) {
    layer_rehash(L, L->tcap * 2);
  } else {
    uint32_t h = hash3(x, y, z) & (uint32_t)(L->tcap - 1);
    while (L->table[h] != -1) h = (h + 1) & (uint32_t)(L->tcap - 1);
    L->table[h] = pos;
  }
  if (is_new) *is_new = 1;
  return pos;
}

static void layer_clear(layer_t* L) {
  for (int i = 0; i < L->n; ++i) free(L->blocks[i].data);
  L->n = 0;
  for (int i = 0; i < L->tcap; ++i) L->table[i] = -1;
}

static void layer_free(layer_t* L) {
  layer_clear(L);
  free(L->blocks);
  free(L->table);
  memset(L, 0, sizeof(*L));
}

/* remove blocks with kill[i] != 0, preserving the order of the survivors */
static void layer_remove(layer_t* L, const uint8_t* kill) {
  int w = 0;
  for (int i = 0; i < L->n; ++i) {
    if (kill[i]) {
      free(L->blocks[i].data);
    } else {
      L->blocks[w++] = L->blocks[i];
    }
  }
  L->n = w;
  layer_rehash(L, L->tcap);
}

/* voxel payloads */
typedef struct {
  float d[VPB];
  float w[VPB];
} tsdf_block;
typedef struct {
  uint8_t rgb[VPB * 3];
  float w[VPB];
} color_block;
/* feature block: uint16 feat[VPB*C] followed by float w[VPB] */
static inline uint16_t* feat_ptr(void* data) { return (uint16_t*)data; }
static inline float* featw_ptr(void* data, int C) { return (float*)((uint8_t*)data + (size_t)VPB * C * 2); }

/* ------------------------------------------------------------------------------- */
/* the mapper                                                                      */
/* ------------------------------------------------------------------------------- */
typedef struct {
  orc_params P;
  float v, bs, inv_bs, inv_v, trunc;
  layer_t tsdf, color, feat;
  /* last candidate lists (for inspection) */
  int* cand;
  int ncand, cand_cap; /* packed triples */
  /* synthetic depth (last rendered) */
  float* synth;
  int synth_W, synth_H;
  /* mesh */
  float* mesh_v;
  uint16_t* mesh_f;
  int mesh_n, mesh_cap;
  int mesh_C;
  uint8_t* mesh_c; /* [mesh_n][3] vertex colours */
  int* mesh_t;     /* [mesh_nt][3] triangles (vertex indices) */
  int mesh_nt, mesh_tcap;
  /* counters of the last integrate call (blocks updated) */
  int last_n_tsdf_blocks, last_n_app_blocks;
} orc_mapper;

orc_mapper* orc_create(const orc_params* p) {
  orc_mapper* m = (orc_mapper*)calloc(1, sizeof(orc_mapper));
  m->P = *p;
  m->v = p->voxel_size;
  m->bs = 8.0f * m->v;
  m->inv_bs = 1.0f / m->bs;
  m->inv_v = 1.0f / m->v;
  m->trunc = p->truncation_distance_vox * m->v;
  layer_init(&m->tsdf, sizeof(tsdf_block));
  layer_init(&m->color, sizeof(color_block));
  layer_init(&m->feat, (size_t)VPB * p->feature_channels * 2 + sizeof(float) * VPB);
  return m;
}

void orc_destroy(orc_mapper* m) {
  if (!m) return;
  layer_free(&m->tsdf);
  layer_free(&m->color);
  layer_free(&m->feat);
  free(m->cand);
  free(m->synth);
  free(m->mesh_v);
  free(m->mesh_f);
  free(m->mesh_c);
  free(m->mesh_t);
  free(m);
}

void orc_clear(orc_mapper* m) {
  layer_clear(&m->tsdf);
  layer_clear(&m->color);
  layer_clear(&m->feat);
  m->mesh_n = 0;
}

/* voxel centre c = b * block_size + (v + 0.5) * voxel_size -> camera frame -> image plane, under orc_params.fma_contraction */
static inline int project_voxel(const orc_mapper* m, const cam_t* cam, const rigid_t* T_C_L, int bx, int by, int bz, int lin, float* p,
                                float* u, float* v) {
  const int fma = m->P.fma_contraction;
  float c[3];
  c[0] = MADD2(fma, (float)bx, m->bs, (float)(lin >> 6) + 0.5f, m->v);
  c[1] = MADD2(fma, (float)by, m->bs, (float)((lin >> 3) & 7) + 0.5f, m->v);
  c[2] = MADD2(fma, (float)bz, m->bs, (float)(lin & 7) + 0.5f, m->v);
  xform_c(T_C_L, c, p, fma);
  return project_c(cam, p, u, v, fma);
}

/* position -> block units / voxel units under orc_params.block_index_by_division */
static inline float to_blocks(const orc_mapper* m, float x) { return m->P.block_index_by_division ? x / m->bs : x * m->inv_bs; }
static inline float to_voxels(const orc_mapper* m, float x) { return m->P.block_index_by_division ? x / m->v : x * m->inv_v; }

/* workspace test on a block index (inclusive index range of the bounds' own blocks) */
static inline int in_workspace(const orc_mapper* m, int x, int y, int z) {
  const orc_params* P = &m->P;
  if (P->workspace_bounds_type == 0) return 1;
  if (z < ifloor(to_blocks(m, P->ws_min[2])) || z > ifloor(to_blocks(m, P->ws_max[2]))) return 0;
  if (P->workspace_bounds_type == 1) return 1;
  if (x < ifloor(to_blocks(m, P->ws_min[0])) || x > ifloor(to_blocks(m, P->ws_max[0]))) return 0;
  if (y < ifloor(to_blocks(m, P->ws_min[1])) || y > ifloor(to_blocks(m, P->ws_max[1]))) return 0;
  return 1;
}

/* ------------------------------------------------------------------------------- */
/* A.2  blocks in view by per-pixel raycast (nvblox ViewCalculator, raycast variant) */
/* ------------------------------------------------------------------------------- */
/* temporary set of int3 */
typedef struct {
  int* keys; /* triples */
  uint8_t* used;
  int cap, n;
} set3;

static void set3_init(set3* s, int cap) {
  s->cap = cap;
  s->n = 0;
  s->keys = (int*)malloc(sizeof(int) * 3 * cap);
  s->used = (uint8_t*)calloc(cap, 1);
}
static void set3_grow(set3* s);
static inline void set3_insert(set3* s, int x, int y, int z) {
  uint32_t h = hash3(x, y, z) & (uint32_t)(s->cap - 1);
  while (s->used[h]) {
    int* k = &s->keys[3 * h];
    if (k[0] == x && k[1] == y && k[2] == z) return;
    h = (h + 1) & (uint32_t)(s->cap - 1);
  }
  s->used[h] = 1;
  s->keys[3 * h] = x;
  s->keys[3 * h + 1] = y;
  s->keys[3 * h + 2] = z;
  s->n++;
  if (s->n * 2 > s->cap) set3_grow(s);
}
static void set3_grow(set3* s) {
  set3 o = *s;
  set3_init(s, o.cap * 2);
  for (int i = 0; i < o.cap; ++i)
    if (o.used[i]) set3_insert(s, o.keys[3 * i], o.keys[3 * i + 1], o.keys[3 * i + 2]);
  free(o.keys);
  free(o.used);
}

static int cmp3(const void* a, const void* b) {
  const int* p = (const int*)a;
  const int* q = (const int*)b;
  for (int i = 0; i < 3; ++i) {
    if (p[i] < q[i]) return -1;
    if (p[i] > q[i]) return 1;
  }
  return 0;
}

/*
 * Grid walk (Amanatides-Woo, block-scaled coordinates) from s to e, visiting the cell of s,
 * the cell of e and every cell between:  n = L1 distance between the two cells, n+1 visits.
 * Per axis: r = e-s; step = sign(r); if step != 0: tmax = ((step>0 ? 1 : 0) - (s - floor(s))) / r,
 * dt = step / r, else tmax = dt = 2.  At each move pick, among the axes whose cell index has
 * not yet reached the goal index, the one with the smallest tmax (ties: x before y before z).
 */
typedef struct {
  int c[3], g[3], st[3], n;
  float tm[3], dt[3];
} walk_t;

static inline void walk_init(walk_t* w, const float* s, const float* e) {
  w->n = 0;
  for (int a = 0; a < 3; ++a) {
    float fs = floorf(s[a]);
    w->c[a] = (int)fs;
    w->g[a] = ifloor(e[a]);
    int diff = w->g[a] - w->c[a];
    w->n += diff < 0 ? -diff : diff;
    float r = e[a] - s[a];
    w->st[a] = r > 0.0f ? 1 : (r < 0.0f ? -1 : 0);
    if (w->st[a] != 0) {
      float corr = w->st[a] > 0 ? 1.0f : 0.0f;
      float dist = corr - (s[a] - fs);
      w->tm[a] = dist / r;
      w->dt[a] = (float)w->st[a] / r;
    } else {
      w->tm[a] = 2.0f;
      w->dt[a] = 2.0f;
    }
  }
}

static inline void walk_step(walk_t* w) {
  int best = -1;
  float bt = 0.0f;
  for (int a = 0; a < 3; ++a) {
    if (w->c[a] == w->g[a]) continue;
    if (best < 0 || w->tm[a] < bt) {
      best = a;
      bt = w->tm[a];
    }
  }
  if (best >= 0) {
    w->c[best] += w->st[best];
    w->tm[best] += w->dt[best];
  }
}

/*
 * Where the grid walk of a ray starts.  Only blocks inside the workspace bounds can be in view, so the part of the segment
 * camera -> end point that lies before the bounds is not walked: per bounded axis the slab parameter of the face the ray enters
 * through, t0 = max over the axes (0 if the camera is inside), stepped back by two cells along the ray's dominant axis so that
 * the walk enters the bounds with the cell sequence of the full walk (the traversed cells inside the bounds are the same
 * either way; what is saved are the steps outside -- a quarter of them for a camera that orbits the box).
 *   kUnbounded: start at the camera.  kHeightBounds: z only.  kBoundingBox: all three axes.
 *   lo_a = (float)ws_lo_a, hi_a = (float)(ws_hi_a + 1)  [block units];  r = e - s0;
 *   t_a = r_a > 0 ? (lo_a - s0_a) / r_a : (r_a < 0 ? (hi_a - s0_a) / r_a : 0);  t0 = max(0, t_a ...);
 *   if t0 > 0: t0 = t0 - 2 / max_a |r_a|;  t0 = clamp(t0, 0, 1);  start = s0 + t0 * r
 */
static void clip_walk_start(const orc_mapper* m, const float* s0, const float* e, float* out) {
  out[0] = s0[0];
  out[1] = s0[1];
  out[2] = s0[2];
  const int type = m->P.workspace_bounds_type;
  if (type == 0 || m->P.raycast_walk_from_camera) return;
  float r[3], t0 = 0.0f, big = 0.0f;
  for (int a = 0; a < 3; ++a) {
    r[a] = e[a] - s0[a];
    float ar = fabsf(r[a]);
    if (ar > big) big = ar;
    if (type == 1 && a < 2) continue;
    float lo = (float)ifloor(to_blocks(m, m->P.ws_min[a])), hi = (float)(ifloor(to_blocks(m, m->P.ws_max[a])) + 1), ta = 0.0f;
    if (r[a] > 0.0f) ta = (lo - s0[a]) / r[a];
    else if (r[a] < 0.0f) ta = (hi - s0[a]) / r[a];
    if (ta > t0) t0 = ta;
  }
  if (!(t0 > 0.0f) || !(big > 0.0f)) return;
  t0 = t0 - 2.0f / big;
  if (!(t0 > 0.0f)) return;
  if (t0 > 1.0f) t0 = 1.0f;
  for (int a = 0; a < 3; ++a) out[a] = s0[a] + t0 * r[a];
}

/*
 * For every raycast_subsampling-th pixel with depth > 0 and mask != 0 (NaN fails "depth > 0"; +inf -- a simulated camera's
 * "no return" -- passes it and is clamped like any far depth; with no maximum distance set, +inf casts no ray):
 *   d = min(depth, max_integration_distance) (if max > 0);  s = d + trunc
 *   ray_C = ((col + .5 - cx)/fx, (row + .5 - cy)/fy, 1);  p_C = s * ray_C;  p_L = T_L_C p_C
 *   walk from clip_walk_start(t_L_C*inv_bs, p_L*inv_bs) to p_L*inv_bs; every visited block inside the workspace is in view.
 * The result is sorted lexicographically (x, then y, then z).
 */
static void blocks_in_view(orc_mapper* m, const float* depth, const uint8_t* mask, const cam_t* cam,
                           const rigid_t* T_L_C) {
  const orc_params* P = &m->P;
  set3 S;
  set3_init(&S, 4096);
  int sub = P->raycast_subsampling < 1 ? 1 : P->raycast_subsampling;
  float s0[3];
  for (int a = 0; a < 3; ++a) s0[a] = to_blocks(m, T_L_C->t[a]);
  /* rows are independent: every thread collects into its own set, the sets are merged afterwards (the result is a
   * sorted set, so neither the thread count nor the merge order can change it) */
#pragma omp parallel
  {
    set3 Sl;
    set3_init(&Sl, 1024);
#pragma omp for schedule(dynamic, 8) nowait
    for (int r = 0; r < cam->H; r += sub) {
      for (int c = 0; c < cam->W; c += sub) {
        float d = depth[(size_t)r * cam->W + c];
        if (!(d > 0.0f)) continue;
        if (mask && !mask[(size_t)r * cam->W + c]) continue;
        if (P->max_integration_distance_m > 0.0f && d > P->max_integration_distance_m)
          d = P->max_integration_distance_m;
        if (!(d <= 3.4028235e38f)) continue; /* +inf with no maximum distance to clamp it to: the pixel casts no ray */
        float s = m->P.raycast_to_truncation ? d + m->trunc : d;
        float ray[3] = {((float)c + 0.5f - cam->cx) / cam->fx, ((float)r + 0.5f - cam->cy) / cam->fy, 1.0f};
        float pC[3] = {s * ray[0], s * ray[1], s * ray[2]};
        float pL[3];
        xform(T_L_C, pC, pL);
        float e[3] = {to_blocks(m, pL[0]), to_blocks(m, pL[1]), to_blocks(m, pL[2])};
        float sc[3];
        clip_walk_start(m, s0, e, sc);
        walk_t w;
        walk_init(&w, sc, e);
        for (int i = 0; i <= w.n; ++i) {
          if (in_workspace(m, w.c[0], w.c[1], w.c[2])) set3_insert(&Sl, w.c[0], w.c[1], w.c[2]);
          walk_step(&w);
        }
        if (P->view_truncation_band_marking) {
          /* the blocks that intersect the cube [p - trunc, p + trunc]^3 around the pixel's surface point p = T_L_C (d * ray) */
          float qC[3] = {d * ray[0], d * ray[1], d * ray[2]}, qL[3];
          int lo[3], hi[3];
          xform(T_L_C, qC, qL);
          for (int a = 0; a < 3; ++a) {
            lo[a] = ifloor(to_blocks(m, qL[a] - m->trunc));
            hi[a] = ifloor(to_blocks(m, qL[a] + m->trunc));
          }
          for (int x = lo[0]; x <= hi[0]; ++x)
            for (int y = lo[1]; y <= hi[1]; ++y)
              for (int z = lo[2]; z <= hi[2]; ++z)
                if (in_workspace(m, x, y, z)) set3_insert(&Sl, x, y, z);
        }
      }
    }
#pragma omp critical(orc_view_merge)
    {
      for (int i = 0; i < Sl.cap; ++i)
        if (Sl.used[i]) set3_insert(&S, Sl.keys[3 * i], Sl.keys[3 * i + 1], Sl.keys[3 * i + 2]);
    }
    free(Sl.keys);
    free(Sl.used);
  }
  if (m->cand_cap < S.n) {
    m->cand_cap = S.n;
    m->cand = (int*)realloc(m->cand, sizeof(int) * 3 * (S.n > 0 ? S.n : 1));
  }
  int k = 0;
  for (int i = 0; i < S.cap; ++i)
    if (S.used[i]) {
      memcpy(&m->cand[3 * k], &S.keys[3 * i], sizeof(int) * 3);
      k++;
    }
  m->ncand = k;
  qsort(m->cand, k, sizeof(int) * 3, cmp3);
  free(S.keys);
  free(S.used);
}

/* ------------------------------------------------------------------------------- */
/* A.3  projective TSDF update                                                      */
/* ------------------------------------------------------------------------------- */
static inline int depth_tap(const float* depth, const uint8_t* mask, int W, int x, int y, float* out) {
  size_t i = (size_t)y * W + x;
  float d = depth[i];
  if (!(d > 0.0f)) return 0;
  if (mask && !mask[i]) return 0;
  *out = d;
  return 1;
}

/*
 * Measured surface depth at image-plane point (u,v):
 *   nearest tap = pixel (floor(u), floor(v)) (must be inside the image and valid);
 *   bilinear value is used iff the 2x2 footprint is inside the image, all four taps are valid
 *   and each differs from the nearest tap by at most lin_interp_max_diff (skipped if <= 0);
 *   otherwise the nearest tap is used; no valid nearest tap -> no measurement.
 */
static inline int sample_depth(const orc_mapper* m, const float* depth, const uint8_t* mask, const cam_t* cam,
                               float u, float v, float* out) {
  int xn = ifloor(u), yn = ifloor(v);
  if (xn > cam->W - 1) xn = cam->W - 1; /* u == W allowed by project() */
  if (yn > cam->H - 1) yn = cam->H - 1;
  float dn;
  if (!depth_tap(depth, mask, cam->W, xn, yn, &dn)) return 0;
  int x0, y0;
  float wx, wy;
  if (bilin_setup(u, v, cam->W, cam->H, &x0, &y0, &wx, &wy)) {
    float a00, a10, a01, a11;
    if (depth_tap(depth, mask, cam->W, x0, y0, &a00) && depth_tap(depth, mask, cam->W, x0 + 1, y0, &a10) &&
        depth_tap(depth, mask, cam->W, x0, y0 + 1, &a01) && depth_tap(depth, mask, cam->W, x0 + 1, y0 + 1, &a11)) {
      float md = m->P.lin_interp_max_diff_vox * m->v;
      int ok = 1;
      if (md > 0.0f) {
        if (fabsf(a00 - dn) > md || fabsf(a10 - dn) > md || fabsf(a01 - dn) > md || fabsf(a11 - dn) > md) ok = 0;
      }
      if (ok) {
        *out = bilin_c(a00, a10, a01, a11, wx, wy, bilin_mode(&m->P));
        return 1;
      }
    }
  }
  *out = dn;
  return 1;
}

/*
 * Measurement weight of a TSDF update: upstream nvblox's WeightingFunctionType family (RECALLED; the reference leaves
 * projective_integrator_weighting_mode at upstream's default, nvblox_mapping_helpers.py:40-46).  d = sampled depth of the
 * pixel, sdf = d - voxel depth, trunc = truncation distance.
 *   0 kConstantWeight                     1
 *   1 kInverseSquareWeight                1 / d^2                                         (this spec's default)
 *   2 kConstantDropoffWeight              dropoff(sdf)
 *   3 kInverseSquareDropoffWeight         (1 / d^2) * dropoff(sdf)
 *   4 kInverseSquareTsdfDistancePenalty   (1 / d^2) * penalty(sdf)
 *   5 kLinearWithMax                      min(1 / d, 1)
 *   dropoff(sdf) = 1 in front of the surface, falling linearly to 0 at -trunc behind it: sdf >= 0 ? 1 : max((trunc + sdf) / trunc, 0)
 *   penalty(sdf) = 1 - 0.5 * min(|sdf|, trunc) / trunc        (voxels far from the measured surface count half)
 * The shapes of dropoff / penalty / linear-with-max are low-confidence recollections: they exist so that the pin kit
 * (tests/pin_report.py) can tell which member upstream's default is, not as a claim about its constants.
 * A measurement of weight <= 0 is skipped (the blend would divide by W alone).
 */
static inline float tsdf_weight(const orc_mapper* m, float d, float sdf) {
  const float trunc = m->trunc;
  switch (m->P.weighting_mode) {
    case 0: return 1.0f;
    case 1: return 1.0f / (d * d);
    case 2: return sdf >= 0.0f ? 1.0f : fmaxf((trunc + sdf) / trunc, 0.0f);
    case 3: return (1.0f / (d * d)) * (sdf >= 0.0f ? 1.0f : fmaxf((trunc + sdf) / trunc, 0.0f));
    case 4: return (1.0f / (d * d)) * (1.0f - 0.5f * (fminf(fabsf(sdf), trunc) / trunc));
    default: return fminf(1.0f / d, 1.0f);
  }
}

/*
 * Per voxel of every block in view:
 *   p = T_C_L c; project (reject); reject p.z > max_integration_distance (if > 0);
 *   d = sample_depth(u,v) (reject); sdf = d - p.z; reject sdf < -trunc;
 *   wm = weight(d);  D' = (sdf*wm + D*W) / (wm + W);  D' = clamp(D', -trunc, trunc);
 *   W' = min(W + wm, max_weight).
 */
static void tsdf_integrate(orc_mapper* m, const float* depth, const uint8_t* mask, const cam_t* cam,
                           const rigid_t* T_C_L, const int* pos, int n) {
  const orc_params* P = &m->P;
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < n; ++i) {
    block_t* B = &m->tsdf.blocks[pos[i]];
    tsdf_block* tb = (tsdf_block*)B->data;
    for (int lin = 0; lin < VPB; ++lin) {
      float p[3], u, v;
      if (!project_voxel(m, cam, T_C_L, B->bx, B->by, B->bz, lin, p, &u, &v)) continue;
      if (P->max_integration_distance_m > 0.0f && p[2] > P->max_integration_distance_m) continue;
      float d;
      if (!sample_depth(m, depth, mask, cam, u, v, &d)) continue;
      float sdf = d - p[2];
      if (sdf < -m->trunc) continue;
      float wm = tsdf_weight(m, d, sdf);
      if (!(wm > 0.0f)) continue;
      float D = tb->d[lin], W = tb->w[lin];
      float Dn = MADD2(P->fma_contraction, sdf, wm, D, W) / (wm + W);
      Dn = Dn > 0.0f ? fminf(m->trunc, Dn) : fmaxf(-m->trunc, Dn);
      tb->d[lin] = Dn;
      tb->w[lin] = fminf(W + wm, P->max_weight);
    }
  }
}

int orc_add_depth_frame(orc_mapper* m, const float* depth, const uint8_t* mask, int H, int W, const float* T_L_C16,
                        const float* K9) {
  cam_t cam;
  cam_from_K(K9, W, H, &cam);
  rigid_t T_L_C, T_C_L;
  rigid_from_T(T_L_C16, &T_L_C);
  rigid_inverse(&T_L_C, &T_C_L);
  blocks_in_view(m, depth, mask, &cam, &T_L_C);
  int* pos = (int*)malloc(sizeof(int) * (m->ncand > 0 ? m->ncand : 1));
  for (int i = 0; i < m->ncand; ++i)
    pos[i] = layer_alloc(&m->tsdf, m->cand[3 * i], m->cand[3 * i + 1], m->cand[3 * i + 2], NULL);
  tsdf_integrate(m, depth, mask, &cam, &T_C_L, pos, m->ncand);
  m->last_n_tsdf_blocks = m->ncand;
  free(pos);
  return 0;
}

/* ------------------------------------------------------------------------------- */
/* A.5  sphere tracing of the TSDF (synthetic depth for the occlusion test)         */
/* ------------------------------------------------------------------------------- */
/* voxel containing p: block floor(p*inv_bs), voxel clamp(floor((p - b*bs)*inv_v), 0, 7) */
static inline int voxel_at(const orc_mapper* m, const layer_t* L, const float* p, int* lin) {
  int b[3], vi[3];
  for (int a = 0; a < 3; ++a) {
    b[a] = ifloor(to_blocks(m, p[a]));
    int q = ifloor(to_voxels(m, p[a] - (float)b[a] * m->bs));
    vi[a] = q < 0 ? 0 : (q > 7 ? 7 : q);
  }
  int pos = layer_find(L, b[0], b[1], b[2]);
  if (pos < 0) return -1;
  *lin = (vi[0] * 8 + vi[1]) * 8 + vi[2];
  return pos;
}

/*
 * Ray (origin o, unit direction dir) marched with t = 0:
 *   for step < max_steps and t < max_ray_length:
 *     p = o + t*dir;  voxel = voxel containing p (valid iff its block exists and W > 1e-4)
 *     invalid: if the previous sample was a valid positive distance -> FAIL, else t += trunc
 *     valid, D < eps: if previous sample was valid positive -> t += D, SUCCESS; else FAIL
 *     valid, D >= eps: t += D, remember "previous positive"
 *   FAIL otherwise.
 */
static int sphere_cast(const orc_mapper* m, const float* o, const float* dir, float* t_out) {
  const orc_params* P = &m->P;
  float eps = P->st_surface_eps_vox * m->v;
  int last_pos = 0;
  float t = 0.0f;
  for (int i = 0; i < P->st_max_steps && t < P->st_max_ray_length_m; ++i) {
    float p[3] = {o[0] + t * dir[0], o[1] + t * dir[1], o[2] + t * dir[2]};
    int lin;
    int pos = voxel_at(m, &m->tsdf, p, &lin);
    int valid = 0;
    float D = 0.0f;
    if (pos >= 0) {
      const tsdf_block* tb = (const tsdf_block*)m->tsdf.blocks[pos].data;
      if (tb->w[lin] > 1e-4f) {
        valid = 1;
        D = tb->d[lin];
      }
    }
    float step;
    if (!valid) {
      if (last_pos) return 0;
      step = m->trunc;
    } else if (D < eps) {
      if (last_pos) {
        *t_out = t + D;
        return 1;
      }
      return 0;
    } else {
      step = D;
      last_pos = 1;
    }
    t += step;
  }
  return 0;
}

/*
 * Synthetic depth image of size (W/sf) x (H/sf): ray of sub-pixel (cs, rs) passes through
 * image-plane point ((cs+.5)*sf, (rs+.5)*sf); dir_C = normalise(((u-cx)/fx, (v-cy)/fy, 1))
 * (n = sqrt((x*x + y*y) + 1), dir = (x/n, y/n, 1/n)); depth = t * dir_C.z, or -1 on failure.
 */
static void render_synthetic_depth(orc_mapper* m, const cam_t* cam, const rigid_t* T_L_C) {
  int sf = m->P.st_subsampling < 1 ? 1 : m->P.st_subsampling;
  int Ws = cam->W / sf, Hs = cam->H / sf;
  if (Ws * Hs > m->synth_W * m->synth_H || !m->synth) {
    free(m->synth);
    m->synth = (float*)malloc(sizeof(float) * (size_t)(Ws * Hs > 0 ? Ws * Hs : 1));
  }
  m->synth_W = Ws;
  m->synth_H = Hs;
#pragma omp parallel for schedule(dynamic, 4)
  for (int rs = 0; rs < Hs; ++rs) {
    for (int cs = 0; cs < Ws; ++cs) {
      float u = ((float)cs + 0.5f) * (float)sf, v = ((float)rs + 0.5f) * (float)sf;
      float x = (u - cam->cx) / cam->fx, y = (v - cam->cy) / cam->fy;
      float n = sqrtf((x * x + y * y) + 1.0f);
      float dC[3] = {x / n, y / n, 1.0f / n};
      float dL[3];
      rotate(T_L_C, dC, dL);
      float t;
      if (sphere_cast(m, T_L_C->t, dL, &t))
        m->synth[(size_t)rs * Ws + cs] = t * dC[2];
      else
        m->synth[(size_t)rs * Ws + cs] = -1.0f;
    }
  }
}

/* ------------------------------------------------------------------------------- */
/* A.5  appearance (colour / feature) integration                                   */
/* ------------------------------------------------------------------------------- */
/*
 * Candidate blocks: every live TSDF block (live order) that has a voxel with W > 0,
 * |D| < trunc whose centre projects into the appearance image with p.z <= max distance.
 */
static int app_candidates(orc_mapper* m, const cam_t* cam, const rigid_t* T_C_L, int** out) {
  const orc_params* P = &m->P;
  int* list = (int*)malloc(sizeof(int) * (m->tsdf.n > 0 ? m->tsdf.n : 1));
  int k = 0;
  for (int i = 0; i < m->tsdf.n; ++i) {
    const block_t* B = &m->tsdf.blocks[i];
    const tsdf_block* tb = (const tsdf_block*)B->data;
    int hit = 0;
    for (int lin = 0; lin < VPB && !hit; ++lin) {
      if (!(tb->w[lin] > 0.0f) || !(fabsf(tb->d[lin]) < m->trunc)) continue;
      float p[3], u, v;
      if (!project_voxel(m, cam, T_C_L, B->bx, B->by, B->bz, lin, p, &u, &v)) continue;
      if (P->max_integration_distance_m > 0.0f && p[2] > P->max_integration_distance_m) continue;
      hit = 1;
    }
    if (hit) list[k++] = i;
  }
  *out = list;
  return k;
}

/*
 * Shared per-voxel gate of both appearance integrators:
 *   p = T_C_L c; project with the appearance camera; reject p.z > max distance;
 *   synthetic depth s = bilinear sample of the sphere-traced image at (u/sf, v/sf), all four
 *   taps must be > 0;  reject |s - p.z| > trunc;
 *   bilinear footprint (x0,y0,wx,wy) of (u,v) in the appearance image must be inside the image
 *   and all four mask taps non-zero.
 */
static inline int app_gate(const orc_mapper* m, const cam_t* cam, const rigid_t* T_C_L, const uint8_t* mask,
                           const block_t* B, int lin, int* x0, int* y0, float* wx, float* wy) {
  const orc_params* P = &m->P;
  float p[3], u, v;
  if (!project_voxel(m, cam, T_C_L, B->bx, B->by, B->bz, lin, p, &u, &v)) return 0;
  if (P->max_integration_distance_m > 0.0f && p[2] > P->max_integration_distance_m) return 0;
  float sf = (float)(P->st_subsampling < 1 ? 1 : P->st_subsampling);
  int sx, sy;
  float swx, swy;
  if (!bilin_setup(u / sf, v / sf, m->synth_W, m->synth_H, &sx, &sy, &swx, &swy)) return 0;
  const float* S = m->synth;
  float s00 = S[(size_t)sy * m->synth_W + sx], s10 = S[(size_t)sy * m->synth_W + sx + 1];
  float s01 = S[(size_t)(sy + 1) * m->synth_W + sx], s11 = S[(size_t)(sy + 1) * m->synth_W + sx + 1];
  if (!(s00 > 0.0f) || !(s10 > 0.0f) || !(s01 > 0.0f) || !(s11 > 0.0f)) return 0;
  float s = bilin_c(s00, s10, s01, s11, swx, swy, bilin_mode(P));
  if (fabsf(s - p[2]) > m->trunc) return 0;
  if (!bilin_setup(u, v, cam->W, cam->H, x0, y0, wx, wy)) return 0;
  if (mask) {
    size_t i = (size_t)(*y0) * cam->W + *x0;
    if (!mask[i] || !mask[i + 1] || !mask[i + cam->W] || !mask[i + cam->W + 1]) return 0;
  }
  return 1;
}

/*
 * Feature update of a gated voxel, per channel k (float32 arithmetic, float16 storage):
 *   a = bilinear(f16->f32 taps);  inv = 1/(W + wm);  A' = f32->f16_rne( (A*W + a*wm) * inv );
 *   W' = min(W + wm, appearance_max_weight);   wm = appearance_measurement_weight.
 */
int orc_add_feature_frame(orc_mapper* m, const uint16_t* feat, const uint8_t* mask, int H, int W, int C,
                          const float* T_L_C16, const float* K9) {
  if (C != m->P.feature_channels) return -1;
  cam_t cam;
  cam_from_K(K9, W, H, &cam);
  rigid_t T_L_C, T_C_L;
  rigid_from_T(T_L_C16, &T_L_C);
  rigid_inverse(&T_L_C, &T_C_L);
  int* cl;
  int nc = app_candidates(m, &cam, &T_C_L, &cl);
  render_synthetic_depth(m, &cam, &T_L_C);
  int* pos = (int*)malloc(sizeof(int) * (nc > 0 ? nc : 1));
  for (int i = 0; i < nc; ++i) {
    const block_t* tb = &m->tsdf.blocks[cl[i]];
    pos[i] = layer_alloc(&m->feat, tb->bx, tb->by, tb->bz, NULL);
  }
  float wm = m->P.appearance_measurement_weight;
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < nc; ++i) {
    block_t* B = &m->feat.blocks[pos[i]];
    uint16_t* A = feat_ptr(B->data);
    float* Wt = featw_ptr(B->data, C);
    for (int lin = 0; lin < VPB; ++lin) {
      int x0, y0;
      float wx, wy;
      if (!app_gate(m, &cam, &T_C_L, mask, B, lin, &x0, &y0, &wx, &wy)) continue;
      const uint16_t* t00 = feat + ((size_t)y0 * W + x0) * C;
      const uint16_t* t10 = t00 + C;
      const uint16_t* t01 = t00 + (size_t)W * C;
      const uint16_t* t11 = t01 + C;
      float Wv = Wt[lin];
      float inv = 1.0f / (Wv + wm);
      uint16_t* Av = A + (size_t)lin * C;
      for (int k = 0; k < C; ++k) {
        float a = bilin_c(h2f(t00[k]), h2f(t10[k]), h2f(t01[k]), h2f(t11[k]), wx, wy, bilin_mode(&m->P));
        float num = MADD2(m->P.fma_contraction, h2f(Av[k]), Wv, a, wm);
        float An = m->P.appearance_blend_division ? num / (Wv + wm) : num * inv;
        Av[k] = f2h(An);
      }
      Wt[lin] = fminf(Wv + wm, m->P.appearance_max_weight);
    }
  }
  m->last_n_app_blocks = nc;
  free(pos);
  free(cl);
  return 0;
}

/* Colour: same gate; per channel A' = (uint8) floor( (A*W + a*wm)*inv + 0.5 ), inv = 1/(W + wm). */
int orc_add_color_frame(orc_mapper* m, const uint8_t* rgb, const uint8_t* mask, int H, int W, const float* T_L_C16,
                        const float* K9) {
  cam_t cam;
  cam_from_K(K9, W, H, &cam);
  rigid_t T_L_C, T_C_L;
  rigid_from_T(T_L_C16, &T_L_C);
  rigid_inverse(&T_L_C, &T_C_L);
  int* cl;
  int nc = app_candidates(m, &cam, &T_C_L, &cl);
  render_synthetic_depth(m, &cam, &T_L_C);
  float wm = m->P.appearance_measurement_weight;
  for (int i = 0; i < nc; ++i) {
    const block_t* tb = &m->tsdf.blocks[cl[i]];
    int pos = layer_alloc(&m->color, tb->bx, tb->by, tb->bz, NULL);
    block_t* B = &m->color.blocks[pos];
    color_block* cb = (color_block*)B->data;
    for (int lin = 0; lin < VPB; ++lin) {
      int x0, y0;
      float wx, wy;
      if (!app_gate(m, &cam, &T_C_L, mask, B, lin, &x0, &y0, &wx, &wy)) continue;
      const uint8_t* t00 = rgb + ((size_t)y0 * W + x0) * 3;
      const uint8_t* t10 = t00 + 3;
      const uint8_t* t01 = t00 + (size_t)W * 3;
      const uint8_t* t11 = t01 + 3;
      float Wv = cb->w[lin];
      float inv = 1.0f / (Wv + wm);
      for (int k = 0; k < 3; ++k) {
        float a = bilin_c((float)t00[k], (float)t10[k], (float)t01[k], (float)t11[k], wx, wy, bilin_mode(&m->P));
        float num = MADD2(m->P.fma_contraction, (float)cb->rgb[lin * 3 + k], Wv, a, wm);
        float An = m->P.appearance_blend_division ? num / (Wv + wm) : num * inv;
        cb->rgb[lin * 3 + k] = (uint8_t)floorf(An + 0.5f);
      }
      cb->w[lin] = fminf(Wv + wm, m->P.appearance_max_weight);
    }
  }
  m->last_n_app_blocks = nc;
  free(cl);
  return 0;
}

/* ------------------------------------------------------------------------------- */
/* A.6  decay                                                                       */
/* ------------------------------------------------------------------------------- */
/* W <- W*factor for every voxel of every TSDF block; a block whose voxels all have
 * W < decayed_weight_threshold is deallocated (live order of the others is preserved).
 * Colour / feature layers are not touched. */
void orc_decay(orc_mapper* m) {
  int n = m->tsdf.n;
  uint8_t* kill = (uint8_t*)calloc(n > 0 ? n : 1, 1);
  int any = 0;
  for (int i = 0; i < n; ++i) {
    tsdf_block* tb = (tsdf_block*)m->tsdf.blocks[i].data;
    int all = 1;
    for (int lin = 0; lin < VPB; ++lin) {
      float w = tb->w[lin] * m->P.tsdf_decay_factor;
      tb->w[lin] = w;
      if (!(w < m->P.decayed_weight_threshold)) all = 0;
    }
    if (all && m->P.deallocate_decayed_blocks) {
      kill[i] = 1;
      any = 1;
    }
  }
  if (any) layer_remove(&m->tsdf, kill);
  free(kill);
  if (m->P.decay_appearance_layers) { /* option: the appearance weights fade with the same factor (no deallocation) */
    for (int i = 0; i < m->color.n; ++i) {
      color_block* cb = (color_block*)m->color.blocks[i].data;
      for (int lin = 0; lin < VPB; ++lin) cb->w[lin] = cb->w[lin] * m->P.tsdf_decay_factor;
    }
    for (int i = 0; i < m->feat.n; ++i) {
      float* w = featw_ptr(m->feat.blocks[i].data, m->P.feature_channels);
      for (int lin = 0; lin < VPB; ++lin) w[lin] = w[lin] * m->P.tsdf_decay_factor;
    }
  }
}

/* ------------------------------------------------------------------------------- */
/* A.7  surface vertices (marching-cubes vertex set, welded per block) + features   */
/* ------------------------------------------------------------------------------- */
/*
 * For each live TSDF block B (live order) build the 9x9x9 lattice of (D, valid) from B and its
 * +x,+y,+z,+xy,+xz,+yz,+xyz neighbours; lattice point valid iff its block exists and
 * W >= mesh_min_weight.  Cube (i,j,k), 0<=i,j,k<=7, is valid iff its 8 corners are valid.
 * A lattice edge from q to q+e_a (both ends inside the lattice) carries a vertex iff
 * (D_q < 0) != (D_q' < 0) and at least one valid cube of B contains the edge.
 * Position: pa = centre(q), pb = centre(q+e_a) with centre_i(q) = (float)b_i*bs + ((float)q_i + .5)*v;
 *   t = Da / (Da - Db);  pos_a = pa_a + t*(pb_a - pa_a), other coordinates = pa's.
 * Vertices are emitted for q in lexicographic (qx, qy, qz) order, axis a = 0,1,2 -- which equals
 * the per-block welded (deduplicated) marching-cubes vertex set.
 * Vertex feature: feature voxel containing pos (if its block exists and W > 0), else zeros.
 */
static void mesh_push(orc_mapper* m, const float* pos, int C) {
  if (m->mesh_n == m->mesh_cap || m->mesh_C != C) {
    if (m->mesh_n == m->mesh_cap) m->mesh_cap = m->mesh_cap ? m->mesh_cap * 2 : 4096;
    m->mesh_C = C;
    m->mesh_v = (float*)realloc(m->mesh_v, sizeof(float) * 3 * m->mesh_cap);
    m->mesh_f = (uint16_t*)realloc(m->mesh_f, sizeof(uint16_t) * (size_t)C * m->mesh_cap);
    m->mesh_c = (uint8_t*)realloc(m->mesh_c, 3 * (size_t)m->mesh_cap);
  }
  memcpy(&m->mesh_v[3 * (size_t)m->mesh_n], pos, sizeof(float) * 3);
  uint16_t* f = &m->mesh_f[(size_t)C * m->mesh_n];
  memset(f, 0, sizeof(uint16_t) * C);
  int lin;
  int p = voxel_at(m, &m->feat, pos, &lin);
  if (p >= 0) {
    const void* data = m->feat.blocks[p].data;
    if (featw_ptr((void*)data, C)[lin] > 0.0f) memcpy(f, feat_ptr((void*)data) + (size_t)lin * C, sizeof(uint16_t) * C);
  }
  /* vertex colour: colour voxel containing pos (if its block exists and W > 0), else black */
  uint8_t* c = &m->mesh_c[3 * (size_t)m->mesh_n];
  c[0] = c[1] = c[2] = 0;
  p = voxel_at(m, &m->color, pos, &lin);
  if (p >= 0) {
    const color_block* cb = (const color_block*)m->color.blocks[p].data;
    if (cb->w[lin] > 0.0f) memcpy(c, &cb->rgb[lin * 3], 3);
  }
  m->mesh_n++;
}

/* Triangles: for each block, after its vertices, the valid cubes (i,j,k) in lexicographic order emit the triangles of
 * their corner pattern (include/mmf_mc_table.h: corner c = dx*4+dy*2+dz inside iff D < 0; cube edge e = a*4+s1*2+s2 is the
 * lattice edge from o + s1*e_(a+1) + s2*e_(a+2) along a), as indices of the block's vertices. */
static void mesh_push_tri(orc_mapper* m, int a, int b, int c) {
  if (m->mesh_nt == m->mesh_tcap) {
    m->mesh_tcap = m->mesh_tcap ? m->mesh_tcap * 2 : 8192;
    m->mesh_t = (int*)realloc(m->mesh_t, sizeof(int) * 3 * m->mesh_tcap);
  }
  int* t = &m->mesh_t[3 * (size_t)m->mesh_nt++];
  t[0] = a;
  t[1] = b;
  t[2] = c;
}

int orc_update_feature_mesh(orc_mapper* m) {
  const int C = m->P.feature_channels;
  m->mesh_n = 0;
  m->mesh_nt = 0;
  m->mesh_C = C;
  static const int E[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  static int vid[9][9][9][3]; /* vertex index of lattice edge (q, a) of the current block, -1 = none */
  for (int bi = 0; bi < m->tsdf.n; ++bi) {
    const block_t* B = &m->tsdf.blocks[bi];
    float D[9][9][9];
    uint8_t V[9][9][9];
    const tsdf_block* nb[2][2][2];
    for (int dx = 0; dx < 2; ++dx)
      for (int dy = 0; dy < 2; ++dy)
        for (int dz = 0; dz < 2; ++dz) {
          int p = layer_find(&m->tsdf, B->bx + dx, B->by + dy, B->bz + dz);
          nb[dx][dy][dz] = p >= 0 ? (const tsdf_block*)m->tsdf.blocks[p].data : NULL;
        }
    for (int x = 0; x < 9; ++x)
      for (int y = 0; y < 9; ++y)
        for (int z = 0; z < 9; ++z) {
          const tsdf_block* tb = nb[x >> 3][y >> 3][z >> 3];
          int lin = ((x & 7) * 8 + (y & 7)) * 8 + (z & 7);
          if (tb && tb->w[lin] >= m->P.mesh_min_weight) {
            V[x][y][z] = 1;
            D[x][y][z] = tb->d[lin];
          } else {
            V[x][y][z] = 0;
            D[x][y][z] = 0.0f;
          }
        }
    uint8_t CV[8][8][8];
    for (int x = 0; x < 8; ++x)
      for (int y = 0; y < 8; ++y)
        for (int z = 0; z < 8; ++z) {
          CV[x][y][z] = V[x][y][z] & V[x + 1][y][z] & V[x][y + 1][z] & V[x + 1][y + 1][z] & V[x][y][z + 1] &
                        V[x + 1][y][z + 1] & V[x][y + 1][z + 1] & V[x + 1][y + 1][z + 1];
        }
    for (int x = 0; x < 9; ++x)
      for (int y = 0; y < 9; ++y)
        for (int z = 0; z < 9; ++z)
          for (int a = 0; a < 3; ++a) {
            vid[x][y][z][a] = -1;
            int q[3] = {x, y, z};
            int r[3] = {x + E[a][0], y + E[a][1], z + E[a][2]};
            if (r[a] > 8) continue;
            float Da = D[x][y][z], Db = D[r[0]][r[1]][r[2]];
            if ((Da < 0.0f) == (Db < 0.0f)) continue;
            /* cubes containing the edge: origin o with o_a = q_a, o_b in {q_b-1, q_b} for b != a */
            int a1 = (a + 1) % 3, a2 = (a + 2) % 3;
            int found = 0;
            for (int s1 = -1; s1 <= 0 && !found; ++s1)
              for (int s2 = -1; s2 <= 0 && !found; ++s2) {
                int o[3];
                o[a] = q[a];
                o[a1] = q[a1] + s1;
                o[a2] = q[a2] + s2;
                if (o[0] < 0 || o[1] < 0 || o[2] < 0 || o[0] > 7 || o[1] > 7 || o[2] > 7) continue;
                if (CV[o[0]][o[1]][o[2]]) found = 1;
              }
            if (!found) continue;
            float pa[3] = {(float)B->bx * m->bs + ((float)x + 0.5f) * m->v,
                           (float)B->by * m->bs + ((float)y + 0.5f) * m->v,
                           (float)B->bz * m->bs + ((float)z + 0.5f) * m->v};
            int bb[3] = {B->bx, B->by, B->bz};
            float pb_a = (float)bb[a] * m->bs + ((float)r[a] + 0.5f) * m->v;
            float t = Da / (Da - Db);
            float pos[3] = {pa[0], pa[1], pa[2]};
            pos[a] = pa[a] + t * (pb_a - pa[a]);
            vid[x][y][z][a] = m->mesh_n;
            mesh_push(m, pos, C);
          }
    for (int x = 0; x < 8; ++x)
      for (int y = 0; y < 8; ++y)
        for (int z = 0; z < 8; ++z) {
          if (!CV[x][y][z]) continue;
          int pat = 0;
          for (int c = 0; c < 8; ++c)
            if (D[x + (c >> 2)][y + ((c >> 1) & 1)][z + (c & 1)] < 0.0f) pat |= 1 << c;
          for (int k = 0; k < mmf_mc_num_tris[pat]; ++k) {
            int v3[3];
            for (int j = 0; j < 3; ++j) {
              const int e = mmf_mc_tris[pat][3 * k + j];
              const int a = e >> 2, s1 = (e >> 1) & 1, s2 = e & 1;
              int q[3] = {x, y, z};
              q[(a + 1) % 3] += s1;
              q[(a + 2) % 3] += s2;
              v3[j] = vid[q[0]][q[1]][q[2]][a];
            }
            mesh_push_tri(m, v3[0], v3[1], v3[2]);
          }
        }
  }
  return m->mesh_n;
}

/* triangles [T,3] and vertex colours [V,3] of the last orc_update_feature_mesh; returns T */
int orc_mesh_num_triangles(const orc_mapper* m) { return m->mesh_nt; }
int orc_get_mesh_topology(const orc_mapper* m, int* tris, uint8_t* colors) {
  if (tris) memcpy(tris, m->mesh_t, sizeof(int) * 3 * (size_t)m->mesh_nt);
  if (colors) memcpy(colors, m->mesh_c, 3 * (size_t)m->mesh_n);
  return m->mesh_nt;
}

int orc_get_feature_mesh(const orc_mapper* m, float* verts, uint16_t* feats) {
  memcpy(verts, m->mesh_v, sizeof(float) * 3 * (size_t)m->mesh_n);
  memcpy(feats, m->mesh_f, sizeof(uint16_t) * (size_t)m->mesh_C * m->mesh_n);
  return m->mesh_n;
}

/* ------------------------------------------------------------------------------- */
/* inspection                                                                       */
/* ------------------------------------------------------------------------------- */
static layer_t* pick_layer(orc_mapper* m, int layer) { return layer == 0 ? &m->tsdf : (layer == 1 ? &m->color : &m->feat); }

int orc_num_blocks(orc_mapper* m, int layer) { return pick_layer(m, layer)->n; }

/* block indices in live (allocation) order */
void orc_get_block_indices(orc_mapper* m, int layer, int* out) {
  layer_t* L = pick_layer(m, layer);
  for (int i = 0; i < L->n; ++i) {
    out[3 * i] = L->blocks[i].bx;
    out[3 * i + 1] = L->blocks[i].by;
    out[3 * i + 2] = L->blocks[i].bz;
  }
}

/* TSDF block i -> out[512][2] (distance, weight) */
void orc_get_tsdf_block(orc_mapper* m, int i, float* out) {
  const tsdf_block* tb = (const tsdf_block*)m->tsdf.blocks[i].data;
  for (int lin = 0; lin < VPB; ++lin) {
    out[2 * lin] = tb->d[lin];
    out[2 * lin + 1] = tb->w[lin];
  }
}

/* all TSDF blocks -> out[n][512][2] */
void orc_get_all_tsdf(orc_mapper* m, float* out) {
  for (int i = 0; i < m->tsdf.n; ++i) orc_get_tsdf_block(m, i, out + (size_t)i * VPB * 2);
}

/* feature block i -> feats[512][C] (f16 bits), weights[512] */
void orc_get_feature_block(orc_mapper* m, int i, uint16_t* feats, float* weights) {
  const int C = m->P.feature_channels;
  memcpy(feats, feat_ptr(m->feat.blocks[i].data), sizeof(uint16_t) * (size_t)VPB * C);
  memcpy(weights, featw_ptr(m->feat.blocks[i].data, C), sizeof(float) * VPB);
}

void orc_get_all_features(orc_mapper* m, uint16_t* feats, float* weights) {
  const int C = m->P.feature_channels;
  for (int i = 0; i < m->feat.n; ++i) orc_get_feature_block(m, i, feats + (size_t)i * VPB * C, weights + (size_t)i * VPB);
}

/* colour block i -> rgb[512][3], weights[512] */
void orc_get_color_block(orc_mapper* m, int i, uint8_t* rgb, float* weights) {
  const color_block* cb = (const color_block*)m->color.blocks[i].data;
  memcpy(rgb, cb->rgb, VPB * 3);
  memcpy(weights, cb->w, sizeof(float) * VPB);
}

void orc_get_all_colors(orc_mapper* m, uint8_t* rgb, float* weights) {
  for (int i = 0; i < m->color.n; ++i) orc_get_color_block(m, i, rgb + (size_t)i * VPB * 3, weights + (size_t)i * VPB);
}

/* candidate (in-view) block list of the last add_depth_frame, sorted */
int orc_last_view_blocks(orc_mapper* m, int* out) {
  if (out) memcpy(out, m->cand, sizeof(int) * 3 * (size_t)m->ncand);
  return m->ncand;
}

int orc_last_counts(orc_mapper* m, int* n_tsdf, int* n_app) {
  *n_tsdf = m->last_n_tsdf_blocks;
  *n_app = m->last_n_app_blocks;
  return 0;
}

/* last synthetic depth image */
int orc_get_synthetic_depth(orc_mapper* m, float* out, int* Ws, int* Hs) {
  *Ws = m->synth_W;
  *Hs = m->synth_H;
  if (out && m->synth) memcpy(out, m->synth, sizeof(float) * (size_t)m->synth_W * m->synth_H);
  return 0;
}

/* render only (for isolated sphere-tracing parity tests) */
int orc_render_synthetic_depth(orc_mapper* m, int H, int W, const float* T_L_C16, const float* K9) {
  cam_t cam;
  cam_from_K(K9, W, H, &cam);
  rigid_t T_L_C;
  rigid_from_T(T_L_C16, &T_L_C);
  render_synthetic_depth(m, &cam, &T_L_C);
  return 0;
}

/* point query: out[n][C+1] f32 (features then weight); zeros where unobserved.
 * (nvblox_torch Mapper.query_layer(QueryType.FEATURE), mindmap/visualization/visualizer.py:678-691) */
void orc_query_features(orc_mapper* m, const float* pts, int n, float* out) {
  const int C = m->P.feature_channels;
  for (int i = 0; i < n; ++i) {
    float* o = out + (size_t)i * (C + 1);
    memset(o, 0, sizeof(float) * (C + 1));
    int lin;
    int p = voxel_at(m, &m->feat, pts + 3 * i, &lin);
    if (p < 0) continue;
    const void* data = m->feat.blocks[p].data;
    const uint16_t* f = feat_ptr((void*)data) + (size_t)lin * C;
    for (int k = 0; k < C; ++k) o[k] = h2f(f[k]);
    o[C] = featw_ptr((void*)data, C)[lin];
  }
}

/* point query on the TSDF layer: out[n][2] (distance, weight); weight 0 where unobserved */
void orc_query_tsdf(orc_mapper* m, const float* pts, int n, float* out) {
  for (int i = 0; i < n; ++i) {
    out[2 * i] = 0.0f;
    out[2 * i + 1] = 0.0f;
    int lin;
    int p = voxel_at(m, &m->tsdf, pts + 3 * i, &lin);
    if (p < 0) continue;
    const tsdf_block* tb = (const tsdf_block*)m->tsdf.blocks[p].data;
    out[2 * i] = tb->d[lin];
    out[2 * i + 1] = tb->w[lin];
  }
}

/*
 * Feature image from the backbone's low-res map (image_processing/feature_extraction.py:188-191,198-210 + the f16 cast of
 * nvblox_mapping_helpers.py:256): bilinear, align_corners = False, [h,w,Cin] f32 channels-last -> [Hf,Wf,Cpad] f16 with zero pad
 * channels; float32 arithmetic in the order of torch's upsample_bilinear2d
 *   s = scale * (dst + 0.5) - 0.5 (clamped at 0);  i0 = min((int)s, n - 1);  i1 = min(i0 + 1, n - 1);  l1 = s - i0;  l0 = 1 - l1;
 *   val = ly0 * (lx0 * a00 + lx1 * a01) + ly1 * (lx0 * a10 + lx1 * a11)
 * with the contraction of orc_params.fma_contraction when `fma` (what a CUDA build of that kernel computes).
 */
void orc_upsample_features(const float* low, int h, int w, int Cin, uint16_t* out, int Hf, int Wf, int Cpad, int fma) {
  const float sh = (float)h / (float)Hf, sw = (float)w / (float)Wf;
#pragma omp parallel for schedule(static)
  for (int yf = 0; yf < Hf; ++yf) {
    float sy = MADD(fma, sh, (float)yf + 0.5f, -0.5f);
    sy = sy < 0.0f ? 0.0f : sy;
    const int y0 = (int)sy < h - 1 ? (int)sy : h - 1;
    const int y1 = y0 < h - 1 ? y0 + 1 : y0;
    const float ly1 = sy - (float)y0, ly0 = 1.0f - ly1;
    for (int xf = 0; xf < Wf; ++xf) {
      float sx = MADD(fma, sw, (float)xf + 0.5f, -0.5f);
      sx = sx < 0.0f ? 0.0f : sx;
      const int x0 = (int)sx < w - 1 ? (int)sx : w - 1;
      const int x1 = x0 < w - 1 ? x0 + 1 : x0;
      const float lx1 = sx - (float)x0, lx0 = 1.0f - lx1;
      const float *a00 = low + ((size_t)y0 * w + x0) * Cin, *a01 = low + ((size_t)y0 * w + x1) * Cin;
      const float *a10 = low + ((size_t)y1 * w + x0) * Cin, *a11 = low + ((size_t)y1 * w + x1) * Cin;
      uint16_t* o = out + ((size_t)yf * Wf + xf) * Cpad;
      for (int k = 0; k < Cin; ++k)
        o[k] = f2h(MADD2(fma, ly0, MADD2(fma, lx0, a00[k], lx1, a01[k]), ly1, MADD2(fma, lx0, a10[k], lx1, a11[k])));
      for (int k = Cin; k < Cpad; ++k) o[k] = 0;
    }
  }
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* bench.py's cpu_baseline sweeps the thread count and reports the best */
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
