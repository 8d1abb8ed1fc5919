/* apriltag_oracle.c -- CPU restatement of AprilRobotics apriltag_detect() (TEST INFRASTRUCTURE).
 * See apriltag_oracle.h for scope and parity status ("parity unpinned" beyond the reference's golden
 * vector).  Plain C99, single-threaded, no dependencies.  Compile with -ffp-contract=off: the HIP
 * path is compared bit-for-bit against this file, so every floating-point operation below is one
 * IEEE-754 operation in the order written.
 *
 * Each function cites the step of the public AprilTag-3 algorithm it restates (SURVEY.md Appendix A
 * numbering, "A.n") and the reference call site whose closed implementation it stands in for
 * (paths relative to /root/reference/isaac_ros_apriltag/).
 */
#include "apriltag_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <pthread.h>
#include <string.h>

#include "../include/apriltag_amd_families.h"

/* ------------------------------------------------------------------------------------------- */
/* parameters and families                                                                      */
/* ------------------------------------------------------------------------------------------- */

/* A.0 detector defaults; tile_size/size/max_tags defaults follow src/apriltag_node.cpp:564-567. */
void ato_default_params(ato_params_t* p) {
  memset(p, 0, sizeof(*p));
  p->decimate = 1;
  p->tile_size = 4;
  p->min_white_black_diff = 5;
  p->min_component_size = 25;
  p->min_cluster_points = 24;
  p->max_nmaxima = 10;
  /* cos(10 deg) as upstream holds it: apriltag_quad_thresh_params.cos_critical_rad is a FLOAT field, so both angle
   * tests compare against (double)(float)cos(10 * M_PI / 180) */
  p->cos_critical_rad = (double)(float)0x1.f838b8c811c17p-1;
  p->max_line_fit_mse = 10.0;
  p->refine_edges = 1;
  p->decode_sharpening = 0.25;
  p->max_hamming = 2;
  p->fx = p->fy = 1000.0;
  p->cx = 960.0;
  p->cy = 540.0;
  p->tag_size = 0.22;
}

/* Accepted family strings: src/apriltag_node.cpp:47-58 (only those with an offline codebook). */
int ato_builtin_family(const char* name, ato_family_t* out) {
  memset(out, 0, sizeof(*out));
  strncpy(out->name, name, sizeof(out->name) - 1);
  if (!strcmp(name, "tag36h11")) {
    out->d = 6; out->ncodes = APRILTAG_AMD_TAG36H11_NCODES; out->codes = apriltag_amd_tag36h11_codes;
#ifdef APRILTAG_AMD_TAG36H10_NCODES
  } else if (!strcmp(name, "tag36h10")) {
    out->d = 6; out->ncodes = APRILTAG_AMD_TAG36H10_NCODES; out->codes = apriltag_amd_tag36h10_codes;
#endif
  } else if (!strcmp(name, "tag25h9")) {
    out->d = 5; out->ncodes = APRILTAG_AMD_TAG25H9_NCODES; out->codes = apriltag_amd_tag25h9_codes;
  } else if (!strcmp(name, "tag16h5")) {
    out->d = 4; out->ncodes = APRILTAG_AMD_TAG16H5_NCODES; out->codes = apriltag_amd_tag16h5_codes;
  } else {
    return -1;
  }
  out->nbits = out->d * out->d;
  out->width_at_border = out->d + 2;
  out->total_width = out->d + 4;
  out->reversed_border = 0;
  for (uint32_t i = 0; i < out->nbits; i++) { out->bit_x[i] = (int8_t)(1 + i % out->d); out->bit_y[i] = (int8_t)(1 + i / out->d); }
  return 0;
}

/* index of the data bit that the 90-degree pattern rotation moves onto bit i: new(x, y) = old(wb - 1 - y, x)
 * (for a classic d x d family this is new(r, c) = old(c, d - 1 - r)); -1 if the layout has no such cell */
static int rot_source(const ato_family_t* f, int i) {
  int sx = (int)f->width_at_border - 1 - f->bit_y[i], sy = f->bit_x[i];
  for (uint32_t j = 0; j < f->nbits; j++)
    if (f->bit_x[j] == sx && f->bit_y[j] == sy) return (int)j;
  return -1;
}

int ato_custom_family(const char* name, uint32_t nbits, const int8_t* bit_x, const int8_t* bit_y, uint32_t width_at_border,
                      uint32_t total_width, int reversed_border, const uint64_t* codes, uint32_t ncodes, ato_family_t* out) {
  memset(out, 0, sizeof(*out));
  if (!name || !bit_x || !bit_y || !codes || nbits == 0 || nbits > 64 || ncodes == 0) return -1;
  if (width_at_border < 3 || total_width < width_at_border || total_width > 12 || ((total_width - width_at_border) & 1)) return -1;
  strncpy(out->name, name, sizeof(out->name) - 1);
  out->nbits = nbits; out->d = 0; out->width_at_border = width_at_border; out->total_width = total_width;
  out->reversed_border = reversed_border ? 1 : 0; out->ncodes = ncodes; out->codes = codes;
  int min_coord = ((int)width_at_border - (int)total_width) / 2;
  for (uint32_t i = 0; i < nbits; i++) {
    if (bit_x[i] < min_coord || bit_x[i] >= min_coord + (int)total_width || bit_y[i] < min_coord || bit_y[i] >= min_coord + (int)total_width) return -1;
    out->bit_x[i] = bit_x[i]; out->bit_y[i] = bit_y[i];
  }
  for (uint32_t i = 0; i < nbits; i++) {
    if (rot_source(out, (int)i) < 0) return -1;
    for (uint32_t j = 0; j < i; j++) if (bit_x[j] == bit_x[i] && bit_y[j] == bit_y[i]) return -1;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* S1  decimation (A.1)  -- stands in for the first step inside cuAprilTagsDetect               */
/*     (src/apriltag_node.cpp:491-493)                                                          */
/* ------------------------------------------------------------------------------------------- */
void ato_decimate(const uint8_t* in, int w, int h, int pitch, int f, uint8_t* out, int* sw, int* sh) {
  int ow = 1 + (w - 1) / f, oh = 1 + (h - 1) / f;
  for (int sy = 0; sy < oh; sy++)
    for (int sx = 0; sx < ow; sx++) out[sy * ow + sx] = in[(size_t)(sy * f) * pitch + sx * f];
  *sw = ow;
  *sh = oh;
}

/* ------------------------------------------------------------------------------------------- */
/* S2  adaptive tile min/max threshold (A.2)                                                    */
/* ------------------------------------------------------------------------------------------- */
void ato_threshold(const uint8_t* im, int w, int h, int tile, int min_diff, uint8_t* out) {
  int tw = w / tile, th = h / tile;
  uint8_t* tmin = (uint8_t*)malloc((size_t)tw * th);
  uint8_t* tmax = (uint8_t*)malloc((size_t)tw * th);
  uint8_t* dmin = (uint8_t*)malloc((size_t)tw * th);
  uint8_t* dmax = (uint8_t*)malloc((size_t)tw * th);
  for (int ty = 0; ty < th; ty++)
    for (int tx = 0; tx < tw; tx++) {
      int mn = 255, mx = 0;
      for (int dy = 0; dy < tile; dy++)
        for (int dx = 0; dx < tile; dx++) {
          int v = im[(ty * tile + dy) * w + tx * tile + dx];
          if (v < mn) mn = v;
          if (v > mx) mx = v;
        }
      tmin[ty * tw + tx] = (uint8_t)mn;
      tmax[ty * tw + tx] = (uint8_t)mx;
    }
  /* 3x3 tile neighbourhood: max of max, min of min, clamped at the borders */
  for (int ty = 0; ty < th; ty++)
    for (int tx = 0; tx < tw; tx++) {
      int mn = 255, mx = 0;
      for (int dy = -1; dy <= 1; dy++) {
        if (ty + dy < 0 || ty + dy >= th) continue;
        for (int dx = -1; dx <= 1; dx++) {
          if (tx + dx < 0 || tx + dx >= tw) continue;
          int a = tmin[(ty + dy) * tw + tx + dx], b = tmax[(ty + dy) * tw + tx + dx];
          if (a < mn) mn = a;
          if (b > mx) mx = b;
        }
      }
      dmin[ty * tw + tx] = (uint8_t)mn;
      dmax[ty * tw + tx] = (uint8_t)mx;
    }
  for (int ty = 0; ty < th; ty++)
    for (int tx = 0; tx < tw; tx++) {
      int mn = dmin[ty * tw + tx], mx = dmax[ty * tw + tx];
      if (mx - mn < min_diff) {
        for (int dy = 0; dy < tile; dy++)
          for (int dx = 0; dx < tile; dx++) out[(ty * tile + dy) * w + tx * tile + dx] = 127;
        continue;
      }
      int thresh = mn + (mx - mn) / 2;
      for (int dy = 0; dy < tile; dy++)
        for (int dx = 0; dx < tile; dx++) {
          int idx = (ty * tile + dy) * w + tx * tile + dx;
          out[idx] = (im[idx] > thresh) ? 255 : 0;
        }
    }
  /* pixels right of / below the last full tile use the nearest tile; no low-contrast rule there */
  for (int y = 0; y < h; y++) {
    int x0 = (y >= th * tile) ? 0 : tw * tile;
    int ty = y / tile;
    if (ty >= th) ty = th - 1;
    for (int x = x0; x < w; x++) {
      int tx = x / tile;
      if (tx >= tw) tx = tw - 1;
      int mn = dmin[ty * tw + tx], mx = dmax[ty * tw + tx];
      int thresh = mn + (mx - mn) / 2;
      out[y * w + x] = (im[y * w + x] > thresh) ? 255 : 0;
    }
  }
  free(tmin); free(tmax); free(dmin); free(dmax);
}

/* ------------------------------------------------------------------------------------------- */
/* S3  union-find connected components (A.3)                                                    */
/*     Links: for x in [1, w-2]: left; up (y>0); and for white pixels up-left / up-right.       */
/*     CANONICAL: representative = smallest pixel index of the component.                       */
/* ------------------------------------------------------------------------------------------- */
static uint32_t uf_find(uint32_t* parent, uint32_t i) {
  uint32_t r = i;
  while (parent[r] != r) r = parent[r];
  while (parent[i] != r) { uint32_t n = parent[i]; parent[i] = r; i = n; }
  return r;
}
static void uf_union(uint32_t* parent, uint32_t a, uint32_t b) {
  a = uf_find(parent, a);
  b = uf_find(parent, b);
  if (a == b) return;
  if (a < b) parent[b] = a; else parent[a] = b;
}

void ato_connected_components(const uint8_t* thr, int w, int h, uint32_t* label, uint32_t* csize) {
  size_t n = (size_t)w * h;
  uint32_t* parent = (uint32_t*)malloc(n * sizeof(uint32_t));
  for (size_t i = 0; i < n; i++) parent[i] = (uint32_t)i;
  for (int y = 0; y < h; y++)
    for (int x = 1; x < w - 1; x++) {
      int v = thr[y * w + x];
      if (v == 127) continue;
      uint32_t i = (uint32_t)(y * w + x);
      if (thr[y * w + x - 1] == v) uf_union(parent, i, i - 1);
      if (y > 0) {
        if (thr[(y - 1) * w + x] == v) uf_union(parent, i, i - w);
        if (v == 255) {
          if (thr[(y - 1) * w + x - 1] == v) uf_union(parent, i, i - w - 1);
          if (thr[(y - 1) * w + x + 1] == v) uf_union(parent, i, i - w + 1);
        }
      }
    }
  memset(csize, 0, n * sizeof(uint32_t));
  for (size_t i = 0; i < n; i++) {
    if (thr[i] == 127) { label[i] = ATO_NO_LABEL; continue; }
    uint32_t r = uf_find(parent, (uint32_t)i);
    label[i] = r;
    csize[r]++;
  }
  free(parent);
}

/* ------------------------------------------------------------------------------------------- */
/* S4  boundary points and clusters (A.4)                                                       */
/* ------------------------------------------------------------------------------------------- */
typedef struct { uint64_t key; uint32_t pt; } kp_t;

static inline uint32_t pack_point(int x, int y, int gx, int gy) {
  /* x,y half-pixel units (< 2^14); gx,gy in {-255,0,255} -> code {0,1,2} */
  return ((uint32_t)x << 18) | ((uint32_t)y << 4) | ((uint32_t)(gx / 255 + 1) << 2) | (uint32_t)(gy / 255 + 1);
}
static inline void unpack_point(uint32_t p, int* x, int* y, int* gx, int* gy) {
  *x = (int)(p >> 18);
  *y = (int)((p >> 4) & 0x3FFF);
  *gx = ((int)((p >> 2) & 3) - 1) * 255;
  *gy = ((int)(p & 3) - 1) * 255;
}

/* Collects every boundary point with its component-pair key (raster order; grouped by group_points). */
static kp_t* gradient_points(const uint8_t* thr, const uint32_t* label, const uint32_t* csize, int w, int h,
                             int min_comp, size_t* nout) {
  static const int DX[4] = {1, 0, -1, 1}, DY[4] = {0, 1, 1, 1};
  size_t cap = 1 << 16, n = 0;
  kp_t* pts = (kp_t*)malloc(cap * sizeof(kp_t));
  for (int y = 1; y < h - 1; y++) {
    /* connected_last: the previous pixel of this row emitted its (1,1) point.  That point and this
     * pixel's (-1,1) point sit on the same half-pixel location, so (-1,1) is skipped then (upstream
     * do_gradient_clusters). */
    int connected_last = 0;
    for (int x = 1; x < w - 1; x++) {
      int v0 = thr[y * w + x];
      if (v0 == 127) { connected_last = 0; continue; }
      uint32_t r0 = label[y * w + x];
      if ((int)csize[r0] < min_comp) { connected_last = 0; continue; }
      int connected = 0;
      for (int k = 0; k < 4; k++) {
        if (k == 2 && connected_last) continue;
        int x1 = x + DX[k], y1 = y + DY[k];
        int v1 = thr[y1 * w + x1];
        if (v0 + v1 != 255) continue;
        uint32_t r1 = label[y1 * w + x1];
        if ((int)csize[r1] < min_comp) continue;   /* upstream: size > 24 */
        uint64_t key = r0 < r1 ? ((uint64_t)r0 << 32) | r1 : ((uint64_t)r1 << 32) | r0;
        if (n == cap) { cap *= 2; pts = (kp_t*)realloc(pts, cap * sizeof(kp_t)); }
        pts[n].key = key;
        pts[n].pt = pack_point(2 * x + DX[k], 2 * y + DY[k], DX[k] * (v1 - v0), DY[k] * (v1 - v0));
        n++;
        if (k == 3) connected = 1;
      }
      connected_last = connected;
    }
  }
  *nout = n;
  return pts;
}

/* Groups the points by key with an open-addressing hash (upstream uses a hash map as well): fills
 * clusters[] (sorted by key afterwards, CANONICAL) and pts_out[] (points of a cluster contiguous). */
static uint32_t u32_cmp_store;
static int u32_cmp(const void* a, const void* b) {
  uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  (void)u32_cmp_store;
  return (x > y) - (x < y);
}
static int cluster_key_cmp(const void* a, const void* b) {
  const ato_cluster_t* x = (const ato_cluster_t*)a; const ato_cluster_t* y = (const ato_cluster_t*)b;
  return (x->key > y->key) - (x->key < y->key);
}
static size_t group_points(const kp_t* kp, size_t n, ato_cluster_t** clusters_out, uint32_t* pts_out, int sort_points) {
  size_t cap = 1024;
  while (cap < n / 4 + 16) cap <<= 1;
  uint64_t* hk = (uint64_t*)malloc(cap * 8);
  uint32_t* hi = (uint32_t*)malloc(cap * 4);
  memset(hk, 0xFF, cap * 8);
  size_t ccap = 1024, nc = 0;
  ato_cluster_t* cl = (ato_cluster_t*)malloc(ccap * sizeof(ato_cluster_t));
  uint32_t* cidx = (uint32_t*)malloc((n ? n : 1) * 4);
  for (size_t i = 0; i < n; i++) {
    uint64_t key = kp[i].key;
    size_t h = (size_t)((key * 0x9E3779B97F4A7C15ULL) >> 20) & (cap - 1);
    while (hk[h] != key && hk[h] != ~0ULL) h = (h + 1) & (cap - 1);
    if (hk[h] == ~0ULL) {
      if (nc * 2 > cap) {  /* cannot happen with cap >= n/4 unless nearly every point is its own pair; grow by rebuild */
        size_t ncap = cap * 4;
        uint64_t* nk = (uint64_t*)malloc(ncap * 8);
        uint32_t* ni = (uint32_t*)malloc(ncap * 4);
        memset(nk, 0xFF, ncap * 8);
        for (size_t j = 0; j < cap; j++)
          if (hk[j] != ~0ULL) {
            size_t g = (size_t)((hk[j] * 0x9E3779B97F4A7C15ULL) >> 20) & (ncap - 1);
            while (nk[g] != ~0ULL) g = (g + 1) & (ncap - 1);
            nk[g] = hk[j]; ni[g] = hi[j];
          }
        free(hk); free(hi);
        hk = nk; hi = ni; cap = ncap;
        h = (size_t)((key * 0x9E3779B97F4A7C15ULL) >> 20) & (cap - 1);
        while (hk[h] != ~0ULL) h = (h + 1) & (cap - 1);
      }
      if (nc == ccap) { ccap *= 2; cl = (ato_cluster_t*)realloc(cl, ccap * sizeof(ato_cluster_t)); }
      hk[h] = key; hi[h] = (uint32_t)nc;
      cl[nc].key = key; cl[nc].start = 0; cl[nc].count = 0;
      nc++;
    }
    cidx[i] = hi[h];
    cl[hi[h]].count++;
  }
  /* canonical cluster order: by key; remap through a rank table */
  uint32_t* rank = (uint32_t*)malloc((nc ? nc : 1) * 4);
  ato_cluster_t* sorted = (ato_cluster_t*)malloc((nc ? nc : 1) * sizeof(ato_cluster_t));
  for (size_t c = 0; c < nc; c++) { sorted[c] = cl[c]; sorted[c].start = (uint32_t)c; }  /* start temporarily = old index */
  qsort(sorted, nc, sizeof(ato_cluster_t), cluster_key_cmp);
  uint32_t off = 0;
  for (size_t c = 0; c < nc; c++) { rank[sorted[c].start] = (uint32_t)c; sorted[c].start = off; off += sorted[c].count; }
  uint32_t* fill = (uint32_t*)calloc(nc ? nc : 1, 4);
  for (size_t i = 0; i < n; i++) {
    uint32_t c = rank[cidx[i]];
    pts_out[sorted[c].start + fill[c]++] = kp[i].pt;
  }
  if (sort_points)
    for (size_t c = 0; c < nc; c++) qsort(pts_out + sorted[c].start, sorted[c].count, 4, u32_cmp);
  free(hk); free(hi); free(cl); free(cidx); free(rank); free(fill);
  *clusters_out = sorted;
  return nc;
}

/* ------------------------------------------------------------------------------------------- */
/* S5  quad fitting (A.5)                                                                       */
/* ------------------------------------------------------------------------------------------- */
typedef struct { double Mx, My, Mxx, Mxy, Myy, W; } lfp_t;

/* Diagnostic counters (not thread-safe; tools/cluster_stats.py only): [r] = clusters that left fit_quad for
 * reason r, [16 + r] = their points.  Reasons: 0 bbox, 1 border direction, 2 < 24 points after duplicate removal,
 * 3 fewer than 4 maxima, 4 no admissible corner choice, 5 total error, 6 final line mse, 7 degenerate
 * intersection, 8 area, 9 angles / winding, 10 accepted. */
long long ato_stats[32];
/* Diagnostic log of the partition bound (tools/early_exit_power.py): when ato_diag_cap > 0, every cluster that
 * reaches the moment prefixes appends {reason, points, ratio[8]}: ratio[k] = (sum of the groups' scatter
 * minimum eigenvalues minus the four largest) / (10 * (W_total + 4 * 362)) for groups of (run << k) kept points,
 * run = the lane run of the HIP kernel's size class.  A ratio above 1 proves that no corner choice is admissible. */
typedef struct { int reason, points; double ratio[8]; } ato_diag_rec_t;
ato_diag_rec_t* ato_diag_log;
int ato_diag_cap, ato_diag_n;
static __thread double g_diag_ratio[8];
static __thread int g_diag_have;
static void ato_diag_push(int reason, int points) {
  if (ato_diag_cap > 0 && g_diag_have && ato_diag_n < ato_diag_cap) {
    ato_diag_rec_t* r = &ato_diag_log[ato_diag_n++];
    r->reason = reason; r->points = points;
    memcpy(r->ratio, g_diag_ratio, sizeof(g_diag_ratio));
  }
}
#define ATO_STAT(r, n) do { ato_stats[(r)]++; ato_stats[16 + (r)] += (n); ato_diag_push((r), (n)); } while (0)
static __thread int g_qsm_reason;


static const float GAUSS7[7] = {0x1.6c0504p-7f, 0x1.152aaap-3f, 0x1.368b3p-1f, 1.0f,
                                0x1.368b3p-1f, 0x1.152aaap-3f, 0x1.6c0504p-7f};

static void fit_line(const lfp_t* lfps, int sz, int i0, int i1, double* lineparm, double* err, double* mse) {
  double Mx, My, Mxx, Mxy, Myy, W;
  int N;
  if (i0 < i1) {
    N = i1 - i0 + 1;
    Mx = lfps[i1].Mx; My = lfps[i1].My; Mxx = lfps[i1].Mxx; Mxy = lfps[i1].Mxy; Myy = lfps[i1].Myy; W = lfps[i1].W;
    if (i0 > 0) {
      Mx -= lfps[i0 - 1].Mx; My -= lfps[i0 - 1].My; Mxx -= lfps[i0 - 1].Mxx;
      Mxy -= lfps[i0 - 1].Mxy; Myy -= lfps[i0 - 1].Myy; W -= lfps[i0 - 1].W;
    }
  } else {
    Mx = lfps[sz - 1].Mx - lfps[i0 - 1].Mx; My = lfps[sz - 1].My - lfps[i0 - 1].My;
    Mxx = lfps[sz - 1].Mxx - lfps[i0 - 1].Mxx; Mxy = lfps[sz - 1].Mxy - lfps[i0 - 1].Mxy;
    Myy = lfps[sz - 1].Myy - lfps[i0 - 1].Myy; W = lfps[sz - 1].W - lfps[i0 - 1].W;
    Mx += lfps[i1].Mx; My += lfps[i1].My; Mxx += lfps[i1].Mxx; Mxy += lfps[i1].Mxy; Myy += lfps[i1].Myy; W += lfps[i1].W;
    N = sz - i0 + i1 + 1;
  }
  double Ex = Mx / W, Ey = My / W;
  double Cxx = Mxx / W - Ex * Ex, Cxy = Mxy / W - Ex * Ey, Cyy = Myy / W - Ey * Ey;
  double disc = (Cxx - Cyy) * (Cxx - Cyy) + 4 * Cxy * Cxy;
  float rootf = sqrtf((float)disc);
  double eig_small = 0.5 * (Cxx + Cyy - (double)rootf);
  if (lineparm) {
    lineparm[0] = Ex; lineparm[1] = Ey;
    double eig = 0.5 * (Cxx + Cyy + (double)rootf);
    double nx1 = Cxx - eig, ny1 = Cxy, M1 = nx1 * nx1 + ny1 * ny1;
    double nx2 = Cxy, ny2 = Cyy - eig, M2 = nx2 * nx2 + ny2 * ny2;
    double nx, ny, M;
    if (M1 > M2) { nx = nx1; ny = ny1; M = M1; } else { nx = nx2; ny = ny2; M = M2; }
    double length = (double)sqrtf((float)M);
    if (fabs(length) < 1e-12) { lineparm[2] = lineparm[3] = 0; }
    else { lineparm[2] = nx / length; lineparm[3] = ny / length; }
  }
  if (err) *err = N * eig_small;
  if (mse) *mse = eig_small;
}

static int dbl_desc(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x < y) - (x > y);
}

static int quad_segment_maxima(const ato_params_t* prm, const lfp_t* lfps, int sz, int indices[4]) {
  int ksz = sz / 12 < 20 ? sz / 12 : 20;
  g_qsm_reason = 3;
  if (ksz < 2) return 0;
  double* errs = (double*)malloc(sizeof(double) * sz);
  double* sm = (double*)malloc(sizeof(double) * sz);
  for (int i = 0; i < sz; i++) fit_line(lfps, sz, (i + sz - ksz) % sz, (i + ksz) % sz, NULL, &errs[i], NULL);
  for (int iy = 0; iy < sz; iy++) {
    double acc = 0;
    for (int i = 0; i < 7; i++) acc += errs[(iy + i - 3 + sz) % sz] * (double)GAUSS7[i];
    sm[iy] = acc;
  }
  memcpy(errs, sm, sizeof(double) * sz);
  free(sm);
  int* maxima = (int*)malloc(sizeof(int) * sz);
  double* maxima_errs = (double*)malloc(sizeof(double) * sz);
  int nmaxima = 0;
  for (int i = 0; i < sz; i++)
    if (errs[i] > errs[(i + 1) % sz] && errs[i] > errs[(i + sz - 1) % sz]) {
      maxima[nmaxima] = i; maxima_errs[nmaxima] = errs[i]; nmaxima++;
    }
  int ok = 0;
  if (nmaxima >= 4) {
    if (nmaxima > prm->max_nmaxima) {
      double* cp = (double*)malloc(sizeof(double) * nmaxima);
      memcpy(cp, maxima_errs, sizeof(double) * nmaxima);
      qsort(cp, nmaxima, sizeof(double), dbl_desc);
      double thresh = cp[prm->max_nmaxima];
      int out = 0;
      for (int in = 0; in < nmaxima; in++) {
        if (maxima_errs[in] <= thresh) continue;
        maxima[out++] = maxima[in];
      }
      nmaxima = out;
      free(cp);
    }
    int best[4] = {0, 0, 0, 0};
    double best_error = (double)HUGE_VALF;
    g_qsm_reason = 4;
    double err01, err12, err23, err30, mse01, mse12, mse23, mse30, p01[4], p12[4];
    double max_dot = prm->cos_critical_rad;
    for (int m0 = 0; m0 < nmaxima - 3; m0++) {
      int i0 = maxima[m0];
      for (int m1 = m0 + 1; m1 < nmaxima - 2; m1++) {
        int i1 = maxima[m1];
        fit_line(lfps, sz, i0, i1, p01, &err01, &mse01);
        if (mse01 > prm->max_line_fit_mse) continue;
        for (int m2 = m1 + 1; m2 < nmaxima - 1; m2++) {
          int i2 = maxima[m2];
          fit_line(lfps, sz, i1, i2, p12, &err12, &mse12);
          if (mse12 > prm->max_line_fit_mse) continue;
          double dot = p01[2] * p12[2] + p01[3] * p12[3];
          if (fabs(dot) > max_dot) continue;
          for (int m3 = m2 + 1; m3 < nmaxima; m3++) {
            int i3 = maxima[m3];
            fit_line(lfps, sz, i2, i3, NULL, &err23, &mse23);
            if (mse23 > prm->max_line_fit_mse) continue;
            fit_line(lfps, sz, i3, i0, NULL, &err30, &mse30);
            if (mse30 > prm->max_line_fit_mse) continue;
            double err = err01 + err12 + err23 + err30;
            if (err < best_error) { best_error = err; best[0] = i0; best[1] = i1; best[2] = i2; best[3] = i3; }
          }
        }
      }
    }
    if (best_error != (double)HUGE_VALF) {
      for (int i = 0; i < 4; i++) indices[i] = best[i];
      g_qsm_reason = 5;
      if (best_error / sz < prm->max_line_fit_mse) ok = 1;
    }
  }
  free(errs); free(maxima); free(maxima_errs);
  return ok;
}

static inline uint32_t float_sortable(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

static int u64_cmp(const void* a, const void* b) {
  uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return (x > y) - (x < y);
}

/* ATO_VAR_FAST_PATHS: the same ascending order of the 64-bit keys as qsort(u64_cmp), by a byte-wise LSD radix sort (keys are
 * unique up to exact duplicates, so stability does not matter).  Small arrays: insertion sort. */
static void sort_keys_fast(uint64_t* k, int n) {
  if (n <= 48) {
    for (int i = 1; i < n; i++) { uint64_t v = k[i]; int j = i - 1; while (j >= 0 && k[j] > v) { k[j + 1] = k[j]; j--; } k[j + 1] = v; }
    return;
  }
  uint64_t* tmp = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
  uint64_t* a = k; uint64_t* b = tmp;
  for (int pass = 0; pass < 8; pass++) {
    const int sh = pass * 8;
    uint32_t cnt[257];
    memset(cnt, 0, sizeof(cnt));
    for (int i = 0; i < n; i++) cnt[((a[i] >> sh) & 0xFF) + 1]++;
    if (cnt[((a[0] >> sh) & 0xFF) + 1] == (uint32_t)n) continue;   /* every key has the same byte here */
    for (int j = 0; j < 256; j++) cnt[j + 1] += cnt[j];
    for (int i = 0; i < n; i++) b[cnt[(a[i] >> sh) & 0xFF]++] = a[i];
    uint64_t* t = a; a = b; b = t;
  }
  if (a != k) memcpy(k, a, sizeof(uint64_t) * (size_t)n);
  free(tmp);
}

/* exact fixed-point view of a double >= 1: value * 2^52 as a 128-bit integer (terms < 2^36 fit easily) */
static unsigned __int128 exact_to_fixed(double t) {
  uint64_t bits;
  memcpy(&bits, &t, 8);
  int exp = (int)((bits >> 52) & 0x7FF);
  uint64_t m = (bits & ((1ULL << 52) - 1)) | (1ULL << 52);
  int shift = exp - 1023;            /* t = m * 2^(exp-1075); fixed = t * 2^52 = m * 2^(exp-1023) */
  if (shift < 0) return (unsigned __int128)(m >> (-shift > 63 ? 63 : -shift));   /* t < 1 never occurs */
  return (unsigned __int128)m << shift;
}
/* nearest-even rounding of fixed / 2^52 to double */
static double exact_from_fixed(unsigned __int128 v) {
  uint64_t hi = (uint64_t)(v >> 64), lo = (uint64_t)v;
  if (hi == 0 && lo < (1ULL << 53)) return (double)(int64_t)lo * 0x1p-52;
  int p = hi ? 127 - __builtin_clzll(hi) : 63 - __builtin_clzll(lo);
  int r = p - 52;
  uint64_t mant = (uint64_t)(v >> r);
  int halfbit = (int)((v >> (r - 1)) & 1);
  int sticky = (v & ((((unsigned __int128)1) << (r - 1)) - 1)) != 0;
  if (halfbit && (sticky || (mant & 1))) mant++;
  if (mant >> 53) { mant >>= 1; r++; }
  uint64_t bits = ((uint64_t)(1023 + r) << 52) | (mant & ((1ULL << 52) - 1));
  double out;
  memcpy(&out, &bits, 8);
  return out;
}

/* fit one quad to the cluster pts[0..sz) (packed points); gray = working image (w x h). */
static int fit_quad(const ato_params_t* prm, const uint8_t* gray, int w, int h, const uint32_t* pts, int sz,
                    int tag_width, int normal_border, int reversed_border, ato_quad_t* quad) {
  if (sz < 24) return 0;
  int xmin = 1 << 30, xmax = -1, ymin = 1 << 30, ymax = -1;
  long long sxg = 0, sgx = 0, sgy = 0;
  for (int i = 0; i < sz; i++) {
    int x, y, gx, gy;
    unpack_point(pts[i], &x, &y, &gx, &gy);
    if (x < xmin) xmin = x;
    if (x > xmax) xmax = x;
    if (y < ymin) ymin = y;
    if (y > ymax) ymax = y;
    sxg += (long long)x * gx + (long long)y * gy;
    sgx += gx; sgy += gy;
  }
  const int sz_in = sz;
  if ((xmax - xmin) * (ymax - ymin) < tag_width) { ATO_STAT(0, sz_in); return 0; }
  double cxd = (xmin + xmax) * 0.5 + 0.05118, cyd = (ymin + ymax) * 0.5 + -0.028581;
  /* CANONICAL: dot = sum(dx*gx + dy*gy) evaluated exactly (upstream: float accumulation in hash order) */
  double dot = (double)sxg - cxd * (double)sgx - cyd * (double)sgy;
  if (prm->variant & ATO_VAR_FLOAT_DOT) {   /* upstream: float accumulation in the cluster's point order */
    float fcx = (float)cxd, fcy = (float)cyd, fdot = 0;
    for (int i = 0; i < sz; i++) {
      int x, y, gx, gy;
      unpack_point(pts[i], &x, &y, &gx, &gy);
      float dx = (float)x - fcx, dy = (float)y - fcy;
      fdot += dx * (float)gx + dy * (float)gy;
    }
    dot = (double)fdot;
  }
  quad->reversed_border = dot < 0;
  if (!reversed_border && quad->reversed_border) { ATO_STAT(1, sz_in); return 0; }
  if (!normal_border && !quad->reversed_border) { ATO_STAT(1, sz_in); return 0; }

  /* slope key: quadrant band + dy/dx after rotating into the first quadrant (float, as upstream) */
  float cx = (float)cxd, cy = (float)cyd;
  uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * sz);
  for (int i = 0; i < sz; i++) {
    int x, y, gx, gy;
    unpack_point(pts[i], &x, &y, &gx, &gy);
    float dx = (float)x - cx, dy = (float)y - cy;
    float quadrant;
    if (dy > 0) quadrant = (dx > 0) ? 65536.0f : 131072.0f;
    else quadrant = (dx > 0) ? 0.0f : -65536.0f;
    if (dy < 0) { dy = -dy; dx = -dx; }
    if (dx < 0) { float tmp = dx; dx = dy; dy = -tmp; }
    float slope = quadrant + dy / dx;
    /* CANONICAL total order: (slope, y, x, gx, gy) */
    keys[i] = ((uint64_t)float_sortable(slope) << 32) | ((uint64_t)y << 18) | ((uint64_t)x << 4) | (pts[i] & 15u);
  }
  if (prm->variant & ATO_VAR_FAST_PATHS) sort_keys_fast(keys, sz);
  else qsort(keys, sz, sizeof(uint64_t), u64_cmp);
  double diag_sector[3] = {0, 0, 0};
  if (ato_diag_cap > 0) {
    /* angular-sector variant of the relaxed feasibility test, computable BEFORE the sort: sectors are intervals of the
     * slope key (contiguous in sorted order); duplicates are not removed, so a point on an odd-odd half-pixel location
     * (the only ones that can occur twice) enters the scatter with half its weight and the weight bound in full */
    for (int v = 0; v < 3; v++) {
      const int SPQ = 8 << v, m = 4 * SPQ;   /* sectors per quadrant: 32 / 64 / 128 sectors */
      const int sub = 1;
      float thrs[128];
      static const float QB[4] = {-65536.0f, 0.0f, 65536.0f, 131072.0f};
      for (int q = 0; q < 4; q++)
        for (int k = 0; k < SPQ; k++) thrs[q * SPQ + k] = QB[q] + (float)tan(k * (M_PI / 2) / SPQ);
      double (*B)[7] = calloc((size_t)m + 1, sizeof(double) * 7);   /* bins, then prefix */
      for (int i = 0; i < sz; i++) {
        uint32_t fs = (uint32_t)(keys[i] >> 32);
        uint32_t fb = (fs & 0x80000000u) ? (fs & 0x7FFFFFFFu) : ~fs;
        float slope; memcpy(&slope, &fb, 4);
        int sec = 0;
        for (int j = 1; j < m; j++) if (slope >= thrs[j]) sec = j;
        int px = (int)((keys[i] >> 4) & 0x3FFF), py = (int)((keys[i] >> 18) & 0x3FFF);
        double x = px * .5 + 0.5, y = py * .5 + 0.5;
        int ix = (int)x, iy = (int)y;
        double Wt = 1;
        if (ix > 0 && ix + 1 < w && iy > 0 && iy + 1 < h) {
          int grad_x = gray[iy * w + ix + 1] - gray[iy * w + ix - 1];
          int grad_y = gray[(iy + 1) * w + ix] - gray[(iy - 1) * w + ix];
          Wt = sqrt((double)(grad_x * grad_x + grad_y * grad_y)) + 1;
        }
        double wl = ((px & 1) && (py & 1)) ? 0.5 * Wt : Wt;
        double* b = B[sec + 1];
        if (i % sub == 0) { b[0] += wl * x; b[1] += wl * y; b[2] += wl * x * x; b[3] += wl * x * y; b[4] += wl * y * y; b[5] += wl; }
        b[6] += Wt;
      }
      for (int j = 1; j <= m; j++) for (int t = 0; t < 7; t++) B[j][t] += B[j - 1][t];
      const double T = 10.5;
      unsigned char* ok = calloc((size_t)m * m, 1);
      unsigned char* okw = calloc((size_t)m * m, 1);
      for (int a = 0; a < m; a++)
        for (int b = a; b < m; b++) {
          int o = 1, ow = 1;
          if (b > a + 1) {
            double M[6]; for (int t = 0; t < 6; t++) M[t] = B[b][t] - B[a + 1][t];
            double Wc = B[b + 1][6] - B[a][6];
            if (M[5] > 0.25) {
              double Sxx = M[2] - M[0] * M[0] / M[5], Sxy = M[3] - M[0] * M[1] / M[5], Syy = M[4] - M[1] * M[1] / M[5];
              double q = 0.5 * (Sxx + Syy - sqrt((Sxx - Syy) * (Sxx - Syy) + 4 * Sxy * Sxy));
              o = !(q > T * Wc);
            }
          }
          {
            double M[6]; for (int t = 0; t < 6; t++) M[t] = (B[m][t] - B[b + 1][t]) + B[a][t];
            double Wc = (B[m][6] - B[b][6]) + B[a + 1][6];
            if (M[5] > 0.25) {
              double Sxx = M[2] - M[0] * M[0] / M[5], Sxy = M[3] - M[0] * M[1] / M[5], Syy = M[4] - M[1] * M[1] / M[5];
              double q = 0.5 * (Sxx + Syy - sqrt((Sxx - Syy) * (Sxx - Syy) + 4 * Sxy * Sxy));
              ow = !(q > T * Wc);
            }
          }
          ok[a * m + b] = (unsigned char)o; okw[b * m + a] = (unsigned char)ow;
        }
      int feasible = 0;
      unsigned char* r1 = malloc(m), *r2 = malloc(m), *r3 = malloc(m);
      for (int a = 0; a < m && !feasible; a++) {
        memset(r1, 0, m); memset(r2, 0, m); memset(r3, 0, m);
        for (int b = a; b < m; b++) if (ok[a * m + b]) r1[b] = 1;
        for (int b = a; b < m; b++) if (r1[b]) for (int c = b; c < m; c++) if (ok[b * m + c]) r2[c] = 1;
        for (int c = a; c < m; c++) if (r2[c]) for (int d = c; d < m; d++) if (ok[c * m + d]) r3[d] = 1;
        for (int d = a; d < m; d++) if (r3[d] && okw[d * m + a]) { feasible = 1; break; }
      }
      free(r1); free(r2); free(r3); free(ok); free(okw); free(B);
      diag_sector[v] = feasible ? 0.0 : 2.0;
    }
  }
  /* remove duplicate points (same half-pixel location, a by-product of the segmentation); the
   * gradient-direction test above has already seen all of them (upstream fit_quad) */
  {
    int outpos = 1;
    for (int i = 1; i < sz; i++)
      if ((keys[i] >> 4) != (keys[i - 1] >> 4)) keys[outpos++] = keys[i];   /* bits 4..31 = (y,x) */
    sz = outpos;
  }
  if (sz < 24) { free(keys); ATO_STAT(2, sz_in); return 0; }

  /* cumulative weighted moments (compute_lfps).  Per-point terms are formed exactly as upstream does
   * (W*x, W*y, (W*x)*x, (W*x)*y, (W*y)*y, W in double).  CANONICAL: the running sums are the EXACT sums
   * of those doubles, rounded once to nearest-even (upstream adds them sequentially in double, which
   * differs by accumulated rounding noise and makes the value depend on the summation order).  Every
   * term is >= 1 and < 2^36, so 52 fractional bits in a 128-bit integer hold each term exactly. */
  lfp_t* lfps = (lfp_t*)malloc(sizeof(lfp_t) * sz);
  unsigned __int128 acc[6] = {0, 0, 0, 0, 0, 0};
  double seq[6] = {0, 0, 0, 0, 0, 0};   /* ATO_VAR_SEQ_MOMENTS: upstream's running double sums */
  for (int i = 0; i < sz; i++) {
    int px = (int)((keys[i] >> 4) & 0x3FFF), py = (int)((keys[i] >> 18) & 0x3FFF);
    double x = px * .5 + 0.5, y = py * .5 + 0.5;
    int ix = (int)x, iy = (int)y;
    double W = 1;
    if (ix > 0 && ix + 1 < w && iy > 0 && iy + 1 < h) {
      int grad_x = gray[iy * w + ix + 1] - gray[iy * w + ix - 1];
      int grad_y = gray[(iy + 1) * w + ix] - gray[(iy - 1) * w + ix];
      W = sqrt((double)(grad_x * grad_x + grad_y * grad_y)) + 1;
    }
    double t[6] = {W * x, W * y, W * x * x, W * x * y, W * y * y, W};
    double r[6];
    if ((prm->variant & (ATO_VAR_FAST_PATHS | ATO_VAR_SEQ_MOMENTS)) == (ATO_VAR_FAST_PATHS | ATO_VAR_SEQ_MOMENTS)) {
      for (int j = 0; j < 6; j++) { seq[j] += t[j]; r[j] = seq[j]; }   /* upstream's compute_lfps, nothing else */
    } else
    for (int j = 0; j < 6; j++) {
      acc[j] += exact_to_fixed(t[j]); r[j] = exact_from_fixed(acc[j]);
      seq[j] += t[j];
      if (prm->variant & ATO_VAR_SEQ_MOMENTS) r[j] = seq[j];
    }
    lfps[i].Mx = r[0]; lfps[i].My = r[1]; lfps[i].Mxx = r[2]; lfps[i].Mxy = r[3]; lfps[i].Myy = r[4]; lfps[i].W = r[5];
  }
  free(keys);
  g_diag_have = 0;
  if (ato_diag_cap > 0) {
    /* relaxed feasibility over m groups of consecutive kept points: cut groups a <= b <= c <= d; an arc between two
     * cut groups is possible only if the scatter of the whole groups strictly between them has
     * lambda_min <= T * (weight of the groups from cut to cut inclusive) */
    static const int MS[8] = {16, 24, 32, 48, 64, 96, 128, 256};
    for (int k = 0; k < 8; k++) {
      int m = MS[k];
      int G = (sz + m - 1) / m;
      m = (sz + G - 1) / G;
      double (*P)[6] = malloc(sizeof(double) * 6 * (m + 1));   /* prefix over groups: P[j] = sums of groups < j */
      for (int j = 0; j <= m; j++) {
        int e = j * G < sz ? j * G : sz;
        if (e == 0) { for (int t = 0; t < 6; t++) P[j][t] = 0; }
        else { P[j][0] = lfps[e-1].Mx; P[j][1] = lfps[e-1].My; P[j][2] = lfps[e-1].Mxx; P[j][3] = lfps[e-1].Mxy; P[j][4] = lfps[e-1].Myy; P[j][5] = lfps[e-1].W; }
      }
      const double T = 10.5;
      unsigned char* ok = calloc((size_t)m * m, 1);   /* ok[a*m+b], a <= b: forward arc; ok[b*m+a] (b>=a) stored at [d*m+a] with d >= a as wrap in okw */
      unsigned char* okw = calloc((size_t)m * m, 1);
      for (int a = 0; a < m; a++)
        for (int b = a; b < m; b++) {
          /* forward: inner groups a+1..b-1 */
          int o = 1;
          if (b > a + 1) {
            double M[6]; for (int t = 0; t < 6; t++) M[t] = P[b][t] - P[a + 1][t];
            double Wc = P[b + 1][5] - P[a][5];
            double Sxx = M[2] - M[0] * M[0] / M[5], Sxy = M[3] - M[0] * M[1] / M[5], Syy = M[4] - M[1] * M[1] / M[5];
            double q = 0.5 * (Sxx + Syy - sqrt((Sxx - Syy) * (Sxx - Syy) + 4 * Sxy * Sxy));
            o = q <= T * Wc;
          }
          ok[a * m + b] = (unsigned char)o;
          /* wrap d=b -> a: inner groups b+1..m-1 and 0..a-1 */
          int ow = 1;
          int ninner = (m - 1 - b) + a;
          if (ninner > 0) {
            double M[6]; for (int t = 0; t < 6; t++) M[t] = (P[m][t] - P[b + 1][t]) + P[a][t];
            double Wc = (P[m][5] - P[b][5]) + P[a + 1][5];
            double Sxx = M[2] - M[0] * M[0] / M[5], Sxy = M[3] - M[0] * M[1] / M[5], Syy = M[4] - M[1] * M[1] / M[5];
            double q = 0.5 * (Sxx + Syy - sqrt((Sxx - Syy) * (Sxx - Syy) + 4 * Sxy * Sxy));
            ow = q <= T * Wc;
          }
          okw[b * m + a] = (unsigned char)ow;
        }
      int feasible = 0;
      unsigned char* r1 = malloc(m), *r2 = malloc(m), *r3 = malloc(m);
      for (int a = 0; a < m && !feasible; a++) {
        memset(r1, 0, m); memset(r2, 0, m); memset(r3, 0, m);
        for (int b = a; b < m; b++) if (ok[a * m + b]) r1[b] = 1;
        for (int b = a; b < m; b++) if (r1[b]) for (int c = b; c < m; c++) if (ok[b * m + c]) r2[c] = 1;
        for (int c = a; c < m; c++) if (r2[c]) for (int d = c; d < m; d++) if (ok[c * m + d]) r3[d] = 1;
        for (int d = a; d < m; d++) if (r3[d] && okw[d * m + a]) { feasible = 1; break; }
      }
      free(r1); free(r2); free(r3); free(ok); free(okw); free(P);
      g_diag_ratio[k] = feasible ? 0.0 : 2.0;
    }
    g_diag_ratio[0] = diag_sector[0];   /* slots 0, 1: the pre-sort sector test with 32 / 64 sectors */
    g_diag_ratio[1] = diag_sector[1];
    g_diag_ratio[2] = diag_sector[2];
    g_diag_have = 1;
  }

  int res = 0, indices[4];
  double lines[4][4];
  if (!quad_segment_maxima(prm, lfps, sz, indices)) { ATO_STAT(g_qsm_reason, sz_in); goto finish; }
  for (int i = 0; i < 4; i++) {
    double mse;
    fit_line(lfps, sz, indices[i], indices[(i + 1) & 3], lines[i], NULL, &mse);
    if (mse > prm->max_line_fit_mse) { ATO_STAT(6, sz_in); goto finish; }
  }
  for (int i = 0; i < 4; i++) {
    double A00 = lines[i][3], A01 = -lines[(i + 1) & 3][3];
    double A10 = -lines[i][2], A11 = lines[(i + 1) & 3][2];
    double B0 = -lines[i][0] + lines[(i + 1) & 3][0];
    double B1 = -lines[i][1] + lines[(i + 1) & 3][1];
    double det = A00 * A11 - A10 * A01;
    if (fabs(det) < 0.001) { ATO_STAT(7, sz_in); goto finish; }
    double W00 = A11 / det, W01 = -A01 / det;
    double L0 = W00 * B0 + W01 * B1;
    quad->p[i][0] = (float)(lines[i][0] + L0 * A00);
    quad->p[i][1] = (float)(lines[i][1] + L0 * A10);
  }
  {
    /* area of the two triangles (Heron) */
    double area = 0, length[3], p;
    for (int i = 0; i < 3; i++) {
      int a = i, b = (i + 1) % 3;
      /* upstream: sq(quad->p[b][0] - quad->p[a][0]) with float p[][] -- the difference is a FLOAT operation, widened for sq() */
      double ddx = (double)(quad->p[b][0] - quad->p[a][0]), ddy = (double)(quad->p[b][1] - quad->p[a][1]);
      length[i] = sqrt(ddx * ddx + ddy * ddy);
    }
    p = (length[0] + length[1] + length[2]) / 2;
    area += sqrt(p * (p - length[0]) * (p - length[1]) * (p - length[2]));
    static const int idxs[4] = {2, 3, 0, 2};
    for (int i = 0; i < 3; i++) {
      int a = idxs[i], b = idxs[i + 1];
      /* upstream: sq(quad->p[b][0] - quad->p[a][0]) with float p[][] -- the difference is a FLOAT operation, widened for sq() */
      double ddx = (double)(quad->p[b][0] - quad->p[a][0]), ddy = (double)(quad->p[b][1] - quad->p[a][1]);
      length[i] = sqrt(ddx * ddx + ddy * ddy);
    }
    p = (length[0] + length[1] + length[2]) / 2;
    area += sqrt(p * (p - length[0]) * (p - length[1]) * (p - length[2]));
    if (area < 0.95 * tag_width * tag_width) { ATO_STAT(8, sz_in); goto finish; }
  }
  for (int i = 0; i < 4; i++) {
    int i0 = i, i1 = (i + 1) & 3, i2 = (i + 2) & 3;
    /* upstream: double dx1 = quad->p[i1][0] - quad->p[i0][0]; -- float differences, then widened */
    double dx1 = (double)(quad->p[i1][0] - quad->p[i0][0]), dy1 = (double)(quad->p[i1][1] - quad->p[i0][1]);
    double dx2 = (double)(quad->p[i2][0] - quad->p[i1][0]), dy2 = (double)(quad->p[i2][1] - quad->p[i1][1]);
    double cos_dtheta = (dx1 * dx2 + dy1 * dy2) / sqrt((dx1 * dx1 + dy1 * dy1) * (dx2 * dx2 + dy2 * dy2));
    const double cos_crit = prm->cos_critical_rad;
    if ((cos_dtheta > cos_crit || cos_dtheta < -cos_crit) || dx1 * dy2 < dy1 * dx2) { ATO_STAT(9, sz_in); goto finish; }
  }
  res = 1;
  ATO_STAT(10, sz_in);
finish:
  free(lfps);
  return res;
}

/* ------------------------------------------------------------------------------------------- */
/* S6  edge refinement on the full-resolution image (A.6)                                       */
/*     Deviation: the refitted line normal is taken from the covariance eigenvector (the fit_line */
/*     formulation upstream's own TODO names) instead of 0.5*atan2f/cosf/sinf, so that no          */
/*     transcendental library call sits between the CPU and HIP results.                          */
/* ------------------------------------------------------------------------------------------- */
static void refine_edges(const ato_params_t* prm, const uint8_t* im, int w, int h, int pitch, ato_quad_t* quad) {
  double lines[4][4];
  for (int edge = 0; edge < 4; edge++) {
    int a = edge, b = (edge + 1) & 3;
    /* upstream: double nx = quad->p[b][1] - quad->p[a][1]; double ny = -quad->p[b][0] + quad->p[a][0]; -- float operations */
    double nx = (double)(quad->p[b][1] - quad->p[a][1]);
    double ny = (double)(-quad->p[b][0] + quad->p[a][0]);
    double mag = sqrt(nx * nx + ny * ny);
    nx /= mag; ny /= mag;
    if (quad->reversed_border) { nx = -nx; ny = -ny; }
    int nsamples = (int)(mag / 8);
    if (nsamples < 16) nsamples = 16;
    double Mx = 0, My = 0, Mxx = 0, Mxy = 0, Myy = 0, N = 0;
    for (int s = 0; s < nsamples; s++) {
      double alpha = (1.0 + s) / (nsamples + 1);
      double x0 = alpha * (double)quad->p[a][0] + (1 - alpha) * (double)quad->p[b][0];
      double y0 = alpha * (double)quad->p[a][1] + (1 - alpha) * (double)quad->p[b][1];
      double Mn = 0, Mcount = 0;
      double range = prm->decimate + 1;
      int steps = (int)(2 * range * 4) + 1;   /* n = -range + 0.25*k, exact in binary */
      for (int k = 0; k < steps; k++) {
        double n = -range + 0.25 * k;
        double grange = 1;
        int x1 = (int)(x0 + (n + grange) * nx);
        int y1 = (int)(y0 + (n + grange) * ny);
        if (x1 < 0 || x1 >= w || y1 < 0 || y1 >= h) continue;
        int x2 = (int)(x0 + (n - grange) * nx);
        int y2 = (int)(y0 + (n - grange) * ny);
        if (x2 < 0 || x2 >= w || y2 < 0 || y2 >= h) continue;
        int g1 = im[(size_t)y1 * pitch + x1];
        int g2 = im[(size_t)y2 * pitch + x2];
        if (g1 < g2) continue;
        double weight = (double)((g2 - g1) * (g2 - g1));
        Mn += weight * n;
        Mcount += weight;
      }
      if (Mcount == 0) continue;
      double n0 = Mn / Mcount;
      double bestx = x0 + n0 * nx, besty = y0 + n0 * ny;
      Mx += bestx; My += besty; Mxx += bestx * bestx; Mxy += bestx * besty; Myy += besty * besty; N++;
    }
    double Ex = Mx / N, Ey = My / N;
    double Cxx = Mxx / N - Ex * Ex, Cxy = Mxy / N - Ex * Ey, Cyy = Myy / N - Ey * Ey;
    double disc = (Cxx - Cyy) * (Cxx - Cyy) + 4 * Cxy * Cxy;
    double eig = 0.5 * (Cxx + Cyy + (double)sqrtf((float)disc));
    double nx1 = Cxx - eig, ny1 = Cxy, M1 = nx1 * nx1 + ny1 * ny1;
    double nx2 = Cxy, ny2 = Cyy - eig, M2 = nx2 * nx2 + ny2 * ny2;
    double M;
    if (M1 > M2) { nx = nx1; ny = ny1; M = M1; } else { nx = nx2; ny = ny2; M = M2; }
    double length = (double)sqrtf((float)M);
    if (fabs(length) < 1e-12) { nx = 0; ny = 0; } else { nx = nx / length; ny = ny / length; }
    if (prm->variant & ATO_VAR_ATAN_NORMAL) {   /* upstream refine_edges */
      float normal_theta = 0.5f * atan2f((float)(-2 * Cxy), (float)(Cyy - Cxx));
      nx = (double)cosf(normal_theta);
      ny = (double)sinf(normal_theta);
    }
    lines[edge][0] = Ex; lines[edge][1] = Ey; lines[edge][2] = nx; lines[edge][3] = ny;
  }
  for (int i = 0; i < 4; i++) {
    double A00 = lines[i][3], A01 = -lines[(i + 1) & 3][3];
    double A10 = -lines[i][2], A11 = lines[(i + 1) & 3][2];
    double B0 = -lines[i][0] + lines[(i + 1) & 3][0];
    double B1 = -lines[i][1] + lines[(i + 1) & 3][1];
    double det = A00 * A11 - A10 * A01;
    if (fabs(det) > 0.001) {
      double W00 = A11 / det, W01 = -A01 / det;
      double L0 = W00 * B0 + W01 * B1;
      quad->p[i][0] = (float)(lines[i][0] + L0 * A00);
      quad->p[i][1] = (float)(lines[i][1] + L0 * A10);
    }
  }
}

/* ------------------------------------------------------------------------------------------- */
/* S6b homography by 8x9 Gaussian elimination (A.7)                                             */
/* ------------------------------------------------------------------------------------------- */
static int homography_compute2(const double c[4][4], double H[9]) {
  double A[72];
  for (int i = 0; i < 4; i++) {
    double x = c[i][0], y = c[i][1], u = c[i][2], v = c[i][3];
    double* r0 = &A[(2 * i) * 9];
    double* r1 = &A[(2 * i + 1) * 9];
    r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -x * u; r0[7] = -y * u; r0[8] = u;
    r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -x * v; r1[7] = -y * v; r1[8] = v;
  }
  for (int col = 0; col < 8; col++) {
    double max_val = 0;
    int max_idx = -1;
    for (int row = col; row < 8; row++) {
      double val = fabs(A[row * 9 + col]);
      if (val > max_val) { max_val = val; max_idx = row; }
    }
    if (max_val < 1e-10) return -1;
    if (max_idx != col)
      for (int i = col; i < 9; i++) { double t = A[col * 9 + i]; A[col * 9 + i] = A[max_idx * 9 + i]; A[max_idx * 9 + i] = t; }
    for (int i = col + 1; i < 8; i++) {
      double f = A[i * 9 + col] / A[col * 9 + col];
      A[i * 9 + col] = 0;
      for (int j = col + 1; j < 9; j++) A[i * 9 + j] -= f * A[col * 9 + j];
    }
  }
  for (int col = 7; col >= 0; col--) {
    double sum = 0;
    for (int i = col + 1; i < 8; i++) sum += A[col * 9 + i] * A[i * 9 + 8];
    A[col * 9 + 8] = (A[col * 9 + 8] - sum) / A[col * 9 + col];
  }
  for (int i = 0; i < 8; i++) H[i] = A[i * 9 + 8];
  H[8] = 1;
  return 0;
}

static inline void homography_project(const double H[9], double x, double y, double* ox, double* oy) {
  double xx = H[0] * x + H[1] * y + H[2];
  double yy = H[3] * x + H[4] * y + H[5];
  double zz = H[6] * x + H[7] * y + H[8];
  *ox = xx / zz;
  *oy = yy / zz;
}

/* ------------------------------------------------------------------------------------------- */
/* S7  decode (A.8)                                                                             */
/* ------------------------------------------------------------------------------------------- */
typedef struct { double A[3][3], B[3], C[3]; } graymodel_t;

static void graymodel_add(graymodel_t* gm, double x, double y, double gray) {
  gm->A[0][0] += x * x; gm->A[0][1] += x * y; gm->A[0][2] += x;
  gm->A[1][1] += y * y; gm->A[1][2] += y; gm->A[2][2] += 1;
  gm->B[0] += x * gray; gm->B[1] += y * gray; gm->B[2] += gray;
}
/* 3x3 symmetric solve via Cholesky, upper-triangle input */
static void graymodel_solve(graymodel_t* gm) {
  double A0 = gm->A[0][0], A1 = gm->A[0][1], A2 = gm->A[0][2], A4 = gm->A[1][1], A5 = gm->A[1][2], A8 = gm->A[2][2];
  double L0 = sqrt(A0);
  double L3 = A1 / L0;
  double L6 = A2 / L0;
  double L4 = sqrt(A4 - L3 * L3);
  double L7 = (A5 - L3 * L6) / L4;
  double L8 = sqrt(A8 - L6 * L6 - L7 * L7);
  double M0 = 1 / L0;
  double M3 = -L3 * M0 / L4;
  double M4 = 1 / L4;
  double M6 = (-L6 * M0 - L7 * M3) / L8;
  double M7 = -L7 * M4 / L8;
  double M8 = 1 / L8;
  double t0 = M0 * gm->B[0];
  double t1 = M3 * gm->B[0] + M4 * gm->B[1];
  double t2 = M6 * gm->B[0] + M7 * gm->B[1] + M8 * gm->B[2];
  gm->C[0] = M0 * t0 + M3 * t1 + M6 * t2;
  gm->C[1] = M4 * t1 + M7 * t2;
  gm->C[2] = M8 * t2;
}
static inline double graymodel_interpolate(const graymodel_t* gm, double x, double y) {
  return gm->C[0] * x + gm->C[1] * y + gm->C[2];
}

static double value_for_pixel(const uint8_t* im, int w, int h, int pitch, double px, double py) {
  int x1 = (int)floor(px - 0.5), x2 = (int)ceil(px - 0.5);
  double x = px - 0.5 - x1;
  int y1 = (int)floor(py - 0.5), y2 = (int)ceil(py - 0.5);
  double y = py - 0.5 - y1;
  if (x1 < 0 || x2 >= w || y1 < 0 || y2 >= h) return -1;
  return im[(size_t)y1 * pitch + x1] * (1 - x) * (1 - y) + im[(size_t)y1 * pitch + x2] * x * (1 - y) +
         im[(size_t)y2 * pitch + x1] * (1 - x) * y + im[(size_t)y2 * pitch + x2] * x * y;
}

/* pattern rotation used by the code lookup (see rot_source) */
static uint64_t rotate90(const ato_family_t* f, uint64_t w) {
  uint64_t o = 0;
  int nb = (int)f->nbits;
  for (int i = 0; i < nb; i++) {
    int j = rot_source(f, i);
    if (j >= 0 && ((w >> (nb - 1 - j)) & 1)) o |= 1ULL << (nb - 1 - i);
  }
  return o;
}

/* quick_decode_codeword semantics: first rotation r in 0..3 for which some code is within
 * max_hamming bits (unique by the family's minimum distance). */
/* ATO_VAR_FAST_PATHS: AprilRobotics' quick_decode -- an open-addressing table of every code word and every word within one or two
 * bit errors of one ({code, id, hamming}); a lookup per rotation instead of a scan over the family.  Built once per family
 * (keyed by the code table's address and size), kept for the life of the process; families whose table would not fit the
 * published layout's assumptions (more than 2 correctable bits asked for) fall back to the scan. */
typedef struct { uint64_t code; uint16_t id; uint8_t hamming, used; } qd_entry_t;
typedef struct { const uint64_t* codes; uint32_t ncodes, nbits; size_t cap; qd_entry_t* tab; } qd_table_t;
static qd_table_t g_qd[8];
static int g_qd_n;
static pthread_mutex_t g_qd_lock = PTHREAD_MUTEX_INITIALIZER;
static void qd_add(qd_table_t* t, uint64_t code, uint32_t id, int hamming) {
  size_t h = (size_t)((code * 0x9E3779B97F4A7C15ULL) >> 24) & (t->cap - 1);
  while (t->tab[h].used) {
    if (t->tab[h].code == code) { if (hamming < t->tab[h].hamming) { t->tab[h].id = (uint16_t)id; t->tab[h].hamming = (uint8_t)hamming; } return; }
    h = (h + 1) & (t->cap - 1);
  }
  t->tab[h].code = code; t->tab[h].id = (uint16_t)id; t->tab[h].hamming = (uint8_t)hamming; t->tab[h].used = 1;
}
static const qd_table_t* qd_get(const ato_family_t* fam) {
  pthread_mutex_lock(&g_qd_lock);
  for (int i = 0; i < g_qd_n; i++)
    if (g_qd[i].codes == fam->codes && g_qd[i].ncodes == fam->ncodes && g_qd[i].nbits == fam->nbits) { pthread_mutex_unlock(&g_qd_lock); return &g_qd[i]; }
  if (g_qd_n == 8 || fam->ncodes > 65535) { pthread_mutex_unlock(&g_qd_lock); return NULL; }
  qd_table_t* t = &g_qd[g_qd_n];
  const size_t entries = (size_t)fam->ncodes * (1 + fam->nbits + (size_t)fam->nbits * (fam->nbits - 1) / 2);
  t->cap = 1024;
  while (t->cap < entries * 3) t->cap <<= 1;
  t->tab = (qd_entry_t*)calloc(t->cap, sizeof(qd_entry_t));
  t->codes = fam->codes; t->ncodes = fam->ncodes; t->nbits = fam->nbits;
  for (uint32_t i = 0; i < fam->ncodes; i++) {
    const uint64_t c = fam->codes[i];
    qd_add(t, c, i, 0);
    for (uint32_t a = 0; a < fam->nbits; a++) {
      qd_add(t, c ^ (1ULL << a), i, 1);
      for (uint32_t b = 0; b < a; b++) qd_add(t, c ^ (1ULL << a) ^ (1ULL << b), i, 2);
    }
  }
  g_qd_n++;
  pthread_mutex_unlock(&g_qd_lock);
  return t;
}
static int decode_codeword_scan(const ato_family_t* fam, uint64_t rcode, int max_hamming, int* id, int* hamming, int* rotation);
static int decode_codeword_fast(const ato_family_t* fam, uint64_t rcode, int max_hamming, int* id, int* hamming, int* rotation) {
  const qd_table_t* t = max_hamming <= 2 ? qd_get(fam) : NULL;
  if (!t) return decode_codeword_scan(fam, rcode, max_hamming, id, hamming, rotation);
  for (int r = 0; r < 4; r++) {
    size_t h = (size_t)((rcode * 0x9E3779B97F4A7C15ULL) >> 24) & (t->cap - 1);
    while (t->tab[h].used) {
      if (t->tab[h].code == rcode) {
        if (t->tab[h].hamming <= max_hamming) { *id = t->tab[h].id; *hamming = t->tab[h].hamming; *rotation = r; return 1; }
        break;
      }
      h = (h + 1) & (t->cap - 1);
    }
    rcode = rotate90(fam, rcode);
  }
  return 0;
}
static int decode_codeword_scan(const ato_family_t* fam, uint64_t rcode, int max_hamming, int* id, int* hamming, int* rotation) {
  for (int r = 0; r < 4; r++) {
    int best = 1 << 30, bid = -1;
    for (uint32_t i = 0; i < fam->ncodes; i++) {
      int hd = __builtin_popcountll(rcode ^ fam->codes[i]);
      if (hd < best) { best = hd; bid = (int)i; }
    }
    if (best <= max_hamming) { *id = bid; *hamming = best; *rotation = r; return 1; }
    rcode = rotate90(fam, rcode);
  }
  return 0;
}

static float quad_decode(const ato_params_t* prm, const ato_family_t* fam, const uint8_t* im, int w, int h, int pitch,
                         const double H[9], int* id, int* hamming, int* rotation, int* found) {
  int wb = (int)fam->width_at_border, tw = (int)fam->total_width;
  float patterns[40] = {
      -0.5f, 0.5f, 0, 1, 1,            /* left white column */
      0.5f, 0.5f, 0, 1, 0,             /* left black column */
      (float)wb + 0.5f, 0.5f, 0, 1, 1, /* right white column */
      (float)wb - 0.5f, 0.5f, 0, 1, 0, /* right black column */
      0.5f, -0.5f, 1, 0, 1,            /* top white row */
      0.5f, 0.5f, 1, 0, 0,             /* top black row */
      0.5f, (float)wb + 0.5f, 1, 0, 1, /* bottom white row */
      0.5f, (float)wb - 0.5f, 1, 0, 0  /* bottom black row */
  };
  graymodel_t whitemodel, blackmodel;
  memset(&whitemodel, 0, sizeof(whitemodel));
  memset(&blackmodel, 0, sizeof(blackmodel));
  *found = 0;
  for (int pi = 0; pi < 8; pi++) {
    const float* pat = &patterns[pi * 5];
    int is_white = (int)pat[4];
    for (int i = 0; i < wb; i++) {
      /* upstream: (pattern[0] + i*pattern[2]) / (family->width_at_border) with float pattern[] and int operands: float
       * multiplication, addition and DIVISION (exact for width 8, not for 5, 6, 7, 9 ...), then widened */
      double tagx01 = (double)((pat[0] + (float)i * pat[2]) / (float)wb);
      double tagy01 = (double)((pat[1] + (float)i * pat[3]) / (float)wb);
      double tagx = 2 * (tagx01 - 0.5), tagy = 2 * (tagy01 - 0.5);
      double px, py;
      homography_project(H, tagx, tagy, &px, &py);
      int ix = (int)px, iy = (int)py;
      if (ix < 0 || iy < 0 || ix >= w || iy >= h) continue;
      int v = im[(size_t)iy * pitch + ix];
      if (is_white) graymodel_add(&whitemodel, tagx, tagy, v);
      else graymodel_add(&blackmodel, tagx, tagy, v);
    }
  }
  graymodel_solve(&whitemodel);
  graymodel_solve(&blackmodel);
  if ((graymodel_interpolate(&whitemodel, 0, 0) - graymodel_interpolate(&blackmodel, 0, 0) < 0) != fam->reversed_border)
    return -1;

  double values[12 * 12];
  memset(values, 0, sizeof(values));
  int min_coord = (wb - tw) / 2;
  int d = (int)fam->d;
  for (int i = 0; i < (int)fam->nbits; i++) {
    int bitx = fam->bit_x[i], bity = fam->bit_y[i];
    double tagx01 = (bitx + 0.5) / wb, tagy01 = (bity + 0.5) / wb;
    double tagx = 2 * (tagx01 - 0.5), tagy = 2 * (tagy01 - 0.5);
    double px, py;
    homography_project(H, tagx, tagy, &px, &py);
    double v = value_for_pixel(im, w, h, pitch, px, py);
    if (v == -1) continue;
    double thresh = (graymodel_interpolate(&blackmodel, tagx, tagy) + graymodel_interpolate(&whitemodel, tagx, tagy)) / 2.0;
    values[tw * (bity - min_coord) + bitx - min_coord] = v - thresh;
  }
  /* sharpen: values += decode_sharpening * Laplacian(values) */
  {
    static const double kernel[9] = {0, -1, 0, -1, 4, -1, 0, -1, 0};
    double sharpened[12 * 12];
    for (int y = 0; y < tw; y++)
      for (int x = 0; x < tw; x++) {
        double s = 0;
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) {
            if ((y + i - 1) < 0 || (y + i - 1) > tw - 1 || (x + j - 1) < 0 || (x + j - 1) > tw - 1) continue;
            s += values[(y + i - 1) * tw + (x + j - 1)] * kernel[i * 3 + j];
          }
        sharpened[y * tw + x] = s;
      }
    for (int i = 0; i < tw * tw; i++) values[i] = values[i] + prm->decode_sharpening * sharpened[i];
  }
  float black_score = 0, white_score = 0, black_count = 1, white_count = 1;
  uint64_t rcode = 0;
  if ((prm->variant & ATO_VAR_AT3_BIT_ORDER) && d > 0) {
    /* AprilTag 3 numbers the data bits quadrant by quadrant: rows y = 1 + l, x = 1 + l .. d - 1 - l of the top triangle,
     * then the same triangle rotated by 90, 180, 270 degrees ((x, y) -> (d + 1 - y, x)), the centre bit of an odd d last.
     * The code word itself stays row-major here (this restatement's tables are); only the order of the float score
     * sums changes. */
    int ox[64], oy[64], n = 0;
    for (int q = 0; q < 4; q++)
      for (int l = 0; 2 * l < d - 1; l++)
        for (int x = 1 + l; x <= d - 1 - l; x++) {
          int bx = x, by = 1 + l;
          for (int r = 0; r < q; r++) { int nx = d + 1 - by, ny = bx; bx = nx; by = ny; }
          ox[n] = bx; oy[n] = by; n++;
        }
    if (d & 1) { ox[n] = (d + 1) / 2; oy[n] = (d + 1) / 2; n++; }
    for (int i = 0; i < n; i++) {
      double v = values[(oy[i] - min_coord) * tw + ox[i] - min_coord];
      if (v > 0) { white_score = (float)((double)white_score + v); white_count++; }
      else { black_score = (float)((double)black_score - v); black_count++; }
    }
    for (int i = 0; i < (int)fam->nbits; i++) {
      int bitx = 1 + i % d, bity = 1 + i / d;
      rcode <<= 1;
      if (values[(bity - min_coord) * tw + bitx - min_coord] > 0) rcode |= 1;
    }
  } else
  for (int i = 0; i < (int)fam->nbits; i++) {
    int bitx = fam->bit_x[i], bity = fam->bit_y[i];
    rcode <<= 1;
    double v = values[(bity - min_coord) * tw + bitx - min_coord];
    if (v > 0) { white_score = (float)((double)white_score + v); white_count++; rcode |= 1; }
    else { black_score = (float)((double)black_score - v); black_count++; }
  }
  *found = (prm->variant & ATO_VAR_FAST_PATHS) ? decode_codeword_fast(fam, rcode, prm->max_hamming, id, hamming, rotation)
                                              : decode_codeword_scan(fam, rcode, prm->max_hamming, id, hamming, rotation);
  float a = white_score / white_count, b = black_score / black_count;
  return a < b ? a : b;
}

/* ------------------------------------------------------------------------------------------- */
/* S8  reconcile duplicates + ordering (A.9)                                                    */
/* ------------------------------------------------------------------------------------------- */
static double orient2d(const double* a, const double* b, const double* c) {
  return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]);
}
static int on_segment(const double* a, const double* b, const double* c) {
  double lox = a[0] < b[0] ? a[0] : b[0], hix = a[0] < b[0] ? b[0] : a[0];
  double loy = a[1] < b[1] ? a[1] : b[1], hiy = a[1] < b[1] ? b[1] : a[1];
  return c[0] >= lox && c[0] <= hix && c[1] >= loy && c[1] <= hiy;
}
static int segments_intersect(const double* p1, const double* p2, const double* q1, const double* q2) {
  double d1 = orient2d(q1, q2, p1), d2 = orient2d(q1, q2, p2);
  double d3 = orient2d(p1, p2, q1), d4 = orient2d(p1, p2, q2);
  if (((d1 > 0 && d2 < 0) || (d1 < 0 && d2 > 0)) && ((d3 > 0 && d4 < 0) || (d3 < 0 && d4 > 0))) return 1;
  if (d1 == 0 && on_segment(q1, q2, p1)) return 1;
  if (d2 == 0 && on_segment(q1, q2, p2)) return 1;
  if (d3 == 0 && on_segment(p1, p2, q1)) return 1;
  if (d4 == 0 && on_segment(p1, p2, q2)) return 1;
  return 0;
}
static int point_in_quad(const double poly[4][2], const double* q) {
  int inside = 0;
  for (int i = 0, j = 3; i < 4; j = i++) {
    if (((poly[i][1] > q[1]) != (poly[j][1] > q[1])) &&
        (q[0] < (poly[j][0] - poly[i][0]) * (q[1] - poly[i][1]) / (poly[j][1] - poly[i][1]) + poly[i][0]))
      inside = !inside;
  }
  return inside;
}
static int quads_overlap(const double a[4][2], const double b[4][2]) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      if (segments_intersect(a[i], a[(i + 1) & 3], b[j], b[(j + 1) & 3])) return 1;
  if (point_in_quad(a, b[0])) return 1;
  if (point_in_quad(b, a[0])) return 1;
  return 0;
}
/* CANONICAL preference order: family, id, lower hamming, higher margin, then corner coordinates */
static int det_cmp(const void* pa, const void* pb) {
  const ato_detection_t* a = (const ato_detection_t*)pa;
  const ato_detection_t* b = (const ato_detection_t*)pb;
  if (a->family != b->family) return a->family < b->family ? -1 : 1;
  if (a->id != b->id) return a->id < b->id ? -1 : 1;
  if (a->hamming != b->hamming) return a->hamming < b->hamming ? -1 : 1;
  if (a->decision_margin != b->decision_margin) return a->decision_margin > b->decision_margin ? -1 : 1;
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 2; k++)
      if (a->p[i][k] != b->p[i][k]) return a->p[i][k] < b->p[i][k] ? -1 : 1;
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* S9  planar pose from the homography (A.10, "reference homography solve")                     */
/* ------------------------------------------------------------------------------------------- */
static void mat33_inv_transpose(const double* M, double* O) {
  double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
  double c10 = M[2] * M[7] - M[1] * M[8], c11 = M[0] * M[8] - M[2] * M[6], c12 = M[1] * M[6] - M[0] * M[7];
  double c20 = M[1] * M[5] - M[2] * M[4], c21 = M[2] * M[3] - M[0] * M[5], c22 = M[0] * M[4] - M[1] * M[3];
  double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
  /* inverse = adj/det = cof^T/det, so inverse-transpose = cof/det */
  O[0] = c00 / det; O[1] = c01 / det; O[2] = c02 / det;
  O[3] = c10 / det; O[4] = c11 / det; O[5] = c12 / det;
  O[6] = c20 / det; O[7] = c21 / det; O[8] = c22 / det;
}

/* Orthogonal polar factor of a 3x3 matrix through its singular value decomposition, M = U S V^T -> U V^T
 * (upstream: matd_svd, then R = U*V').  U V^T = M V S^-1 V^T with V, S^2 from the Jacobi eigen-decomposition of
 * M^T M. */
static void polar_by_svd(const double* M, double* O) {
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i * 3 + j] = M[0 * 3 + i] * M[0 * 3 + j] + M[1 * 3 + i] * M[1 * 3 + j] + M[2 * 3 + i] * M[2 * 3 + j];
  for (int sweep = 0; sweep < 30; sweep++) {
    double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double apq = A[p * 3 + q];
        if (apq == 0) continue;
        double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2 * apq);
        double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
        double c = 1 / sqrt(tt * tt + 1), sn = tt * c;
        for (int k = 0; k < 3; k++) {  /* A <- A J */
          double akp = A[k * 3 + p], akq = A[k * 3 + q];
          A[k * 3 + p] = c * akp - sn * akq; A[k * 3 + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {  /* A <- J^T A */
          double apk = A[p * 3 + k], aqk = A[q * 3 + k];
          A[p * 3 + k] = c * apk - sn * aqk; A[q * 3 + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
          V[k * 3 + p] = c * vkp - sn * vkq; V[k * 3 + q] = sn * vkp + c * vkq;
        }
      }
  }
  double is[3] = {1 / sqrt(A[0]), 1 / sqrt(A[4]), 1 / sqrt(A[8])};
  double T[9];  /* V S^-1 V^T */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[i * 3 + j] = V[i * 3 + 0] * is[0] * V[j * 3 + 0] + V[i * 3 + 1] * is[1] * V[j * 3 + 1] + V[i * 3 + 2] * is[2] * V[j * 3 + 2];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) O[i * 3 + j] = M[i * 3 + 0] * T[0 * 3 + j] + M[i * 3 + 1] * T[1 * 3 + j] + M[i * 3 + 2] * T[2 * 3 + j];
}

void ato_pose_from_homography_ex(const double H[9], double fx_in, double fy, double cx, double cy, double skew,
                                 double tag_size, int variant, double R[9], double t[3]) {
  double fx = -fx_in; /* upstream calls homography_to_pose(H, -fx, fy, cx, cy) */
  double R20 = H[6], R21 = H[7], TZ = H[8];
  double R10 = (H[3] - cy * R20) / fy, R11 = (H[4] - cy * R21) / fy, TY = (H[5] - cy * TZ) / fy;
  double R00, R01, TX;
  if (skew == 0.0) {
    R00 = (H[0] - cx * R20) / fx; R01 = (H[1] - cx * R21) / fx; TX = (H[2] - cx * TZ) / fx;
  } else {  /* first row of K = (fx, skew, cx); upstream has no skew term (the VPI path of the reference does) */
    R00 = (H[0] - cx * R20 - skew * R10) / fx; R01 = (H[1] - cx * R21 - skew * R11) / fx; TX = (H[2] - cx * TZ - skew * TY) / fx;
  }
  double length1 = (double)sqrtf((float)(R00 * R00 + R10 * R10 + R20 * R20));
  double length2 = (double)sqrtf((float)(R01 * R01 + R11 * R11 + R21 * R21));
  double s = 1.0 / (double)sqrtf((float)(length1 * length2));
  if (TZ > 0) s *= -1;
  R20 *= s; R21 *= s; TZ *= s; R00 *= s; R01 *= s; TX *= s; R10 *= s; R11 *= s; TY *= s;
  double R02 = R10 * R21 - R20 * R11, R12 = R20 * R01 - R00 * R21, R22 = R00 * R11 - R10 * R01;
  /* polar decomposition (upstream: R = U*V' from the SVD); canonically the orthogonal polar factor by a
   * fixed number of Newton steps X <- (X + X^-T)/2, which converges to the same matrix. */
  double X[9] = {R00, R01, R02, R10, R11, R12, R20, R21, R22};
  if (variant & ATO_VAR_SVD_POLAR) {
    double Y[9];
    polar_by_svd(X, Y);
    memcpy(X, Y, sizeof(X));
  } else {
    for (int it = 0; it < 12; it++) {
      double Y[9];
      mat33_inv_transpose(X, Y);
      for (int i = 0; i < 9; i++) X[i] = 0.5 * (X[i] + Y[i]);
    }
  }
  double scale = tag_size / 2.0;
  TX *= scale; TY *= scale; TZ *= scale;
  /* fix = diag(1,-1,-1): camera looks along +z, y down */
  R[0] = X[0]; R[1] = X[1]; R[2] = X[2];
  R[3] = -X[3]; R[4] = -X[4]; R[5] = -X[5];
  R[6] = -X[6]; R[7] = -X[7]; R[8] = -X[8];
  t[0] = TX; t[1] = -TY; t[2] = -TZ;
}


void ato_pose_from_homography(const double H[9], double fx, double fy, double cx, double cy, double tag_size,
                              double R[9], double t[3]) {
  ato_pose_from_homography_ex(H, fx, fy, cx, cy, 0.0, tag_size, 0, R, t);
}

/* ------------------------------------------------------------------------------------------- */
/* full pipeline -- the CPU stand-in for one cuAprilTagsDetect call                             */
/* (src/apriltag_node.cpp:491-493)                                                              */
/* ------------------------------------------------------------------------------------------- */
static int quad_key_cmp(const void* a, const void* b) {
  const ato_quad_t* x = (const ato_quad_t*)a; const ato_quad_t* y = (const ato_quad_t*)b;
  return (x->key > y->key) - (x->key < y->key);
}

int ato_detect(const ato_params_t* prm, const ato_family_t* fams, int nfam, const uint8_t* image, int width,
               int height, int pitch, ato_detection_t* out, int max_det, ato_dump_t* dump) {
  if (nfam < 1 || nfam > ATO_MAX_FAMILIES || prm->decimate < 1 || prm->tile_size < 1) return -1;
  int f = prm->decimate;
  int w = 1 + (width - 1) / f, h = 1 + (height - 1) / f;
  if (w / prm->tile_size < 1 || h / prm->tile_size < 1 || 2 * w + 1 >= (1 << 14) || 2 * h + 1 >= (1 << 14)) return -2;
  size_t n = (size_t)w * h;
  uint8_t* gray = (uint8_t*)malloc(n);
  uint8_t* thr = (uint8_t*)malloc(n);
  uint32_t* label = (uint32_t*)malloc(n * 4);
  uint32_t* csize = (uint32_t*)malloc(n * 4);
  int sw, sh;
  ato_decimate(image, width, height, pitch, f, gray, &sw, &sh);
  ato_threshold(gray, w, h, prm->tile_size, prm->min_white_black_diff, thr);
  ato_connected_components(thr, w, h, label, csize);

  size_t npts;
  kp_t* kp = gradient_points(thr, label, csize, w, h, prm->min_component_size, &npts);

  /* family-derived quad parameters */
  int min_tag_width = 1000000, normal_border = 0, reversed_border = 0;
  for (int i = 0; i < nfam; i++) {
    if ((int)fams[i].width_at_border < min_tag_width) min_tag_width = (int)fams[i].width_at_border;
    normal_border |= !fams[i].reversed_border;
    reversed_border |= fams[i].reversed_border;
  }
  min_tag_width = (int)((float)min_tag_width / (float)f);
  if (min_tag_width < 3) min_tag_width = 3;

  /* clusters -> quads */
  uint32_t maxpts = (uint32_t)(3 * (2 * w + 2 * h));
  size_t qcap = 64, nq = 0;
  ato_quad_t* quads = (ato_quad_t*)malloc(qcap * sizeof(ato_quad_t));
  uint32_t* ptsbuf = (uint32_t*)malloc((npts ? npts : 1) * 4);
  ato_cluster_t* all_clusters = NULL;
  size_t nall = group_points(kp, npts, &all_clusters, ptsbuf, dump != NULL);
  /* kept clusters are compacted to the front of ptsbuf (their points stay contiguous) */
  size_t nc = 0, npk = 0;
  ato_cluster_t* clusters = (ato_cluster_t*)malloc((nall ? nall : 1) * sizeof(ato_cluster_t));
  for (size_t ci = 0; ci < nall; ci++) {
    size_t cnt = all_clusters[ci].count;
    if (cnt >= (size_t)prm->min_cluster_points && cnt <= maxpts) {
      if (npk != all_clusters[ci].start) memmove(&ptsbuf[npk], &ptsbuf[all_clusters[ci].start], cnt * 4);
      clusters[nc].key = all_clusters[ci].key; clusters[nc].start = (uint32_t)npk; clusters[nc].count = (uint32_t)cnt; nc++;
      npk += cnt;
      ato_quad_t q;
      memset(&q, 0, sizeof(q));
      q.key = all_clusters[ci].key;
      if (fit_quad(prm, gray, w, h, &ptsbuf[npk - cnt], (int)cnt, min_tag_width, normal_border, reversed_border, &q)) {
        if (f > 1)
          for (int c = 0; c < 4; c++) {
            q.p[c][0] = (float)(((double)q.p[c][0] - 0.5) * (double)(float)f + 0.5);
            q.p[c][1] = (float)(((double)q.p[c][1] - 0.5) * (double)(float)f + 0.5);
          }
        if (nq == qcap) { qcap *= 2; quads = (ato_quad_t*)realloc(quads, qcap * sizeof(ato_quad_t)); }
        quads[nq++] = q;
      }
    }
  }
  free(all_clusters);
  free(kp);
  qsort(quads, nq, sizeof(ato_quad_t), quad_key_cmp);

  /* decode */
  size_t dcap = 64, nd = 0;
  ato_detection_t* dets = (ato_detection_t*)malloc(dcap * sizeof(ato_detection_t));
  for (size_t qi = 0; qi < nq; qi++) {
    ato_quad_t q = quads[qi];
    if (prm->refine_edges) refine_edges(prm, image, width, height, pitch, &q);
    double corr[4][4], H[9];
    for (int i = 0; i < 4; i++) {
      corr[i][0] = (i == 0 || i == 3) ? -1 : 1;
      corr[i][1] = (i == 0 || i == 1) ? -1 : 1;
      corr[i][2] = (double)q.p[i][0];
      corr[i][3] = (double)q.p[i][1];
    }
    if (homography_compute2(corr, H) != 0) continue;
    for (int fi = 0; fi < nfam; fi++) {
      if (fams[fi].reversed_border != q.reversed_border) continue;
      int id = 0, hamming = 0, rotation = 0, found = 0;
      float margin = quad_decode(prm, &fams[fi], image, width, height, pitch, H, &id, &hamming, &rotation, &found);
      if (!(margin >= 0 && found)) continue;
      if (nd == dcap) { dcap *= 2; dets = (ato_detection_t*)realloc(dets, dcap * sizeof(ato_detection_t)); }
      ato_detection_t* det = &dets[nd++];
      memset(det, 0, sizeof(*det));
      det->family = fi; det->id = id; det->hamming = hamming; det->decision_margin = margin;
      /* H' = H * Rz(rotation * 90 deg) as upstream forms it: c = cos(theta), s = sin(theta) of theta = rotation * M_PI / 2.0
       * from libm -- the correctly rounded values are written out as literals (cos(pi/2) is 6.1e-17, not 0) -- and the full
       * 3x3 product of matd_op("M*M"): acc = 0; acc += a[i][k] * b[k][j] for k = 0, 1, 2 */
      static const double CS[4][2] = {{1.0, 0.0},
                                      {6.123233995736766e-17, 1.0},
                                      {-1.0, 1.2246467991473532e-16},
                                      {-1.8369701987210297e-16, -1.0}};
      {
        const double c = CS[rotation][0], sn = CS[rotation][1];
        const double Rz[9] = {c, -sn, 0, sn, c, 0, 0, 0, 1};
        for (int r = 0; r < 3; r++)
          for (int cc = 0; cc < 3; cc++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += H[r * 3 + k] * Rz[k * 3 + cc];
            det->H[r * 3 + cc] = acc;
          }
      }
      homography_project(det->H, 0, 0, &det->c[0], &det->c[1]);
      for (int i = 0; i < 4; i++) {
        double tcx = (i == 1 || i == 2) ? 1 : -1, tcy = (i < 2) ? 1 : -1;
        homography_project(det->H, tcx, tcy, &det->p[i][0], &det->p[i][1]);
      }
    }
  }
  /* reconcile + order */
  qsort(dets, nd, sizeof(ato_detection_t), det_cmp);
  size_t nk = 0;
  for (size_t i = 0; i < nd; i++) {
    int dead = 0;
    for (size_t j = 0; j < nk && !dead; j++)
      if (dets[j].family == dets[i].family && dets[j].id == dets[i].id && quads_overlap(dets[j].p, dets[i].p)) dead = 1;
    if (!dead) dets[nk++] = dets[i];
  }
  nd = nk;
  for (size_t i = 0; i < nd; i++)
    ato_pose_from_homography_ex(dets[i].H, prm->fx, prm->fy, prm->cx, prm->cy, prm->skew, prm->tag_size, prm->variant, dets[i].R, dets[i].t);

  int nout = (int)(nd < (size_t)max_det ? nd : (size_t)max_det);
  for (int i = 0; i < nout; i++) out[i] = dets[i];

  if (dump) {
    dump->w = w; dump->h = h; dump->gray = gray; dump->thr = thr; dump->label = label; dump->csize = csize;
    dump->nclusters = (uint32_t)nc; dump->clusters = clusters; dump->npoints = (uint32_t)npk; dump->points = ptsbuf;
    dump->nquads = (uint32_t)nq; dump->quads = quads; dump->ndet = (uint32_t)nd; dump->dets = dets;
  } else {
    free(gray); free(thr); free(label); free(csize); free(clusters); free(ptsbuf); free(quads); free(dets);
  }
  return nout;
}

void ato_dump_free(ato_dump_t* d) {
  free(d->gray); free(d->thr); free(d->label); free(d->csize); free(d->clusters); free(d->points); free(d->quads);
  free(d->dets);
  memset(d, 0, sizeof(*d));
}

/* ------------------------------------------------------------------------------------------- */
/* front steps: resize and rectify (reference README.md:16-29; launch/..._usb_cam.launch.py:43-63) */
/* ------------------------------------------------------------------------------------------- */
void ato_resize_mono8(const uint8_t* src, int spitch, int sw, int sh, uint8_t* dst, int dpitch, int dw, int dh) {
  for (int y = 0; y < dh; y++)
    for (int x = 0; x < dw; x++) {
      long long fx = ((long long)(2 * x + 1) * sw * 1024) / dw - 1024;
      long long fy = ((long long)(2 * y + 1) * sh * 1024) / dh - 1024;
      if (fx < 0) fx = 0;
      if (fy < 0) fy = 0;
      int x0 = (int)(fx >> 11), y0 = (int)(fy >> 11), wx = (int)(fx & 2047), wy = (int)(fy & 2047);
      if (x0 >= sw - 1) { x0 = sw - 1; wx = 0; }
      if (y0 >= sh - 1) { y0 = sh - 1; wy = 0; }
      int x1 = x0 + 1 < sw ? x0 + 1 : sw - 1, y1 = y0 + 1 < sh ? y0 + 1 : sh - 1;
      uint32_t p00 = src[(size_t)y0 * spitch + x0], p01 = src[(size_t)y0 * spitch + x1];
      uint32_t p10 = src[(size_t)y1 * spitch + x0], p11 = src[(size_t)y1 * spitch + x1];
      uint32_t top = p00 * (2048 - wx) + p01 * wx, bot = p10 * (2048 - wx) + p11 * wx;
      uint64_t v = (uint64_t)top * (2048 - wy) + (uint64_t)bot * wy;
      dst[(size_t)y * dpitch + x] = (uint8_t)((v + (1ull << 21)) >> 22);
    }
}

void ato_rectify_mono8(const uint8_t* src, int spitch, uint8_t* dst, int dpitch, int w, int h, const double K[9],
                       const double D[5], const double Knew[9]) {
  double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
  double nfx = Knew[0], nfy = Knew[4], ncx = Knew[2], ncy = Knew[5];
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      double xn = ((double)x - ncx) / nfx, yn = ((double)y - ncy) / nfy;
      double r2 = xn * xn + yn * yn;
      double radial = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
      double xd = xn * radial + (2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn));
      double yd = yn * radial + (p1 * (r2 + 2.0 * yn * yn) + 2.0 * p2 * xn * yn);
      double u = fx * xd + cx, v = fy * yd + cy;
      uint8_t out = 0;
      if (u >= 0.0 && v >= 0.0 && u <= (double)(w - 1) && v <= (double)(h - 1)) {
        int fu = (int)(u * 32.0 + 0.5), fv = (int)(v * 32.0 + 0.5);
        int x0 = fu >> 5, y0 = fv >> 5, wx = fu & 31, wy = fv & 31;
        if (x0 >= w - 1) { x0 = w - 1; wx = 0; }
        if (y0 >= h - 1) { y0 = h - 1; wy = 0; }
        int x1 = x0 + 1 < w ? x0 + 1 : w - 1, y1 = y0 + 1 < h ? y0 + 1 : h - 1;
        uint32_t p00 = src[(size_t)y0 * spitch + x0], p01 = src[(size_t)y0 * spitch + x1];
        uint32_t p10 = src[(size_t)y1 * spitch + x0], p11 = src[(size_t)y1 * spitch + x1];
        uint32_t top = p00 * (32 - wx) + p01 * wx, bot = p10 * (32 - wx) + p11 * wx;
        out = (uint8_t)((top * (32 - wy) + bot * wy + 512) >> 10);
      }
      dst[(size_t)y * dpitch + x] = out;
    }
}
