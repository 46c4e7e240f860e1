/* ORACLE — test infrastructure only: CPU restatement of the reference's rotated-box BEV overlap / IoU / NMS.
 *
 * Follows /root/reference/mmdet3d/ops/iou3d/src/iou3d_kernel.cu in fp32, step for step:
 *   :35-43   cross products            :45-52  bounding-rectangle rejection of two segments
 *   :54-77   point-in-rotated-box test (MARGIN 1e-5, rotation by -angle about the box centre)
 *   :79-107  segment intersection (strict straddle test, two formulas split on |s5 - s1| > EPS)
 *   :109-118 corner rotation about the centre by +angle
 *   :120-124 ordering of polygon vertices by atan2 about their centroid (bubble sort, :198-208)
 *   :126-222 box_overlap: 16 edge pairs, then corners of b inside a / of a inside b, centroid, sort, shoelace
 *   :224-229 iou_bev = overlap / max(sa + sb - overlap, EPS)
 *   :291-299 iou_normal (axis-aligned)
 * and the host side of the two NMS entry points (iou3d.cpp:96-137,139-180): boxes arrive sorted by score,
 * box i is kept unless an earlier kept box overlaps it by more than the threshold.
 * Boxes are [x1, y1, x2, y2, angle] (5 floats). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define EPSF 1e-8f

typedef struct { float x, y; } Pt;

static float cross2(Pt a, Pt b) { return a.x * b.y - a.y * b.x; }
static float cross3(Pt p1, Pt p2, Pt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }

static int rect_cross(Pt p1, Pt p2, Pt q1, Pt q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

static int in_box2d(const float* box, Pt p) {
  const float MARGIN = 1e-5f;
  float cx = (box[0] + box[2]) / 2, cy = (box[1] + box[3]) / 2;
  float c = cosf(-box[4]), s = sinf(-box[4]);
  float rx = (p.x - cx) * c + (p.y - cy) * s + cx;
  float ry = -(p.x - cx) * s + (p.y - cy) * c + cy;
  return rx > box[0] - MARGIN && rx < box[2] + MARGIN && ry > box[1] - MARGIN && ry < box[3] + MARGIN;
}

static int seg_intersection(Pt p1, Pt p0, Pt q1, Pt q0, Pt* ans) {
  if (!rect_cross(p0, p1, q0, q1)) return 0;
  float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > EPSF) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

static void rotate_about(Pt c, float ac, float as, Pt* p) {
  float nx = (p->x - c.x) * ac + (p->y - c.y) * as + c.x;
  float ny = -(p->x - c.x) * as + (p->y - c.y) * ac + c.y;
  p->x = nx;
  p->y = ny;
}

static int pt_cmp(Pt a, Pt b, Pt c) { return atan2f(a.y - c.y, a.x - c.x) > atan2f(b.y - c.y, b.x - c.x); }

float iou3d_box_overlap(const float* A, const float* B) {
  Pt ca = {(A[0] + A[2]) / 2, (A[1] + A[3]) / 2}, cb = {(B[0] + B[2]) / 2, (B[1] + B[3]) / 2};
  Pt a[5] = {{A[0], A[1]}, {A[2], A[1]}, {A[2], A[3]}, {A[0], A[3]}, {0, 0}};
  Pt b[5] = {{B[0], B[1]}, {B[2], B[1]}, {B[2], B[3]}, {B[0], B[3]}, {0, 0}};
  float aco = cosf(A[4]), asi = sinf(A[4]), bco = cosf(B[4]), bsi = sinf(B[4]);
  for (int k = 0; k < 4; ++k) {
    rotate_about(ca, aco, asi, &a[k]);
    rotate_about(cb, bco, bsi, &b[k]);
  }
  a[4] = a[0];
  b[4] = b[0];
  Pt poly[16], centre = {0, 0};
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (seg_intersection(a[i + 1], a[i], b[j + 1], b[j], &poly[cnt])) {
        centre.x += poly[cnt].x;
        centre.y += poly[cnt].y;
        ++cnt;
      }
  for (int k = 0; k < 4; ++k) {
    if (in_box2d(A, b[k])) {
      centre.x += b[k].x;
      centre.y += b[k].y;
      poly[cnt++] = b[k];
    }
    if (in_box2d(B, a[k])) {
      centre.x += a[k].x;
      centre.y += a[k].y;
      poly[cnt++] = a[k];
    }
  }
  centre.x /= cnt;   /* cnt == 0 -> NaN centre, loops below do not run, area stays 0 (as in the reference) */
  centre.y /= cnt;
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (pt_cmp(poly[i], poly[i + 1], centre)) {
        Pt t = poly[i];
        poly[i] = poly[i + 1];
        poly[i + 1] = t;
      }
  float area = 0;
  for (int k = 0; k < cnt - 1; ++k) {
    Pt u = {poly[k].x - poly[0].x, poly[k].y - poly[0].y}, v = {poly[k + 1].x - poly[0].x, poly[k + 1].y - poly[0].y};
    area += cross2(u, v);
  }
  return fabsf(area) / 2.0f;
}

float iou3d_iou_bev(const float* A, const float* B) {
  float sa = (A[2] - A[0]) * (A[3] - A[1]), sb = (B[2] - B[0]) * (B[3] - B[1]);
  float so = iou3d_box_overlap(A, B);
  return so / fmaxf(sa + sb - so, EPSF);
}

float iou3d_iou_normal(const float* a, const float* b) {
  float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]), top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f), inter = w * h;
  float sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
  return inter / fmaxf(sa + sb - inter, EPSF);
}

/* mode 0: overlap area, 1: rotated IoU, 2: axis-aligned IoU.  out [m, n]. */
void iou3d_pairwise(const float* boxes_a, int64_t m, const float* boxes_b, int64_t n, int mode, float* out) {
  for (int64_t i = 0; i < m; ++i)
    for (int64_t j = 0; j < n; ++j) {
      const float *A = boxes_a + i * 5, *B = boxes_b + j * 5;
      out[i * n + j] = mode == 0 ? iou3d_box_overlap(A, B) : mode == 1 ? iou3d_iou_bev(A, B) : iou3d_iou_normal(A, B);
    }
}

/* greedy NMS over boxes already sorted by descending score; returns the number kept, indices in keep[] */
int64_t iou3d_nms(const float* boxes, int64_t n, float thresh, int normal, int64_t* keep) {
  unsigned char* removed = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
  int64_t kept = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (removed[i]) continue;
    keep[kept++] = i;
    for (int64_t j = i + 1; j < n; ++j) {
      if (removed[j]) continue;
      float v = normal ? iou3d_iou_normal(boxes + i * 5, boxes + j * 5) : iou3d_iou_bev(boxes + i * 5, boxes + j * 5);
      if (v > thresh) removed[j] = 1;
    }
  }
  free(removed);
  return kept;
}
