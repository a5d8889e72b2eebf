"""CPU restatement of the reference's rotated BEV overlap / IoU / NMS (TEST INFRASTRUCTURE ONLY - imported by tests/, never by
the product path).

Follows pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:30-235 (`box_overlap`, `iou_bev`, `iou_normal`) and the greedy mask scan
of iou3d_nms.cpp (`nms_gpu` / `nms_normal_gpu`): intersection polygon = proper edge crossings + corners inside the other box
with a 1e-2 margin, ordered by angle around the centroid, shoelace area, fp32 arithmetic.

**parity unpinned**: the reference implementation is a CUDA extension (its CPU twin includes <cuda.h>, absent here) and the
reference holds no test vectors for it.  The restatement is cross-checked in tests against an independent exact convex
clipping (Sutherland-Hodgman in float64, `exact_overlap`) away from the margin cases."""
import math

import numpy as np

F = np.float32
EPS, MARGIN = F(1e-8), F(1e-2)


def _corners(b):
    hx, hy = F(b[3]) * F(0.5), F(b[4]) * F(0.5)
    cs, sn = F(math.cos(float(b[6]))), F(math.sin(float(b[6])))
    out = []
    for lx, ly in ((-hx, -hy), (hx, -hy), (hx, hy), (-hx, hy)):
        px, py = (F(b[0]) + lx) - F(b[0]), (F(b[1]) + ly) - F(b[1])
        out.append((F(px * cs + py * (-sn) + F(b[0])), F(px * sn + py * cs + F(b[1]))))
    out.append(out[0])
    return out


def _cross3(a, b, o):
    return F((a[0] - o[0]) * (b[1] - o[1]) - (b[0] - o[0]) * (a[1] - o[1]))


def _inside(b, p):
    cs, sn = F(math.cos(-float(b[6]))), F(math.sin(-float(b[6])))
    rx = F((p[0] - F(b[0])) * cs + (p[1] - F(b[1])) * (-sn))
    ry = F((p[0] - F(b[0])) * sn + (p[1] - F(b[1])) * cs)
    return abs(rx) < F(b[3]) * F(0.5) + MARGIN and abs(ry) < F(b[4]) * F(0.5) + MARGIN


def _seg(p1, p0, q1, q0):
    if not (min(p0[0], p1[0]) <= max(q0[0], q1[0]) and min(q0[0], q1[0]) <= max(p0[0], p1[0]) and
            min(p0[1], p1[1]) <= max(q0[1], q1[1]) and min(q0[1], q1[1]) <= max(p0[1], p1[1])):
        return None
    s1, s2, s3, s4 = _cross3(q0, p1, p0), _cross3(p1, q1, p0), _cross3(p0, q1, q0), _cross3(q1, p1, q0)
    if not (s1 * s2 > 0 and s3 * s4 > 0):
        return None
    s5 = _cross3(q1, p1, p0)
    if abs(s5 - s1) > EPS:
        return (F((s5 * q0[0] - s1 * q1[0]) / (s5 - s1)), F((s5 * q0[1] - s1 * q1[1]) / (s5 - s1)))
    a0, b0, c0 = p0[1] - p1[1], p1[0] - p0[0], p0[0] * p1[1] - p1[0] * p0[1]
    a1, b1, c1 = q0[1] - q1[1], q1[0] - q0[0], q0[0] * q1[1] - q1[0] * q0[1]
    D = a0 * b1 - a1 * b0
    return (F((b0 * c1 - b1 * c0) / D), F((a1 * c0 - a0 * c1) / D))


def overlap(a, b):
    ca, cb = _corners(a), _corners(b)
    pts = []
    for i in range(4):
        for j in range(4):
            x = _seg(ca[i + 1], ca[i], cb[j + 1], cb[j])
            if x is not None:
                pts.append(x)
    for k in range(4):
        if _inside(a, cb[k]):
            pts.append(cb[k])
        if _inside(b, ca[k]):
            pts.append(ca[k])
    n = len(pts)
    if n < 3:
        return F(0)
    mx, my = F(sum(p[0] for p in pts) / F(n)), F(sum(p[1] for p in pts) / F(n))
    pts = sorted(pts, key=lambda p: math.atan2(float(p[1] - my), float(p[0] - mx)))
    area = F(0)
    for k in range(n - 1):
        ux, uy, vx, vy = pts[k][0] - pts[0][0], pts[k][1] - pts[0][1], pts[k + 1][0] - pts[0][0], pts[k + 1][1] - pts[0][1]
        area = F(area + (ux * vy - uy * vx))
    return F(abs(area) * F(0.5))


def iou_bev(a, b):
    ov = overlap(a, b)
    return F(ov / max(F(a[3]) * F(a[4]) + F(b[3]) * F(b[4]) - ov, EPS))


def iou_axis(a, b):
    l, r = max(a[0] - a[3] / 2, b[0] - b[3] / 2), min(a[0] + a[3] / 2, b[0] + b[3] / 2)
    t, bo = max(a[1] - a[4] / 2, b[1] - b[4] / 2), min(a[1] + a[4] / 2, b[1] + b[4] / 2)
    inter = max(r - l, 0) * max(bo - t, 0)
    return F(inter / max(a[3] * a[4] + b[3] * b[4] - inter, 1e-8))


def pairs(A, B, mode):
    return np.array([[overlap(a, b) if mode == 0 else iou_bev(a, b) for b in B] for a in A], dtype=np.float32).reshape(len(A), len(B))


def nms(boxes_sorted, thresh, rotated=True):
    """Greedy NMS over boxes already sorted by descending score -> kept indices (ascending)."""
    n = len(boxes_sorted)
    f = iou_bev if rotated else iou_axis
    removed = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        for j in range(i + 1, n):
            if not removed[j] and f(boxes_sorted[i], boxes_sorted[j]) > thresh:
                removed[j] = True
    return np.array(keep, dtype=np.int64)


def exact_overlap(a, b):
    """Independent check: exact area of the intersection of two rotated rectangles (Sutherland-Hodgman, float64)."""
    def corners(bx):
        c, s = math.cos(bx[6]), math.sin(bx[6])
        return [(bx[0] + lx * c - ly * s, bx[1] + lx * s + ly * c)
                for lx, ly in ((-bx[3] / 2, -bx[4] / 2), (bx[3] / 2, -bx[4] / 2), (bx[3] / 2, bx[4] / 2), (-bx[3] / 2, bx[4] / 2))]
    poly, clip = corners([float(v) for v in a]), corners([float(v) for v in b])
    for i in range(4):
        p0, p1 = clip[i], clip[(i + 1) % 4]
        inside = lambda q: (p1[0] - p0[0]) * (q[1] - p0[1]) - (p1[1] - p0[1]) * (q[0] - p0[0]) >= 0     # noqa: E731
        new = []
        for k in range(len(poly)):
            c, nx = poly[k], poly[(k + 1) % len(poly)]
            ic, inx = inside(c), inside(nx)
            if ic != inx:
                dx, dy = nx[0] - c[0], nx[1] - c[1]
                ex, ey = p1[0] - p0[0], p1[1] - p0[1]
                t = (ex * (c[1] - p0[1]) - ey * (c[0] - p0[0])) / (ey * dx - ex * dy)
                hit = (c[0] + t * dx, c[1] + t * dy)
                if ic:
                    new += [c, hit]
                else:
                    new += [hit]
            elif ic:
                new.append(c)
        poly = new
        if not poly:
            return 0.0
    return abs(sum(poly[k][0] * poly[(k + 1) % len(poly)][1] - poly[(k + 1) % len(poly)][0] * poly[k][1] for k in range(len(poly)))) / 2
