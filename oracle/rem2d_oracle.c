/*
 * rem2d_oracle.c -- CPU restatement (plain C, one world per creature) of the
 * hot path of gym_rem2D:  Modular2D.step() -> world.Step(1/50, 180, 60).
 *
 * TEST INFRASTRUCTURE ONLY (see rem2d_oracle.h).  PARITY UNPINNED against the
 * real Box2D 2.3.10 wheel, which is not available anywhere this can be built.
 *
 * What is restated, and from where:
 *   - Modular2D.step / PID / WallOfDeath      gym_rem2D/envs/Modular2DEnv.py:102-108,600-653
 *   - Controller.update                       Controller/m_controller.py:17-21
 *   - the pybox2d call surface used by reset  Modular2DEnv.py:144,226-306,572;
 *                                             simple_module.py:286-298; circular_module.py:191-202;
 *                                             module_utility.py:19-32
 *   - evaluate() fitness rule                 REM2D_main.py:350-378
 *   - Box2D 2.3.x b2World::Step and everything under it: SURVEY.md Appendix A
 *     [B2D-recalled]; section tags below (A.2 .. A.10) refer to it.
 *
 * All engine arithmetic is IEEE binary32 without fused multiply-add (build with
 * -ffp-contract=off), evaluated in the operand order of the Box2D expressions.
 * sinf/cosf are replaced by one documented algorithm (rem2d_oracle_sincosf, built
 * from explicit fmaf steps) so that the HIP path can reproduce it bit for bit;
 * Python's math.sin by o_sin.
 */
#include "rem2d_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* REM2D_ORACLE_F64: the "truth" build (librem2d_oracle_f64.so).  Every engine quantity is binary64 (positions,
 * velocities, impulses, manifolds, GJK / TOI), constants keep Box2D's binary32 literals, b2Rot::Set uses the binary64
 * sine / cosine.  Same ABI (float arrays in and out; the conversion happens at the boundary).  It quantifies how much
 * of a trajectory difference is inherent to binary32 arithmetic (SURVEY.md 8c protocol iv); it is NOT bit-comparable
 * with anything. */
#ifdef REM2D_ORACLE_F64
typedef double f32;
#define sqrtf sqrt
#define floorf floor
#else
typedef float f32;
#endif

/* ---- A.1 constants (b2Settings.h) ---- */
#define B2_PI 3.14159265359f
#define B2_EPSILON FLT_EPSILON
#define B2_MAXFLOAT FLT_MAX
#define B2_LINEAR_SLOP 0.005f
#define B2_ANGULAR_SLOP (2.0f / 180.0f * B2_PI)
#define B2_POLYGON_RADIUS (2.0f * B2_LINEAR_SLOP)
#define B2_AABB_EXTENSION 0.1f
#define B2_AABB_MULTIPLIER 2.0f
#define B2_VELOCITY_THRESHOLD 1.0f
#define B2_MAX_LINEAR_CORRECTION 0.2f
#define B2_MAX_ANGULAR_CORRECTION (8.0f / 180.0f * B2_PI)
#define B2_MAX_TRANSLATION 2.0f
#define B2_MAX_TRANSLATION_SQ (B2_MAX_TRANSLATION * B2_MAX_TRANSLATION)
#define B2_MAX_ROTATION (0.5f * B2_PI)
#define B2_MAX_ROTATION_SQ (B2_MAX_ROTATION * B2_MAX_ROTATION)
#define B2_BAUMGARTE 0.2f
#define B2_TOI_BAUGARTE 0.75f
#define B2_TIME_TO_SLEEP 0.5f
#define B2_LINEAR_SLEEP_TOL 0.01f
#define B2_ANGULAR_SLEEP_TOL (2.0f / 180.0f * B2_PI)
#define B2_MAX_SUB_STEPS 8
#define B2_MAX_TOI_CONTACTS 32
#define B2_MAX_POLY 8

/* ---- small vector algebra (b2Math.h), operand order as in Box2D ---- */
typedef struct { f32 x, y; } v2;
typedef struct { f32 x, y, z; } v3;
typedef struct { f32 s, c; } rot_t;
typedef struct { v2 p; rot_t q; } xf_t;
typedef struct { v2 lower, upper; } aabb_t;

static inline v2 V2(f32 x, f32 y) { v2 r; r.x = x; r.y = y; return r; }
static inline v2 vadd(v2 a, v2 b) { return V2(a.x + b.x, a.y + b.y); }
static inline v2 vsub(v2 a, v2 b) { return V2(a.x - b.x, a.y - b.y); }
static inline v2 vneg(v2 a) { return V2(-a.x, -a.y); }
static inline v2 vscale(f32 s, v2 a) { return V2(s * a.x, s * a.y); }
static inline f32 vdot(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }
static inline f32 vcross(v2 a, v2 b) { return a.x * b.y - a.y * b.x; }
static inline v2 vcross_vs(v2 a, f32 s) { return V2(s * a.y, -s * a.x); }
static inline v2 vcross_sv(f32 s, v2 a) { return V2(-s * a.y, s * a.x); }
static inline f32 vlen(v2 a) { return sqrtf(a.x * a.x + a.y * a.y); }
static inline f32 vlen2(v2 a) { return a.x * a.x + a.y * a.y; }
static inline f32 vdist2(v2 a, v2 b) { v2 c = vsub(a, b); return vdot(c, c); }
static inline f32 fmin32(f32 a, f32 b) { return a < b ? a : b; }
static inline f32 fmax32(f32 a, f32 b) { return a > b ? a : b; }
static inline f32 fabs32(f32 a) { return a > 0.0f ? a : -a; }
static inline f32 fclamp(f32 a, f32 lo, f32 hi) { return fmax32(lo, fmin32(a, hi)); }
static inline v2 vmin(v2 a, v2 b) { return V2(fmin32(a.x, b.x), fmin32(a.y, b.y)); }
static inline v2 vmax(v2 a, v2 b) { return V2(fmax32(a.x, b.x), fmax32(a.y, b.y)); }
static inline f32 vnormalize(v2 *a) {
    f32 length = vlen(*a);
    if (length < B2_EPSILON) return 0.0f;
    f32 inv = 1.0f / length;
    a->x *= inv;
    a->y *= inv;
    return length;
}
static inline v2 rmul(rot_t q, v2 v) { return V2(q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y); }
static inline v2 rmulT(rot_t q, v2 v) { return V2(q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y); }
static inline v2 xmul(xf_t T, v2 v) {
    f32 x = (T.q.c * v.x - T.q.s * v.y) + T.p.x;
    f32 y = (T.q.s * v.x + T.q.c * v.y) + T.p.y;
    return V2(x, y);
}
static inline v2 xmulT(xf_t T, v2 v) {
    f32 px = v.x - T.p.x, py = v.y - T.p.y;
    return V2(T.q.c * px + T.q.s * py, -T.q.s * px + T.q.c * py);
}
static inline rot_t rrmulT(rot_t q, rot_t r) {
    rot_t o;
    o.s = q.c * r.s - q.s * r.c;
    o.c = q.c * r.c + q.s * r.s;
    return o;
}
static inline xf_t xxmulT(xf_t A, xf_t B) {
    xf_t C;
    C.q = rrmulT(A.q, B.q);
    C.p = rmulT(A.q, vsub(B.p, A.p));
    return C;
}

/* ---- trig: one documented algorithm shared (by specification, not by code) with the HIP path.
 * Argument reduction: n = rint(x*2/pi); r = (x - n*PIO2_1) - n*PIO2_1T (fdlibm's 33-bit split),
 * then the fdlibm kernel polynomials on [-pi/4, pi/4], all in binary64 without FMA; the binary32
 * variant rounds the binary64 result once.  Valid for |x| < 2^20*pi/2. ---- */
static const double INV_PIO2 = 6.36619772367581382433e-01;
static const double PIO2_1 = 1.57079632673412561417e+00;
static const double PIO2_1T = 6.07710050650619224932e-11;
static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                    S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                    S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                    C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                    C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;

static void o_sincos_d(double x, double *s, double *c) {
    double fn = rint(x * INV_PIO2);
    int n = (int)fn;
    double r = (x - fn * PIO2_1) - fn * PIO2_1T;
    double z = r * r;
    double ps = r + r * (z * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))))));
    double pc = (1.0 - 0.5 * z) + z * z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    switch (n & 3) {
    case 0: *s = ps; *c = pc; break;
    case 1: *s = pc; *c = -ps; break;
    case 2: *s = -ps; *c = -pc; break;
    default: *s = -pc; *c = ps; break;
    }
}
/* b2Rot::Set: Box2D calls libm sinf/cosf, whose bits are platform specific.  "rem2d trig", binary32
 * form: n = rintf(x*2/pi); r = x - n*DP1 - n*DP2 - n*DP3 (Cody-Waite, pi/2 in three parts); Cephes
 * sinf/cosf polynomials on [-pi/4, pi/4] in Horner form.  Every step of the reduction and of the
 * polynomials is ONE fused multiply-add (fmaf: a single, exactly defined rounding -- the same bits from
 * the hardware instruction, from libm's software fallback and from the GPU's v_fma_f32); this is the
 * only place where the engine arithmetic fuses anything.  Within ~1 ulp of libm for |x| < 1e3
 * (tests/test_oracle_kat.py).  (Round 3: the separately rounded multiply / add form it replaces cost
 * the HIP path 11 more instructions per b2Rot::Set, of which the position solver runs two per joint and
 * one per manifold point in every iteration.) */
#if defined(__GNUC__) && defined(__x86_64__) && !defined(__clang__)
__attribute__((target_clones("fma", "default")))
#endif
void rem2d_oracle_sincosf(float x, float *s, float *c) {
    const float TWO_OVER_PI = 0.63661977236758134308f;
    const float DP1 = 1.5703125f, DP2 = 4.837512969970703125e-4f, DP3 = 7.54978995489188216e-8f;
    const float FS1 = -1.6666654611e-1f, FS2 = 8.3321608736e-3f, FS3 = -1.9515295891e-4f;
    const float FC1 = 4.166664568298827e-2f, FC2 = -1.388731625493765e-3f, FC3 = 2.443315711809948e-5f;
    float fn = rintf(x * TWO_OVER_PI);
    int n = (int)fn;
    float r = fmaf(-fn, DP3, fmaf(-fn, DP2, fmaf(-fn, DP1, x)));
    float z = r * r;
    float ps = fmaf(r, z * fmaf(z, fmaf(z, FS3, FS2), FS1), r);
    float pc = fmaf(z * z, fmaf(z, fmaf(z, FC3, FC2), FC1), fmaf(z, -0.5f, 1.0f));
    switch (n & 3) {
    case 0: *s = ps; *c = pc; break;
    case 1: *s = pc; *c = -ps; break;
    case 2: *s = -ps; *c = -pc; break;
    default: *s = -pc; *c = ps; break;
    }
}
double rem2d_oracle_sin(double x) {
    double ds, dc;
    o_sincos_d(x, &ds, &dc);
    return ds;
}
static inline rot_t rot_set(f32 a) {
    rot_t q;
#ifdef REM2D_ORACLE_F64
    double ds, dc;
    o_sincos_d(a, &ds, &dc);
    q.s = ds;
    q.c = dc;
#else
    rem2d_oracle_sincosf(a, &q.s, &q.c);
#endif
    return q;
}

/* ---- shapes (A.10) ---- */
enum { SH_EDGE = 0, SH_POLY = 1, SH_CIRCLE = 2 };
typedef struct {
    int type;
    f32 radius;
    int count;
    v2 verts[B2_MAX_POLY], normals[B2_MAX_POLY], centroid; /* polygon */
    v2 v1, v2_;                                             /* edge */
    v2 p;                                                   /* circle */
} shape_t;

typedef union {
    struct { uint8_t indexA, indexB, typeA, typeB; } cf;
    uint32_t key;
} cid_t;
enum { CF_VERTEX = 0, CF_FACE = 1 };
enum { MF_CIRCLES = 0, MF_FACE_A = 1, MF_FACE_B = 2 };
typedef struct { v2 localPoint; f32 normalImpulse, tangentImpulse; cid_t id; } mpoint_t;
typedef struct { mpoint_t points[2]; v2 localNormal, localPoint; int type, pointCount; } manifold_t;

/* b2PolygonShape::SetAsBox */
static void shape_set_box(shape_t *s, f32 hx, f32 hy) {
    memset(s, 0, sizeof(*s));
    s->type = SH_POLY;
    s->radius = B2_POLYGON_RADIUS;
    s->count = 4;
    s->verts[0] = V2(-hx, -hy);
    s->verts[1] = V2(hx, -hy);
    s->verts[2] = V2(hx, hy);
    s->verts[3] = V2(-hx, hy);
    s->normals[0] = V2(0.0f, -1.0f);
    s->normals[1] = V2(1.0f, 0.0f);
    s->normals[2] = V2(0.0f, 1.0f);
    s->normals[3] = V2(-1.0f, 0.0f);
    s->centroid = V2(0.0f, 0.0f);
}
static v2 poly_centroid(const v2 *vs, int count) {
    v2 c = V2(0.0f, 0.0f);
    f32 area = 0.0f;
    v2 pRef = V2(0.0f, 0.0f);
    const f32 inv3 = 1.0f / 3.0f;
    for (int i = 0; i < count; ++i) {
        v2 p1 = pRef, p2 = vs[i], p3 = i + 1 < count ? vs[i + 1] : vs[0];
        v2 e1 = vsub(p2, p1), e2 = vsub(p3, p1);
        f32 D = vcross(e1, e2);
        f32 triangleArea = 0.5f * D;
        area += triangleArea;
        c = vadd(c, vscale(triangleArea * inv3, vadd(vadd(p1, p2), p3)));
    }
    c = vscale(1.0f / area, c);
    return c;
}
/* b2PolygonShape::Set (2.3.1: weld, gift-wrap hull from the right-most/lowest point, CCW) */
static int shape_set_poly(shape_t *s, const v2 *vertices, int count) {
    memset(s, 0, sizeof(*s));
    s->type = SH_POLY;
    s->radius = B2_POLYGON_RADIUS;
    if (count < 3) return -1;
    int n = count < B2_MAX_POLY ? count : B2_MAX_POLY;
    v2 ps[B2_MAX_POLY];
    int tempCount = 0;
    for (int i = 0; i < n; ++i) {
        v2 v = vertices[i];
        int unique = 1;
        for (int j = 0; j < tempCount; ++j)
            if (vdist2(v, ps[j]) < 0.5f * B2_LINEAR_SLOP) { unique = 0; break; }
        if (unique) ps[tempCount++] = v;
    }
    n = tempCount;
    if (n < 3) return -1;
    int i0 = 0;
    f32 x0 = ps[0].x;
    for (int i = 1; i < n; ++i) {
        f32 x = ps[i].x;
        if (x > x0 || (x == x0 && ps[i].y < ps[i0].y)) { i0 = i; x0 = x; }
    }
    int hull[B2_MAX_POLY], m = 0, ih = i0;
    for (;;) {
        hull[m] = ih;
        int ie = 0;
        for (int j = 1; j < n; ++j) {
            if (ie == ih) { ie = j; continue; }
            v2 r = vsub(ps[ie], ps[hull[m]]), v = vsub(ps[j], ps[hull[m]]);
            f32 c = vcross(r, v);
            if (c < 0.0f) ie = j;
            if (c == 0.0f && vlen2(v) > vlen2(r)) ie = j;
        }
        ++m;
        ih = ie;
        if (ie == i0) break;
        if (m >= B2_MAX_POLY) break;
    }
    s->count = m;
    for (int i = 0; i < m; ++i) s->verts[i] = ps[hull[i]];
    for (int i = 0; i < m; ++i) {
        int i2 = i + 1 < m ? i + 1 : 0;
        v2 edge = vsub(s->verts[i2], s->verts[i]);
        s->normals[i] = vcross_vs(edge, 1.0f);
        vnormalize(&s->normals[i]);
    }
    s->centroid = poly_centroid(s->verts, m);
    return 0;
}
/* b2PolygonShape::ComputeMass (2.3.1), density 1 applied by caller */
static void poly_mass(const shape_t *s, f32 density, f32 *mass, v2 *center_out, f32 *I_out) {
    v2 center = V2(0.0f, 0.0f);
    f32 area = 0.0f, I = 0.0f;
    v2 sref = V2(0.0f, 0.0f);
    for (int i = 0; i < s->count; ++i) sref = vadd(sref, s->verts[i]);
    sref = vscale(1.0f / (f32)s->count, sref);
    const f32 k_inv3 = 1.0f / 3.0f;
    for (int i = 0; i < s->count; ++i) {
        v2 e1 = vsub(s->verts[i], sref);
        v2 e2 = i + 1 < s->count ? vsub(s->verts[i + 1], sref) : vsub(s->verts[0], sref);
        f32 D = vcross(e1, e2);
        f32 triangleArea = 0.5f * D;
        area += triangleArea;
        center = vadd(center, vscale(triangleArea * k_inv3, vadd(e1, e2)));
        f32 ex1 = e1.x, ey1 = e1.y, ex2 = e2.x, ey2 = e2.y;
        f32 intx2 = ex1 * ex1 + ex2 * ex1 + ex2 * ex2;
        f32 inty2 = ey1 * ey1 + ey2 * ey1 + ey2 * ey2;
        I += (0.25f * k_inv3 * D) * (intx2 + inty2);
    }
    *mass = density * area;
    center = vscale(1.0f / area, center);
    v2 mc = vadd(center, sref);
    *center_out = mc;
    f32 Iout = density * I;
    Iout += *mass * (vdot(mc, mc) - vdot(center, center));
    *I_out = Iout;
}
static void circle_mass(f32 r, v2 p, f32 density, f32 *mass, f32 *I) {
    *mass = density * B2_PI * r * r;
    *I = *mass * (0.5f * r * r + vdot(p, p));
}

static aabb_t shape_aabb(const shape_t *s, xf_t xf) {
    aabb_t bb;
    if (s->type == SH_POLY) {
        v2 lower = xmul(xf, s->verts[0]), upper = lower;
        for (int i = 1; i < s->count; ++i) {
            v2 v = xmul(xf, s->verts[i]);
            lower = vmin(lower, v);
            upper = vmax(upper, v);
        }
        v2 r = V2(s->radius, s->radius);
        bb.lower = vsub(lower, r);
        bb.upper = vadd(upper, r);
    } else if (s->type == SH_CIRCLE) {
        v2 p = vadd(xf.p, rmul(xf.q, s->p));
        bb.lower = V2(p.x - s->radius, p.y - s->radius);
        bb.upper = V2(p.x + s->radius, p.y + s->radius);
    } else {
        v2 a = xmul(xf, s->v1), b = xmul(xf, s->v2_);
        v2 lower = vmin(a, b), upper = vmax(a, b);
        v2 r = V2(s->radius, s->radius);
        bb.lower = vsub(lower, r);
        bb.upper = vadd(upper, r);
    }
    return bb;
}
static inline aabb_t aabb_combine(aabb_t a, aabb_t b) {
    aabb_t r;
    r.lower = vmin(a.lower, b.lower);
    r.upper = vmax(a.upper, b.upper);
    return r;
}
static inline int aabb_contains(aabb_t a, aabb_t b) {
    int result = 1;
    result = result && a.lower.x <= b.lower.x;
    result = result && a.lower.y <= b.lower.y;
    result = result && b.upper.x <= a.upper.x;
    result = result && b.upper.y <= a.upper.y;
    return result;
}
static inline int aabb_overlap(aabb_t a, aabb_t b) {
    v2 d1 = vsub(b.lower, a.upper), d2 = vsub(a.lower, b.upper);
    if (d1.x > 0.0f || d1.y > 0.0f) return 0;
    if (d2.x > 0.0f || d2.y > 0.0f) return 0;
    return 1;
}
static inline aabb_t aabb_fatten(aabb_t a) {
    v2 r = V2(B2_AABB_EXTENSION, B2_AABB_EXTENSION);
    aabb_t f;
    f.lower = vsub(a.lower, r);
    f.upper = vadd(a.upper, r);
    return f;
}

/* ---- terrain: static bodies at the origin with one fixture each, shared by all worlds ---- */
typedef struct { shape_t shape; aabb_t fat; } static_t;
struct o_terrain {
    int nstatic, npoly, nedge;
    static_t *statics; /* creation order == broadphase proxy order: hardcore polygons, then edges */
    f32 friction;
    f32 x0, pitch; /* for the candidate range of the edge scan */
};

o_terrain *rem2d_oracle_terrain_create(const float *xs, const float *ys, int npts, const float *polys,
                                       int npoly, float friction) {
    o_terrain *t = (o_terrain *)calloc(1, sizeof(*t));
    t->npoly = npoly;
    t->nedge = npts > 1 ? npts - 1 : 0;
    t->nstatic = t->npoly + t->nedge;
    t->statics = (static_t *)calloc((size_t)(t->nstatic > 0 ? t->nstatic : 1), sizeof(static_t));
    t->friction = friction;
    xf_t id;
    id.p = V2(0.0f, 0.0f);
    id.q = rot_set(0.0f);
    for (int i = 0; i < npoly; ++i) {
        v2 vs[4];
        for (int k = 0; k < 4; ++k) vs[k] = V2(polys[(i * 4 + k) * 2], polys[(i * 4 + k) * 2 + 1]);
        shape_set_poly(&t->statics[i].shape, vs, 4);
        t->statics[i].fat = aabb_fatten(shape_aabb(&t->statics[i].shape, id));
    }
    for (int i = 0; i < t->nedge; ++i) {
        static_t *s = &t->statics[npoly + i];
        memset(&s->shape, 0, sizeof(s->shape));
        s->shape.type = SH_EDGE;
        s->shape.radius = B2_POLYGON_RADIUS;
        s->shape.v1 = V2(xs[i], ys[i]);
        s->shape.v2_ = V2(xs[i + 1], ys[i + 1]);
        s->fat = aabb_fatten(shape_aabb(&s->shape, id));
    }
    return t;
}
void rem2d_oracle_terrain_destroy(o_terrain *t) {
    if (!t) return;
    free(t->statics);
    free(t);
}

/* ---- world ---- */
typedef struct {
    shape_t shape;
    xf_t xf;
    v2 localCenter, c0, c;
    f32 a0, a, alpha0;
    v2 v;
    f32 w;
    f32 mass, invMass, I, invI;
    f32 friction;
    int awake, islandFlag, islandIndex;
    f32 sleepTime;
    aabb_t fat;
    int ncontacts, contacts[O_MAX_BODY_CONTACTS]; /* m_contactList order, head first */
    int njoints, joints[O_MAX_BODIES];            /* m_jointList order, head first */
} body_t;

enum { LIM_INACTIVE = 0, LIM_AT_LOWER = 1, LIM_AT_UPPER = 2, LIM_EQUAL = 3 };
typedef struct {
    int bodyA, bodyB;
    v2 localAnchorA, localAnchorB;
    f32 referenceAngle;
    v3 impulse;
    f32 motorImpulse;
    int enableMotor, enableLimit;
    f32 lowerAngle, upperAngle, maxMotorTorque, motorSpeed;
    int limitState;
    int islandFlag;
    /* solver temp */
    int indexA, indexB;
    v2 rA, rB, localCenterA, localCenterB;
    f32 invMassA, invMassB, invIA, invIB;
    v3 mex, mey, mez; /* m_mass columns */
    f32 motorMass;
    /* controller (Controller/m_controller.py) of the node that owns body B */
    double amp, phase, freq, offset, istate;
} joint_t;

typedef struct {
    int used;
    int staticIdx, body; /* fixture A = terrain static, fixture B = module */
    manifold_t m;
    int touching, enabled, islandFlag, toiFlag, toiCount;
    f32 toi;
    f32 friction, restitution;
} contact_t;

#define O_MAX_CONTACTS (O_MAX_BODIES * O_MAX_BODY_CONTACTS)
#define O_MAX_STATICS 512

typedef struct { v2 c; f32 a; } pos_t;
typedef struct { v2 v; f32 w; } vel_t;

struct o_world {
    const o_terrain *terrain;
    unsigned flags;
    v2 gravity;
    int nbody, njoint;
    body_t bodies[O_MAX_BODIES];
    joint_t joints[O_MAX_BODIES];
    contact_t contacts[O_MAX_CONTACTS];
    int wcount, wlist[O_MAX_CONTACTS]; /* world m_contactList order, head first */
    int nmoved, moved[O_MAX_BODIES];   /* broadphase move buffer (body proxies only) */
    int newFixture, stepComplete;
    f32 inv_dt0;
    /* last island bookkeeping (for index-parity tests) */
    int islandJointCount, islandJoints[O_MAX_BODIES];
    int lastPositionIterations;
    int toiEvents;
    /* b2World::SolveTOI bookkeeping on the static (terrain) bodies: every terrain edge / hardcore box is its
     * own b2Body, so each carries m_sweep.alpha0 and e_islandFlag (A.8) */
    f32 staticAlpha0[O_MAX_STATICS];
    unsigned char staticIslandFlag[O_MAX_STATICS];
    int toiDynamicAdvances; /* times a static's alpha0 forced bB->m_sweep.Advance (provably never) */
    /* Modular2D state */
    double wod;
    int overflow;
    /* bookkeeping only (no arithmetic reads them): the most pairs / touching manifolds any body held right after a
     * Collide or inside a TOI island -- what a fixed-capacity engine (the HIP builds: 24 / 6 and 32 / 12 slots per body)
     * would have needed; tests that exercise the overflow tiers ask for them (rem2d_oracle_batch_run_caps) */
    int maxBodyPairs, maxBodyTouching;
};

static void body_set_awake(o_world *w, body_t *b, int flag) {
    if (flag) {
        if (w->flags & O_FLAG_SLEEP_RESET_ALWAYS) {
            b->awake = 1;
            b->sleepTime = 0.0f;
        } else if (!b->awake) {
            b->awake = 1;
            b->sleepTime = 0.0f;
        }
    } else {
        b->awake = 0;
        b->sleepTime = 0.0f;
        b->v = V2(0.0f, 0.0f);
        b->w = 0.0f;
    }
}

o_world *rem2d_oracle_world_create(const o_terrain *t, unsigned flags) {
    o_world *w = (o_world *)calloc(1, sizeof(*w));
    w->terrain = t;
    w->flags = flags;
    w->gravity = V2(0.0f, -10.0f); /* pybox2d b2World() default, A.1 */
    w->stepComplete = 1;
    w->inv_dt0 = 0.0f;
    return w;
}
void rem2d_oracle_world_destroy(o_world *w) { free(w); }
void rem2d_oracle_set_gravity(o_world *w, float gx, float gy) { w->gravity = V2(gx, gy); }

/* world.CreateDynamicBody(position, angle, fixtures=fixtureDef(shape, density=1, friction=0.1)) */
static int add_body(o_world *w, const shape_t *shape, f32 x, f32 y, f32 angle) {
    if (w->nbody >= O_MAX_BODIES) return -1;
    int idx = w->nbody++;
    body_t *b = &w->bodies[idx];
    memset(b, 0, sizeof(*b));
    b->shape = *shape;
    b->xf.p = V2(x, y);
    b->xf.q = rot_set(angle);
    b->a0 = b->a = angle;
    b->c0 = b->c = b->xf.p;
    b->awake = 1;
    b->friction = 0.1f;
    /* b2Body::ResetMassData */
    f32 mass, I;
    v2 center;
    if (shape->type == SH_POLY) poly_mass(shape, 1.0f, &mass, &center, &I);
    else { circle_mass(shape->radius, shape->p, 1.0f, &mass, &I); center = shape->p; }
    b->mass = 0.0f;
    b->I = 0.0f;
    v2 localCenter = V2(0.0f, 0.0f);
    b->mass += mass;
    localCenter = vadd(localCenter, vscale(mass, center));
    b->I += I;
    if (b->mass > 0.0f) {
        b->invMass = 1.0f / b->mass;
        localCenter = vscale(b->invMass, localCenter);
    } else {
        b->mass = 1.0f;
        b->invMass = 1.0f;
    }
    if (b->I > 0.0f) {
        b->I -= b->mass * vdot(localCenter, localCenter);
        b->invI = 1.0f / b->I;
    } else {
        b->I = 0.0f;
        b->invI = 0.0f;
    }
    b->localCenter = localCenter;
    b->c0 = b->c = xmul(b->xf, b->localCenter);
    /* b2Fixture::CreateProxies: fat AABB, BufferMove */
    b->fat = aabb_fatten(shape_aabb(&b->shape, b->xf));
    w->moved[w->nmoved++] = idx;
    w->newFixture = 1;
    return idx;
}
int rem2d_oracle_add_box(o_world *w, float hx, float hy, float x, float y, float angle) {
    shape_t s;
    shape_set_box(&s, hx, hy);
    return add_body(w, &s, x, y, angle);
}
int rem2d_oracle_add_circle(o_world *w, float r, float x, float y, float angle) {
    shape_t s;
    memset(&s, 0, sizeof(s));
    s.type = SH_CIRCLE;
    s.radius = r;
    s.p = V2(0.0f, 0.0f);
    return add_body(w, &s, x, y, angle);
}
/* world.CreateJoint(revoluteJointDef(bodyA, bodyB, localAnchorA, localAnchorB, enableMotor=True,
 * enableLimit=True, maxMotorTorque, motorSpeed=0, lowerAngle, upperAngle)), referenceAngle 0 */
int rem2d_oracle_add_joint(o_world *w, int bodyA, int bodyB, float ax, float ay, float bx, float by,
                           float maxTorque, float lower, float upper) {
    if (w->njoint >= O_MAX_BODIES) return -1;
    int idx = w->njoint++;
    joint_t *j = &w->joints[idx];
    memset(j, 0, sizeof(*j));
    j->bodyA = bodyA;
    j->bodyB = bodyB;
    j->localAnchorA = V2(ax, ay);
    j->localAnchorB = V2(bx, by);
    j->referenceAngle = 0.0f;
    j->enableMotor = 1;
    j->enableLimit = 1;
    j->lowerAngle = lower;
    j->upperAngle = upper;
    j->maxMotorTorque = maxTorque;
    j->motorSpeed = 0.0f;
    j->limitState = LIM_INACTIVE;
    /* head-insert the joint edges (A.3) */
    body_t *A = &w->bodies[bodyA], *B = &w->bodies[bodyB];
    memmove(A->joints + 1, A->joints, sizeof(int) * (size_t)A->njoints);
    A->joints[0] = idx;
    A->njoints++;
    memmove(B->joints + 1, B->joints, sizeof(int) * (size_t)B->njoints);
    B->joints[0] = idx;
    B->njoints++;
    return idx;
}
void rem2d_oracle_set_controller(o_world *w, int joint, double amp, double phase, double freq,
                                 double offset, double istate) {
    joint_t *j = &w->joints[joint];
    j->amp = amp;
    j->phase = phase;
    j->freq = freq;
    j->offset = offset;
    j->istate = istate;
}
/* b2RevoluteJoint::SetMotorSpeed */
void rem2d_oracle_set_motor_speed(o_world *w, int joint, float speed) {
    joint_t *j = &w->joints[joint];
    body_set_awake(w, &w->bodies[j->bodyA], 1);
    body_set_awake(w, &w->bodies[j->bodyB], 1);
    j->motorSpeed = speed;
}
void rem2d_oracle_set_velocity(o_world *w, int body, float vx, float vy, float wz) {
    body_t *b = &w->bodies[body];
    if (vx != 0.0f || vy != 0.0f || wz != 0.0f) body_set_awake(w, b, 1);
    b->v = V2(vx, vy);
    b->w = wz;
}

/* ---- state re-synchronisation (tests/test_box2d_pin.py: SURVEY 8c protocol (i), "identical full state in, one step out") ----
 * Setters for exactly the state pybox2d lets a script READ from a b2World (body pose / velocity / awake flag, the revolute joint's
 * accumulated impulses, a manifold's warm-start impulses): the pin test runs this oracle beside a recorded trajectory of the real
 * engine and, before every step, overwrites these floats with the recording's so that differences cannot accumulate across
 * steps.  What pybox2d does not expose (sleep timers, fat AABBs, limit states, the contact lists' order) stays the oracle's own.
 * No arithmetic of the step reads anything these functions compute except b2Rot::Set of the angle, which b2Body::SetTransform
 * does the same way.  Test infrastructure like the rest of this file. */
void rem2d_oracle_set_body_state(o_world *w, int body, float x, float y, float angle, float vx, float vy, float wz, int awake) {
    body_t *b = &w->bodies[body];
    /* b2Body::SetTransform: xf, then sweep.c = xf * localCenter, a, c0 = c, a0 = a (fixtures' proxies are NOT re-synchronised
     * here: the fat AABB is the oracle's own, see above) */
    b->xf.q = rot_set(angle);
    b->xf.p = V2(x, y);
    b->c = xmul(b->xf, b->localCenter);
    b->a = angle;
    b->c0 = b->c;
    b->a0 = angle;
    b->v = V2(vx, vy);
    b->w = wz;
    b->awake = awake ? 1 : 0;
}
void rem2d_oracle_set_joint_impulses(o_world *w, int joint, float ix, float iy, float iz, float motorImpulse) {
    joint_t *j = &w->joints[joint];
    j->impulse.x = ix;
    j->impulse.y = iy;
    j->impulse.z = iz;
    j->motorImpulse = motorImpulse;
}
/* contact k of the body's list (rem2d_oracle_get_contacts order) */
void rem2d_oracle_set_contact_impulses(o_world *w, int body, int k, float n0, float n1, float t0, float t1) {
    body_t *b = &w->bodies[body];
    if (k < 0 || k >= b->ncontacts) return;
    contact_t *c = &w->contacts[b->contacts[k]];
    c->m.points[0].normalImpulse = n0;
    c->m.points[1].normalImpulse = n1;
    c->m.points[0].tangentImpulse = t0;
    c->m.points[1].tangentImpulse = t1;
}
double rem2d_oracle_get_wod(const o_world *w) { return w->wod; }
/* out[njoint] = the controllers' i_state (Controller/m_controller.py:17-21) */
void rem2d_oracle_get_controller_state(const o_world *w, double *out) {
    for (int i = 0; i < w->njoint; ++i) out[i] = w->joints[i].istate;
}

/* ---- A.7 narrowphase ---- */
static void collide_edge_circle(manifold_t *m, const shape_t *edgeA, xf_t xfA, const shape_t *circleB,
                                xf_t xfB) {
    m->pointCount = 0;
    v2 Q = xmulT(xfA, xmul(xfB, circleB->p));
    v2 A = edgeA->v1, B = edgeA->v2_;
    v2 e = vsub(B, A);
    f32 u = vdot(e, vsub(B, Q));
    f32 v = vdot(e, vsub(Q, A));
    f32 radius = edgeA->radius + circleB->radius;
    cid_t id;
    id.key = 0;
    id.cf.indexB = 0;
    id.cf.typeB = CF_VERTEX;
    if (v <= 0.0f) {
        v2 P = A;
        v2 d = vsub(Q, P);
        f32 dd = vdot(d, d);
        if (dd > radius * radius) return;
        id.cf.indexA = 0;
        id.cf.typeA = CF_VERTEX;
        m->pointCount = 1;
        m->type = MF_CIRCLES;
        m->localNormal = V2(0.0f, 0.0f);
        m->localPoint = P;
        m->points[0].id = id;
        m->points[0].localPoint = circleB->p;
        return;
    }
    if (u <= 0.0f) {
        v2 P = B;
        v2 d = vsub(Q, P);
        f32 dd = vdot(d, d);
        if (dd > radius * radius) return;
        id.cf.indexA = 1;
        id.cf.typeA = CF_VERTEX;
        m->pointCount = 1;
        m->type = MF_CIRCLES;
        m->localNormal = V2(0.0f, 0.0f);
        m->localPoint = P;
        m->points[0].id = id;
        m->points[0].localPoint = circleB->p;
        return;
    }
    f32 den = vdot(e, e);
    v2 P = vscale(1.0f / den, vadd(vscale(u, A), vscale(v, B)));
    v2 d = vsub(Q, P);
    f32 dd = vdot(d, d);
    if (dd > radius * radius) return;
    v2 n = V2(-e.y, e.x);
    if (vdot(n, vsub(Q, A)) < 0.0f) n = V2(-n.x, -n.y);
    vnormalize(&n);
    id.cf.indexA = 0;
    id.cf.typeA = CF_FACE;
    m->pointCount = 1;
    m->type = MF_FACE_A;
    m->localNormal = n;
    m->localPoint = A;
    m->points[0].id = id;
    m->points[0].localPoint = circleB->p;
}

typedef struct { v2 v; cid_t id; } clipv_t;
static int clip_segment(clipv_t vOut[2], const clipv_t vIn[2], v2 normal, f32 offset, int vertexIndexA) {
    int numOut = 0;
    f32 distance0 = vdot(normal, vIn[0].v) - offset;
    f32 distance1 = vdot(normal, vIn[1].v) - offset;
    if (distance0 <= 0.0f) vOut[numOut++] = vIn[0];
    if (distance1 <= 0.0f) vOut[numOut++] = vIn[1];
    if (distance0 * distance1 < 0.0f) {
        f32 interp = distance0 / (distance0 - distance1);
        vOut[numOut].v = vadd(vIn[0].v, vscale(interp, vsub(vIn[1].v, vIn[0].v)));
        vOut[numOut].id.cf.indexA = (uint8_t)vertexIndexA;
        vOut[numOut].id.cf.indexB = vIn[0].id.cf.indexB;
        vOut[numOut].id.cf.typeA = CF_VERTEX;
        vOut[numOut].id.cf.typeB = CF_FACE;
        ++numOut;
    }
    return numOut;
}

enum { EP_UNKNOWN = 0, EP_EDGE_A = 1, EP_EDGE_B = 2 };
/* b2EPCollider::Collide for an isolated edge (m_hasVertex0 = m_hasVertex3 = false) */
static void collide_edge_polygon(manifold_t *manifold, const shape_t *edgeA, xf_t xfA,
                                 const shape_t *polygonB, xf_t xfB) {
    xf_t xf = xxmulT(xfA, xfB);
    v2 centroidB = xmul(xf, polygonB->centroid);
    v2 v1 = edgeA->v1, v2p = edgeA->v2_;
    v2 edge1 = vsub(v2p, v1);
    vnormalize(&edge1);
    v2 normal1 = V2(edge1.y, -edge1.x);
    f32 offset1 = vdot(normal1, vsub(centroidB, v1));
    int front = offset1 >= 0.0f;
    v2 normal, lowerLimit, upperLimit;
    if (front) {
        normal = normal1;
        lowerLimit = vneg(normal1);
        upperLimit = vneg(normal1);
    } else {
        normal = vneg(normal1);
        lowerLimit = normal1;
        upperLimit = normal1;
    }
    int count = polygonB->count;
    v2 pv[B2_MAX_POLY], pn[B2_MAX_POLY];
    for (int i = 0; i < count; ++i) {
        pv[i] = xmul(xf, polygonB->verts[i]);
        pn[i] = rmul(xf.q, polygonB->normals[i]);
    }
    f32 radius = 2.0f * B2_POLYGON_RADIUS;
    manifold->pointCount = 0;
    /* ComputeEdgeSeparation */
    int edgeAxisIndex = front ? 0 : 1;
    f32 edgeSep = FLT_MAX;
    (void)edgeAxisIndex;
    for (int i = 0; i < count; ++i) {
        f32 s = vdot(normal, vsub(pv[i], v1));
        if (s < edgeSep) edgeSep = s;
    }
    if (edgeSep > radius) return;
    /* ComputePolygonSeparation */
    int polyType = EP_UNKNOWN, polyIndex = -1;
    f32 polySep = -FLT_MAX;
    v2 perp = V2(-normal.y, normal.x);
    for (int i = 0; i < count; ++i) {
        v2 n = vneg(pn[i]);
        f32 s1 = vdot(n, vsub(pv[i], v1));
        f32 s2 = vdot(n, vsub(pv[i], v2p));
        f32 s = fmin32(s1, s2);
        if (s > radius) {
            polyType = EP_EDGE_B;
            polyIndex = i;
            polySep = s;
            break;
        }
        if (vdot(n, perp) >= 0.0f) {
            if (vdot(vsub(n, upperLimit), normal) < -B2_ANGULAR_SLOP) continue;
        } else {
            if (vdot(vsub(n, lowerLimit), normal) < -B2_ANGULAR_SLOP) continue;
        }
        if (s > polySep) {
            polyType = EP_EDGE_B;
            polyIndex = i;
            polySep = s;
        }
    }
    if (polyType != EP_UNKNOWN && polySep > radius) return;
    const f32 k_relativeTol = 0.98f, k_absoluteTol = 0.001f;
    int primaryType, primaryIndex;
    if (polyType == EP_UNKNOWN) {
        primaryType = EP_EDGE_A;
        primaryIndex = edgeAxisIndex;
    } else if (polySep > k_relativeTol * edgeSep + k_absoluteTol) {
        primaryType = EP_EDGE_B;
        primaryIndex = polyIndex;
    } else {
        primaryType = EP_EDGE_A;
        primaryIndex = edgeAxisIndex;
    }
    clipv_t ie[2];
    int rf_i1, rf_i2;
    v2 rf_v1, rf_v2, rf_normal;
    if (primaryType == EP_EDGE_A) {
        manifold->type = MF_FACE_A;
        int bestIndex = 0;
        f32 bestValue = vdot(normal, pn[0]);
        for (int i = 1; i < count; ++i) {
            f32 value = vdot(normal, pn[i]);
            if (value < bestValue) { bestValue = value; bestIndex = i; }
        }
        int i1 = bestIndex, i2 = i1 + 1 < count ? i1 + 1 : 0;
        ie[0].v = pv[i1];
        ie[0].id.cf.indexA = 0;
        ie[0].id.cf.indexB = (uint8_t)i1;
        ie[0].id.cf.typeA = CF_FACE;
        ie[0].id.cf.typeB = CF_VERTEX;
        ie[1].v = pv[i2];
        ie[1].id.cf.indexA = 0;
        ie[1].id.cf.indexB = (uint8_t)i2;
        ie[1].id.cf.typeA = CF_FACE;
        ie[1].id.cf.typeB = CF_VERTEX;
        if (front) {
            rf_i1 = 0; rf_i2 = 1; rf_v1 = v1; rf_v2 = v2p; rf_normal = normal1;
        } else {
            rf_i1 = 1; rf_i2 = 0; rf_v1 = v2p; rf_v2 = v1; rf_normal = vneg(normal1);
        }
    } else {
        manifold->type = MF_FACE_B;
        ie[0].v = v1;
        ie[0].id.cf.indexA = 0;
        ie[0].id.cf.indexB = (uint8_t)primaryIndex;
        ie[0].id.cf.typeA = CF_VERTEX;
        ie[0].id.cf.typeB = CF_FACE;
        ie[1].v = v2p;
        ie[1].id.cf.indexA = 0;
        ie[1].id.cf.indexB = (uint8_t)primaryIndex;
        ie[1].id.cf.typeA = CF_VERTEX;
        ie[1].id.cf.typeB = CF_FACE;
        rf_i1 = primaryIndex;
        rf_i2 = rf_i1 + 1 < count ? rf_i1 + 1 : 0;
        rf_v1 = pv[rf_i1];
        rf_v2 = pv[rf_i2];
        rf_normal = pn[rf_i1];
    }
    v2 sideNormal1 = V2(rf_normal.y, -rf_normal.x);
    v2 sideNormal2 = vneg(sideNormal1);
    f32 sideOffset1 = vdot(sideNormal1, rf_v1);
    f32 sideOffset2 = vdot(sideNormal2, rf_v2);
    clipv_t clip1[2], clip2[2];
    int np = clip_segment(clip1, ie, sideNormal1, sideOffset1, rf_i1);
    if (np < 2) return;
    np = clip_segment(clip2, clip1, sideNormal2, sideOffset2, rf_i2);
    if (np < 2) return;
    if (primaryType == EP_EDGE_A) {
        manifold->localNormal = rf_normal;
        manifold->localPoint = rf_v1;
    } else {
        manifold->localNormal = polygonB->normals[rf_i1];
        manifold->localPoint = polygonB->verts[rf_i1];
    }
    int pointCount = 0;
    for (int i = 0; i < 2; ++i) {
        f32 separation = vdot(rf_normal, vsub(clip2[i].v, rf_v1));
        if (separation <= radius) {
            mpoint_t *cp = &manifold->points[pointCount];
            if (primaryType == EP_EDGE_A) {
                cp->localPoint = xmulT(xf, clip2[i].v);
                cp->id = clip2[i].id;
            } else {
                cp->localPoint = clip2[i].v;
                cp->id.cf.typeA = clip2[i].id.cf.typeB;
                cp->id.cf.typeB = clip2[i].id.cf.typeA;
                cp->id.cf.indexA = clip2[i].id.cf.indexB;
                cp->id.cf.indexB = clip2[i].id.cf.indexA;
            }
            ++pointCount;
        }
    }
    manifold->pointCount = pointCount;
}

/* b2FindMaxSeparation (2.3.1 exhaustive form) */
static f32 find_max_separation(int *edgeIndex, const shape_t *poly1, xf_t xf1, const shape_t *poly2,
                               xf_t xf2) {
    xf_t xf = xxmulT(xf2, xf1);
    int bestIndex = 0;
    f32 maxSeparation = -B2_MAXFLOAT;
    for (int i = 0; i < poly1->count; ++i) {
        v2 n = rmul(xf.q, poly1->normals[i]);
        v2 v1 = xmul(xf, poly1->verts[i]);
        f32 si = B2_MAXFLOAT;
        for (int j = 0; j < poly2->count; ++j) {
            f32 sij = vdot(n, vsub(poly2->verts[j], v1));
            if (sij < si) si = sij;
        }
        if (si > maxSeparation) { maxSeparation = si; bestIndex = i; }
    }
    *edgeIndex = bestIndex;
    return maxSeparation;
}
static void find_incident_edge(clipv_t c[2], const shape_t *poly1, xf_t xf1, int edge1, const shape_t *poly2,
                               xf_t xf2) {
    v2 normal1 = rmulT(xf2.q, rmul(xf1.q, poly1->normals[edge1]));
    int index = 0;
    f32 minDot = B2_MAXFLOAT;
    for (int i = 0; i < poly2->count; ++i) {
        f32 dot = vdot(normal1, poly2->normals[i]);
        if (dot < minDot) { minDot = dot; index = i; }
    }
    int i1 = index, i2 = i1 + 1 < poly2->count ? i1 + 1 : 0;
    c[0].v = xmul(xf2, poly2->verts[i1]);
    c[0].id.cf.indexA = (uint8_t)edge1;
    c[0].id.cf.indexB = (uint8_t)i1;
    c[0].id.cf.typeA = CF_FACE;
    c[0].id.cf.typeB = CF_VERTEX;
    c[1].v = xmul(xf2, poly2->verts[i2]);
    c[1].id.cf.indexA = (uint8_t)edge1;
    c[1].id.cf.indexB = (uint8_t)i2;
    c[1].id.cf.typeA = CF_FACE;
    c[1].id.cf.typeB = CF_VERTEX;
}
/* b2CollidePolygons (2.3.1 flip rule: sepB > sepA + 0.1*linearSlop) */
static void collide_polygons(manifold_t *manifold, const shape_t *polyA, xf_t xfA, const shape_t *polyB,
                             xf_t xfB) {
    manifold->pointCount = 0;
    f32 totalRadius = polyA->radius + polyB->radius;
    int edgeA = 0;
    f32 separationA = find_max_separation(&edgeA, polyA, xfA, polyB, xfB);
    if (separationA > totalRadius) return;
    int edgeB = 0;
    f32 separationB = find_max_separation(&edgeB, polyB, xfB, polyA, xfA);
    if (separationB > totalRadius) return;
    const shape_t *poly1, *poly2;
    xf_t xf1, xf2;
    int edge1, flip;
    const f32 k_tol = 0.1f * B2_LINEAR_SLOP;
    if (separationB > separationA + k_tol) {
        poly1 = polyB; poly2 = polyA; xf1 = xfB; xf2 = xfA; edge1 = edgeB;
        manifold->type = MF_FACE_B;
        flip = 1;
    } else {
        poly1 = polyA; poly2 = polyB; xf1 = xfA; xf2 = xfB; edge1 = edgeA;
        manifold->type = MF_FACE_A;
        flip = 0;
    }
    clipv_t incidentEdge[2];
    find_incident_edge(incidentEdge, poly1, xf1, edge1, poly2, xf2);
    int count1 = poly1->count;
    int iv1 = edge1, iv2 = edge1 + 1 < count1 ? edge1 + 1 : 0;
    v2 v11 = poly1->verts[iv1], v12 = poly1->verts[iv2];
    v2 localTangent = vsub(v12, v11);
    vnormalize(&localTangent);
    v2 localNormal = vcross_vs(localTangent, 1.0f);
    v2 planePoint = vscale(0.5f, vadd(v11, v12));
    v2 tangent = rmul(xf1.q, localTangent);
    v2 normal = vcross_vs(tangent, 1.0f);
    v11 = xmul(xf1, v11);
    v12 = xmul(xf1, v12);
    f32 frontOffset = vdot(normal, v11);
    f32 sideOffset1 = -vdot(tangent, v11) + totalRadius;
    f32 sideOffset2 = vdot(tangent, v12) + totalRadius;
    clipv_t clip1[2], clip2[2];
    int np = clip_segment(clip1, incidentEdge, vneg(tangent), sideOffset1, iv1);
    if (np < 2) return;
    np = clip_segment(clip2, clip1, tangent, sideOffset2, iv2);
    if (np < 2) return;
    manifold->localNormal = localNormal;
    manifold->localPoint = planePoint;
    int pointCount = 0;
    for (int i = 0; i < 2; ++i) {
        f32 separation = vdot(normal, clip2[i].v) - frontOffset;
        if (separation <= totalRadius) {
            mpoint_t *cp = &manifold->points[pointCount];
            cp->localPoint = xmulT(xf2, clip2[i].v);
            cp->id = clip2[i].id;
            if (flip) {
                cid_t cf = cp->id;
                cp->id.cf.indexA = cf.cf.indexB;
                cp->id.cf.indexB = cf.cf.indexA;
                cp->id.cf.typeA = cf.cf.typeB;
                cp->id.cf.typeB = cf.cf.typeA;
            }
            ++pointCount;
        }
    }
    manifold->pointCount = pointCount;
}
/* b2CollidePolygonAndCircle */
static void collide_polygon_circle(manifold_t *m, const shape_t *polygonA, xf_t xfA, const shape_t *circleB,
                                   xf_t xfB) {
    m->pointCount = 0;
    v2 c = xmul(xfB, circleB->p);
    v2 cLocal = xmulT(xfA, c);
    int normalIndex = 0;
    f32 separation = -B2_MAXFLOAT;
    f32 radius = polygonA->radius + circleB->radius;
    int vertexCount = polygonA->count;
    for (int i = 0; i < vertexCount; ++i) {
        f32 s = vdot(polygonA->normals[i], vsub(cLocal, polygonA->verts[i]));
        if (s > radius) return;
        if (s > separation) { separation = s; normalIndex = i; }
    }
    int vertIndex1 = normalIndex, vertIndex2 = vertIndex1 + 1 < vertexCount ? vertIndex1 + 1 : 0;
    v2 v1 = polygonA->verts[vertIndex1], v2p = polygonA->verts[vertIndex2];
    if (separation < B2_EPSILON) {
        m->pointCount = 1;
        m->type = MF_FACE_A;
        m->localNormal = polygonA->normals[normalIndex];
        m->localPoint = vscale(0.5f, vadd(v1, v2p));
        m->points[0].localPoint = circleB->p;
        m->points[0].id.key = 0;
        return;
    }
    f32 u1 = vdot(vsub(cLocal, v1), vsub(v2p, v1));
    f32 u2 = vdot(vsub(cLocal, v2p), vsub(v1, v2p));
    if (u1 <= 0.0f) {
        if (vdist2(cLocal, v1) > radius * radius) return;
        m->pointCount = 1;
        m->type = MF_FACE_A;
        m->localNormal = vsub(cLocal, v1);
        vnormalize(&m->localNormal);
        m->localPoint = v1;
        m->points[0].localPoint = circleB->p;
        m->points[0].id.key = 0;
    } else if (u2 <= 0.0f) {
        if (vdist2(cLocal, v2p) > radius * radius) return;
        m->pointCount = 1;
        m->type = MF_FACE_A;
        m->localNormal = vsub(cLocal, v2p);
        vnormalize(&m->localNormal);
        m->localPoint = v2p;
        m->points[0].localPoint = circleB->p;
        m->points[0].id.key = 0;
    } else {
        v2 faceCenter = vscale(0.5f, vadd(v1, v2p));
        f32 sep = vdot(vsub(cLocal, faceCenter), polygonA->normals[vertIndex1]);
        if (sep > radius) return;
        m->pointCount = 1;
        m->type = MF_FACE_A;
        m->localNormal = polygonA->normals[vertIndex1];
        m->localPoint = faceCenter;
        m->points[0].localPoint = circleB->p;
        m->points[0].id.key = 0;
    }
}

static xf_t static_xf(void) {
    xf_t id;
    id.p = V2(0.0f, 0.0f);
    id.q.s = 0.0f;
    id.q.c = 1.0f;
    return id;
}
static void contact_evaluate(const o_world *w, const contact_t *c, manifold_t *m, xf_t xfA, xf_t xfB) {
    const shape_t *sa = &w->terrain->statics[c->staticIdx].shape;
    const shape_t *sb = &w->bodies[c->body].shape;
    if (sa->type == SH_EDGE) {
        if (sb->type == SH_POLY) collide_edge_polygon(m, sa, xfA, sb, xfB);
        else collide_edge_circle(m, sa, xfA, sb, xfB);
    } else {
        if (sb->type == SH_POLY) collide_polygons(m, sa, xfA, sb, xfB);
        else collide_polygon_circle(m, sa, xfA, sb, xfB);
    }
}
/* b2Contact::Update */
static void contact_update(o_world *w, contact_t *c) {
    manifold_t oldManifold = c->m;
    c->enabled = 1;
    int wasTouching = c->touching;
    body_t *B = &w->bodies[c->body];
    contact_evaluate(w, c, &c->m, static_xf(), B->xf);
    int touching = c->m.pointCount > 0;
    for (int i = 0; i < c->m.pointCount; ++i) {
        mpoint_t *mp2 = &c->m.points[i];
        mp2->normalImpulse = 0.0f;
        mp2->tangentImpulse = 0.0f;
        cid_t id2 = mp2->id;
        for (int j = 0; j < oldManifold.pointCount; ++j) {
            mpoint_t *mp1 = &oldManifold.points[j];
            if (mp1->id.key == id2.key) {
                mp2->normalImpulse = mp1->normalImpulse;
                mp2->tangentImpulse = mp1->tangentImpulse;
                break;
            }
        }
    }
    if (touching != wasTouching) body_set_awake(w, B, 1);
    c->touching = touching;
}

/* ---- A.2/A.3 contact manager ---- */
static void contact_destroy(o_world *w, int ci) {
    contact_t *c = &w->contacts[ci];
    body_t *B = &w->bodies[c->body];
    int k;
    for (k = 0; k < w->wcount; ++k)
        if (w->wlist[k] == ci) break;
    if (k < w->wcount) {
        memmove(w->wlist + k, w->wlist + k + 1, sizeof(int) * (size_t)(w->wcount - k - 1));
        w->wcount--;
    }
    for (k = 0; k < B->ncontacts; ++k)
        if (B->contacts[k] == ci) break;
    if (k < B->ncontacts) {
        memmove(B->contacts + k, B->contacts + k + 1, sizeof(int) * (size_t)(B->ncontacts - k - 1));
        B->ncontacts--;
    }
    if (c->m.pointCount > 0) body_set_awake(w, B, 1);
    c->used = 0;
}
/* b2ContactManager::AddPair for (terrain static s, body b) */
static void add_pair(o_world *w, int s, int b) {
    body_t *B = &w->bodies[b];
    for (int k = 0; k < B->ncontacts; ++k)
        if (w->contacts[B->contacts[k]].staticIdx == s) return;
    if (B->ncontacts >= O_MAX_BODY_CONTACTS) { w->overflow++; return; }
    int ci = -1;
    for (int k = 0; k < O_MAX_CONTACTS; ++k)
        if (!w->contacts[k].used) { ci = k; break; }
    if (ci < 0) { w->overflow++; return; }
    contact_t *c = &w->contacts[ci];
    memset(c, 0, sizeof(*c));
    c->used = 1;
    c->staticIdx = s;
    c->body = b;
    c->enabled = 1;
    c->toi = 1.0f;
    c->friction = sqrtf(w->terrain->friction * B->friction); /* b2MixFriction */
    c->restitution = 0.0f;                                    /* b2MixRestitution = max(0,0) */
    memmove(w->wlist + 1, w->wlist, sizeof(int) * (size_t)w->wcount);
    w->wlist[0] = ci;
    w->wcount++;
    memmove(B->contacts + 1, B->contacts, sizeof(int) * (size_t)B->ncontacts);
    B->contacts[0] = ci;
    B->ncontacts++;
    body_set_awake(w, B, 1);
}
/* b2ContactManager::FindNewContacts -> b2BroadPhase::UpdatePairs restricted to (static, dynamic)
 * pairs: module/module pairs are rejected by the 0x0020/0x0001 filter (simple_module.py:291-292),
 * static/static pairs by b2Body::ShouldCollide.  Pairs are created in ascending (proxyA, proxyB). */
static void find_new_contacts(o_world *w) {
    if (w->nmoved == 0) return;
    const o_terrain *t = w->terrain;
    int ismoved[O_MAX_BODIES];
    memset(ismoved, 0, sizeof(ismoved));
    for (int k = 0; k < w->nmoved; ++k) ismoved[w->moved[k]] = 1;
    for (int s = 0; s < t->nstatic; ++s) {
        for (int b = 0; b < w->nbody; ++b) {
            if (!ismoved[b]) continue;
            if (aabb_overlap(t->statics[s].fat, w->bodies[b].fat)) add_pair(w, s, b);
        }
    }
    w->nmoved = 0;
}
/* On the first Step the terrain proxies are in the move buffer too; their queries return the
 * module proxies they overlap, which is the same pair set as the module queries. */

/* b2ContactManager::Collide */
static void collide(o_world *w) {
    int k = 0;
    while (k < w->wcount) {
        int ci = w->wlist[k];
        contact_t *c = &w->contacts[ci];
        body_t *B = &w->bodies[c->body];
        int activeB = B->awake;
        if (!activeB) { ++k; continue; }
        if (!aabb_overlap(w->terrain->statics[c->staticIdx].fat, B->fat)) {
            contact_destroy(w, ci); /* list shifts down; k now addresses the next contact */
            continue;
        }
        contact_update(w, c);
        ++k;
    }
}

/* ---- A.5 contact solver ---- */
typedef struct { v2 rA, rB; f32 normalImpulse, tangentImpulse, normalMass, tangentMass, velocityBias; } vcp_t;
typedef struct {
    vcp_t points[2];
    v2 normal;
    f32 nm_exx, nm_exy, nm_eyx, nm_eyy; /* normalMass (b2Mat22) */
    f32 K_exx, K_exy, K_eyx, K_eyy;
    int indexA, indexB;
    f32 invMassA, invMassB, invIA, invIB, friction, restitution, tangentSpeed;
    int pointCount, contactIndex;
} vc_t;
typedef struct {
    v2 localPoints[2], localNormal, localPoint;
    int indexA, indexB;
    f32 invMassA, invMassB;
    v2 localCenterA, localCenterB;
    f32 invIA, invIB;
    int type;
    f32 radiusA, radiusB;
    int pointCount;
} pc_t;

#define IDX_STATIC O_MAX_BODIES /* solver slot of the (shared) static body: c=(0,0), a=0, v=0, w=0 */

typedef struct {
    int nbody, bodies[O_MAX_BODIES];
    int ncontact, contacts[O_MAX_CONTACTS];
    int njoint, joints[O_MAX_BODIES];
    pos_t positions[O_MAX_BODIES + 1];
    vel_t velocities[O_MAX_BODIES + 1];
    vc_t vcs[O_MAX_CONTACTS];
    pc_t pcs[O_MAX_CONTACTS];
} island_t;

typedef struct { f32 dt, inv_dt, dtRatio; int velocityIterations, positionIterations, warmStarting; } step_t;

static void world_manifold(const manifold_t *m, xf_t xfA, f32 radiusA, xf_t xfB, f32 radiusB, v2 *normal,
                           v2 points[2]) {
    if (m->pointCount == 0) return;
    switch (m->type) {
    case MF_CIRCLES: {
        *normal = V2(1.0f, 0.0f);
        v2 pointA = xmul(xfA, m->localPoint);
        v2 pointB = xmul(xfB, m->points[0].localPoint);
        if (vdist2(pointA, pointB) > B2_EPSILON * B2_EPSILON) {
            *normal = vsub(pointB, pointA);
            vnormalize(normal);
        }
        v2 cA = vadd(pointA, vscale(radiusA, *normal));
        v2 cB = vsub(pointB, vscale(radiusB, *normal));
        points[0] = vscale(0.5f, vadd(cA, cB));
    } break;
    case MF_FACE_A: {
        *normal = rmul(xfA.q, m->localNormal);
        v2 planePoint = xmul(xfA, m->localPoint);
        for (int i = 0; i < m->pointCount; ++i) {
            v2 clipPoint = xmul(xfB, m->points[i].localPoint);
            v2 cA = vadd(clipPoint, vscale(radiusA - vdot(vsub(clipPoint, planePoint), *normal), *normal));
            v2 cB = vsub(clipPoint, vscale(radiusB, *normal));
            points[i] = vscale(0.5f, vadd(cA, cB));
        }
    } break;
    case MF_FACE_B: {
        *normal = rmul(xfB.q, m->localNormal);
        v2 planePoint = xmul(xfB, m->localPoint);
        for (int i = 0; i < m->pointCount; ++i) {
            v2 clipPoint = xmul(xfA, m->points[i].localPoint);
            v2 cB = vadd(clipPoint, vscale(radiusB - vdot(vsub(clipPoint, planePoint), *normal), *normal));
            v2 cA = vsub(clipPoint, vscale(radiusA, *normal));
            points[i] = vscale(0.5f, vadd(cA, cB));
        }
        *normal = vneg(*normal);
    } break;
    }
}

static void contact_solver_setup(o_world *w, island_t *is, const step_t *step) {
    for (int i = 0; i < is->ncontact; ++i) {
        contact_t *c = &w->contacts[is->contacts[i]];
        const shape_t *shapeA = &w->terrain->statics[c->staticIdx].shape;
        body_t *B = &w->bodies[c->body];
        vc_t *vc = &is->vcs[i];
        pc_t *pc = &is->pcs[i];
        memset(vc, 0, sizeof(*vc));
        memset(pc, 0, sizeof(*pc));
        vc->friction = c->friction;
        vc->restitution = c->restitution;
        vc->tangentSpeed = 0.0f;
        vc->indexA = IDX_STATIC;
        vc->indexB = B->islandIndex;
        vc->invMassA = 0.0f;
        vc->invMassB = B->invMass;
        vc->invIA = 0.0f;
        vc->invIB = B->invI;
        vc->contactIndex = is->contacts[i];
        vc->pointCount = c->m.pointCount;
        pc->indexA = IDX_STATIC;
        pc->indexB = B->islandIndex;
        pc->invMassA = 0.0f;
        pc->invMassB = B->invMass;
        pc->localCenterA = V2(0.0f, 0.0f);
        pc->localCenterB = B->localCenter;
        pc->invIA = 0.0f;
        pc->invIB = B->invI;
        pc->localNormal = c->m.localNormal;
        pc->localPoint = c->m.localPoint;
        pc->pointCount = c->m.pointCount;
        pc->radiusA = shapeA->radius;
        pc->radiusB = B->shape.radius;
        pc->type = c->m.type;
        for (int j = 0; j < c->m.pointCount; ++j) {
            mpoint_t *cp = &c->m.points[j];
            vcp_t *vcp = &vc->points[j];
            if (step->warmStarting) {
                vcp->normalImpulse = step->dtRatio * cp->normalImpulse;
                vcp->tangentImpulse = step->dtRatio * cp->tangentImpulse;
            } else {
                vcp->normalImpulse = 0.0f;
                vcp->tangentImpulse = 0.0f;
            }
            pc->localPoints[j] = cp->localPoint;
        }
    }
}
/* the per-point / per-contact effective masses of b2ContactSolver::InitializeVelocityConstraints, from the world
 * manifold points (shared with the known-answer entry point rem2d_oracle_kat_contact_solve) */
static void vc_init_masses(vc_t *vc, const v2 *wmPoints, v2 cA, v2 cB, v2 vA, f32 wA, v2 vB, f32 wB) {
    f32 mA = vc->invMassA, mB = vc->invMassB, iA = vc->invIA, iB = vc->invIB;
    int pointCount = vc->pointCount;
    for (int j = 0; j < pointCount; ++j) {
        vcp_t *vcp = &vc->points[j];
        vcp->rA = vsub(wmPoints[j], cA);
        vcp->rB = vsub(wmPoints[j], cB);
        f32 rnA = vcross(vcp->rA, vc->normal);
        f32 rnB = vcross(vcp->rB, vc->normal);
        f32 kNormal = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
        vcp->normalMass = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
        v2 tangent = vcross_vs(vc->normal, 1.0f);
        f32 rtA = vcross(vcp->rA, tangent);
        f32 rtB = vcross(vcp->rB, tangent);
        f32 kTangent = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
        vcp->tangentMass = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
        vcp->velocityBias = 0.0f;
        f32 vRel = vdot(vc->normal,
                        vsub(vsub(vadd(vB, vcross_sv(wB, vcp->rB)), vA), vcross_sv(wA, vcp->rA)));
        if (vRel < -B2_VELOCITY_THRESHOLD) vcp->velocityBias = -vc->restitution * vRel;
    }
    if (vc->pointCount == 2) {
        vcp_t *vcp1 = &vc->points[0], *vcp2 = &vc->points[1];
        f32 rn1A = vcross(vcp1->rA, vc->normal);
        f32 rn1B = vcross(vcp1->rB, vc->normal);
        f32 rn2A = vcross(vcp2->rA, vc->normal);
        f32 rn2B = vcross(vcp2->rB, vc->normal);
        f32 k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B;
        f32 k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
        f32 k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;
        const f32 k_maxConditionNumber = 1000.0f;
        if (k11 * k11 < k_maxConditionNumber * (k11 * k22 - k12 * k12)) {
            vc->K_exx = k11; vc->K_exy = k12; vc->K_eyx = k12; vc->K_eyy = k22;
            f32 a = vc->K_exx, b = vc->K_eyx, c = vc->K_exy, d = vc->K_eyy;
            f32 det = a * d - b * c;
            if (det != 0.0f) det = 1.0f / det;
            vc->nm_exx = det * d;
            vc->nm_eyx = -det * b;
            vc->nm_exy = -det * c;
            vc->nm_eyy = det * a;
        } else {
            vc->pointCount = 1;
        }
    }
}
static void contact_solver_init_velocity(o_world *w, island_t *is) {
    for (int i = 0; i < is->ncontact; ++i) {
        vc_t *vc = &is->vcs[i];
        pc_t *pc = &is->pcs[i];
        f32 radiusA = pc->radiusA, radiusB = pc->radiusB;
        const manifold_t *manifold = &w->contacts[vc->contactIndex].m;
        int indexA = vc->indexA, indexB = vc->indexB;
        v2 localCenterA = pc->localCenterA, localCenterB = pc->localCenterB;
        v2 cA = is->positions[indexA].c;
        f32 aA = is->positions[indexA].a;
        v2 vA = is->velocities[indexA].v;
        f32 wA = is->velocities[indexA].w;
        v2 cB = is->positions[indexB].c;
        f32 aB = is->positions[indexB].a;
        v2 vB = is->velocities[indexB].v;
        f32 wB = is->velocities[indexB].w;
        xf_t xfA, xfB;
        xfA.q = rot_set(aA);
        xfB.q = rot_set(aB);
        xfA.p = vsub(cA, rmul(xfA.q, localCenterA));
        xfB.p = vsub(cB, rmul(xfB.q, localCenterB));
        v2 wmPoints[2];
        v2 wmNormal = V2(0.0f, 0.0f);
        world_manifold(manifold, xfA, radiusA, xfB, radiusB, &wmNormal, wmPoints);
        vc->normal = wmNormal;
        vc_init_masses(vc, wmPoints, cA, cB, vA, wA, vB, wB);
    }
}
static void contact_solver_warm_start(island_t *is) {
    for (int i = 0; i < is->ncontact; ++i) {
        vc_t *vc = &is->vcs[i];
        int indexA = vc->indexA, indexB = vc->indexB;
        f32 mA = vc->invMassA, iA = vc->invIA, mB = vc->invMassB, iB = vc->invIB;
        v2 vA = is->velocities[indexA].v;
        f32 wA = is->velocities[indexA].w;
        v2 vB = is->velocities[indexB].v;
        f32 wB = is->velocities[indexB].w;
        v2 normal = vc->normal;
        v2 tangent = vcross_vs(normal, 1.0f);
        for (int j = 0; j < vc->pointCount; ++j) {
            vcp_t *vcp = &vc->points[j];
            v2 P = vadd(vscale(vcp->normalImpulse, normal), vscale(vcp->tangentImpulse, tangent));
            wA -= iA * vcross(vcp->rA, P);
            vA = vsub(vA, vscale(mA, P));
            wB += iB * vcross(vcp->rB, P);
            vB = vadd(vB, vscale(mB, P));
        }
        is->velocities[indexA].v = vA;
        is->velocities[indexA].w = wA;
        is->velocities[indexB].v = vB;
        is->velocities[indexB].w = wB;
    }
}
static void contact_solver_solve_velocity(island_t *is) {
    for (int i = 0; i < is->ncontact; ++i) {
        vc_t *vc = &is->vcs[i];
        int indexA = vc->indexA, indexB = vc->indexB;
        f32 mA = vc->invMassA, iA = vc->invIA, mB = vc->invMassB, iB = vc->invIB;
        int pointCount = vc->pointCount;
        v2 vA = is->velocities[indexA].v;
        f32 wA = is->velocities[indexA].w;
        v2 vB = is->velocities[indexB].v;
        f32 wB = is->velocities[indexB].w;
        v2 normal = vc->normal;
        v2 tangent = vcross_vs(normal, 1.0f);
        f32 friction = vc->friction;
        for (int j = 0; j < pointCount; ++j) {
            vcp_t *vcp = &vc->points[j];
            v2 dv = vsub(vsub(vadd(vB, vcross_sv(wB, vcp->rB)), vA), vcross_sv(wA, vcp->rA));
            f32 vt = vdot(dv, tangent) - vc->tangentSpeed;
            f32 lambda = vcp->tangentMass * (-vt);
            f32 maxFriction = friction * vcp->normalImpulse;
            f32 newImpulse = fclamp(vcp->tangentImpulse + lambda, -maxFriction, maxFriction);
            lambda = newImpulse - vcp->tangentImpulse;
            vcp->tangentImpulse = newImpulse;
            v2 P = vscale(lambda, tangent);
            vA = vsub(vA, vscale(mA, P));
            wA -= iA * vcross(vcp->rA, P);
            vB = vadd(vB, vscale(mB, P));
            wB += iB * vcross(vcp->rB, P);
        }
        if (vc->pointCount == 1) {
            vcp_t *vcp = &vc->points[0];
            v2 dv = vsub(vsub(vadd(vB, vcross_sv(wB, vcp->rB)), vA), vcross_sv(wA, vcp->rA));
            f32 vn = vdot(dv, normal);
            f32 lambda = -vcp->normalMass * (vn - vcp->velocityBias);
            f32 newImpulse = fmax32(vcp->normalImpulse + lambda, 0.0f);
            lambda = newImpulse - vcp->normalImpulse;
            vcp->normalImpulse = newImpulse;
            v2 P = vscale(lambda, normal);
            vA = vsub(vA, vscale(mA, P));
            wA -= iA * vcross(vcp->rA, P);
            vB = vadd(vB, vscale(mB, P));
            wB += iB * vcross(vcp->rB, P);
        } else {
            vcp_t *cp1 = &vc->points[0], *cp2 = &vc->points[1];
            v2 a = V2(cp1->normalImpulse, cp2->normalImpulse);
            v2 dv1 = vsub(vsub(vadd(vB, vcross_sv(wB, cp1->rB)), vA), vcross_sv(wA, cp1->rA));
            v2 dv2 = vsub(vsub(vadd(vB, vcross_sv(wB, cp2->rB)), vA), vcross_sv(wA, cp2->rA));
            f32 vn1 = vdot(dv1, normal), vn2 = vdot(dv2, normal);
            v2 b;
            b.x = vn1 - cp1->velocityBias;
            b.y = vn2 - cp2->velocityBias;
            /* b -= b2Mul(K, a) */
            b = vsub(b, V2(vc->K_exx * a.x + vc->K_eyx * a.y, vc->K_exy * a.x + vc->K_eyy * a.y));
            for (;;) {
                v2 x;
                /* Case 1 */
                x = vneg(V2(vc->nm_exx * b.x + vc->nm_eyx * b.y, vc->nm_exy * b.x + vc->nm_eyy * b.y));
                if (x.x >= 0.0f && x.y >= 0.0f) {
                    v2 d = vsub(x, a);
                    v2 P1 = vscale(d.x, normal), P2 = vscale(d.y, normal);
                    vA = vsub(vA, vscale(mA, vadd(P1, P2)));
                    wA -= iA * (vcross(cp1->rA, P1) + vcross(cp2->rA, P2));
                    vB = vadd(vB, vscale(mB, vadd(P1, P2)));
                    wB += iB * (vcross(cp1->rB, P1) + vcross(cp2->rB, P2));
                    cp1->normalImpulse = x.x;
                    cp2->normalImpulse = x.y;
                    break;
                }
                /* Case 2 */
                x.x = -cp1->normalMass * b.x;
                x.y = 0.0f;
                vn1 = 0.0f;
                vn2 = vc->K_exy * x.x + b.y;
                if (x.x >= 0.0f && vn2 >= 0.0f) {
                    v2 d = vsub(x, a);
                    v2 P1 = vscale(d.x, normal), P2 = vscale(d.y, normal);
                    vA = vsub(vA, vscale(mA, vadd(P1, P2)));
                    wA -= iA * (vcross(cp1->rA, P1) + vcross(cp2->rA, P2));
                    vB = vadd(vB, vscale(mB, vadd(P1, P2)));
                    wB += iB * (vcross(cp1->rB, P1) + vcross(cp2->rB, P2));
                    cp1->normalImpulse = x.x;
                    cp2->normalImpulse = x.y;
                    break;
                }
                /* Case 3 */
                x.x = 0.0f;
                x.y = -cp2->normalMass * b.y;
                vn1 = vc->K_eyx * x.y + b.x;
                vn2 = 0.0f;
                if (x.y >= 0.0f && vn1 >= 0.0f) {
                    v2 d = vsub(x, a);
                    v2 P1 = vscale(d.x, normal), P2 = vscale(d.y, normal);
                    vA = vsub(vA, vscale(mA, vadd(P1, P2)));
                    wA -= iA * (vcross(cp1->rA, P1) + vcross(cp2->rA, P2));
                    vB = vadd(vB, vscale(mB, vadd(P1, P2)));
                    wB += iB * (vcross(cp1->rB, P1) + vcross(cp2->rB, P2));
                    cp1->normalImpulse = x.x;
                    cp2->normalImpulse = x.y;
                    break;
                }
                /* Case 4 */
                x.x = 0.0f;
                x.y = 0.0f;
                vn1 = b.x;
                vn2 = b.y;
                if (vn1 >= 0.0f && vn2 >= 0.0f) {
                    v2 d = vsub(x, a);
                    v2 P1 = vscale(d.x, normal), P2 = vscale(d.y, normal);
                    vA = vsub(vA, vscale(mA, vadd(P1, P2)));
                    wA -= iA * (vcross(cp1->rA, P1) + vcross(cp2->rA, P2));
                    vB = vadd(vB, vscale(mB, vadd(P1, P2)));
                    wB += iB * (vcross(cp1->rB, P1) + vcross(cp2->rB, P2));
                    cp1->normalImpulse = x.x;
                    cp2->normalImpulse = x.y;
                    break;
                }
                break;
            }
        }
        is->velocities[indexA].v = vA;
        is->velocities[indexA].w = wA;
        is->velocities[indexB].v = vB;
        is->velocities[indexB].w = wB;
    }
}
static void contact_solver_store(o_world *w, island_t *is) {
    for (int i = 0; i < is->ncontact; ++i) {
        vc_t *vc = &is->vcs[i];
        manifold_t *m = &w->contacts[vc->contactIndex].m;
        for (int j = 0; j < vc->pointCount; ++j) {
            m->points[j].normalImpulse = vc->points[j].normalImpulse;
            m->points[j].tangentImpulse = vc->points[j].tangentImpulse;
        }
    }
}
/* b2PositionSolverManifold::Initialize */
static void psm_init(const pc_t *pc, xf_t xfA, xf_t xfB, int index, v2 *normal, v2 *point, f32 *separation) {
    switch (pc->type) {
    case MF_CIRCLES: {
        v2 pointA = xmul(xfA, pc->localPoint);
        v2 pointB = xmul(xfB, pc->localPoints[0]);
        *normal = vsub(pointB, pointA);
        vnormalize(normal);
        *point = vscale(0.5f, vadd(pointA, pointB));
        *separation = vdot(vsub(pointB, pointA), *normal) - pc->radiusA - pc->radiusB;
    } break;
    case MF_FACE_A: {
        *normal = rmul(xfA.q, pc->localNormal);
        v2 planePoint = xmul(xfA, pc->localPoint);
        v2 clipPoint = xmul(xfB, pc->localPoints[index]);
        *separation = vdot(vsub(clipPoint, planePoint), *normal) - pc->radiusA - pc->radiusB;
        *point = clipPoint;
    } break;
    default: {
        *normal = rmul(xfB.q, pc->localNormal);
        v2 planePoint = xmul(xfB, pc->localPoint);
        v2 clipPoint = xmul(xfA, pc->localPoints[index]);
        *separation = vdot(vsub(clipPoint, planePoint), *normal) - pc->radiusA - pc->radiusB;
        *point = clipPoint;
        *normal = vneg(*normal);
    } break;
    }
}
static int contact_solver_solve_position(island_t *is) {
    f32 minSeparation = 0.0f;
    for (int i = 0; i < is->ncontact; ++i) {
        pc_t *pc = &is->pcs[i];
        int indexA = pc->indexA, indexB = pc->indexB;
        v2 localCenterA = pc->localCenterA, localCenterB = pc->localCenterB;
        f32 mA = pc->invMassA, iA = pc->invIA, mB = pc->invMassB, iB = pc->invIB;
        int pointCount = pc->pointCount;
        v2 cA = is->positions[indexA].c;
        f32 aA = is->positions[indexA].a;
        v2 cB = is->positions[indexB].c;
        f32 aB = is->positions[indexB].a;
        for (int j = 0; j < pointCount; ++j) {
            xf_t xfA, xfB;
            xfA.q = rot_set(aA);
            xfB.q = rot_set(aB);
            xfA.p = vsub(cA, rmul(xfA.q, localCenterA));
            xfB.p = vsub(cB, rmul(xfB.q, localCenterB));
            v2 normal, point;
            f32 separation;
            psm_init(pc, xfA, xfB, j, &normal, &point, &separation);
            v2 rA = vsub(point, cA), rB = vsub(point, cB);
            minSeparation = fmin32(minSeparation, separation);
            f32 C = fclamp(B2_BAUMGARTE * (separation + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
            f32 rnA = vcross(rA, normal), rnB = vcross(rB, normal);
            f32 K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
            f32 impulse = K > 0.0f ? -C / K : 0.0f;
            v2 P = vscale(impulse, normal);
            cA = vsub(cA, vscale(mA, P));
            aA -= iA * vcross(rA, P);
            cB = vadd(cB, vscale(mB, P));
            aB += iB * vcross(rB, P);
        }
        is->positions[indexA].c = cA;
        is->positions[indexA].a = aA;
        is->positions[indexB].c = cB;
        is->positions[indexB].a = aB;
    }
    return minSeparation >= -3.0f * B2_LINEAR_SLOP;
}

/* ---- A.6 revolute joint ---- */
static inline v3 v3cross(v3 a, v3 b) {
    v3 r;
    r.x = a.y * b.z - a.z * b.y;
    r.y = a.z * b.x - a.x * b.z;
    r.z = a.x * b.y - a.y * b.x;
    return r;
}
static inline f32 v3dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 mat33_solve33(v3 ex, v3 ey, v3 ez, v3 b) {
    f32 det = v3dot(ex, v3cross(ey, ez));
    if (det != 0.0f) det = 1.0f / det;
    v3 x;
    x.x = det * v3dot(b, v3cross(ey, ez));
    x.y = det * v3dot(ex, v3cross(b, ez));
    x.z = det * v3dot(ex, v3cross(ey, b));
    return x;
}
static v2 mat_solve22(f32 a11, f32 a12, f32 a21, f32 a22, v2 b) {
    f32 det = a11 * a22 - a12 * a21;
    if (det != 0.0f) det = 1.0f / det;
    v2 x;
    x.x = det * (a22 * b.x - a12 * b.y);
    x.y = det * (a11 * b.y - a21 * b.x);
    return x;
}
static void joint_init_velocity(o_world *w, island_t *is, joint_t *j, const step_t *step) {
    body_t *bA = &w->bodies[j->bodyA], *bB = &w->bodies[j->bodyB];
    j->indexA = bA->islandIndex;
    j->indexB = bB->islandIndex;
    j->localCenterA = bA->localCenter;
    j->localCenterB = bB->localCenter;
    j->invMassA = bA->invMass;
    j->invMassB = bB->invMass;
    j->invIA = bA->invI;
    j->invIB = bB->invI;
    f32 aA = is->positions[j->indexA].a;
    v2 vA = is->velocities[j->indexA].v;
    f32 wA = is->velocities[j->indexA].w;
    f32 aB = is->positions[j->indexB].a;
    v2 vB = is->velocities[j->indexB].v;
    f32 wB = is->velocities[j->indexB].w;
    rot_t qA = rot_set(aA), qB = rot_set(aB);
    j->rA = rmul(qA, vsub(j->localAnchorA, j->localCenterA));
    j->rB = rmul(qB, vsub(j->localAnchorB, j->localCenterB));
    f32 mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
    int fixedRotation = (iA + iB == 0.0f);
    j->mex.x = mA + mB + j->rA.y * j->rA.y * iA + j->rB.y * j->rB.y * iB;
    j->mey.x = -j->rA.y * j->rA.x * iA - j->rB.y * j->rB.x * iB;
    j->mez.x = -j->rA.y * iA - j->rB.y * iB;
    j->mex.y = j->mey.x;
    j->mey.y = mA + mB + j->rA.x * j->rA.x * iA + j->rB.x * j->rB.x * iB;
    j->mez.y = j->rA.x * iA + j->rB.x * iB;
    j->mex.z = j->mez.x;
    j->mey.z = j->mez.y;
    j->mez.z = iA + iB;
    j->motorMass = iA + iB;
    if (j->motorMass > 0.0f) j->motorMass = 1.0f / j->motorMass;
    if (j->enableMotor == 0 || fixedRotation) j->motorImpulse = 0.0f;
    if (j->enableLimit && fixedRotation == 0) {
        f32 jointAngle = aB - aA - j->referenceAngle;
        if (fabs32(j->upperAngle - j->lowerAngle) < 2.0f * B2_ANGULAR_SLOP) {
            j->limitState = LIM_EQUAL;
        } else if (jointAngle <= j->lowerAngle) {
            if (j->limitState != LIM_AT_LOWER) j->impulse.z = 0.0f;
            j->limitState = LIM_AT_LOWER;
        } else if (jointAngle >= j->upperAngle) {
            if (j->limitState != LIM_AT_UPPER) j->impulse.z = 0.0f;
            j->limitState = LIM_AT_UPPER;
        } else {
            j->limitState = LIM_INACTIVE;
            j->impulse.z = 0.0f;
        }
    } else {
        j->limitState = LIM_INACTIVE;
    }
    if (step->warmStarting) {
        j->impulse.x *= step->dtRatio;
        j->impulse.y *= step->dtRatio;
        j->impulse.z *= step->dtRatio;
        j->motorImpulse *= step->dtRatio;
        v2 P = V2(j->impulse.x, j->impulse.y);
        vA = vsub(vA, vscale(mA, P));
        wA -= iA * (vcross(j->rA, P) + j->motorImpulse + j->impulse.z);
        vB = vadd(vB, vscale(mB, P));
        wB += iB * (vcross(j->rB, P) + j->motorImpulse + j->impulse.z);
    } else {
        j->impulse.x = j->impulse.y = j->impulse.z = 0.0f;
        j->motorImpulse = 0.0f;
    }
    is->velocities[j->indexA].v = vA;
    is->velocities[j->indexA].w = wA;
    is->velocities[j->indexB].v = vB;
    is->velocities[j->indexB].w = wB;
}
static void joint_solve_velocity(island_t *is, joint_t *j, const step_t *step) {
    v2 vA = is->velocities[j->indexA].v;
    f32 wA = is->velocities[j->indexA].w;
    v2 vB = is->velocities[j->indexB].v;
    f32 wB = is->velocities[j->indexB].w;
    f32 mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
    int fixedRotation = (iA + iB == 0.0f);
    if (j->enableMotor && j->limitState != LIM_EQUAL && fixedRotation == 0) {
        f32 Cdot = wB - wA - j->motorSpeed;
        f32 impulse = -j->motorMass * Cdot;
        f32 oldImpulse = j->motorImpulse;
        f32 maxImpulse = step->dt * j->maxMotorTorque;
        j->motorImpulse = fclamp(oldImpulse + impulse, -maxImpulse, maxImpulse);
        impulse = j->motorImpulse - oldImpulse;
        wA -= iA * impulse;
        wB += iB * impulse;
    }
    if (j->enableLimit && j->limitState != LIM_INACTIVE && fixedRotation == 0) {
        v2 Cdot1 = vsub(vsub(vadd(vB, vcross_sv(wB, j->rB)), vA), vcross_sv(wA, j->rA));
        f32 Cdot2 = wB - wA;
        v3 Cdot;
        Cdot.x = Cdot1.x;
        Cdot.y = Cdot1.y;
        Cdot.z = Cdot2;
        v3 impulse = mat33_solve33(j->mex, j->mey, j->mez, Cdot);
        impulse.x = -impulse.x;
        impulse.y = -impulse.y;
        impulse.z = -impulse.z;
        if (j->limitState == LIM_EQUAL) {
            j->impulse.x += impulse.x;
            j->impulse.y += impulse.y;
            j->impulse.z += impulse.z;
        } else if (j->limitState == LIM_AT_LOWER) {
            f32 newImpulse = j->impulse.z + impulse.z;
            if (newImpulse < 0.0f) {
                v2 rhs = vadd(vneg(Cdot1), vscale(j->impulse.z, V2(j->mez.x, j->mez.y)));
                v2 reduced = mat_solve22(j->mex.x, j->mey.x, j->mex.y, j->mey.y, rhs);
                impulse.x = reduced.x;
                impulse.y = reduced.y;
                impulse.z = -j->impulse.z;
                j->impulse.x += reduced.x;
                j->impulse.y += reduced.y;
                j->impulse.z = 0.0f;
            } else {
                j->impulse.x += impulse.x;
                j->impulse.y += impulse.y;
                j->impulse.z += impulse.z;
            }
        } else if (j->limitState == LIM_AT_UPPER) {
            f32 newImpulse = j->impulse.z + impulse.z;
            if (newImpulse > 0.0f) {
                v2 rhs = vadd(vneg(Cdot1), vscale(j->impulse.z, V2(j->mez.x, j->mez.y)));
                v2 reduced = mat_solve22(j->mex.x, j->mey.x, j->mex.y, j->mey.y, rhs);
                impulse.x = reduced.x;
                impulse.y = reduced.y;
                impulse.z = -j->impulse.z;
                j->impulse.x += reduced.x;
                j->impulse.y += reduced.y;
                j->impulse.z = 0.0f;
            } else {
                j->impulse.x += impulse.x;
                j->impulse.y += impulse.y;
                j->impulse.z += impulse.z;
            }
        }
        v2 P = V2(impulse.x, impulse.y);
        vA = vsub(vA, vscale(mA, P));
        wA -= iA * (vcross(j->rA, P) + impulse.z);
        vB = vadd(vB, vscale(mB, P));
        wB += iB * (vcross(j->rB, P) + impulse.z);
    } else {
        v2 Cdot = vsub(vsub(vadd(vB, vcross_sv(wB, j->rB)), vA), vcross_sv(wA, j->rA));
        v2 impulse = mat_solve22(j->mex.x, j->mey.x, j->mex.y, j->mey.y, vneg(Cdot));
        j->impulse.x += impulse.x;
        j->impulse.y += impulse.y;
        vA = vsub(vA, vscale(mA, impulse));
        wA -= iA * vcross(j->rA, impulse);
        vB = vadd(vB, vscale(mB, impulse));
        wB += iB * vcross(j->rB, impulse);
    }
    is->velocities[j->indexA].v = vA;
    is->velocities[j->indexA].w = wA;
    is->velocities[j->indexB].v = vB;
    is->velocities[j->indexB].w = wB;
}
static int joint_solve_position(island_t *is, joint_t *j) {
    v2 cA = is->positions[j->indexA].c;
    f32 aA = is->positions[j->indexA].a;
    v2 cB = is->positions[j->indexB].c;
    f32 aB = is->positions[j->indexB].a;
    f32 angularError = 0.0f, positionError = 0.0f;
    int fixedRotation = (j->invIA + j->invIB == 0.0f);
    if (j->enableLimit && j->limitState != LIM_INACTIVE && fixedRotation == 0) {
        f32 angle = aB - aA - j->referenceAngle;
        f32 limitImpulse = 0.0f;
        if (j->limitState == LIM_EQUAL) {
            f32 C = fclamp(angle - j->lowerAngle, -B2_MAX_ANGULAR_CORRECTION, B2_MAX_ANGULAR_CORRECTION);
            limitImpulse = -j->motorMass * C;
            angularError = fabs32(C);
        } else if (j->limitState == LIM_AT_LOWER) {
            f32 C = angle - j->lowerAngle;
            angularError = -C;
            C = fclamp(C + B2_ANGULAR_SLOP, -B2_MAX_ANGULAR_CORRECTION, 0.0f);
            limitImpulse = -j->motorMass * C;
        } else if (j->limitState == LIM_AT_UPPER) {
            f32 C = angle - j->upperAngle;
            angularError = C;
            C = fclamp(C - B2_ANGULAR_SLOP, 0.0f, B2_MAX_ANGULAR_CORRECTION);
            limitImpulse = -j->motorMass * C;
        }
        aA -= j->invIA * limitImpulse;
        aB += j->invIB * limitImpulse;
    }
    {
        rot_t qA = rot_set(aA), qB = rot_set(aB);
        v2 rA = rmul(qA, vsub(j->localAnchorA, j->localCenterA));
        v2 rB = rmul(qB, vsub(j->localAnchorB, j->localCenterB));
        v2 C = vsub(vsub(vadd(cB, rB), cA), rA);
        positionError = vlen(C);
        f32 mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
        f32 Kexx = mA + mB + iA * rA.y * rA.y + iB * rB.y * rB.y;
        f32 Kexy = -iA * rA.x * rA.y - iB * rB.x * rB.y;
        f32 Keyx = Kexy;
        f32 Keyy = mA + mB + iA * rA.x * rA.x + iB * rB.x * rB.x;
        v2 impulse = vneg(mat_solve22(Kexx, Keyx, Kexy, Keyy, C));
        cA = vsub(cA, vscale(mA, impulse));
        aA -= iA * vcross(rA, impulse);
        cB = vadd(cB, vscale(mB, impulse));
        aB += iB * vcross(rB, impulse);
    }
    is->positions[j->indexA].c = cA;
    is->positions[j->indexA].a = aA;
    is->positions[j->indexB].c = cB;
    is->positions[j->indexB].a = aB;
    return positionError <= B2_LINEAR_SLOP && angularError <= B2_ANGULAR_SLOP;
}

static void body_sync_transform(body_t *b) {
    b->xf.q = rot_set(b->a);
    b->xf.p = vsub(b->c, rmul(b->xf.q, b->localCenter));
}
/* b2Body::SynchronizeFixtures -> b2Fixture::Synchronize -> b2DynamicTree::MoveProxy */
static void body_sync_fixtures(o_world *w, int bi) {
    body_t *b = &w->bodies[bi];
    xf_t xf1;
    xf1.q = rot_set(b->a0);
    xf1.p = vsub(b->c0, rmul(xf1.q, b->localCenter));
    aabb_t aabb1 = shape_aabb(&b->shape, xf1), aabb2 = shape_aabb(&b->shape, b->xf);
    aabb_t aabb = aabb_combine(aabb1, aabb2);
    v2 displacement = vsub(b->xf.p, xf1.p);
    if (aabb_contains(b->fat, aabb)) return;
    aabb_t f = aabb_fatten(aabb);
    v2 d = vscale(B2_AABB_MULTIPLIER, displacement);
    if (d.x < 0.0f) f.lower.x += d.x; else f.upper.x += d.x;
    if (d.y < 0.0f) f.lower.y += d.y; else f.upper.y += d.y;
    b->fat = f;
    w->moved[w->nmoved++] = bi;
}

/* ---- A.4 b2Island::Solve ---- */

#ifdef REM2D_ORACLE_PROBE
/* Diagnostic build only (tools/probe_fixed_point.py): at which velocity iteration does a sweep stop changing any
 * bit of the solver state?  From there on every further sweep is a no-op. */
static int g_probe_hist[512];
#define PROBE_TOI_P 64
static int g_probe_toi[PROBE_TOI_P + 1][256]; /* [period][sweep at which the cycle was first seen]; [0][255] = none within the budget */
static _Thread_local f32 t_probe[2][8 * O_MAX_BODIES + 4 * O_MAX_CONTACTS];
static int probe_snapshot(o_world *w, island_t *is, int which) {
    f32 *o = t_probe[which];
    int n = 0;
    for (int i = 0; i < is->nbody; ++i) { o[n++] = is->velocities[i].v.x; o[n++] = is->velocities[i].v.y; o[n++] = is->velocities[i].w; }
    for (int i = 0; i < is->njoint; ++i) {
        joint_t *j = &w->joints[is->joints[i]];
        o[n++] = j->impulse.x; o[n++] = j->impulse.y; o[n++] = j->impulse.z; o[n++] = j->motorImpulse;
    }
    for (int i = 0; i < is->ncontact; ++i)
        for (int k = 0; k < is->vcs[i].pointCount; ++k) { o[n++] = is->vcs[i].points[k].normalImpulse; o[n++] = is->vcs[i].points[k].tangentImpulse; }
    return which == 1 && memcmp(t_probe[0], t_probe[1], (size_t)n * sizeof(f32)) == 0;
}
static void probe_record(int it, int maxIt) {
    (void)maxIt;
    if (it > 511) it = 511;
    __atomic_fetch_add(&g_probe_hist[it], 1, __ATOMIC_RELAXED);
}
void rem2d_oracle_probe_toi(int *out, int reset) {
    for (int p = 0; p <= PROBE_TOI_P; ++p)
        for (int i = 0; i < 256; ++i) { out[p * 256 + i] = g_probe_toi[p][i]; if (reset) g_probe_toi[p][i] = 0; }
}
void rem2d_oracle_probe_hist(int *out, int reset) {
    for (int i = 0; i < 512; ++i) { out[i] = g_probe_hist[i]; if (reset) g_probe_hist[i] = 0; }
}
/* position iterations: [0..63] iterations used by islands that passed the tolerance test; [64..127] iteration at which an
 * island that never passes it stopped changing any bit of any position (127 = never); [128] contacts failing at the last
 * iteration, [129] joints failing, [130] both */
static int g_probe_pos[160];
void rem2d_oracle_probe_pos(int *out, int reset) {
    for (int i = 0; i < 160; ++i) { out[i] = g_probe_pos[i]; if (reset) g_probe_pos[i] = 0; }
}
static int probe_pos_snapshot(island_t *is, int which) {
    f32 *o = t_probe[which];
    int n = 0;
    for (int i = 0; i < is->nbody; ++i) { o[n++] = is->positions[i].c.x; o[n++] = is->positions[i].c.y; o[n++] = is->positions[i].a; }
    return which == 1 && memcmp(t_probe[0], t_probe[1], (size_t)n * sizeof(f32)) == 0;
}
#endif

static void island_solve(o_world *w, island_t *is, const step_t *step) {
    f32 h = step->dt;
    is->positions[IDX_STATIC].c = V2(0.0f, 0.0f);
    is->positions[IDX_STATIC].a = 0.0f;
    is->velocities[IDX_STATIC].v = V2(0.0f, 0.0f);
    is->velocities[IDX_STATIC].w = 0.0f;
    for (int i = 0; i < is->nbody; ++i) {
        body_t *b = &w->bodies[is->bodies[i]];
        v2 c = b->c;
        f32 a = b->a;
        v2 v = b->v;
        f32 wz = b->w;
        b->c0 = b->c;
        b->a0 = b->a;
        /* v += h * (gravityScale * gravity + invMass * force); force = torque = 0, damping 0 */
        v2 acc = vadd(vscale(1.0f, w->gravity), vscale(b->invMass, V2(0.0f, 0.0f)));
        v = vadd(v, vscale(h, acc));
        wz += h * b->invI * 0.0f;
        v = vscale(1.0f / (1.0f + h * 0.0f), v);
        wz *= 1.0f / (1.0f + h * 0.0f);
        is->positions[i].c = c;
        is->positions[i].a = a;
        is->velocities[i].v = v;
        is->velocities[i].w = wz;
    }
    contact_solver_setup(w, is, step);
    contact_solver_init_velocity(w, is);
    if (step->warmStarting) contact_solver_warm_start(is);
    for (int i = 0; i < is->njoint; ++i) joint_init_velocity(w, is, &w->joints[is->joints[i]], step);
    for (int it = 0; it < step->velocityIterations; ++it) {
#ifdef REM2D_ORACLE_PROBE
        probe_snapshot(w, is, 0);
#endif
        for (int i = 0; i < is->njoint; ++i) joint_solve_velocity(is, &w->joints[is->joints[i]], step);
        contact_solver_solve_velocity(is);
#ifdef REM2D_ORACLE_PROBE
        if (probe_snapshot(w, is, 1)) { probe_record(it, step->velocityIterations); break; } /* fixed point: later sweeps are no-ops */
        if (it + 1 == step->velocityIterations) probe_record(it + 1, step->velocityIterations);
#endif
    }
    contact_solver_store(w, is);
    for (int i = 0; i < is->nbody; ++i) {
        v2 c = is->positions[i].c;
        f32 a = is->positions[i].a;
        v2 v = is->velocities[i].v;
        f32 wz = is->velocities[i].w;
        v2 translation = vscale(h, v);
        if (vdot(translation, translation) > B2_MAX_TRANSLATION_SQ) {
            f32 ratio = B2_MAX_TRANSLATION / vlen(translation);
            v = vscale(ratio, v);
        }
        f32 rotation = h * wz;
        if (rotation * rotation > B2_MAX_ROTATION_SQ) {
            f32 ratio = B2_MAX_ROTATION / fabs32(rotation);
            wz *= ratio;
        }
        c = vadd(c, vscale(h, v));
        a += h * wz;
        is->positions[i].c = c;
        is->positions[i].a = a;
        is->velocities[i].v = v;
        is->velocities[i].w = wz;
    }
    int positionSolved = 0;
    w->lastPositionIterations = step->positionIterations;
#ifdef REM2D_ORACLE_PROBE
    int probeFixed = -1, probeC = 1, probeJ = 1;
#endif
    for (int it = 0; it < step->positionIterations; ++it) {
#ifdef REM2D_ORACLE_PROBE
        probe_pos_snapshot(is, 0);
#endif
        int contactsOkay = contact_solver_solve_position(is);
        int jointsOkay = 1;
        for (int i = 0; i < is->njoint; ++i) {
            int jointOkay = joint_solve_position(is, &w->joints[is->joints[i]]);
            jointsOkay = jointsOkay && jointOkay;
        }
#ifdef REM2D_ORACLE_PROBE
        if (probeFixed < 0 && probe_pos_snapshot(is, 1)) probeFixed = it;
        probeC = contactsOkay; probeJ = jointsOkay;
#endif
        if (contactsOkay && jointsOkay) {
            positionSolved = 1;
            w->lastPositionIterations = it + 1;
            break;
        }
    }
#ifdef REM2D_ORACLE_PROBE
    if (positionSolved) __atomic_fetch_add(&g_probe_pos[w->lastPositionIterations < 63 ? w->lastPositionIterations : 63], 1, __ATOMIC_RELAXED);
    else {
        __atomic_fetch_add(&g_probe_pos[64 + (probeFixed < 0 ? 63 : probeFixed)], 1, __ATOMIC_RELAXED);
        __atomic_fetch_add(&g_probe_pos[!probeC && !probeJ ? 130 : !probeC ? 128 : 129], 1, __ATOMIC_RELAXED);
    }
#endif
    for (int i = 0; i < is->nbody; ++i) {
        body_t *b = &w->bodies[is->bodies[i]];
        b->c = is->positions[i].c;
        b->a = is->positions[i].a;
        b->v = is->velocities[i].v;
        b->w = is->velocities[i].w;
        body_sync_transform(b);
    }
    if (!(w->flags & O_FLAG_NO_SLEEP)) {
        f32 minSleepTime = B2_MAXFLOAT;
        const f32 linTolSqr = B2_LINEAR_SLEEP_TOL * B2_LINEAR_SLEEP_TOL;
        const f32 angTolSqr = B2_ANGULAR_SLEEP_TOL * B2_ANGULAR_SLEEP_TOL;
        for (int i = 0; i < is->nbody; ++i) {
            body_t *b = &w->bodies[is->bodies[i]];
            if (b->w * b->w > angTolSqr || vdot(b->v, b->v) > linTolSqr) {
                b->sleepTime = 0.0f;
                minSleepTime = 0.0f;
            } else {
                b->sleepTime += h;
                minSleepTime = fmin32(minSleepTime, b->sleepTime);
            }
        }
        if (minSleepTime >= B2_TIME_TO_SLEEP && positionSolved) {
            for (int i = 0; i < is->nbody; ++i) body_set_awake(w, &w->bodies[is->bodies[i]], 0);
        }
    }
}

/* ---- b2World::Solve ---- */
static void world_solve(o_world *w, const step_t *step) {
    static _Thread_local island_t island;
    island_t *is = &island;
    for (int i = 0; i < w->nbody; ++i) w->bodies[i].islandFlag = 0;
    for (int k = 0; k < w->wcount; ++k) w->contacts[w->wlist[k]].islandFlag = 0;
    for (int i = 0; i < w->njoint; ++i) w->joints[i].islandFlag = 0;
    w->islandJointCount = 0;
    int stack[O_MAX_BODIES];
    /* m_bodyList is LIFO: the last created body is the first seed candidate (A.3) */
    for (int seed = w->nbody - 1; seed >= 0; --seed) {
        body_t *sb = &w->bodies[seed];
        if (sb->islandFlag) continue;
        if (!sb->awake) continue;
        is->nbody = is->ncontact = is->njoint = 0;
        int stackCount = 0;
        stack[stackCount++] = seed;
        sb->islandFlag = 1;
        while (stackCount > 0) {
            int bi = stack[--stackCount];
            body_t *b = &w->bodies[bi];
            b->islandIndex = is->nbody;
            is->bodies[is->nbody++] = bi;
            body_set_awake(w, b, 1);
            for (int k = 0; k < b->ncontacts; ++k) {
                contact_t *c = &w->contacts[b->contacts[k]];
                if (c->islandFlag) continue;
                if (!c->enabled || !c->touching) continue;
                is->contacts[is->ncontact++] = b->contacts[k];
                c->islandFlag = 1;
                /* other body is static: it joins the island but does not propagate */
            }
            for (int k = 0; k < b->njoints; ++k) {
                joint_t *j = &w->joints[b->joints[k]];
                if (j->islandFlag) continue;
                int other = j->bodyA == bi ? j->bodyB : j->bodyA;
                is->joints[is->njoint++] = b->joints[k];
                j->islandFlag = 1;
                if (w->bodies[other].islandFlag) continue;
                stack[stackCount++] = other;
                w->bodies[other].islandFlag = 1;
            }
        }
        for (int k = 0; k < is->njoint; ++k) w->islandJoints[w->islandJointCount++] = is->joints[k];
        island_solve(w, is, step);
    }
    /* Synchronize fixtures in m_bodyList order, then look for new contacts. */
    for (int bi = w->nbody - 1; bi >= 0; --bi) {
        if (!w->bodies[bi].islandFlag) continue;
        body_sync_fixtures(w, bi);
    }
    find_new_contacts(w);
}

static void world_solve_toi(o_world *w, const step_t *step); /* A.8, below */

/* ---- A.2 b2World::Step ---- */
void rem2d_oracle_world_step(o_world *w, float dt, int velIters, int posIters) {
    if (w->newFixture) {
        find_new_contacts(w);
        w->newFixture = 0;
    }
    step_t step;
    step.dt = dt;
    step.velocityIterations = velIters;
    step.positionIterations = posIters;
    step.inv_dt = dt > 0.0f ? 1.0f / dt : 0.0f;
    step.dtRatio = w->inv_dt0 * dt;
    step.warmStarting = 1;
    collide(w);
    for (int i = 0; i < w->nbody; ++i) { /* (bookkeeping: capacity high-water marks) */
        const body_t *b = &w->bodies[i];
        int touching = 0;
        for (int k = 0; k < b->ncontacts; ++k) touching += w->contacts[b->contacts[k]].touching != 0;
        if (b->ncontacts > w->maxBodyPairs) w->maxBodyPairs = b->ncontacts;
        if (touching > w->maxBodyTouching) w->maxBodyTouching = touching;
    }
    if (w->stepComplete && step.dt > 0.0f) world_solve(w, &step);
    if ((w->flags & O_FLAG_CONTINUOUS) && step.dt > 0.0f) world_solve_toi(w, &step);
    if (step.dt > 0.0f) w->inv_dt0 = step.inv_dt;
    /* ClearForces: forces are never applied on this path */
}

/* ---- A.8/A.9 continuous collision: b2Distance, b2TimeOfImpact, b2World::SolveTOI ---- */
typedef struct { v2 verts[B2_MAX_POLY]; int count; f32 radius; } proxy_t;
static void proxy_set(proxy_t *p, const shape_t *s) {
    if (s->type == SH_CIRCLE) {
        p->verts[0] = s->p;
        p->count = 1;
        p->radius = s->radius;
    } else if (s->type == SH_POLY) {
        for (int i = 0; i < s->count; ++i) p->verts[i] = s->verts[i];
        p->count = s->count;
        p->radius = s->radius;
    } else {
        p->verts[0] = s->v1;
        p->verts[1] = s->v2_;
        p->count = 2;
        p->radius = s->radius;
    }
}
static int proxy_support(const proxy_t *p, v2 d) {
    int bestIndex = 0;
    f32 bestValue = vdot(p->verts[0], d);
    for (int i = 1; i < p->count; ++i) {
        f32 value = vdot(p->verts[i], d);
        if (value > bestValue) { bestIndex = i; bestValue = value; }
    }
    return bestIndex;
}
typedef struct { f32 metric; uint16_t count; uint8_t indexA[3], indexB[3]; } simplex_cache_t;
typedef struct { v2 wA, wB, w; f32 a; int indexA, indexB; } sv_t;
typedef struct { sv_t v[3]; int count; } simplex_t;

static f32 simplex_metric(const simplex_t *s) {
    switch (s->count) {
    case 1: return 0.0f;
    case 2: return sqrtf(vdist2(s->v[0].w, s->v[1].w));
    case 3: return vcross(vsub(s->v[1].w, s->v[0].w), vsub(s->v[2].w, s->v[0].w));
    default: return 0.0f;
    }
}
static void simplex_read_cache(simplex_t *s, const simplex_cache_t *cache, const proxy_t *pA, xf_t xfA,
                               const proxy_t *pB, xf_t xfB) {
    s->count = cache->count;
    for (int i = 0; i < s->count; ++i) {
        sv_t *v = &s->v[i];
        v->indexA = cache->indexA[i];
        v->indexB = cache->indexB[i];
        v2 wALocal = pA->verts[v->indexA], wBLocal = pB->verts[v->indexB];
        v->wA = xmul(xfA, wALocal);
        v->wB = xmul(xfB, wBLocal);
        v->w = vsub(v->wB, v->wA);
        v->a = 0.0f;
    }
    if (s->count > 1) {
        f32 metric1 = cache->metric;
        f32 metric2 = simplex_metric(s);
        if (metric2 < 0.5f * metric1 || 2.0f * metric1 < metric2 || metric2 < B2_EPSILON) s->count = 0;
    }
    if (s->count == 0) {
        sv_t *v = &s->v[0];
        v->indexA = 0;
        v->indexB = 0;
        v2 wALocal = pA->verts[0], wBLocal = pB->verts[0];
        v->wA = xmul(xfA, wALocal);
        v->wB = xmul(xfB, wBLocal);
        v->w = vsub(v->wB, v->wA);
        v->a = 1.0f;
        s->count = 1;
    }
}
static void simplex_write_cache(const simplex_t *s, simplex_cache_t *cache) {
    cache->metric = simplex_metric(s);
    cache->count = (uint16_t)s->count;
    for (int i = 0; i < s->count; ++i) {
        cache->indexA[i] = (uint8_t)s->v[i].indexA;
        cache->indexB[i] = (uint8_t)s->v[i].indexB;
    }
}
static v2 simplex_search_direction(const simplex_t *s) {
    switch (s->count) {
    case 1: return vneg(s->v[0].w);
    case 2: {
        v2 e12 = vsub(s->v[1].w, s->v[0].w);
        f32 sgn = vcross(e12, vneg(s->v[0].w));
        if (sgn > 0.0f) return vcross_sv(1.0f, e12);
        else return vcross_vs(e12, 1.0f);
    }
    default: return V2(0.0f, 0.0f);
    }
}
static v2 simplex_closest_point(const simplex_t *s) {
    switch (s->count) {
    case 1: return s->v[0].w;
    case 2: return vadd(vscale(s->v[0].a, s->v[0].w), vscale(s->v[1].a, s->v[1].w));
    default: return V2(0.0f, 0.0f);
    }
}
static void simplex_witness(const simplex_t *s, v2 *pA, v2 *pB) {
    switch (s->count) {
    case 1:
        *pA = s->v[0].wA;
        *pB = s->v[0].wB;
        break;
    case 2:
        *pA = vadd(vscale(s->v[0].a, s->v[0].wA), vscale(s->v[1].a, s->v[1].wA));
        *pB = vadd(vscale(s->v[0].a, s->v[0].wB), vscale(s->v[1].a, s->v[1].wB));
        break;
    case 3:
        *pA = vadd(vadd(vscale(s->v[0].a, s->v[0].wA), vscale(s->v[1].a, s->v[1].wA)),
                   vscale(s->v[2].a, s->v[2].wA));
        *pB = *pA;
        break;
    default: break;
    }
}
static void simplex_solve2(simplex_t *s) {
    v2 w1 = s->v[0].w, w2 = s->v[1].w;
    v2 e12 = vsub(w2, w1);
    f32 d12_2 = -vdot(w1, e12);
    if (d12_2 <= 0.0f) {
        s->v[0].a = 1.0f;
        s->count = 1;
        return;
    }
    f32 d12_1 = vdot(w2, e12);
    if (d12_1 <= 0.0f) {
        s->v[1].a = 1.0f;
        s->count = 1;
        s->v[0] = s->v[1];
        return;
    }
    f32 inv_d12 = 1.0f / (d12_1 + d12_2);
    s->v[0].a = d12_1 * inv_d12;
    s->v[1].a = d12_2 * inv_d12;
    s->count = 2;
}
static void simplex_solve3(simplex_t *s) {
    v2 w1 = s->v[0].w, w2 = s->v[1].w, w3 = s->v[2].w;
    v2 e12 = vsub(w2, w1);
    f32 w1e12 = vdot(w1, e12), w2e12 = vdot(w2, e12);
    f32 d12_1 = w2e12, d12_2 = -w1e12;
    v2 e13 = vsub(w3, w1);
    f32 w1e13 = vdot(w1, e13), w3e13 = vdot(w3, e13);
    f32 d13_1 = w3e13, d13_2 = -w1e13;
    v2 e23 = vsub(w3, w2);
    f32 w2e23 = vdot(w2, e23), w3e23 = vdot(w3, e23);
    f32 d23_1 = w3e23, d23_2 = -w2e23;
    f32 n123 = vcross(e12, e13);
    f32 d123_1 = n123 * vcross(w2, w3);
    f32 d123_2 = n123 * vcross(w3, w1);
    f32 d123_3 = n123 * vcross(w1, w2);
    if (d12_2 <= 0.0f && d13_2 <= 0.0f) {
        s->v[0].a = 1.0f;
        s->count = 1;
        return;
    }
    if (d12_1 > 0.0f && d12_2 > 0.0f && d123_3 <= 0.0f) {
        f32 inv_d12 = 1.0f / (d12_1 + d12_2);
        s->v[0].a = d12_1 * inv_d12;
        s->v[1].a = d12_2 * inv_d12;
        s->count = 2;
        return;
    }
    if (d13_1 > 0.0f && d13_2 > 0.0f && d123_2 <= 0.0f) {
        f32 inv_d13 = 1.0f / (d13_1 + d13_2);
        s->v[0].a = d13_1 * inv_d13;
        s->v[2].a = d13_2 * inv_d13;
        s->count = 2;
        s->v[1] = s->v[2];
        return;
    }
    if (d12_1 <= 0.0f && d23_2 <= 0.0f) {
        s->v[1].a = 1.0f;
        s->count = 1;
        s->v[0] = s->v[1];
        return;
    }
    if (d13_1 <= 0.0f && d23_1 <= 0.0f) {
        s->v[2].a = 1.0f;
        s->count = 1;
        s->v[0] = s->v[2];
        return;
    }
    if (d23_1 > 0.0f && d23_2 > 0.0f && d123_1 <= 0.0f) {
        f32 inv_d23 = 1.0f / (d23_1 + d23_2);
        s->v[1].a = d23_1 * inv_d23;
        s->v[2].a = d23_2 * inv_d23;
        s->count = 2;
        s->v[0] = s->v[2];
        return;
    }
    f32 inv_d123 = 1.0f / (d123_1 + d123_2 + d123_3);
    s->v[0].a = d123_1 * inv_d123;
    s->v[1].a = d123_2 * inv_d123;
    s->v[2].a = d123_3 * inv_d123;
    s->count = 3;
}
typedef struct { v2 pointA, pointB; f32 distance; int iterations; } dist_out_t;
/* b2Distance with useRadii = false */
static void b2distance(dist_out_t *out, simplex_cache_t *cache, const proxy_t *proxyA, xf_t xfA,
                       const proxy_t *proxyB, xf_t xfB) {
    simplex_t simplex;
    out->pointA = V2(0.0f, 0.0f);
    out->pointB = V2(0.0f, 0.0f);
    simplex_read_cache(&simplex, cache, proxyA, xfA, proxyB, xfB);
    sv_t *vertices = simplex.v;
    const int k_maxIters = 20;
    int saveA[3], saveB[3], saveCount = 0;
    f32 distanceSqr1 = B2_MAXFLOAT, distanceSqr2 = distanceSqr1;
    (void)distanceSqr2;
    int iter = 0;
    while (iter < k_maxIters) {
        saveCount = simplex.count;
        for (int i = 0; i < saveCount; ++i) {
            saveA[i] = vertices[i].indexA;
            saveB[i] = vertices[i].indexB;
        }
        switch (simplex.count) {
        case 1: break;
        case 2: simplex_solve2(&simplex); break;
        case 3: simplex_solve3(&simplex); break;
        default: break;
        }
        if (simplex.count == 3) break;
        v2 p = simplex_closest_point(&simplex);
        distanceSqr2 = vlen2(p);
        distanceSqr1 = distanceSqr2;
        v2 d = simplex_search_direction(&simplex);
        if (vlen2(d) < B2_EPSILON * B2_EPSILON) break;
        sv_t *vertex = vertices + simplex.count;
        vertex->indexA = proxy_support(proxyA, rmulT(xfA.q, vneg(d)));
        vertex->wA = xmul(xfA, proxyA->verts[vertex->indexA]);
        vertex->indexB = proxy_support(proxyB, rmulT(xfB.q, d));
        vertex->wB = xmul(xfB, proxyB->verts[vertex->indexB]);
        vertex->w = vsub(vertex->wB, vertex->wA);
        ++iter;
        int duplicate = 0;
        for (int i = 0; i < saveCount; ++i) {
            if (vertex->indexA == saveA[i] && vertex->indexB == saveB[i]) { duplicate = 1; break; }
        }
        if (duplicate) break;
        ++simplex.count;
    }
    (void)distanceSqr1;
    simplex_witness(&simplex, &out->pointA, &out->pointB);
    out->distance = sqrtf(vdist2(out->pointA, out->pointB));
    out->iterations = iter;
    simplex_write_cache(&simplex, cache);
}

typedef struct { v2 localCenter, c0, c; f32 a0, a, alpha0; } sweep_t;
static xf_t sweep_xf(const sweep_t *s, f32 beta) {
    xf_t xf;
    xf.p = vadd(vscale(1.0f - beta, s->c0), vscale(beta, s->c));
    f32 angle = (1.0f - beta) * s->a0 + beta * s->a;
    xf.q = rot_set(angle);
    xf.p = vsub(xf.p, rmul(xf.q, s->localCenter));
    return xf;
}
static void sweep_advance(sweep_t *s, f32 alpha) {
    f32 beta = (alpha - s->alpha0) / (1.0f - s->alpha0);
    s->c0 = vadd(s->c0, vscale(beta, vsub(s->c, s->c0)));
    s->a0 += beta * (s->a - s->a0);
    s->alpha0 = alpha;
}
static void sweep_normalize(sweep_t *s) {
    f32 twoPi = 2.0f * B2_PI;
    f32 d = twoPi * floorf(s->a0 / twoPi);
    s->a0 -= d;
    s->a -= d;
}
enum { SEP_POINTS = 0, SEP_FACE_A = 1, SEP_FACE_B = 2 };
typedef struct {
    const proxy_t *proxyA, *proxyB;
    sweep_t sweepA, sweepB;
    int type;
    v2 localPoint, axis;
} sepfn_t;
static f32 sepfn_init(sepfn_t *f, const simplex_cache_t *cache, const proxy_t *proxyA, const sweep_t *sweepA,
                      const proxy_t *proxyB, const sweep_t *sweepB, f32 t1) {
    f->proxyA = proxyA;
    f->proxyB = proxyB;
    int count = cache->count;
    f->sweepA = *sweepA;
    f->sweepB = *sweepB;
    xf_t xfA = sweep_xf(&f->sweepA, t1), xfB = sweep_xf(&f->sweepB, t1);
    if (count == 1) {
        f->type = SEP_POINTS;
        v2 localPointA = proxyA->verts[cache->indexA[0]];
        v2 localPointB = proxyB->verts[cache->indexB[0]];
        v2 pointA = xmul(xfA, localPointA), pointB = xmul(xfB, localPointB);
        f->axis = vsub(pointB, pointA);
        f32 s = vnormalize(&f->axis);
        return s;
    } else if (cache->indexA[0] == cache->indexA[1]) {
        f->type = SEP_FACE_B;
        v2 localPointB1 = proxyB->verts[cache->indexB[0]];
        v2 localPointB2 = proxyB->verts[cache->indexB[1]];
        f->axis = vcross_vs(vsub(localPointB2, localPointB1), 1.0f);
        vnormalize(&f->axis);
        v2 normal = rmul(xfB.q, f->axis);
        f->localPoint = vscale(0.5f, vadd(localPointB1, localPointB2));
        v2 pointB = xmul(xfB, f->localPoint);
        v2 localPointA = proxyA->verts[cache->indexA[0]];
        v2 pointA = xmul(xfA, localPointA);
        f32 s = vdot(vsub(pointA, pointB), normal);
        if (s < 0.0f) {
            f->axis = vneg(f->axis);
            s = -s;
        }
        return s;
    } else {
        f->type = SEP_FACE_A;
        v2 localPointA1 = proxyA->verts[cache->indexA[0]];
        v2 localPointA2 = proxyA->verts[cache->indexA[1]];
        f->axis = vcross_vs(vsub(localPointA2, localPointA1), 1.0f);
        vnormalize(&f->axis);
        v2 normal = rmul(xfA.q, f->axis);
        f->localPoint = vscale(0.5f, vadd(localPointA1, localPointA2));
        v2 pointA = xmul(xfA, f->localPoint);
        v2 localPointB = proxyB->verts[cache->indexB[0]];
        v2 pointB = xmul(xfB, localPointB);
        f32 s = vdot(vsub(pointB, pointA), normal);
        if (s < 0.0f) {
            f->axis = vneg(f->axis);
            s = -s;
        }
        return s;
    }
}
static f32 sepfn_find_min(const sepfn_t *f, int *indexA, int *indexB, f32 t) {
    xf_t xfA = sweep_xf(&f->sweepA, t), xfB = sweep_xf(&f->sweepB, t);
    switch (f->type) {
    case SEP_POINTS: {
        v2 axisA = rmulT(xfA.q, f->axis);
        v2 axisB = rmulT(xfB.q, vneg(f->axis));
        *indexA = proxy_support(f->proxyA, axisA);
        *indexB = proxy_support(f->proxyB, axisB);
        v2 pointA = xmul(xfA, f->proxyA->verts[*indexA]);
        v2 pointB = xmul(xfB, f->proxyB->verts[*indexB]);
        return vdot(vsub(pointB, pointA), f->axis);
    }
    case SEP_FACE_A: {
        v2 normal = rmul(xfA.q, f->axis);
        v2 pointA = xmul(xfA, f->localPoint);
        v2 axisB = rmulT(xfB.q, vneg(normal));
        *indexA = -1;
        *indexB = proxy_support(f->proxyB, axisB);
        v2 pointB = xmul(xfB, f->proxyB->verts[*indexB]);
        return vdot(vsub(pointB, pointA), normal);
    }
    default: {
        v2 normal = rmul(xfB.q, f->axis);
        v2 pointB = xmul(xfB, f->localPoint);
        v2 axisA = rmulT(xfA.q, vneg(normal));
        *indexB = -1;
        *indexA = proxy_support(f->proxyA, axisA);
        v2 pointA = xmul(xfA, f->proxyA->verts[*indexA]);
        return vdot(vsub(pointA, pointB), normal);
    }
    }
}
static f32 sepfn_evaluate(const sepfn_t *f, int indexA, int indexB, f32 t) {
    xf_t xfA = sweep_xf(&f->sweepA, t), xfB = sweep_xf(&f->sweepB, t);
    switch (f->type) {
    case SEP_POINTS: {
        v2 pointA = xmul(xfA, f->proxyA->verts[indexA]);
        v2 pointB = xmul(xfB, f->proxyB->verts[indexB]);
        return vdot(vsub(pointB, pointA), f->axis);
    }
    case SEP_FACE_A: {
        v2 normal = rmul(xfA.q, f->axis);
        v2 pointA = xmul(xfA, f->localPoint);
        v2 pointB = xmul(xfB, f->proxyB->verts[indexB]);
        return vdot(vsub(pointB, pointA), normal);
    }
    default: {
        v2 normal = rmul(xfB.q, f->axis);
        v2 pointB = xmul(xfB, f->localPoint);
        v2 pointA = xmul(xfA, f->proxyA->verts[indexA]);
        return vdot(vsub(pointA, pointB), normal);
    }
    }
}
enum { TOI_UNKNOWN = 0, TOI_FAILED, TOI_OVERLAPPED, TOI_TOUCHING, TOI_SEPARATED };
/* b2TimeOfImpact */
static void time_of_impact(int *state_out, f32 *t_out, const proxy_t *proxyA, const sweep_t *sweepA_in,
                           const proxy_t *proxyB, const sweep_t *sweepB_in, f32 tMax) {
    *state_out = TOI_UNKNOWN;
    *t_out = tMax;
    sweep_t sweepA = *sweepA_in, sweepB = *sweepB_in;
    sweep_normalize(&sweepA);
    sweep_normalize(&sweepB);
    f32 totalRadius = proxyA->radius + proxyB->radius;
    f32 target = fmax32(B2_LINEAR_SLOP, totalRadius - 3.0f * B2_LINEAR_SLOP);
    f32 tolerance = 0.25f * B2_LINEAR_SLOP;
    f32 t1 = 0.0f;
    const int k_maxIterations = 20;
    int iter = 0;
    simplex_cache_t cache;
    memset(&cache, 0, sizeof(cache));
    cache.count = 0;
    for (;;) {
        xf_t xfA = sweep_xf(&sweepA, t1), xfB = sweep_xf(&sweepB, t1);
        dist_out_t dout;
        b2distance(&dout, &cache, proxyA, xfA, proxyB, xfB);
        if (dout.distance <= 0.0f) {
            *state_out = TOI_OVERLAPPED;
            *t_out = 0.0f;
            break;
        }
        if (dout.distance < target + tolerance) {
            *state_out = TOI_TOUCHING;
            *t_out = t1;
            break;
        }
        sepfn_t fcn;
        sepfn_init(&fcn, &cache, proxyA, &sweepA, proxyB, &sweepB, t1);
        int done = 0;
        f32 t2 = tMax;
        int pushBackIter = 0;
        for (;;) {
            int indexA, indexB;
            f32 s2 = sepfn_find_min(&fcn, &indexA, &indexB, t2);
            if (s2 > target + tolerance) {
                *state_out = TOI_SEPARATED;
                *t_out = tMax;
                done = 1;
                break;
            }
            if (s2 > target - tolerance) {
                t1 = t2;
                break;
            }
            f32 s1 = sepfn_evaluate(&fcn, indexA, indexB, t1);
            if (s1 < target - tolerance) {
                *state_out = TOI_FAILED;
                *t_out = t1;
                done = 1;
                break;
            }
            if (s1 <= target + tolerance) {
                *state_out = TOI_TOUCHING;
                *t_out = t1;
                done = 1;
                break;
            }
            int rootIterCount = 0;
            f32 a1 = t1, a2 = t2;
            for (;;) {
                f32 t;
                if (rootIterCount & 1) t = a1 + (target - s1) * (a2 - a1) / (s2 - s1);
                else t = 0.5f * (a1 + a2);
                ++rootIterCount;
                f32 s = sepfn_evaluate(&fcn, indexA, indexB, t);
                if (fabs32(s - target) < tolerance) {
                    t2 = t;
                    break;
                }
                if (s > target) {
                    a1 = t;
                    s1 = s;
                } else {
                    a2 = t;
                    s2 = s;
                }
                if (rootIterCount == 50) break;
            }
            ++pushBackIter;
            if (pushBackIter == B2_MAX_POLY) break;
        }
        ++iter;
        if (done) break;
        if (iter == k_maxIterations) {
            *state_out = TOI_FAILED;
            *t_out = t1;
            break;
        }
    }
}

static sweep_t body_sweep(const body_t *b) {
    sweep_t s;
    s.localCenter = b->localCenter;
    s.c0 = b->c0;
    s.c = b->c;
    s.a0 = b->a0;
    s.a = b->a;
    s.alpha0 = b->alpha0;
    return s;
}
static void body_set_sweep(body_t *b, const sweep_t *s) {
    b->c0 = s->c0;
    b->c = s->c;
    b->a0 = s->a0;
    b->a = s->a;
    b->alpha0 = s->alpha0;
}
/* b2Body::Advance */
static void body_advance(body_t *b, f32 alpha) {
    sweep_t s = body_sweep(b);
    sweep_advance(&s, alpha);
    s.c = s.c0;
    s.a = s.a0;
    body_set_sweep(b, &s);
    b->xf.q = rot_set(b->a);
    b->xf.p = vsub(b->c, rmul(b->xf.q, b->localCenter));
}

/* b2ContactSolver::SolveTOIPositionConstraints(toiIndexA, toiIndexB): the static body never moves. */
static int contact_solver_solve_toi_position(island_t *is, int toiIndexB) {
    f32 minSeparation = 0.0f;
    for (int i = 0; i < is->ncontact; ++i) {
        pc_t *pc = &is->pcs[i];
        int indexA = pc->indexA, indexB = pc->indexB;
        v2 localCenterA = pc->localCenterA, localCenterB = pc->localCenterB;
        int pointCount = pc->pointCount;
        f32 mA = 0.0f, iA = 0.0f; /* indexA is the static body: never a TOI index on this path */
        f32 mB = 0.0f, iB = 0.0f;
        if (indexB == toiIndexB) {
            mB = pc->invMassB;
            iB = pc->invIB;
        }
        v2 cA = is->positions[indexA].c;
        f32 aA = is->positions[indexA].a;
        v2 cB = is->positions[indexB].c;
        f32 aB = is->positions[indexB].a;
        for (int j = 0; j < pointCount; ++j) {
            xf_t xfA, xfB;
            xfA.q = rot_set(aA);
            xfB.q = rot_set(aB);
            xfA.p = vsub(cA, rmul(xfA.q, localCenterA));
            xfB.p = vsub(cB, rmul(xfB.q, localCenterB));
            v2 normal, point;
            f32 separation;
            psm_init(pc, xfA, xfB, j, &normal, &point, &separation);
            v2 rA = vsub(point, cA), rB = vsub(point, cB);
            minSeparation = fmin32(minSeparation, separation);
            f32 C = fclamp(B2_TOI_BAUGARTE * (separation + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
            f32 rnA = vcross(rA, normal), rnB = vcross(rB, normal);
            f32 K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
            f32 impulse = K > 0.0f ? -C / K : 0.0f;
            v2 P = vscale(impulse, normal);
            cA = vsub(cA, vscale(mA, P));
            aA -= iA * vcross(rA, P);
            cB = vadd(cB, vscale(mB, P));
            aB += iB * vcross(rB, P);
        }
        is->positions[indexA].c = cA;
        is->positions[indexA].a = aA;
        is->positions[indexB].c = cB;
        is->positions[indexB].a = aB;
    }
    return minSeparation >= -1.5f * B2_LINEAR_SLOP;
}
/* b2Island::SolveTOI(subStep, toiIndexA, toiIndexB) */
static void island_solve_toi(o_world *w, island_t *is, const step_t *subStep, int toiIndexB) {
    is->positions[IDX_STATIC].c = V2(0.0f, 0.0f);
    is->positions[IDX_STATIC].a = 0.0f;
    is->velocities[IDX_STATIC].v = V2(0.0f, 0.0f);
    is->velocities[IDX_STATIC].w = 0.0f;
    for (int i = 0; i < is->nbody; ++i) {
        body_t *b = &w->bodies[is->bodies[i]];
        is->positions[i].c = b->c;
        is->positions[i].a = b->a;
        is->velocities[i].v = b->v;
        is->velocities[i].w = b->w;
    }
    contact_solver_setup(w, is, subStep);
    for (int i = 0; i < subStep->positionIterations; ++i) {
        int contactsOkay = contact_solver_solve_toi_position(is, toiIndexB);
        if (contactsOkay) break;
    }
    /* Leap of faith to new safe state. */
    w->bodies[is->bodies[toiIndexB]].c0 = is->positions[toiIndexB].c;
    w->bodies[is->bodies[toiIndexB]].a0 = is->positions[toiIndexB].a;
    contact_solver_init_velocity(w, is);
#ifdef REM2D_ORACLE_PROBE
    {   /* diagnostic: after how many sweeps does the TOI sub-step's state revisit a state of 1..PROBE_TOI_P sweeps ago? */
        static _Thread_local f32 ring[PROBE_TOI_P + 1][8 + 4 * B2_MAX_TOI_CONTACTS];
        int found = 0;
        for (int i = 0; i < subStep->velocityIterations; ++i) {
            contact_solver_solve_velocity(is);
            f32 *o = ring[i % (PROBE_TOI_P + 1)];
            int n = 0;
            o[n++] = is->velocities[0].v.x; o[n++] = is->velocities[0].v.y; o[n++] = is->velocities[0].w;
            for (int c = 0; c < is->ncontact; ++c)
                for (int k = 0; k < is->vcs[c].pointCount; ++k) { o[n++] = is->vcs[c].points[k].normalImpulse; o[n++] = is->vcs[c].points[k].tangentImpulse; }
            if (!found)
                for (int p = 1; p <= PROBE_TOI_P && p <= i; ++p)
                    if (memcmp(ring[i % (PROBE_TOI_P + 1)], ring[(i - p) % (PROBE_TOI_P + 1)], (size_t)n * sizeof(f32)) == 0) {
                        __atomic_fetch_add(&g_probe_toi[p][i < 255 ? i : 255], 1, __ATOMIC_RELAXED);
                        found = 1;
                        break;
                    }
        }
        if (!found) __atomic_fetch_add(&g_probe_toi[0][255], 1, __ATOMIC_RELAXED);
    }
#else
    for (int i = 0; i < subStep->velocityIterations; ++i) contact_solver_solve_velocity(is);
#endif
    f32 h = subStep->dt;
    for (int i = 0; i < is->nbody; ++i) {
        v2 c = is->positions[i].c;
        f32 a = is->positions[i].a;
        v2 v = is->velocities[i].v;
        f32 wz = is->velocities[i].w;
        v2 translation = vscale(h, v);
        if (vdot(translation, translation) > B2_MAX_TRANSLATION_SQ) {
            f32 ratio = B2_MAX_TRANSLATION / vlen(translation);
            v = vscale(ratio, v);
        }
        f32 rotation = h * wz;
        if (rotation * rotation > B2_MAX_ROTATION_SQ) {
            f32 ratio = B2_MAX_ROTATION / fabs32(rotation);
            wz *= ratio;
        }
        c = vadd(c, vscale(h, v));
        a += h * wz;
        is->positions[i].c = c;
        is->positions[i].a = a;
        is->velocities[i].v = v;
        is->velocities[i].w = wz;
        body_t *b = &w->bodies[is->bodies[i]];
        b->c = c;
        b->a = a;
        b->v = v;
        b->w = wz;
        body_sync_transform(b);
    }
}
/* b2Sweep::Advance on a static terrain body: c0 == c and a0 == a (the body never moves), so
 * c0 += beta * (c - c0) leaves the pose bit-identical and only alpha0 changes. */
static void static_advance(o_world *w, int s, f32 alpha) { w->staticAlpha0[s] = alpha; }

/* b2World::SolveTOI.  Every contact on this path is (static A, dynamic non-bullet B).  Each terrain edge /
 * hardcore box is its own static b2Body (Modular2DEnv.py:294-306, 217-275), so the sweep.alpha0 / e_islandFlag
 * bookkeeping Box2D does on body A is modelled per static (O_FLAG_TOI_TRANSPARENT_STATICS switches it off: the
 * round-1 form, kept to show that both give the same bits -- DESIGN.md section 2 has the argument). */
static void world_solve_toi(o_world *w, const step_t *step) {
    static _Thread_local island_t island;
    island_t *is = &island;
    const int bookkeeping = (w->flags & O_FLAG_TOI_TRANSPARENT_STATICS) == 0;
    const int nstatic = w->terrain->nstatic;
    if (w->stepComplete) {
        for (int i = 0; i < w->nbody; ++i) {
            w->bodies[i].islandFlag = 0;
            w->bodies[i].alpha0 = 0.0f;
        }
        for (int s = 0; s < nstatic; ++s) { /* the static bodies are in m_bodyList too */
            w->staticIslandFlag[s] = 0;
            w->staticAlpha0[s] = 0.0f;
        }
        for (int k = 0; k < w->wcount; ++k) {
            contact_t *c = &w->contacts[w->wlist[k]];
            c->toiFlag = 0;
            c->islandFlag = 0;
            c->toiCount = 0;
            c->toi = 1.0f;
        }
    }
    for (;;) {
        int minContact = -1;
        f32 minAlpha = 1.0f;
        for (int k = 0; k < w->wcount; ++k) {
            contact_t *c = &w->contacts[w->wlist[k]];
            if (!c->enabled) continue;
            if (c->toiCount > B2_MAX_SUB_STEPS) continue;
            f32 alpha = 1.0f;
            if (c->toiFlag) {
                alpha = c->toi;
            } else {
                body_t *bB = &w->bodies[c->body];
                int activeB = bB->awake;
                if (!activeB) continue; /* static A is never active */
                /* collideA = bulletA || typeB != dynamic = false; collideB = typeA != dynamic = true */
                /* Put the sweeps onto the same time interval. */
                f32 alphaA = bookkeeping ? w->staticAlpha0[c->staticIdx] : 0.0f;
                f32 alpha0 = alphaA;
                if (alphaA < bB->alpha0) {
                    alpha0 = bB->alpha0;
                    if (bookkeeping) static_advance(w, c->staticIdx, alpha0); /* bA->m_sweep.Advance(alpha0) */
                } else if (bB->alpha0 < alphaA) {
                    alpha0 = alphaA;
                    sweep_t sb = body_sweep(bB);
                    sweep_advance(&sb, alpha0); /* bB->m_sweep.Advance(alpha0): moves c0 / a0 of B */
                    body_set_sweep(bB, &sb);
                    w->toiDynamicAdvances++;
                }
                proxy_t pA, pB;
                proxy_set(&pA, &w->terrain->statics[c->staticIdx].shape);
                proxy_set(&pB, &bB->shape);
                sweep_t sA;
                memset(&sA, 0, sizeof(sA));
                sA.alpha0 = alpha0; /* static sweep advanced to alpha0: c0 = c = 0, a0 = a = 0 */
                sweep_t sB = body_sweep(bB);
                int state;
                f32 t;
                time_of_impact(&state, &t, &pA, &sA, &pB, &sB, 1.0f);
                f32 beta = t;
                if (state == TOI_TOUCHING) alpha = fmin32(alpha0 + (1.0f - alpha0) * beta, 1.0f);
                else alpha = 1.0f;
                c->toi = alpha;
                c->toiFlag = 1;
            }
            if (alpha < minAlpha) {
                minContact = w->wlist[k];
                minAlpha = alpha;
            }
        }
        if (minContact < 0 || 1.0f - 10.0f * B2_EPSILON < minAlpha) {
            w->stepComplete = 1;
            break;
        }
        contact_t *mc = &w->contacts[minContact];
        int bBi = mc->body;
        body_t *bB = &w->bodies[bBi];
        const int sA0 = mc->staticIdx;
        f32 backup1 = w->staticAlpha0[sA0]; /* b2Sweep backup1 = bA->m_sweep (only alpha0 can differ) */
        sweep_t backup2 = body_sweep(bB);
        if (bookkeeping) static_advance(w, sA0, minAlpha); /* bA->Advance(minAlpha) */
        body_advance(bB, minAlpha);
        contact_update(w, mc);
        mc->toiFlag = 0;
        ++mc->toiCount;
        if (!mc->enabled || !mc->touching) {
            mc->enabled = 0;
            w->staticAlpha0[sA0] = backup1;
            body_set_sweep(bB, &backup2);
            body_sync_transform(bB);
            continue;
        }
        body_set_awake(w, bB, 1);
        w->toiEvents++;
        /* Build the island: static A (slot IDX_STATIC), B, the contact, then B's other contacts */
        is->nbody = is->ncontact = is->njoint = 0;
        bB->islandIndex = 0;
        is->bodies[is->nbody++] = bBi;
        is->contacts[is->ncontact++] = minContact;
        bB->islandFlag = 1;
        mc->islandFlag = 1;
        int islandStatics[B2_MAX_TOI_CONTACTS + 1], nIslandStatics = 0;
        w->staticIslandFlag[sA0] = 1;
        islandStatics[nIslandStatics++] = sA0;
        /* bodies[2] = {bA, bB}: bA is static (m_type != b2_dynamicBody), only bB's contact list is walked */
        for (int k = 0; k < bB->ncontacts; ++k) {
            /* island.m_bodyCapacity = 2 * b2_maxTOIContacts is never reached before the contact capacity:
             * every static added below comes with a contact */
            if (is->ncontact == B2_MAX_TOI_CONTACTS) break;
            int ci = bB->contacts[k];
            contact_t *c = &w->contacts[ci];
            if (c->islandFlag) continue;
            /* other is static: always allowed (only dynamic non-bullet others are skipped) */
            const int so = c->staticIdx;
            /* Tentatively advance the body to the TOI. */
            f32 backup = w->staticAlpha0[so];
            if (bookkeeping && !w->staticIslandFlag[so]) static_advance(w, so, minAlpha);
            contact_update(w, c);
            if (!c->enabled) { w->staticAlpha0[so] = backup; continue; }
            if (!c->touching) { w->staticAlpha0[so] = backup; continue; }
            c->islandFlag = 1;
            is->contacts[is->ncontact++] = ci;
            if (w->staticIslandFlag[so]) continue;
            w->staticIslandFlag[so] = 1; /* other->SetAwake is skipped for static bodies */
            islandStatics[nIslandStatics++] = so;
        }
        step_t subStep;
        subStep.dt = (1.0f - minAlpha) * step->dt;
        subStep.inv_dt = 1.0f / subStep.dt;
        subStep.dtRatio = 1.0f;
        subStep.positionIterations = 20;
        subStep.velocityIterations = step->velocityIterations;
        subStep.warmStarting = 0;
        island_solve_toi(w, is, &subStep, 0);
        for (int i = 0; i < nIslandStatics; ++i) w->staticIslandFlag[islandStatics[i]] = 0;
        for (int i = 0; i < is->nbody; ++i) {
            body_t *b = &w->bodies[is->bodies[i]];
            b->islandFlag = 0;
            body_sync_fixtures(w, is->bodies[i]);
            for (int k = 0; k < b->ncontacts; ++k) {
                contact_t *c = &w->contacts[b->contacts[k]];
                c->toiFlag = 0;
                c->islandFlag = 0;
            }
        }
        find_new_contacts(w);
        /* m_subStepping is false */
    }
}

/* ---- Modular2D.step (Modular2DEnv.py:607-653) ---- */
#define FPS 50
#define WOD_SPEED 0.04
void rem2d_oracle_env_step(o_world *w, double *reward_out, int *done_out) {
    rem2d_oracle_env_step_ex(w, (float)(1.0 / FPS), 6 * 30, 2 * 30, reward_out, done_out); /* :634 */
}
/* the same step with explicit b2World::Step arguments (mirror of rem2d_world_step_ex) */
void rem2d_oracle_env_step_ex(o_world *w, float dt, int velIters, int posIters, double *reward_out, int *done_out) {
    w->wod += WOD_SPEED; /* :613-614 */
    /* controller sweep (:620-623) in node order, then PID -> motorSpeed (:631-632) */
    double cvals[O_MAX_BODIES];
    for (int i = 0; i < w->njoint; ++i) {
        joint_t *j = &w->joints[i];
        j->istate += j->freq; /* m_controller.py:18-19 (input is always 0) */
        cvals[i] = (j->amp * rem2d_oracle_sin(j->istate + j->phase)) + j->offset;
    }
    for (int i = 0; i < w->njoint; ++i) {
        joint_t *j = &w->joints[i];
        /* joint.angle = GetJointAngle(): float32 */
        f32 jointAngle = w->bodies[j->bodyB].a - w->bodies[j->bodyA].a - j->referenceAngle;
        double angleDifference = cvals[i] - (double)jointAngle;
        double speed = angleDifference * 1.9;
        rem2d_oracle_set_motor_speed(w, i, (float)speed);
    }
    rem2d_oracle_world_step(w, dt, velIters, posIters);
    double reward = 0.0;
    int done = 0;
    if (w->nbody > 0) {
        double rootx = (double)w->bodies[0].xf.p.x;
        reward = rootx; /* :642 */
        if (rootx < 0.0) { reward = -100.0; done = 1; }
        if (w->wod > rootx) { reward = -100.0; done = 1; }
    }
    if (reward_out) *reward_out = reward;
    if (done_out) *done_out = done;
}

/* ---- getters ---- */
int rem2d_oracle_num_bodies(const o_world *w) { return w->nbody; }
int rem2d_oracle_num_joints(const o_world *w) { return w->njoint; }
void rem2d_oracle_get_bodies(const o_world *w, float *out) {
    for (int i = 0; i < w->nbody; ++i) {
        const body_t *b = &w->bodies[i];
        float *o = out + i * 8;
        o[0] = b->xf.p.x;
        o[1] = b->xf.p.y;
        o[2] = b->a;
        o[3] = b->v.x;
        o[4] = b->v.y;
        o[5] = b->w;
        o[6] = b->sleepTime;
        o[7] = (float)b->awake;
    }
}
void rem2d_oracle_get_mass(const o_world *w, float *out) {
    for (int i = 0; i < w->nbody; ++i) {
        const body_t *b = &w->bodies[i];
        out[i * 4 + 0] = b->invMass;
        out[i * 4 + 1] = b->invI;
        out[i * 4 + 2] = b->mass;
        out[i * 4 + 3] = b->I;
    }
}
void rem2d_oracle_get_joints(const o_world *w, float *out) {
    for (int i = 0; i < w->njoint; ++i) {
        const joint_t *j = &w->joints[i];
        float *o = out + i * 6;
        o[0] = j->impulse.x;
        o[1] = j->impulse.y;
        o[2] = j->impulse.z;
        o[3] = j->motorImpulse;
        o[4] = j->motorSpeed;
        o[5] = (float)j->limitState;
    }
}
int rem2d_oracle_get_island_joint_order(const o_world *w, int32_t *out) {
    for (int i = 0; i < w->islandJointCount; ++i) out[i] = w->islandJoints[i];
    return w->islandJointCount;
}
int rem2d_oracle_get_contacts(const o_world *w, int body, int32_t *out, float *fout) {
    const body_t *b = &w->bodies[body];
    for (int k = 0; k < b->ncontacts; ++k) {
        const contact_t *c = &w->contacts[b->contacts[k]];
        int32_t *o = out + k * 8;
        o[0] = c->staticIdx;
        o[1] = c->m.pointCount;
        o[2] = c->m.type;
        o[3] = c->touching;
        o[4] = (int32_t)c->m.points[0].id.key;
        o[5] = (int32_t)c->m.points[1].id.key;
        o[6] = 0;
        o[7] = 0;
        if (fout) {
            fout[k * 4 + 0] = c->m.points[0].normalImpulse;
            fout[k * 4 + 1] = c->m.points[1].normalImpulse;
            fout[k * 4 + 2] = c->m.points[0].tangentImpulse;
            fout[k * 4 + 3] = c->m.points[1].tangentImpulse;
        }
    }
    return b->ncontacts;
}
void rem2d_oracle_get_manifold(const o_world *w, int body, int k, float *out) {
    const contact_t *c = &w->contacts[w->bodies[body].contacts[k]];
    out[0] = c->m.localNormal.x;
    out[1] = c->m.localNormal.y;
    out[2] = c->m.localPoint.x;
    out[3] = c->m.localPoint.y;
    out[4] = c->m.points[0].localPoint.x;
    out[5] = c->m.points[0].localPoint.y;
    out[6] = c->m.points[1].localPoint.x;
    out[7] = c->m.points[1].localPoint.y;
}
void rem2d_oracle_get_fat_aabb(const o_world *w, int body, float *out) {
    const body_t *b = &w->bodies[body];
    out[0] = b->fat.lower.x;
    out[1] = b->fat.lower.y;
    out[2] = b->fat.upper.x;
    out[3] = b->fat.upper.y;
}
int rem2d_oracle_position_iterations(const o_world *w) { return w->lastPositionIterations; }
int rem2d_oracle_toi_events(const o_world *w) { return w->toiEvents; }
int rem2d_oracle_toi_dynamic_advances(const o_world *w) { return w->toiDynamicAdvances; }

int rem2d_oracle_is_f64(void) { return (int)(sizeof(f32) == 8); }
void rem2d_oracle_box_mass(float hx, float hy, float *mass, float *I) {
    shape_t s;
    shape_set_box(&s, hx, hy);
    v2 c;
    f32 m, i;
    poly_mass(&s, 1.0f, &m, &c, &i);
    *mass = (float)m;
    *I = (float)i;
}
void rem2d_oracle_circle_mass(float r, float *mass, float *I) {
    f32 m, i;
    circle_mass(r, V2(0.0f, 0.0f), 1.0f, &m, &i);
    *mass = (float)m;
    *I = (float)i;
}

/* ---- standalone entry points for the known-answer tests (tests/test_oracle_kat.py) ---- */
/* shape spec: {0, x1, y1, x2, y2} edge | {1, hx, hy} box | {2, r} circle | {3, n, x0, y0, ...} convex polygon */
static int kat_shape(shape_t *s, const float *spec) {
    int type = (int)spec[0];
    memset(s, 0, sizeof(*s));
    if (type == 0) {
        s->type = SH_EDGE;
        s->radius = B2_POLYGON_RADIUS;
        s->v1 = V2(spec[1], spec[2]);
        s->v2_ = V2(spec[3], spec[4]);
    } else if (type == 1) {
        shape_set_box(s, spec[1], spec[2]);
    } else if (type == 2) {
        s->type = SH_CIRCLE;
        s->radius = spec[1];
        s->p = V2(0.0f, 0.0f);
    } else if (type == 3) {
        int n = (int)spec[1];
        v2 vs[B2_MAX_POLY];
        if (n < 3 || n > B2_MAX_POLY) return -1;
        for (int i = 0; i < n; ++i) vs[i] = V2(spec[2 + 2 * i], spec[3 + 2 * i]);
        if (shape_set_poly(s, vs, n) != 0) return -1;
    } else return -1;
    return 0;
}
static xf_t kat_xf(const float *x) {
    xf_t t;
    t.p = V2(x[0], x[1]);
    t.q = rot_set(x[2]);
    return t;
}
/* b2Distance (useRadii = false): out = pointA.xy pointB.xy distance iterations */
int rem2d_oracle_kat_distance(const float *specA, const float *xfA, const float *specB, const float *xfB, float *out) {
    shape_t a, b;
    if (kat_shape(&a, specA) || kat_shape(&b, specB)) return -1;
    proxy_t pa, pb;
    proxy_set(&pa, &a);
    proxy_set(&pb, &b);
    simplex_cache_t cache;
    memset(&cache, 0, sizeof(cache));
    dist_out_t d;
    b2distance(&d, &cache, &pa, kat_xf(xfA), &pb, kat_xf(xfB));
    out[0] = (float)d.pointA.x; out[1] = (float)d.pointA.y; out[2] = (float)d.pointB.x; out[3] = (float)d.pointB.y;
    out[4] = (float)d.distance; out[5] = (float)d.iterations;
    return 0;
}
/* b2TimeOfImpact: sweep = c0.xy a0 c.xy a (localCenter 0, alpha0 0); out = state (TOI_*), t */
int rem2d_oracle_kat_toi(const float *specA, const float *sweepA, const float *specB, const float *sweepB, float tMax,
                         float *out) {
    shape_t a, b;
    if (kat_shape(&a, specA) || kat_shape(&b, specB)) return -1;
    proxy_t pa, pb;
    proxy_set(&pa, &a);
    proxy_set(&pb, &b);
    sweep_t sa, sb;
    memset(&sa, 0, sizeof(sa));
    memset(&sb, 0, sizeof(sb));
    sa.c0 = V2(sweepA[0], sweepA[1]); sa.a0 = sweepA[2]; sa.c = V2(sweepA[3], sweepA[4]); sa.a = sweepA[5];
    sb.c0 = V2(sweepB[0], sweepB[1]); sb.a0 = sweepB[2]; sb.c = V2(sweepB[3], sweepB[4]); sb.a = sweepB[5];
    int state;
    f32 t;
    time_of_impact(&state, &t, &pa, &sa, &pb, &sb, tMax);
    out[0] = (float)state;
    out[1] = (float)t;
    return 0;
}
/* narrowphase dispatch of contact_evaluate; iout = type pointCount key0 key1; fout = localNormal.xy localPoint.xy
 * p0.xy p1.xy */
int rem2d_oracle_kat_collide(const float *specA, const float *xfA, const float *specB, const float *xfB, int32_t *iout,
                             float *fout) {
    shape_t a, b;
    if (kat_shape(&a, specA) || kat_shape(&b, specB)) return -1;
    manifold_t m;
    memset(&m, 0, sizeof(m));
    if (a.type == SH_EDGE && b.type == SH_POLY) collide_edge_polygon(&m, &a, kat_xf(xfA), &b, kat_xf(xfB));
    else if (a.type == SH_EDGE && b.type == SH_CIRCLE) collide_edge_circle(&m, &a, kat_xf(xfA), &b, kat_xf(xfB));
    else if (a.type == SH_POLY && b.type == SH_POLY) collide_polygons(&m, &a, kat_xf(xfA), &b, kat_xf(xfB));
    else if (a.type == SH_POLY && b.type == SH_CIRCLE) collide_polygon_circle(&m, &a, kat_xf(xfA), &b, kat_xf(xfB));
    else return -1;
    iout[0] = m.type; iout[1] = m.pointCount; iout[2] = (int32_t)m.points[0].id.key; iout[3] = (int32_t)m.points[1].id.key;
    fout[0] = (float)m.localNormal.x; fout[1] = (float)m.localNormal.y; fout[2] = (float)m.localPoint.x; fout[3] = (float)m.localPoint.y;
    fout[4] = (float)m.points[0].localPoint.x; fout[5] = (float)m.points[0].localPoint.y;
    fout[6] = (float)m.points[1].localPoint.x; fout[7] = (float)m.points[1].localPoint.y;
    return 0;
}
/* One b2ContactSolver::SolveVelocityConstraints sweep over ONE contact between the static body (A) and a body B at
 * cB: in = normal.xy, nPoints, p0.xy, p1.xy (world manifold points), cB.xy, invMassB, invIB, friction, vB.xy, wB,
 * accumulated normalImpulse0/1, tangentImpulse0/1.  out = vB.xy wB normalImpulse0/1 tangentImpulse0/1 pointCount
 * (the block solver may have dropped to one point). */
int rem2d_oracle_kat_contact_solve(const float *in, float *out) {
    static _Thread_local island_t island;
    island_t *is = &island;
    is->nbody = 1;
    is->ncontact = 1;
    is->njoint = 0;
    vc_t *vc = &is->vcs[0];
    memset(vc, 0, sizeof(*vc));
    vc->normal = V2(in[0], in[1]);
    vc->pointCount = (int)in[2];
    v2 pts[2] = {V2(in[3], in[4]), V2(in[5], in[6])};
    v2 cB = V2(in[7], in[8]);
    vc->indexA = IDX_STATIC;
    vc->indexB = 0;
    vc->invMassA = 0.0f; vc->invIA = 0.0f;
    vc->invMassB = in[9]; vc->invIB = in[10];
    vc->friction = in[11];
    is->velocities[IDX_STATIC].v = V2(0.0f, 0.0f);
    is->velocities[IDX_STATIC].w = 0.0f;
    is->velocities[0].v = V2(in[12], in[13]);
    is->velocities[0].w = in[14];
    vc_init_masses(vc, pts, V2(0.0f, 0.0f), cB, V2(0.0f, 0.0f), 0.0f, is->velocities[0].v, is->velocities[0].w);
    vc->points[0].normalImpulse = in[15]; vc->points[1].normalImpulse = in[16];
    vc->points[0].tangentImpulse = in[17]; vc->points[1].tangentImpulse = in[18];
    contact_solver_solve_velocity(is);
    out[0] = (float)is->velocities[0].v.x; out[1] = (float)is->velocities[0].v.y; out[2] = (float)is->velocities[0].w;
    out[3] = (float)vc->points[0].normalImpulse; out[4] = (float)vc->points[1].normalImpulse;
    out[5] = (float)vc->points[0].tangentImpulse; out[6] = (float)vc->points[1].tangentImpulse;
    out[7] = (float)vc->pointCount;
    return 0;
}
/* b2Min / b2Max / b2Clamp as the step uses them (fmin32 / fmax32 / fclamp above), element by element: out = [3][n]
 * (min(a, b), max(a, b), clamp(a, lo = b, hi = c)).  The checker of the HIP library's rem2d_selftest_scalar, which
 * evaluates the same three through v_med3_f32 (tests/test_parity_gpu.py::test_scalar_helpers_special_values). */
int rem2d_oracle_kat_scalar(const float *a, const float *b, const float *c, int32_t n, float *out) {
    for (int32_t i = 0; i < n; ++i) {
        out[i] = (float)fmin32((f32)a[i], (f32)b[i]);
        out[(size_t)n + i] = (float)fmax32((f32)a[i], (f32)b[i]);
        out[2 * (size_t)n + i] = (float)fclamp((f32)a[i], (f32)b[i], (f32)c[i]);
    }
    return 0;
}

/* ---- batch driver ---- */
static long long g_batchToiEvents, g_batchToiDynamicAdvances; /* summed over the worlds of rem2d_oracle_batch_run */
void rem2d_oracle_batch_toi_stats(long long *events, long long *dynamic_advances, int reset) {
    if (events) *events = __atomic_load_n(&g_batchToiEvents, __ATOMIC_RELAXED);
    if (dynamic_advances) *dynamic_advances = __atomic_load_n(&g_batchToiDynamicAdvances, __ATOMIC_RELAXED);
    if (reset) {
        __atomic_store_n(&g_batchToiEvents, 0, __ATOMIC_RELAXED);
        __atomic_store_n(&g_batchToiDynamicAdvances, 0, __ATOMIC_RELAXED);
    }
}
o_world *rem2d_oracle_world_from_morph(const o_terrain *t, const o_morph *m, int e, unsigned flags) {
    o_world *w = rem2d_oracle_world_create(t, flags);
    int K = m->lanes;
    int slot2body[O_MAX_BODIES];
    for (int s = 0; s < K && s < O_MAX_BODIES; ++s) {
        int i = e * K + s;
        slot2body[s] = -1;
        if (m->shape[i] == 1) slot2body[s] = rem2d_oracle_add_box(w, m->hx[i], m->hy[i], m->x[i], m->y[i], m->angle[i]);
        else if (m->shape[i] == 2) slot2body[s] = rem2d_oracle_add_circle(w, m->hx[i], m->x[i], m->y[i], m->angle[i]);
        else continue;
        if (m->parent[i] >= 0) {
            int ji = rem2d_oracle_add_joint(w, slot2body[m->parent[i]], slot2body[s], m->ax[i], m->ay[i],
                                            m->bx[i], m->by[i], m->torque[i], m->lower[i], m->upper[i]);
            rem2d_oracle_set_controller(w, ji, m->amp[i], m->phase[i], m->freq[i], m->offset[i], m->istate[i]);
        }
    }
    return w;
}

static int batch_run_impl(const o_terrain *t, const o_morph *m, int n_steps, int n_threads, unsigned flags,
                          float *bodies_out, double *reward_out, int32_t *done_out, double *fitness_out,
                          float *trace_out, int32_t *caps_out);
int rem2d_oracle_batch_run(const o_terrain *t, const o_morph *m, int n_steps, int n_threads, unsigned flags,
                           float *bodies_out, double *reward_out, int32_t *done_out, double *fitness_out,
                           float *trace_out) {
    return batch_run_impl(t, m, n_steps, n_threads, flags, bodies_out, reward_out, done_out, fitness_out, trace_out, NULL);
}
/* the same, plus caps_out [N][3]: most pairs on one body, most touching manifolds on one body (both sampled while
 * evaluate()'s loop would still be running for that creature, REM2D_main.py:362-377), refused pairs (> O_MAX_BODY_CONTACTS) */
int rem2d_oracle_batch_run_caps(const o_terrain *t, const o_morph *m, int n_steps, int n_threads, unsigned flags,
                                float *bodies_out, double *reward_out, int32_t *done_out, double *fitness_out,
                                int32_t *caps_out) {
    return batch_run_impl(t, m, n_steps, n_threads, flags, bodies_out, reward_out, done_out, fitness_out, NULL, caps_out);
}
static int batch_run_impl(const o_terrain *t, const o_morph *m, int n_steps, int n_threads, unsigned flags,
                          float *bodies_out, double *reward_out, int32_t *done_out, double *fitness_out,
                          float *trace_out, int32_t *caps_out) {
    int N = m->n_envs, K = m->lanes;
    if (K > O_MAX_BODIES) return -1;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int e = 0; e < N; ++e) {
        o_world *w = rem2d_oracle_world_from_morph(t, m, e, flags);
        double reward = 0.0, fitness = 0.0;
        int done = 0, everDone = 0, fitnessFrozen = 0;
        /* slot -> body map: bodies are created in slot order, skipping empty slots */
        int slotBody[O_MAX_BODIES], nb = 0;
        for (int s = 0; s < K; ++s) slotBody[s] = (m->shape[e * K + s] != 0) ? nb++ : -1;
        float st[O_MAX_BODIES * 8];
        int capPairs = 0, capTouch = 0;
        for (int step = 0; step < n_steps; ++step) {
            rem2d_oracle_env_step(w, &reward, &done);
            everDone |= done;
            if (!fitnessFrozen) { capPairs = w->maxBodyPairs; capTouch = w->maxBodyTouching; }
            /* evaluate() (REM2D_main.py:362-377) with EVALUATION_STEPS = 10000, ENV_LENGTH = 100 */
            if (!fitnessFrozen) {
                if (reward < -10.0) fitnessFrozen = 1;
                else if (reward > 100.0) {
                    fitness = reward + (double)(10000 - step) / 10000.0;
                    fitnessFrozen = 1;
                } else if (reward > 0.0) fitness = reward;
            }
            if (trace_out) {
                rem2d_oracle_get_bodies(w, st);
                for (int s = 0; s < K; ++s) {
                    float *o = trace_out + (((size_t)step * N + e) * K + s) * 3;
                    if (slotBody[s] >= 0) {
                        o[0] = st[slotBody[s] * 8 + 0];
                        o[1] = st[slotBody[s] * 8 + 1];
                        o[2] = st[slotBody[s] * 8 + 2];
                    } else o[0] = o[1] = o[2] = 0.0f;
                }
            }
        }
        if (bodies_out) {
            rem2d_oracle_get_bodies(w, st);
            for (int s = 0; s < K; ++s) {
                float *o = bodies_out + ((size_t)e * K + s) * 8;
                for (int q = 0; q < 8; ++q) o[q] = slotBody[s] >= 0 ? st[slotBody[s] * 8 + q] : 0.0f;
            }
        }
        if (reward_out) reward_out[e] = reward;
        if (done_out) done_out[e] = everDone;
        if (fitness_out) fitness_out[e] = fitness;
        if (caps_out) { caps_out[3 * e] = capPairs; caps_out[3 * e + 1] = capTouch; caps_out[3 * e + 2] = w->overflow; }
        __atomic_fetch_add(&g_batchToiEvents, (long long)w->toiEvents, __ATOMIC_RELAXED);
        __atomic_fetch_add(&g_batchToiDynamicAdvances, (long long)w->toiDynamicAdvances, __ATOMIC_RELAXED);
        rem2d_oracle_world_destroy(w);
    }
    return 0;
}

/* bench.py's cpu_baseline leg: ONE continuous timed window.  Every creature of the batch is first stepped `settle` times
 * (untimed, worlds kept), then the wall clock runs around `window` further steps of all of them (OpenMP over creatures, as
 * in rem2d_oracle_batch_run).  *seconds_out = wall time of the window only; returns 0, or -1 on a bad batch / allocation. */
#include <time.h>
static double wall_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
int rem2d_oracle_batch_window(const o_terrain *t, const o_morph *m, int settle, int window, int n_threads, unsigned flags,
                              double *seconds_out, double *reward_out) {
    int N = m->n_envs;
    if (m->lanes > O_MAX_BODIES || N <= 0 || !seconds_out) return -1;
    o_world **ws = (o_world **)calloc((size_t)N, sizeof(o_world *));
    if (!ws) return -1;
    (void)n_threads;
    int bad = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int e = 0; e < N; ++e) {
        double reward = 0.0;
        int done = 0;
        ws[e] = rem2d_oracle_world_from_morph(t, m, e, flags);
        if (!ws[e]) { __atomic_store_n(&bad, 1, __ATOMIC_RELAXED); continue; }
        for (int step = 0; step < settle; ++step) rem2d_oracle_env_step(ws[e], &reward, &done);
    }
    if (!bad) {
        const double t0 = wall_seconds();
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
#endif
        for (int e = 0; e < N; ++e) {
            double reward = 0.0;
            int done = 0;
            for (int step = 0; step < window; ++step) rem2d_oracle_env_step(ws[e], &reward, &done);
            if (reward_out) reward_out[e] = reward;
        }
        *seconds_out = wall_seconds() - t0;
    }
    for (int e = 0; e < N; ++e)
        if (ws[e]) rem2d_oracle_world_destroy(ws[e]);
    free(ws);
    return bad ? -1 : 0;
}
