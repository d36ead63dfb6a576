// rem2d_math.h -- device math in b2Math.h operand order, "rem2d trig", K-lane group reductions.
// Part of the single translation unit rem2d.hip (see its header comment); not a stand-alone header.
#ifndef REM2D_MATH_H
#define REM2D_MATH_H

// =====================================================================================
// device math (b2Math.h operand order)
// =====================================================================================
struct V2 { float x, y; };
#define DEV __device__ __forceinline__
DEV V2 mk(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
DEV V2 vadd(V2 a, V2 b) { return mk(a.x + b.x, a.y + b.y); }
DEV V2 vsub(V2 a, V2 b) { return mk(a.x - b.x, a.y - b.y); }
DEV V2 vneg(V2 a) { return mk(-a.x, -a.y); }
DEV V2 vscale(float s, V2 a) { return mk(s * a.x, s * a.y); }
DEV float vdot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
DEV float vcross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
DEV V2 vcross_vs(V2 a, float s) { return mk(s * a.y, -s * a.x); }
DEV V2 vcross_sv(float s, V2 a) { return mk(-s * a.y, s * a.x); }
DEV float vlen(V2 a) { return sqrtf(a.x * a.x + a.y * a.y); }
DEV float vdist2(V2 a, V2 b) { V2 c = vsub(a, b); return vdot(c, c); }
// b2Min / b2Max (a < b ? a : b, a > b ? a : b) as the median with -inf / +inf: one v_med3_f32 instead of a compare, a wait
// state and a select.  The same value for every pair of numbers; of two zeros of different sign the other one may come
// out (see fclamp below).
DEV float fmin32(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -__builtin_inff()); }
DEV float fmax32(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
DEV float fabs32(float a) { return a > 0.0f ? a : -a; }
// b2Clamp(a, lo, hi) = b2Max(lo, b2Min(a, hi)) for lo <= hi: the median of the three -- one v_med3_f32 instead of two
// compare / select pairs (each with a wait state for the SGPR condition).  Same bits for every input but a zero: where
// Box2D's form picks a bound's +0 the median may keep the argument's -0; a zero stays a zero through everything that
// follows (sums, products, comparisons against 0 -- there is no division by it), and -0 == +0 in every comparison the
// tests make (the digests add +0.0 first).
DEV float fclamp(float a, float lo, float hi) { return __builtin_amdgcn_fmed3f(a, lo, hi); }
DEV V2 vmin2(V2 a, V2 b) { return mk(fmin32(a.x, b.x), fmin32(a.y, b.y)); }
DEV V2 vmax2(V2 a, V2 b) { return mk(fmax32(a.x, b.x), fmax32(a.y, b.y)); }
DEV float vnormalize(V2 &a) {
    float length = vlen(a);
    if (length < B2_EPSILON) return 0.0f;
    float inv = 1.0f / length;
    a.x *= inv;
    a.y *= inv;
    return length;
}
struct Rot { float s, c; };
DEV V2 rmul(Rot q, V2 v) { return mk(q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y); }
DEV V2 rmulT(Rot q, V2 v) { return mk(q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y); }
DEV V2 xmul(Rot q, V2 p, V2 v) {
    float x = (q.c * v.x - q.s * v.y) + p.x;
    float y = (q.s * v.x + q.c * v.y) + p.y;
    return mk(x, y);
}
DEV V2 xmulT(Rot q, V2 p, V2 v) {
    float px = v.x - p.x, py = v.y - p.y;
    return mk(q.c * px + q.s * py, -q.s * px + q.c * py);
}

// ---- trig (DESIGN.md "rem2d trig"): binary64 form for the controller's math.sin ----
DEV void dev_sincos_d(double x, double &s, double &c) {
    const double INV_PIO2 = 6.36619772367581382433e-01, PIO2_1 = 1.57079632673412561417e+00,
                 PIO2_1T = 6.07710050650619224932e-11;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double fn = rint(x * INV_PIO2);
    int n = (int)fn;
    double r = (x - fn * PIO2_1) - fn * PIO2_1T;
    double z = r * r;
    double ps = r + r * (z * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))))));
    double pc = (1.0 - 0.5 * z) + z * z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    int q = n & 3;
    double ss = (q & 1) ? pc : ps, cc = (q & 1) ? ps : pc;
    s = (q == 2 || q == 3) ? -ss : ss;
    c = (q == 1 || q == 2) ? -cc : cc;
}
// b2Rot::Set -- "rem2d trig" binary32 form: 3-term Cody-Waite reduction by pi/2 and the
// Cephes sinf/cosf minimax polynomials on [-pi/4, pi/4], every operation a separately rounded binary32 op.
DEV Rot rot_set(float x) {
    const float TWO_OVER_PI = 0.63661977236758134308f;
    const float DP1 = 1.5703125f, DP2 = 4.837512969970703125e-4f, DP3 = 7.54978995489188216e-8f;
    const float S1 = -1.6666654611e-1f, S2 = 8.3321608736e-3f, S3 = -1.9515295891e-4f;
    const float C1 = 4.166664568298827e-2f, C2 = -1.388731625493765e-3f, C3 = 2.443315711809948e-5f;
    float fn = rintf(x * TWO_OVER_PI);
    int n = (int)fn;
    // (rem2d trig: every step of the reduction and of the two polynomials is ONE fused multiply-add -- a single, exactly
    // defined rounding each, the same on v_fma_f32 and on the oracle's fmaf)
    float r = __builtin_fmaf(-fn, DP3, __builtin_fmaf(-fn, DP2, __builtin_fmaf(-fn, DP1, x)));
    float z = r * r;
    float ps = __builtin_fmaf(r, z * __builtin_fmaf(z, __builtin_fmaf(z, S3, S2), S1), r);
    float pc = __builtin_fmaf(z * z, __builtin_fmaf(z, __builtin_fmaf(z, C3, C2), C1), __builtin_fmaf(z, -0.5f, 1.0f));
    float ss = (n & 1) ? pc : ps, cc = (n & 1) ? ps : pc;
    Rot o;
    // quadrants 2, 3 negate the sine, quadrants 1, 2 the cosine: the sign bit flipped by bit 1 of n / of n + 1 (the same
    // bits as the compare-and-select form, in three integer instructions each)
    o.s = __uint_as_float(__float_as_uint(ss) ^ (((unsigned)n & 2u) << 30));
    o.c = __uint_as_float(__float_as_uint(cc) ^ ((((unsigned)n + 1u) & 2u) << 30));
    return o;
}
DEV double dev_sin(double x) {
    double s, c;
    dev_sincos_d(x, s, c);
    return s;
}

// ---- K-lane group reductions (K consecutive lanes = one creature) ----
template <int K> DEV float group_min(float v) {
#pragma unroll
    for (int o = 1; o < K; o <<= 1) {
        float t = __shfl_xor(v, o);
        v = t < v ? t : v;
    }
    return v;
}
template <int K> DEV int group_and(int v) {
#pragma unroll
    for (int o = 1; o < K; o <<= 1) v &= __shfl_xor(v, o);
    return v;
}
template <int K> DEV int group_or(int v) {
#pragma unroll
    for (int o = 1; o < K; o <<= 1) v |= __shfl_xor(v, o);
    return v;
}
// Mailbox hand-off between lanes of ONE wave (workgroup == wavefront).  The LDS operations of a wave execute in issue
// order, so a ds_read issued after a ds_write of another lane of the same wave sees the written value without any
// wait in between; all that is needed is that the compiler neither reorders the LDS accesses across this point nor
// forwards stale values: release + acquire at WAVEFRONT scope (no s_waitcnt is emitted for it -- the only waits left
// are the ones in front of the first use of loaded data; at workgroup scope every hand-off drained lgkmcnt, ~100
// cycles of exposed latency per slot of the solver loops).  REM2D_LDS_SYNC_WORKGROUP restores the conservative form.
DEV void lds_sync() {
#ifdef REM2D_LDS_SYNC_WORKGROUP
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
// the same for code that only SOME lanes of the wavefront execute (a shuffle would read the registers of the inactive lanes):
// the maximum of 0 <= v < 2^bits over the ACTIVE lanes, bit by bit from the top with ballots; a scalar.
DEV int wave_max_active(int v, int bits) {
    int r = 0;
    bool cand = true;
    for (int b = bits - 1; b >= 0; --b) {
        const bool has = cand && ((v >> b) & 1) != 0;
        if (__ballot(has)) { // some candidate has this bit: the maximum has it, candidates without it drop out
            r |= 1 << b;
            cand = has;
        }
    }
    return r;
}
DEV int wave_max(int v) {
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        int t = __shfl_xor(v, o);
        v = t > v ? t : v;
    }
    // (every lane holds the same value now: hand it out as a scalar, so that the loops and branches it controls are
    // scalar loops and branches instead of lane-mask juggling around every slot of the solver loops)
    return __builtin_amdgcn_readfirstlane(v);
}

#endif
