/* rt_strict_math.h — bit-reproducible f32 math shared by host and gfx950 device code.
 *
 * The reference (Rust, f32) reaches the platform libm through
 *   f32::sin / f32::cos   src/material.rs:180            (diffuse scatter, phi in [0, 2*pi])
 *   f32::atan2 / f32::acos src/shape.rs:102-103           (sphere surface uv)
 * and libm is not pinned by Cargo.lock (glibc vs musl builds already differ in the last ulp,
 * SURVEY.md §8c iii).  Discrete decisions downstream of these calls (checker parity, texel
 * index, next-hit selection) re-route a pixel's whole PCG stream, so host oracle and device
 * kernel must agree to the last bit.  This header fixes that by construction: every function is
 * evaluated in IEEE binary64 with plain mul/add/div/sqrt plus EXPLICIT fused multiply-adds
 * (rt_fma_d; no implicit contraction: build with -ffp-contract=off on both sides) and rounded once
 * to binary32.  The binary64 evaluation error
 * is < 2^-48 relative, so the result is the correctly rounded f32 value except when the exact
 * value lies within ~2^-24 ulp of a rounding boundary.
 *
 * Everything else the hot path needs (sqrt, /, floor, copysign, fabs) is exact IEEE on both
 * sides and is used directly.
 *
 * Compiles as C++ (g++) and as HIP device code (hipcc --offload-arch=gfx950).
 */
#ifndef RT_STRICT_MATH_H
#define RT_STRICT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD static inline
#endif

/* Binary64 literals of the polynomial kernels.  On the device each one is pinned to a scalar register
 * pair at its use (s_mov_b32 x2 on the scalar unit, then the SGPR operand of v_fma_f64): left alone,
 * the compiler parks every literal in a VGPR pair for the whole render loop (36 VGPRs) and feeds
 * v_fmac_f64 through a v_mov_b64 per Horner step. */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double rt_k_sgpr(double c) { asm volatile("" : "+s"(c)); return c; }
#define RT_K(x) rt_k_sgpr(x)
#else
#define RT_K(x) (x)
#endif

/* ---- small exact helpers ------------------------------------------------------------------ */

/* Rust `x as u32` for f32: saturating, NaN -> 0 (material.rs:56-57, :88-89). */
RT_HD uint32_t rt_f32_to_u32_sat(float x) {
    if (!(x > 0.0f)) return 0u;               /* NaN, -x, +-0 */
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;                       /* truncation toward zero */
}

/* Rust f32::max: if one operand is NaN the other is returned (render.rs:148, :150). */
RT_HD float rt_maxf(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    /* v_max_f32: IEEE maxNum, the same value (the sign of a zero result is never observed by the
     * renderers: lum > 0, 1 - lum and max(.., 0.05) do not depend on it) */
    return __builtin_fmaxf(a, b);
#else
    if (a != a) return b;
    if (b != b) return a;
    return (a > b) ? a : b;
#endif
}

/* ---- sin / cos ---------------------------------------------------------------------------- */

/* Cody-Waite reduction by pi/2 in three 33-bit pieces (n*RT_PIO2_1 is exact for |n| < 2^20),
 * then Taylor kernels on |r| <= pi/4, Horner steps fused (rt_fma_d).  Accurate domain |x| <= 1e6;
 * the hot path only calls it with x = 2*pi*rf, rf in [0,1]. */
#define RT_INV_PIO2 0.6366197723675814      /* 2/pi                                        */
#define RT_PIO2_1 1.5707963267341256        /* 0x1.921fb54400000p+0  first 33 bits of pi/2 */
#define RT_PIO2_2 6.077100506303966e-11     /* 0x1.0b4611a600000p-34 next 33 bits          */
#define RT_PIO2_3 2.0222662487959506e-21    /* 0x1.3198a2e037073p-69 remainder             */

RT_HD double rt_rint_d(double x) { return __builtin_rint(x); }
/* Correctly rounded fused multiply-add in binary64: v_fma_f64 on the device, fma() (hardware FMA
 * through glibc's ifunc where the CPU has it, exact software otherwise) on the host.  Fusion here
 * is part of the definition of these functions; the reference's own f32 arithmetic is never fused. */
RT_HD double rt_fma_d(double a, double b, double c) { return __builtin_fma(a, b, c); }

/* sin(r) = r + r^3 * P(r^2), Taylor through r^15: truncation < 4.7e-17 on |r| <= pi/4. */
RT_HD double rt_ksin_d(double r, double z) {
    double p = RT_K(-7.647163731819816e-13);                /* -1/15! */
    p = rt_fma_d(p, z, RT_K(1.6059043836821613e-10));       /*  1/13! */
    p = rt_fma_d(p, z, RT_K(-2.505210838544172e-08));       /* -1/11! */
    p = rt_fma_d(p, z, RT_K(2.7557319223985893e-06));       /*  1/9!  */
    p = rt_fma_d(p, z, RT_K(-0.0001984126984126984));       /* -1/7!  */
    p = rt_fma_d(p, z, RT_K(0.008333333333333333));         /*  1/5!  */
    p = rt_fma_d(p, z, RT_K(-0.16666666666666666));         /* -1/3!  */
    return rt_fma_d(r * z, p, r);
}

/* cos(r) = 1 + r^2 * Q(r^2), Taylor through r^16: truncation < 2e-18 on |r| <= pi/4. */
RT_HD double rt_kcos_d(double z) {
    double p = RT_K(4.779477332387385e-14);                 /*  1/16! */
    p = rt_fma_d(p, z, RT_K(-1.1470745597729725e-11));      /* -1/14! */
    p = rt_fma_d(p, z, RT_K(2.08767569878681e-09));         /*  1/12! */
    p = rt_fma_d(p, z, RT_K(-2.755731922398589e-07));       /* -1/10! */
    p = rt_fma_d(p, z, RT_K(2.48015873015873e-05));         /*  1/8!  */
    p = rt_fma_d(p, z, RT_K(-0.001388888888888889));        /* -1/6!  */
    p = rt_fma_d(p, z, RT_K(0.041666666666666664));         /*  1/4!  */
    p = rt_fma_d(p, z, RT_K(-0.5));                         /* -1/2!  */
    return rt_fma_d(z, p, 1.0);
}

/* sin and cos of an f32 argument, each rounded once to f32 (the quadrant's sign change and swap
 * commute with that rounding). */
RT_HD void rt_sincosf(float xf, float* s, float* c) {
    const double x = (double)xf;
    const double fn = rt_rint_d(x * RT_K(RT_INV_PIO2));
    double r = rt_fma_d(-fn, RT_K(RT_PIO2_1), x);
    r = rt_fma_d(-fn, RT_K(RT_PIO2_2), r);
    r = rt_fma_d(-fn, RT_K(RT_PIO2_3), r);
    const double z = r * r;
    const float ks = (float)rt_ksin_d(r, z);
    const float kc = (float)rt_kcos_d(z);
    const int q = (int)fn & 3;
    const float sv = (q & 1) ? kc : ks;
    const float cv = (q & 1) ? ks : kc;
    *s = (q & 2) ? -sv : sv;                  /* q = 2, 3: sin < 0 side */
    *c = ((q + 1) & 2) ? -cv : cv;            /* q = 1, 2 */
}

RT_HD float rt_sinf(float x) { float s, c; rt_sincosf(x, &s, &c); return s; }
RT_HD float rt_cosf(float x) { float s, c; rt_sincosf(x, &s, &c); return c; }

/* ---- atan2 / acos ------------------------------------------------------------------------- */

#define RT_PI_D 3.141592653589793
#define RT_PIO2_D 1.5707963267948966

/* atan(a) for a in [0,1]: a = c + d with c = i/8, atan(a) = atan(c) + atan((a-c)/(1+a*c)),
 * |t| <= 1/16, odd Taylor series through t^17. */
RT_HD double rt_atan01_d(double a) {
    const double k = rt_rint_d(a * 8.0);
    const int i = (int)k;
    double base;
    switch (i) {
        case 0: base = 0.0; break;
        case 1: base = RT_K(0.12435499454676144); break;
        case 2: base = RT_K(0.24497866312686414); break;
        case 3: base = RT_K(0.35877067027057225); break;
        case 4: base = RT_K(0.4636476090008061); break;
        case 5: base = RT_K(0.5585993153435624); break;
        case 6: base = RT_K(0.6435011087932844); break;
        case 7: base = RT_K(0.7188299996216245); break;
        default: base = RT_K(0.7853981633974483); break;
    }
    const double c = k * 0.125;
    const double t = (a - c) / (1.0 + a * c);
    const double z = t * t;
    double p = RT_K(0.058823529411764705);          /*  1/17 */
    p = p * z - RT_K(0.06666666666666667);          /* -1/15 */
    p = p * z + RT_K(0.07692307692307693);          /*  1/13 */
    p = p * z - RT_K(0.09090909090909091);          /* -1/11 */
    p = p * z + RT_K(0.1111111111111111);           /*  1/9  */
    p = p * z - RT_K(0.14285714285714285);          /* -1/7  */
    p = p * z + RT_K(0.2);                          /*  1/5  */
    p = p * z - RT_K(0.3333333333333333);           /* -1/3  */
    return base + (t + (t * z) * p);
}

/* atan2 in binary64 for finite or infinite operands; NaN in -> NaN out. */
RT_HD double rt_atan2_d(double y, double x) {
    if (x != x || y != y) return x + y;
    const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    double r;
    if (ay == 0.0) {
        r = 0.0;                               /* atan2(+-0, x) */
    } else if (ax == ay) {
        r = RT_K(0.7853981633974483);          /* includes inf/inf */
    } else if (ax > ay) {
        r = rt_atan01_d(ay / ax);              /* ay/inf = 0 */
    } else {
        r = RT_K(RT_PIO2_D) - rt_atan01_d(ax / ay);
    }
    if (__builtin_signbit(x)) r = RT_K(RT_PI_D) - r;
    return __builtin_signbit(y) ? -r : r;
}

RT_HD float rt_atan2f(float y, float x) { return (float)rt_atan2_d((double)y, (double)x); }

/* acos(z) = atan2(sqrt((1-z)(1+z)), z); NaN for |z| > 1 like libm (shape.rs:103 relies on the
 * NaN, which Rust's saturating cast then turns into texel/checker index 0). */
RT_HD float rt_acosf(float zf) {
    const double z = (double)zf;
    if (!(__builtin_fabs(z) <= 1.0)) return __builtin_nanf("");
    const double s = __builtin_sqrt((1.0 - z) * (1.0 + z));
    return (float)rt_atan2_d(s, z);
}

#endif /* RT_STRICT_MATH_H */
