// ora_det.cpp — TEST INFRASTRUCTURE (CPU oracle).  The transcendental functions of the step path as fixed operation sequences.
//
// The reference calls the C runtime's acos / atan2 / sin / cos (src/physics/constraints.cpp:1162, 1835, 1874, 1895;
// src/core/math.cpp:582; src/core/math.h:934-935).  MSVC's CRT, glibc and the GPU's ocml differ in the last ulp, so the
// oracle, the product (d3d12renderer_amd/csrc/dmath.hpp, same sequence) and the reference build under oracle/_ref (which
// links THIS file and routes the reference's calls here, oracle/refbuild/ref_pch.h) all use these; their error against
// libm is bounded in tests/test_transcendentals.py (< 1e-13 before the final rounding to float).
#include <cmath>
#include "ora_math.h"

namespace ora {

// atan(t) = t*Q(t^2) on [0, tan(pi/8)] with Q a degree-8 Chebyshev-fitted polynomial evaluated in
// double by Horner without FMA (max abs error 1e-14, i.e. the float rounding of the result
// dominates); [tan(pi/8), 1] is folded with atan(a) = pi/4 + atan((a-1)/(a+1)); a > 1 with
// atan(a) = pi/2 - atan(1/a).
static double atan_poly(double t) {
    double u = t * t;
    double q = 3.07024230903805012e-02;
    q = q * u + -5.87725109466260137e-02;
    q = q * u + 7.56446973448029469e-02;
    q = q * u + -9.07850346883833786e-02;
    q = q * u + 1.11103940245759952e-01;
    q = q * u + -1.42856908172007746e-01;
    q = q * u + 1.99999996133686908e-01;
    q = q * u + -3.33333333308731938e-01;
    q = q * u + 9.99999999999974576e-01;
    return q * t;
}
static double atan01(double a) {  // a in [0,1]
    if (a > 0.41421356237309503) return 0.78539816339744828 + atan_poly((a - 1.0) / (a + 1.0));
    return atan_poly(a);
}
float det_atan2f(float y, float x) {
    double ax = std::fabs((double)x), ay = std::fabs((double)y);
    double r;
    if (ax == 0.0 && ay == 0.0) r = 0.0;
    else if (ay <= ax) r = atan01(ay / ax);
    else r = 1.5707963267948966 - atan01(ax / ay);
    if (x < 0.f) r = 3.141592653589793 - r;
    if (y < 0.f) r = -r;
    return (float)r;
}
float det_acosf(float x) {
    float c = clampf(x, -1.f, 1.f);
    double d = (double)c;
    double s = std::sqrt((1.0 - d) * (1.0 + d));
    return det_atan2f((float)s, c);
}
// sin/cos on [-pi, pi] via degree-limited Taylor/minimax in double after quadrant folding.
static double sin_core(double x) {  // |x| <= pi/4
    double x2 = x * x;
    double p = -2.5052108385441720e-08;
    p = p * x2 + 2.7557319223985893e-06;
    p = p * x2 + -1.9841269841269841e-04;
    p = p * x2 + 8.3333333333333332e-03;
    p = p * x2 + -1.6666666666666666e-01;
    return x + x * x2 * p;
}
static double cos_core(double x) {  // |x| <= pi/4
    double x2 = x * x;
    double p = 2.0876756987868100e-09;
    p = p * x2 + -2.7557319223985888e-07;
    p = p * x2 + 2.4801587301587302e-05;
    p = p * x2 + -1.3888888888888889e-03;
    p = p * x2 + 4.1666666666666664e-02;
    p = p * x2 + -0.5;
    return 1.0 + x2 * p;
}
static void sincos_det(double x, double& s, double& c) {
    double q = std::floor(x * 0.6366197723675814 + 0.5);  // nearest multiple of pi/2
    double r = x - q * 1.5707963267948966;
    r = r - q * 6.123233995736766e-17;
    long long k = (long long)q;
    double sr = sin_core(r), cr = cos_core(r);
    switch (k & 3) {
        case 0: s = sr; c = cr; break;
        case 1: s = cr; c = -sr; break;
        case 2: s = -sr; c = -cr; break;
        default: s = -cr; c = sr; break;
    }
}
float det_sinf(float x) { double s, c; sincos_det((double)x, s, c); return (float)s; }
float det_cosf(float x) { double s, c; sincos_det((double)x, s, c); return (float)c; }

}  // namespace ora
