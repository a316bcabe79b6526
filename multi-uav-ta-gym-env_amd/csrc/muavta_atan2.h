// muavta_atan2.h — atan2(y, x) as the host's libm computes it (included by muavta_math.h inside namespace muavta; the host-side check
// tests/atan2_host_check.c includes it too, with MUAVTA_ATAN2_HOST defined).
//
// Why it is here: core_sim's avoid_obstacles (core_sim/src/sim_core.rs:25-59) turns an agent left or right of an obstacle by the SIGN of
//     ang = (atan2(mov.y, mov.x) - atan2(dy, dx) + PI) % (2 PI) - PI,
// f64::atan2 being the platform libm's atan2.  An agent whose task lies behind an obstacle is steered ONTO the line through the
// obstacle's centre (the rule pushes it back across that line every step), where the two angles agree to an ulp and the sign of
// `ang` is decided by the last bit of each atan2 — so an atan2 that differs from the host's in the last bit (ocml's does, for about
// one argument in a few hundred) sends the agent round the other side.  Found by the device fuzz (tests/fuzz_device.py config
// 32517, scored leg, t = 182: both angles -0x1.978fec4a46805p+0 / ...806p+0).  glibc 2.35's atan2 is NOT correctly rounded (about
// one result in a thousand is the other neighbour: tools/atan2_cr_probe.py, profiles/r05_atan2_glibc_vs_cr.txt), so matching it means
// restating ITS algorithm, not rounding better:
//
// glibc 2.35 sysdeps/ieee754/dbl-64/e_atan2.c (IBM Accurate Mathematical Library; since glibc 2.34 only the first, ~0.55 ulp, stage
// is left): with ax = |x|, ay = |y| (scaled by 2^+-500 when tiny / huge), u = min / max as a double-double (u, du) by one
// division, an exact product and a second division;  u < 1/16: the odd Taylor polynomial d3..d13 in v = u^2;  otherwise the node
// table cij[i] = {x_i, atan(x_i), c2..c6} with i = round(256 u) - 16 and atan(u) = atan(x_i) + (v c2 + (dv c2 + v^2 (c3 + v (c4 +
// v (c5 + v c6))))) for (v, dv) = (u - x_i) + du;  then atan, pi/2 -+ atan or pi - atan by quadrant, pi and pi/2 as double-doubles
// (opi + opi1, hpi + hpi1), and the sign of y.  The multiply-adds are fused exactly where the FMA build of this image's libm
// (the ifunc variant every FMA + AVX2 host picks) fuses them — read off its instruction stream, noted per line below — and nowhere
// else (the library is compiled with -ffp-contract=off).  The table is data of that libm (tools/gen_atan2_table.py).
// tests: tests/test_abi_cpu.py::test_atan2_restatement_vs_host_libm (this file compiled for the host against the host's atan2,
// 9 M arguments over every branch) and tests/test_gpu_parity.py::test_libm_atan2_bit_exact (the device function against the host's).
#ifdef MUAVTA_ATAN2_HOST
#define ATAN2_FN static inline
#define ATAN2_TAB static const
#define ATAN2_BITS(x) ({ double x_ = (x); unsigned long long u_; __builtin_memcpy(&u_, &x_, 8); u_; })
#else
#define ATAN2_FN DEV
#define ATAN2_TAB __constant__
#define ATAN2_BITS(x) ((unsigned long long)__double_as_longlong(x))
#endif

ATAN2_TAB double LIBM_ATAN2_TAB[241][7] = {
#include "muavta_atan2_tab.inc"
};

ATAN2_FN double libm_atan2(double y, double x) {
  const double hpi = 0x1.921fb54442d18p+0, hpi1 = 0x1.1a62633145c07p-54, opi = 0x1.921fb54442d18p+1, opi1 = 0x1.1a62633145c07p-53;
  const double qpi = 0x1.921fb54442d18p-1, tqpi = 0x1.2d97c7f3321d2p+1;
  const double d3 = -0x1.5555555555555p-2, d5 = 0x1.99999999997fdp-3, d7 = -0x1.24924923f7603p-3, d9 = 0x1.c71c6e5129a3bp-4,
               d11 = -0x1.7458022b13c25p-4, d13 = 0x1.375f08b31cbcep-4;
  const unsigned long long bx = ATAN2_BITS(x), by = ATAN2_BITS(y);
  const unsigned int ux = (unsigned int)(bx >> 32), uy = (unsigned int)(by >> 32);
  const bool xneg = (bx >> 63) != 0, yneg = (by >> 63) != 0;
  // ---- special operands (e_atan2.c's first block; C99 F.9.1.4)
  if (x != x || y != y) return x + y;
  if ((by << 1) == 0) return xneg ? (yneg ? -opi : opi) : y;                      // y = +-0: +-0 for x >= +0, +-pi for x <= -0
  if ((bx << 1) == 0) return yneg ? -hpi : hpi;                                   // x = +-0
  const bool xinf = (bx << 1) == 0xffe0000000000000ull, yinf = (by << 1) == 0xffe0000000000000ull;
  if (xinf) {
    const double r = xneg ? (yinf ? tqpi : opi) : (yinf ? qpi : 0.0);
    return yneg ? -r : r;
  }
  if (yinf) return yneg ? -hpi : hpi;
  double ax = xneg ? -x : x, ay = yneg ? -y : y;
  const int de = (int)(uy & 0x7ff00000u) - (int)(ux & 0x7ff00000u);
  if (de >= 0x3900000) return yneg ? -hpi : hpi;                                  // |y / x| beyond 2^57
  if (de <= -0x3900000) {                                                         // ... below 2^-57
    const double r = xneg ? opi : ay / ax;
    return yneg ? -r : r;
  }
  if (ax < 0x1p-500 || ay < 0x1p-500) { ax *= 0x1p500; ay *= 0x1p500; }
  if (ax > 0x1p500 || ay > 0x1p500) { ax *= 0x1p-500; ay *= 0x1p-500; }
  const bool flat = ay < ax;                                                      // atan of ay / ax, else of ax / ay
  const double num = flat ? ay : ax, den = flat ? ax : ay;
  const double u = num / den;
  const double v0 = den * u;
  const double vv = __builtin_fma(den, u, -v0);                                   // EMULV: vfmsub132sd
  const double du = ((num - v0) - vv) / den;
  const bool c1_ = !xneg && flat, c3_ = xneg && ax < ay, c4_ = xneg && !c3_;      // (i) x>0, ay<ax  (ii) x>0, ax<=ay  (iii) x<0, ax<ay  (iv) x<0, ay<=ax
  double z;
  if (u < 0x1p-4) {
    const double v = u * u;
    double p = __builtin_fma(v, d13, d11);                                        // five vfmadd213sd
    p = __builtin_fma(v, p, d9); p = __builtin_fma(v, p, d7); p = __builtin_fma(v, p, d5); p = __builtin_fma(v, p, d3);
    const double uv = u * v;
    if (c1_) {                                                                    // (i)  atan(ay / ax)
      z = u + __builtin_fma(uv, p, du);                                           // zz = du + u v p: vfmadd132sd
    } else {
      const double zz = uv * p;
      const double big = c4_ ? opi : hpi, small = c4_ ? opi1 : hpi1;              // (ii), (iii): pi/2 -+ ...;  (iv): pi - ...
      if (c3_) {                                                                  // (iii) EADD(hpi, u, t2, cor)
        const double t2 = big + u;
        const double cor = big > u ? (big - t2) + u : (u - t2) + big;
        z = t2 + (((small + cor) + du) + zz);
      } else {                                                                    // (ii), (iv) ESUB(big, u, t2, cor)
        const double t2 = big - u;
        const double cor = big > u ? (big - t2) - u : big - (u + t2);
        z = t2 + (((small + cor) - du) - zz);
      }
    }
  } else {
    const int i = (int)(__builtin_fma(u, 0x1p8, 0x1p52) - 0x1p52) - 16;           // (TWO52 + TWO8 u) - TWO52: vfmadd132sd (exact either way)
    const double c0 = LIBM_ATAN2_TAB[i][0], c1 = LIBM_ATAN2_TAB[i][1], c2 = LIBM_ATAN2_TAB[i][2], c3 = LIBM_ATAN2_TAB[i][3],
                 c4 = LIBM_ATAN2_TAB[i][4], c5 = LIBM_ATAN2_TAB[i][5], c6 = LIBM_ATAN2_TAB[i][6];
    if (c1_) {                                                                    // (i)
      const double t3 = u - c0;
      const double v = t3 + du;                                                   // EADD(t3, du, v, dv)
      const double t3a = t3 < 0 ? -t3 : t3, dua = du < 0 ? -du : du;
      const double dv = t3a > dua ? (t3 - v) + du : (du - v) + t3;
      double q = __builtin_fma(v, c6, c5);                                        // three vfmadd213sd
      q = __builtin_fma(v, q, c4); q = __builtin_fma(v, q, c3);
      double zz = (v * v) * q;
      zz = __builtin_fma(dv, c2, zz);                                             // vfmadd231sd
      zz = __builtin_fma(v, c2, zz);                                              // vfmadd132sd
      z = c1 + zz;
    } else {
      const double v = (u - c0) + du;
      double r = __builtin_fma(v, c6, c5);                                        // four vfmadd213sd
      r = __builtin_fma(v, r, c4); r = __builtin_fma(v, r, c3); r = __builtin_fma(v, r, c2);
      if (c3_) z = (hpi + c1) + __builtin_fma(v, r, hpi1);              // (iii) vfmadd213sd
      else if (c4_) z = (opi - c1) + __builtin_fma(-v, r, opi1);                 // (iv)  vfnmadd213sd
      else z = (hpi - c1) + __builtin_fma(-v, r, hpi1);                           // (ii)  vfnmadd213sd
    }
  }
  z = z < 0 ? -z : z;                                                             // signArctan2(y, z)
  return yneg ? -z : z;
}
#undef ATAN2_FN
#undef ATAN2_TAB
#undef ATAN2_BITS
