// Device-side arithmetic of the SWE right-hand side for gfx950.
//
// Each function keeps the formulas and operand order of the reference
// expression it stands for; the differences from the CPU path are (a) FMA
// contraction, (b) square / cube roots and reciprocals computed from
// v_rsq_f64 / v_rcp_f64 with Newton refinement (<= 1 ulp) instead of
// pow(x,0.5), pow(x,+-k/3) and IEEE division, with divisions that share a
// denominator sharing one reciprocal.  All of it stays ~1e-15 relative; the
// parity bar is 1e-10 (tests/test_gpu_parity.py).  A cell's edge contributions
// are summed in the reference's edge-loop order (slot ordering, rdyhip_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rdyhip {

// src/swe/swe_types_petsc.h:7
constexpr double GRAVITY      = 9.806;
constexpr double SQRT_GRAVITY = 3.1314533367112465;  // sqrt(9.806) correctly rounded

constexpr int32_t NBR_EMPTY = INT32_MIN;    // unused slot (triangle in a 4-slot layout)
constexpr int32_t NBR_GHOST = 1 << 30;      // neighbour is a ghost (non-owned) cell
constexpr int32_t NBR_MASK  = (1 << 30) - 1;

// Reciprocal by v_rcp_f64 + two Newton steps (the same refinement the IEEE
// division expansion performs, without its scaling / fix-up wrapper): <= 1 ulp
// for normal operands; a zero or non-finite operand yields a non-finite
// result, as the division would (0 -> the IEEE 1/0 = inf becomes NaN, which
// only occurs on dry-dry edges whose flux is discarded or reported as NaN by
// the reference too).
__device__ __forceinline__ double rdy_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r        = fma(r, e, r);
  e        = fma(-x, r, 1.0);
  r        = fma(r, e, r);
  return r;
}

// Square root by v_rsq_f64 + one Goldschmidt step + one residual correction
// (the IEEE expansion minus its denormal-range scaling and its second
// correction): <= 1 ulp for normal operands, exact for 0 and +inf, NaN for
// negative operands.
__device__ __forceinline__ double rdy_sqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double       g = x * y;
  double       h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g              = fma(g, r, g);
  h              = fma(h, r, h);
  const double d = fma(-g, g, x);
  g              = fma(d, h, g);
  return (x == 0.0 || x == __builtin_inf()) ? x : g;
}

// sqrt(x) and 1/sqrt(x) from ONE v_rsq_f64: the Goldschmidt pair (g -> sqrt x, h -> 1/(2 sqrt x)) refined together;
// the square root gets the residual correction of rdy_sqrt (<= 1 ulp), the reciprocal root is 2 h (a few ulp).
// Saves the separate reciprocal where both are wanted: 1/h = (1/sqrt h)^2 next to sqrt(h), 1/chat next to chat.
struct SqrtPair {
  double s, r;
};
__device__ __forceinline__ SqrtPair rdy_sqrt_rsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double       g = x * y;
  double       h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g              = fma(g, r, g);
  h              = fma(h, r, h);
  const double d = fma(-g, g, x);
  g              = fma(d, h, g);
  SqrtPair o;
  o.s = (x == 0.0 || x == __builtin_inf()) ? x : g;
  o.r = h + h;
  return o;
}

struct RoeFlux {
  double f0, f1, f2, amax;
};

// a b + c d with each product rounded on its own: the expression is then symmetric under (a, b) <-> (c, d), as it is
// in the reference's arithmetic (no contraction there) -- fma(a, b, c d) is not.  The largest wave speed of an edge
// must not depend on which of its cells is called left, or on a mirror image of the state: edges that tie in the
// reference (uniform or symmetric states) have to tie here as well, or the Courant diagnostic names another edge.
__device__ __forceinline__ double sum_of_products(double a, double b, double c, double d) {
#pragma clang fp contract(off)
  const double p = a * b, q = c * d;
  return p + q;
}

// One side of a Riemann problem with everything that depends on that side
// alone: velocities (ComputeRiemannVelocities, src/swe/swe_petsc.c:57-73) and
// the two square roots of swe_roe_flux_petsc.h:21-24.  A cell's own state is
// prepared once and reused by all of its edges.
struct RiemannSide {
  double h, u, v;
  double sqh;  // sqrt(h)      "duml/dumr"
  double c;    // sqrt(g h)    "cl/cr"
};

__device__ __forceinline__ RiemannSide riemann_side(double h, double hu, double hv, double tiny_h, double h_anuga_sq) {
  RiemannSide s;
  s.h = h;
  if (h_anuga_sq == 0.0) {
    // the default (h_anuga_regular = 0, src/yaml_input.c:855): hu h / (h^2 + 0) = hu / h, and 1/h = (1/sqrt h)^2 comes with
    // the square root the Roe solver needs anyway (wave-uniform branch)
    const SqrtPair p   = rdy_sqrt_rsqrt(h);
    const double   inv = p.r * p.r;
    s.u                = (h < tiny_h) ? 0.0 : hu * inv;
    s.v                = (h < tiny_h) ? 0.0 : hv * inv;
    s.sqh              = p.s;
    s.c                = SQRT_GRAVITY * s.sqh;
    return s;
  }
  if (h < tiny_h) {
    s.u = 0.0;
    s.v = 0.0;
  } else {
    // hu*h/denom and hv*h/denom share one division: r = h/denom
    const double r = h * rdy_rcp(h * h + h_anuga_sq);
    s.u            = hu * r;
    s.v            = hv * r;
  }
  s.sqh = rdy_sqrt(h);
  s.c   = SQRT_GRAVITY * s.sqh;  // sqrt(g h) = sqrt(g) sqrt(h): one multiply instead of a second square root
  return s;
}

// ComputeSWERoeEigenspectrum + ComputeSWERoeFlux for one edge,
// src/swe/swe_roe_flux_petsc.h:15-81, 103-128.  Same formulas; the divisions
// by (duml+dumr) and by chat are each done once as a reciprocal.
__device__ __forceinline__ RoeFlux roe_flux(const RiemannSide &L, const RiemannSide &R, double sn, double cn) {
  const double hl = L.h, ul = L.u, vl = L.v, hr = R.h, ur = R.u, vr = R.v;
  const double duml = L.sqh, dumr = R.sqh, cl = L.c, cr = R.c;
  const double hhat    = duml * dumr;
  const double inv_sum = rdy_rcp(duml + dumr);
  const double uhat    = sum_of_products(duml, ul, dumr, ur) * inv_sum;  // (these three feed amax: see sum_of_products)
  const double vhat    = sum_of_products(duml, vl, dumr, vr) * inv_sum;
  const SqrtPair cp       = rdy_sqrt_rsqrt(0.5 * GRAVITY * (hl + hr));
  const double   chat     = cp.s;
  const double   inv_chat = cp.r;  // 1/chat with the square root, instead of a separate reciprocal
  const double uperp   = sum_of_products(uhat, cn, vhat, sn);

  const double dh     = hr - hl;
  const double du     = ur - ul;
  const double dv     = vr - vl;
  const double dupar  = -du * sn + dv * cn;
  const double duperp = du * cn + dv * sn;

  const double r10 = uhat - chat * cn, r11 = -sn, r12 = uhat + chat * cn;
  const double r20 = vhat - chat * sn, r21 = cn, r22 = vhat + chat * sn;

  const double uperpl = ul * cn + vl * sn;
  const double uperpr = ur * cn + vr * sn;
  double       a1     = fabs(uperp - chat);
  const double a2     = fabs(uperp);
  double       a3     = fabs(uperp + chat);

  // critical flow fix (rarely taken: keep the divisions behind real branches)
  const double da1 = fmax(0.0, 2.0 * ((uperpr - cr) - (uperpl - cl)));
  if (a1 < da1) a1 = 0.5 * (a1 * a1 / da1 + da1);
  const double da3 = fmax(0.0, 2.0 * ((uperpr + cr) - (uperpl + cl)));
  if (a3 < da3) a3 = 0.5 * (a3 * a3 / da3 + da3);

  const double t   = hhat * duperp * inv_chat;
  const double dw0 = 0.5 * (dh - t);
  const double dw1 = hhat * dupar;
  const double dw2 = 0.5 * (dh + t);

  RoeFlux out;
  out.amax = chat + fabs(uperp);

  const double gh2l = 0.5 * GRAVITY * hl * hl, gh2r = 0.5 * GRAVITY * hr * hr;
  const double ql = uperpl * hl, qr = uperpr * hr;
  const double fl0 = ql;
  const double fl1 = ul * ql + gh2l * cn;
  const double fl2 = vl * ql + gh2l * sn;
  const double fr0 = qr;
  const double fr1 = ur * qr + gh2r * cn;
  const double fr2 = vr * qr + gh2r * sn;

  const double w0 = a1 * dw0, w1 = a2 * dw1, w2 = a3 * dw2;
  // R[0][] = {1, 0, 1}; 0*w1 is kept so a non-finite w1 poisons f0 as in the reference
  out.f0 = 0.5 * (fl0 + fr0 - w0 - 0.0 * w1 - w2);
  out.f1 = 0.5 * (fl1 + fr1 - r10 * w0 - r11 * w1 - r12 * w2);
  out.f2 = 0.5 * (fl2 + fr2 - r20 * w0 - r21 * w1 - r22 * w2);
  return out;
}

struct BoundaryFlux {
  RoeFlux flux;
  bool    wet;  // !(hl < tiny_h && hr < tiny_h), src/swe/swe_petsc.c:593
};

// Right state of a boundary edge + its Roe flux: ApplyBoundaryFlux's
// condition switch (src/swe/swe_petsc.c:549-576) with ApplyReflectingBC
// (434-461) and ApplyCriticalOutflowBC (465-503).  (hl,ul,vl) is the left
// cell's state with Riemann velocities.  For reflecting / outflow edges whose
// left cell is not owned the reference leaves its zero-initialised scratch
// untouched (449, 480), i.e. the right state is (0,0,0).
__device__ __forceinline__ BoundaryFlux boundary_flux(int type, bool left_owned, RiemannSide L, const double *__restrict__ bval, double sn, double cn,
                                                      double tiny_h, double h_anuga_sq) {
  RiemannSide R;
  R.h = R.u = R.v = R.sqh = R.c = 0.0;
  if (type == 0 /* CONDITION_DIRICHLET */) {
    R = riemann_side(bval[0], bval[1], bval[2], tiny_h, h_anuga_sq);
  } else if (type == 2 /* CONDITION_REFLECTING */) {
    if (left_owned) {
      const double dum1 = sn * sn - cn * cn;
      const double dum2 = 2.0 * sn * cn;
      R.h               = L.h;
      R.u               = L.u * dum1 - L.v * dum2;
      R.v               = -L.u * dum2 - L.v * dum1;
      R.sqh             = L.sqh;
      R.c               = L.c;
    }
  } else /* CONDITION_CRITICAL_OUTFLOW */ {
    if (left_owned) {
      const double uperp = L.u * cn + L.v * sn;
      if (uperp < 0.0) {
        L.h = L.u = L.v = L.sqh = L.c = 0.0;
      } else {
        const double q = L.h * fabs(uperp);
        R.h            = cbrt(q * q / GRAVITY);
        R.sqh          = rdy_sqrt(R.h);
        R.c            = SQRT_GRAVITY * R.sqh;
        R.u            = R.c * cn;
        R.v            = R.c * sn;
      }
    }
  }
  BoundaryFlux out;
  out.flux = roe_flux(L, R, sn, cn);
  out.wet  = !(L.h < tiny_h && R.h < tiny_h);
  return out;
}

// Friction term of ApplySourceSemiImplicit, src/swe/swe_petsc.c:764-780
__device__ __forceinline__ void friction_semi_implicit(double h, double hu, double hv, double n, double dt, double fsum_x, double fsum_y, double bedx,
                                                       double bedy, double &tbx, double &tby) {
  const double inv_h  = rdy_rcp(h);
  const double u      = hu * inv_h;
  const double v      = hv * inv_h;
  const double Cd     = GRAVITY * (n * n) * rcbrt(h);  // g n^2 h^(-1/3)
  const double vel    = rdy_sqrt(u * u + v * v);
  const double tb     = Cd * vel * inv_h;
  const double factor = tb * rdy_rcp(1.0 + dt * tb);
  tbx                 = (hu + dt * fsum_x - dt * bedx) * factor;
  tby                 = (hv + dt * fsum_y - dt * bedy) * factor;
}

// Friction term of ApplySourceImplicitXQ2018, src/swe/swe_petsc.c:876-907
__device__ __forceinline__ void friction_xq2018(double h, double hu, double hv, double n, double dt, double thresh, double fsum_x, double fsum_y,
                                                double bedx, double bedy, double &tbx, double &tby) {
  const double Ax     = fsum_x - bedx;
  const double Ay     = fsum_y - bedy;
  const double mx     = hu + Ax * dt;
  const double my     = hv + Ay * dt;
  const double rcb    = rcbrt(h);                // h^(-1/3)
  const double inv_h  = rdy_rcp(h);
  const double gn2    = GRAVITY * (n * n);
  const double mxh    = mx * inv_h;
  const double myh    = my * inv_h;
  const double lambda = gn2 * (inv_h * rcb) * rdy_sqrt(mxh * mxh + myh * myh);  // h^(-4/3)
  double       qx, qy;
  if (dt * lambda < thresh) {
    qx = mx;
    qy = my;
  } else {
    const double root = rdy_sqrt(1.0 + 4.0 * dt * lambda);
    const double inv  = rdy_rcp(-2.0 * dt * lambda);
    qx                = (mx - mx * root) * inv;
    qy                = (my - my * root) * inv;
  }
  const double qmag = rdy_sqrt(qx * qx + qy * qy);
  const double hm73 = inv_h * inv_h * rcb;  // h^(-7/3)
  tbx               = gn2 * hm73 * qx * qmag;
  tby               = gn2 * hm73 * qy * qmag;
}

}  // namespace rdyhip
