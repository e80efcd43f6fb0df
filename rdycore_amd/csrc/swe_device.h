// Device-side arithmetic of the SWE right-hand side for gfx950.
//
// Each function keeps the operand order of the reference expression it stands
// for, so that the only differences from the CPU path are (a) FMA contraction,
// (b) sqrt()/cbrt() in place of pow(x,0.5)/pow(x,+-k/3) and (c) the order in
// which a cell's edge contributions are summed -- which is made the same as the
// reference's edge-loop order by the slot ordering built in rdyhip_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rdyhip {

// src/swe/swe_types_petsc.h:7
constexpr double GRAVITY = 9.806;

constexpr int32_t NBR_EMPTY = INT32_MIN;    // unused slot (triangle in a 4-slot layout)
constexpr int32_t NBR_GHOST = 1 << 30;      // neighbour is a ghost (non-owned) cell
constexpr int32_t NBR_MASK  = (1 << 30) - 1;

struct RoeFlux {
  double f0, f1, f2, amax;
};

// ComputeRiemannVelocities, src/swe/swe_petsc.c:57-73 (one state)
__device__ __forceinline__ void riemann_velocity(double h, double hu, double hv, double tiny_h, double h_anuga_sq, double &u, double &v) {
  if (h < tiny_h) {
    u = 0.0;
    v = 0.0;
  } else {
    const double denom = h * h + h_anuga_sq;
    u                  = hu * h / denom;
    v                  = hv * h / denom;
  }
}

// ComputeSWERoeEigenspectrum + ComputeSWERoeFlux for one edge,
// src/swe/swe_roe_flux_petsc.h:15-81, 103-128
__device__ __forceinline__ RoeFlux roe_flux(double hl, double ul, double vl, double hr, double ur, double vr, double sn, double cn) {
  const double duml  = sqrt(hl);
  const double dumr  = sqrt(hr);
  const double cl    = sqrt(GRAVITY * hl);
  const double cr    = sqrt(GRAVITY * hr);
  const double hhat  = duml * dumr;
  const double uhat  = (duml * ul + dumr * ur) / (duml + dumr);
  const double vhat  = (duml * vl + dumr * vr) / (duml + dumr);
  const double chat  = sqrt(0.5 * GRAVITY * (hl + hr));
  const double uperp = uhat * cn + vhat * sn;

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

  // critical flow fix
  const double al1 = uperpl - cl;
  const double ar1 = uperpr - cr;
  const double da1 = fmax(0.0, 2.0 * (ar1 - al1));
  if (a1 < da1) a1 = 0.5 * (a1 * a1 / da1 + da1);
  const double al3 = uperpl + cl;
  const double ar3 = uperpr + cr;
  const double da3 = fmax(0.0, 2.0 * (ar3 - al3));
  if (a3 < da3) a3 = 0.5 * (a3 * a3 / da3 + da3);

  const double dw0 = 0.5 * (dh - hhat * duperp / chat);
  const double dw1 = hhat * dupar;
  const double dw2 = 0.5 * (dh + hhat * duperp / chat);

  RoeFlux out;
  out.amax = chat + fabs(uperp);

  const double fl0 = uperpl * hl;
  const double fl1 = ul * uperpl * hl + 0.5 * GRAVITY * hl * hl * cn;
  const double fl2 = vl * uperpl * hl + 0.5 * GRAVITY * hl * hl * sn;
  const double fr0 = uperpr * hr;
  const double fr1 = ur * uperpr * hr + 0.5 * GRAVITY * hr * hr * cn;
  const double fr2 = vr * uperpr * hr + 0.5 * GRAVITY * hr * hr * sn;

  // R[0][] = {1, 0, 1}: the reference multiplies by these constants; 0*x is
  // kept so a non-finite a2*dw1 poisons the result exactly as it does there.
  out.f0 = 0.5 * (fl0 + fr0 - a1 * dw0 - 0.0 * a2 * dw1 - a3 * dw2);
  out.f1 = 0.5 * (fl1 + fr1 - r10 * a1 * dw0 - r11 * a2 * dw1 - r12 * a3 * dw2);
  out.f2 = 0.5 * (fl2 + fr2 - r20 * a1 * dw0 - r21 * a2 * dw1 - r22 * a3 * dw2);
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
__device__ __forceinline__ BoundaryFlux boundary_flux(int type, bool left_owned, double hl, double ul, double vl, const double *__restrict__ bval,
                                                      double sn, double cn, double tiny_h, double h_anuga_sq) {
  double hr = 0.0, ur = 0.0, vr = 0.0;
  if (type == 0 /* CONDITION_DIRICHLET */) {
    hr = bval[0];
    riemann_velocity(hr, bval[1], bval[2], tiny_h, h_anuga_sq, ur, vr);
  } else if (type == 2 /* CONDITION_REFLECTING */) {
    if (left_owned) {
      hr                = hl;
      const double dum1 = sn * sn - cn * cn;
      const double dum2 = 2.0 * sn * cn;
      ur                = ul * dum1 - vl * dum2;
      vr                = -ul * dum2 - vl * dum1;
    }
  } else /* CONDITION_CRITICAL_OUTFLOW */ {
    if (left_owned) {
      const double uperp = ul * cn + vl * sn;
      if (uperp < 0.0) {
        hl = ul = vl = 0.0;
      } else {
        const double q   = hl * fabs(uperp);
        hr               = cbrt(q * q / GRAVITY);
        const double vel = sqrt(GRAVITY * hr);
        ur               = vel * cn;
        vr               = vel * sn;
      }
    }
  }
  BoundaryFlux out;
  out.flux = roe_flux(hl, ul, vl, hr, ur, vr, sn, cn);
  out.wet  = !(hl < tiny_h && hr < tiny_h);
  return out;
}

// Friction term of ApplySourceSemiImplicit, src/swe/swe_petsc.c:764-780
__device__ __forceinline__ void friction_semi_implicit(double h, double hu, double hv, double n, double dt, double fsum_x, double fsum_y, double bedx,
                                                       double bedy, double &tbx, double &tby) {
  const double u      = hu / h;
  const double v      = hv / h;
  const double Cd     = GRAVITY * (n * n) * (1.0 / cbrt(h));  // g n^2 h^(-1/3)
  const double vel    = sqrt(u * u + v * v);
  const double tb     = Cd * vel / h;
  const double factor = tb / (1.0 + dt * tb);
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
  const double cb     = cbrt(h);
  const double gn2    = GRAVITY * (n * n);
  const double mxh    = mx / h;
  const double myh    = my / h;
  const double lambda = gn2 * (1.0 / (h * cb)) * sqrt(mxh * mxh + myh * myh);  // h^(-4/3)
  double       qx, qy;
  if (dt * lambda < thresh) {
    qx = mx;
    qy = my;
  } else {
    const double root = sqrt(1.0 + 4.0 * dt * lambda);
    qx                = (mx - mx * root) / (-2.0 * dt * lambda);
    qy                = (my - my * root) / (-2.0 * dt * lambda);
  }
  const double qmag = sqrt(qx * qx + qy * qy);
  const double hm73 = 1.0 / (h * h * cb);  // h^(-7/3)
  tbx               = gn2 * hm73 * qx * qmag;
  tby               = gn2 * hm73 * qy * qmag;
}

}  // namespace rdyhip
