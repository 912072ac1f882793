// oracle/rigid_body.h — TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Inverse dynamics with compliant contact for a tree of rigid bodies: the
// physics the reference delegates to Drake v1.30.0 (not present in
// /root/reference), restated in plain C++.  Call sites being restated
// (reference optimizer/trajectory_optimizer.cc):
//   :228-245  CalcInverseDynamicsSingleTimeStep  -> InverseDynamics()
//   :232      plant.CalcForceElementsContribution (gravity + joint damping)
//   :244      plant.CalcInverseDynamics  (tau = M a + C v - applied forces)
//   :247-386  CalcContactForceContribution        -> ContactForces()
//   :279      ComputeSignedDistancePairwiseClosestPoints -> SignedDistance()
//   :559      plant.CalcMassMatrix                 -> MassMatrix()
//   :1645     plant.MakeQDotToVelocityMap          -> Nplus()
// Conventions confirmed by the reference's closed-form tests
// (optimizer/test/trajectory_optimizer_test.cc:1351-1363, 939-973, 1104-1136):
//   tau = m l^2 a + m g l sin(q) + b v for the damped pendulum.
//
// PARITY STATUS: the formulation (world-frame recursive Newton-Euler about the
// body origins) is algebraically what Drake computes, but Drake's exact
// operation order is not reproducible without its source => results agree with
// the reference to round-off only, and for the multi-DoF contact models
// (hopper, mini_cheetah, allegro) the conventions of SURVEY.md Appendix D
// marked "unverified" are *defined* here ("parity unpinned" for those).
//
// The floating-point association order of every sum is part of the spec
//  so that the lane-parallel HIP kernel can reproduce the
// same bits.
#pragma once

#include <cmath>
#include <cstring>
#include <vector>

#include "idto/detmath.h"
#include "idto_model.h"

namespace oracle {

#ifdef IDTO_ORACLE_LIBM
inline void m_sincos(double x, double* s, double* c) { *s = std::sin(x); *c = std::cos(x); }
inline double m_exp(double x) { return std::exp(x); }
inline double m_log(double x) { return std::log(x); }
#else
inline void m_sincos(double x, double* s, double* c) { idto::detmath::sincos(x, s, c); }
inline double m_exp(double x) { return idto::detmath::exp(x); }
inline double m_log(double x) { return idto::detmath::log(x); }
#endif

struct V3 {
  double x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
// (explicit fused multiply-adds in a fixed form, identically in idto_amd/csrc/dev_math.h)
inline double fma3(double a0, double b0, double a1, double b1, double a2, double b2) {
  return std::fma(a2, b2, std::fma(a1, b1, a0 * b0));
}
inline double fms(double a, double b, double c, double d) { return std::fma(a, b, -(c * d)); }  // a b - c d
inline double dot(V3 a, V3 b) { return fma3(a.x, b.x, a.y, b.y, a.z, b.z); }
inline V3 cross(V3 a, V3 b) {
  return {fms(a.y, b.z, a.z, b.y), fms(a.z, b.x, a.x, b.z), fms(a.x, b.y, a.y, b.x)};
}
struct M3 {
  double m[9];  // row-major
};
inline M3 Identity3() { return {{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
inline V3 operator*(const M3& R, V3 v) {
  return {fma3(R.m[0], v.x, R.m[1], v.y, R.m[2], v.z), fma3(R.m[3], v.x, R.m[4], v.y, R.m[5], v.z),
          fma3(R.m[6], v.x, R.m[7], v.y, R.m[8], v.z)};
}
inline V3 TMul(const M3& R, V3 v) {  // R^T v
  return {fma3(R.m[0], v.x, R.m[3], v.y, R.m[6], v.z), fma3(R.m[1], v.x, R.m[4], v.y, R.m[7], v.z),
          fma3(R.m[2], v.x, R.m[5], v.y, R.m[8], v.z)};
}
inline M3 operator*(const M3& A, const M3& B) {
  M3 C;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      C.m[3 * r + c] = fma3(A.m[3 * r], B.m[c], A.m[3 * r + 1], B.m[3 + c], A.m[3 * r + 2], B.m[6 + c]);
  return C;
}
inline V3 Col(const M3& R, int c) { return {R.m[c], R.m[3 + c], R.m[6 + c]}; }

// Rotation by angle (given sin, cos) about the unit axis a (Rodrigues).
inline M3 AxisAngle(V3 a, double s, double c) {
  const V3 sa = a * s;
  const V3 ca = a * (1.0 - c);
  M3 R;
  double t;
  t = ca.x * a.y; R.m[1] = t - sa.z; R.m[3] = t + sa.z;
  t = ca.x * a.z; R.m[2] = t + sa.y; R.m[6] = t - sa.y;
  t = ca.y * a.z; R.m[5] = t - sa.x; R.m[7] = t + sa.x;
  R.m[0] = ca.x * a.x + c;
  R.m[4] = ca.y * a.y + c;
  R.m[8] = ca.z * a.z + c;
  return R;
}

// Rotation matrix of a (possibly un-normalised) quaternion [w x y z]; the
// 2/|q|^2 form normalises implicitly ("N+ for a quaternion").
inline M3 QuatToRot(const double* q) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double n2 = ((w * w + x * x) + y * y) + z * z;
  const double s2 = 2.0 / n2;
  const double sx = s2 * x, sy = s2 * y, sz = s2 * z;
  const double swx = sx * w, swy = sy * w, swz = sz * w;
  const double sxx = sx * x, sxy = sy * x, sxz = sz * x;
  const double syy = sy * y, syz = sz * y, szz = sz * z;
  M3 R;
  R.m[0] = (1.0 - syy) - szz; R.m[1] = sxy - swz;         R.m[2] = sxz + swy;
  R.m[3] = sxy + swz;         R.m[4] = (1.0 - sxx) - szz; R.m[5] = syz - swx;
  R.m[6] = sxz - swy;         R.m[7] = syz + swx;         R.m[8] = (1.0 - sxx) - syy;
  return R;
}

struct Model {
  int nb = 0, nq = 0, nv = 0;
  std::vector<int> parent, jtype, qstart, vstart;
  std::vector<M3> R_PF;
  std::vector<V3> p_PF, axis, com;
  std::vector<double> mass;
  std::vector<double> inertia;  // 6 per body
  std::vector<double> damping;
  std::vector<int> actuated;
  V3 gravity{0, 0, -9.81};
  int ngeoms = 0;
  std::vector<int> geom_body, geom_type;
  std::vector<M3> geom_R;
  std::vector<V3> geom_p, geom_size;
  int npairs = 0;
  std::vector<int> pair_a, pair_b;
  int npaths = 1, common_body = -1;
  std::vector<int> body_path, pair_path;

  void FromC(const idto_model_t& m) {
    nb = m.nbodies; nq = m.nq; nv = m.nv;
    parent.assign(m.parent, m.parent + nb);
    jtype.assign(m.jtype, m.jtype + nb);
    qstart.assign(m.qstart, m.qstart + nb);
    vstart.assign(m.vstart, m.vstart + nb);
    R_PF.resize(nb); p_PF.resize(nb); axis.resize(nb); com.resize(nb);
    mass.assign(m.mass, m.mass + nb);
    inertia.assign(m.inertia, m.inertia + 6 * nb);
    for (int i = 0; i < nb; ++i) {
      std::memcpy(R_PF[i].m, m.X_PF + 12 * i, 9 * sizeof(double));
      p_PF[i] = {m.X_PF[12 * i + 9], m.X_PF[12 * i + 10], m.X_PF[12 * i + 11]};
      axis[i] = {m.axis[3 * i], m.axis[3 * i + 1], m.axis[3 * i + 2]};
      com[i] = {m.com[3 * i], m.com[3 * i + 1], m.com[3 * i + 2]};
    }
    damping.assign(m.damping, m.damping + nv);
    actuated.assign(m.actuated, m.actuated + nv);
    gravity = {m.gravity[0], m.gravity[1], m.gravity[2]};
    ngeoms = m.ngeoms;
    geom_body.assign(m.geom_body, m.geom_body + ngeoms);
    geom_type.assign(m.geom_type, m.geom_type + ngeoms);
    geom_R.resize(ngeoms); geom_p.resize(ngeoms); geom_size.resize(ngeoms);
    for (int g = 0; g < ngeoms; ++g) {
      std::memcpy(geom_R[g].m, m.geom_X + 12 * g, 9 * sizeof(double));
      geom_p[g] = {m.geom_X[12 * g + 9], m.geom_X[12 * g + 10], m.geom_X[12 * g + 11]};
      geom_size[g] = {m.geom_size[3 * g], m.geom_size[3 * g + 1], m.geom_size[3 * g + 2]};
    }
    npairs = m.npairs;
    pair_a.assign(m.pair_a, m.pair_a + npairs);
    pair_b.assign(m.pair_b, m.pair_b + npairs);
    npaths = m.npaths; common_body = m.common_body;
    body_path.assign(m.body_path, m.body_path + nb);
    pair_path.assign(m.pair_path, m.pair_path + npairs);
  }
};

struct ContactParams {
  double k = 100, vd = 0.1, vs = 0.05, mu = 0.5, sigma = 0.1;
  double threshold = 0;  // distance beyond which a pair exerts no force
  // reference TO.cc:266-269
  void Finalize() {
    const double eps = std::sqrt(2.220446049250313e-16);
    threshold = -sigma * m_log(m_exp(eps / (sigma * k)) - 1.0);
  }
};

struct BodyKin {
  M3 R;               // R_WB
  V3 p;               // origin in world
  V3 w, v;            // angular velocity, origin velocity (world)
  V3 al, a;           // angular acceleration, origin (classical) acceleration
  V3 r;               // p - p_parent
  M3 R_WF;            // joint frame F in world
  V3 hW;              // revolute/prismatic axis in world
};

struct Wrench {
  V3 f{0, 0, 0}, n{0, 0, 0};  // force, torque about the body origin (world frame)
};

struct SignedDistanceResult {
  bool valid = false;
  double phi = 0;
  V3 n{0, 0, 1};  // unit, from A towards B ("outwards from A", TO.cc:282-283)
  V3 Ca{0, 0, 0}, Cb{0, 0, 0};  // witness points in world
};

// Signed distance between two primitives given their world poses.  Restates what
// the reference obtains from Drake's SceneGraph (TO.cc:279,300-312): witness
// points, distance < 0 in penetration, normal from B into A negated.
// Supported: sphere-sphere, sphere-box (either order), box-vs-world-fixed-box
// (lowest vertex of A against B's top face; see DESIGN.md "contact geometry").
inline SignedDistanceResult SignedDistance(int typeA, const M3& RA, V3 pA, V3 sA, int typeB, const M3& RB,
                                           V3 pB, V3 sB) {
  SignedDistanceResult out;
  if (typeA == IDTO_GEOM_SPHERE && typeB == IDTO_GEOM_SPHERE) {
    const V3 d = pB - pA;
    const double dist = std::sqrt(dot(d, d));
    out.phi = (dist - sA.x) - sB.x;
    out.n = d / dist;
    out.Ca = pA + out.n * sA.x;
    out.Cb = pB - out.n * sB.x;
    out.valid = true;
    return out;
  }
  if ((typeA == IDTO_GEOM_SPHERE && typeB == IDTO_GEOM_BOX) ||
      (typeA == IDTO_GEOM_BOX && typeB == IDTO_GEOM_SPHERE)) {
    const bool sphere_is_A = (typeA == IDTO_GEOM_SPHERE);
    const V3 pS = sphere_is_A ? pA : pB;
    const double rad = sphere_is_A ? sA.x : sB.x;
    const M3& RX = sphere_is_A ? RB : RA;
    const V3 pX = sphere_is_A ? pB : pA;
    const V3 h = sphere_is_A ? sB : sA;
    const V3 c = TMul(RX, pS - pX);  // sphere centre in the box frame
    V3 pc = c;
    bool outside = false;
    if (pc.x > h.x) { pc.x = h.x; outside = true; } else if (pc.x < -h.x) { pc.x = -h.x; outside = true; }
    if (pc.y > h.y) { pc.y = h.y; outside = true; } else if (pc.y < -h.y) { pc.y = -h.y; outside = true; }
    if (pc.z > h.z) { pc.z = h.z; outside = true; } else if (pc.z < -h.z) { pc.z = -h.z; outside = true; }
    V3 g;  // unit, from the box towards the sphere centre, box frame
    double phi;
    if (outside) {
      const V3 dv = c - pc;
      const double dist = std::sqrt(dot(dv, dv));
      g = dv / dist;
      phi = dist - rad;
    } else {
      const double dx = h.x - std::fabs(c.x), dy = h.y - std::fabs(c.y), dz = h.z - std::fabs(c.z);
      g = {0, 0, 0};
      double depth;
      if (dx <= dy && dx <= dz) { depth = dx; g.x = (c.x >= 0) ? 1.0 : -1.0; pc.x = g.x * h.x; }
      else if (dy <= dz) { depth = dy; g.y = (c.y >= 0) ? 1.0 : -1.0; pc.y = g.y * h.y; }
      else { depth = dz; g.z = (c.z >= 0) ? 1.0 : -1.0; pc.z = g.z * h.z; }
      phi = -depth - rad;
    }
    const V3 gW = RX * g;                  // box -> sphere, world
    const V3 boxW = pX + RX * pc;          // witness on the box
    const V3 sphW = pS - gW * rad;         // witness on the sphere
    out.phi = phi;
    if (sphere_is_A) { out.n = -gW; out.Ca = sphW; out.Cb = boxW; }
    else { out.n = gW; out.Ca = boxW; out.Cb = sphW; }
    out.valid = true;
    return out;
  }
  if (typeA == IDTO_GEOM_BOX && typeB == IDTO_GEOM_BOX) {
    // A: moving box; B: world-fixed, axis-aligned box used as ground: only its top face.
    const double ztop = pB.z + sB.z;
    double zmin = 0;
    V3 best{0, 0, 0};
    bool first = true;
    for (int ix = 0; ix < 2; ++ix)
      for (int iy = 0; iy < 2; ++iy)
        for (int iz = 0; iz < 2; ++iz) {
          const V3 cb = {ix ? sA.x : -sA.x, iy ? sA.y : -sA.y, iz ? sA.z : -sA.z};
          const V3 cw = pA + RA * cb;
          if (first || cw.z < zmin) { zmin = cw.z; best = cw; first = false; }
        }
    out.phi = zmin - ztop;
    out.n = {0, 0, -1};
    out.Ca = best;
    out.Cb = {best.x, best.y, ztop};
    out.valid = true;
    return out;
  }
  return out;
}

// Pairwise butterfly sum over npaths (power of two) values: the association
// order the HIP kernel's cross-lane xor-reduction produces.
inline V3 TreeSum(const V3* x, int n) {
  V3 t[IDTO_MAX_PATHS];
  for (int i = 0; i < n; ++i) t[i] = x[i];
  for (int stride = 1; stride < n; stride *= 2) {
    V3 u[IDTO_MAX_PATHS];
    for (int i = 0; i < n; ++i) u[i] = t[i] + t[i ^ stride];
    for (int i = 0; i < n; ++i) t[i] = u[i];
  }
  return t[0];
}

class Dynamics {
 public:
  Model model;
  ContactParams contact;

  // Forward kinematics + velocities + accelerations of every body.
  void Kinematics(const double* q, const double* v, const double* a, std::vector<BodyKin>* kin_out) const {
    const Model& m = model;
    std::vector<BodyKin>& kin = *kin_out;
    kin.resize(m.nb);
    const V3 zero{0, 0, 0};
    for (int i = 0; i < m.nb; ++i) {
      const int lam = m.parent[i];
      const M3 Rp = lam < 0 ? Identity3() : kin[lam].R;
      const V3 pp = lam < 0 ? zero : kin[lam].p;
      const V3 wp = lam < 0 ? zero : kin[lam].w;
      const V3 vp = lam < 0 ? zero : kin[lam].v;
      const V3 alp = lam < 0 ? zero : kin[lam].al;
      const V3 ap = lam < 0 ? zero : kin[lam].a;
      BodyKin& b = kin[i];
      b.R_WF = Rp * m.R_PF[i];
      const V3 d1 = Rp * m.p_PF[i];
      const double* qi = q + m.qstart[i];
      const double* vi = v + m.vstart[i];
      const double* ai = a + m.vstart[i];
      M3 R_FM = Identity3();
      V3 d2 = zero, w_rel = zero, v_rel = zero, al_rel = zero, a_rel = zero;
      b.hW = zero;
      switch (m.jtype[i]) {
        case IDTO_JOINT_REVOLUTE: {
          double s, c;
          m_sincos(qi[0], &s, &c);
          R_FM = AxisAngle(m.axis[i], s, c);
          b.hW = b.R_WF * m.axis[i];
          w_rel = b.hW * vi[0];
          al_rel = b.hW * ai[0];
        } break;
        case IDTO_JOINT_PRISMATIC: {
          b.hW = b.R_WF * m.axis[i];
          d2 = b.hW * qi[0];
          v_rel = b.hW * vi[0];
          a_rel = b.hW * ai[0];
        } break;
        case IDTO_JOINT_PLANAR: {
          double s, c;
          m_sincos(qi[2], &s, &c);
          R_FM = {{c, -s, 0, s, c, 0, 0, 0, 1}};
          const V3 ex = Col(b.R_WF, 0), ey = Col(b.R_WF, 1), ez = Col(b.R_WF, 2);
          d2 = ex * qi[0] + ey * qi[1];
          v_rel = ex * vi[0] + ey * vi[1];
          a_rel = ex * ai[0] + ey * ai[1];
          w_rel = ez * vi[2];
          al_rel = ez * ai[2];
        } break;
        case IDTO_JOINT_FLOATING: {
          R_FM = QuatToRot(qi);
          d2 = b.R_WF * V3{qi[4], qi[5], qi[6]};
          w_rel = b.R_WF * V3{vi[0], vi[1], vi[2]};
          v_rel = b.R_WF * V3{vi[3], vi[4], vi[5]};
          al_rel = b.R_WF * V3{ai[0], ai[1], ai[2]};
          a_rel = b.R_WF * V3{ai[3], ai[4], ai[5]};
        } break;
      }
      b.R = b.R_WF * R_FM;
      b.r = d1 + d2;
      b.p = pp + b.r;
      b.w = wp + w_rel;
      b.v = (vp + cross(wp, b.r)) + v_rel;
      b.al = (alp + al_rel) + cross(wp, w_rel);
      b.a = (((ap + cross(alp, b.r)) + cross(wp, cross(wp, b.r))) + cross(wp, v_rel) * 2.0) + a_rel;
    }
  }

  // Contact wrenches on every body (about the body origins, world frame).
  // Restates TO.cc:247-386 line by line; see SignedDistance() for :279.
  void ContactForces(const std::vector<BodyKin>& kin, std::vector<Wrench>* ext_out) const {
    const Model& m = model;
    const ContactParams& cp = contact;
    std::vector<Wrench>& ext = *ext_out;
    ext.assign(m.nb, Wrench());
    // per-path partial sums for the common body (summation-order spec)
    V3 cf[IDTO_MAX_PATHS], cn[IDTO_MAX_PATHS];
    for (int k = 0; k < IDTO_MAX_PATHS; ++k) cf[k] = cn[k] = {0, 0, 0};
    const V3 zero{0, 0, 0};
    for (int pi = 0; pi < m.npairs; ++pi) {
      const int ga = m.pair_a[pi], gb = m.pair_b[pi];
      const int ba = m.geom_body[ga], bb = m.geom_body[gb];
      const M3 RbA = ba < 0 ? Identity3() : kin[ba].R;
      const V3 pbA = ba < 0 ? zero : kin[ba].p;
      const M3 RbB = bb < 0 ? Identity3() : kin[bb].R;
      const V3 pbB = bb < 0 ? zero : kin[bb].p;
      // geometry poses in world: X_WG = X_WB * X_BG (TO.cc:308,311)
      const M3 RgA = RbA * m.geom_R[ga];
      const V3 pgA = pbA + RbA * m.geom_p[ga];
      const M3 RgB = RbB * m.geom_R[gb];
      const V3 pgB = pbB + RbB * m.geom_p[gb];
      const SignedDistanceResult sd =
          SignedDistance(m.geom_type[ga], RgA, pgA, m.geom_size[ga], m.geom_type[gb], RgB, pgB, m.geom_size[gb]);
      if (!sd.valid) continue;
      if (sd.phi > cp.threshold) continue;  // max_distance = threshold (:279)
      const V3 nhat = sd.n;                              // :283
      const V3 pC = (sd.Ca + sd.Cb) * 0.5;               // :316
      const V3 pAC = pC - pbA, pBC = pC - pbB;           // :319-320
      const V3 wA = ba < 0 ? zero : kin[ba].w, vA = ba < 0 ? zero : kin[ba].v;
      const V3 wB = bb < 0 ? zero : kin[bb].w, vB = bb < 0 ? zero : kin[bb].v;
      const V3 vAc = vA + cross(wA, pAC);                // :327
      const V3 vBc = vB + cross(wB, pBC);                // :328
      const V3 vrel = vBc - vAc;                         // :331-332
      const double vn = dot(nhat, vrel);                 // :335
      const V3 vt = vrel - nhat * vn;                    // :336
      double dissipation = 0.0;                          // :339-345
      const double s = vn / cp.vd;
      if (s < 0) dissipation = 1 - s;
      else if (s < 2) dissipation = (s - 2) * (s - 2) / 4;
      double compliant_fn;                               // :349-359
      const double exponent = -sd.phi / cp.sigma;
      if (exponent >= 37) compliant_fn = -cp.k * sd.phi;
      else compliant_fn = cp.sigma * cp.k * m_log(1 + m_exp(exponent));
      const double fn = compliant_fn * dissipation;      // :360
      const V3 that = (-vt) / std::sqrt(cp.vs * cp.vs + dot(vt, vt));  // :368-369
      const V3 ft = (that * cp.mu) * fn;                 // :370
      const V3 fB = nhat * fn + ft;                      // :373
      const V3 fA = -fB;                                 // :379
      const V3 nB = cross(pBC, fB);                      // Shift(-p_BC) :377
      const V3 nA = cross(pAC, fA);                      // :380
      const int path = m.pair_path[pi];
      if (ba >= 0) {
        if (ba == m.common_body) { cf[path] = cf[path] + fA; cn[path] = cn[path] + nA; }
        else { ext[ba].f = ext[ba].f + fA; ext[ba].n = ext[ba].n + nA; }
      }
      if (bb >= 0) {
        if (bb == m.common_body) { cf[path] = cf[path] + fB; cn[path] = cn[path] + nB; }
        else { ext[bb].f = ext[bb].f + fB; ext[bb].n = ext[bb].n + nB; }
      }
    }
    if (m.common_body >= 0) {
      ext[m.common_body].f = TreeSum(cf, m.npaths);
      ext[m.common_body].n = TreeSum(cn, m.npaths);
    }
  }

  // tau = ID(q, v, a).  `full` = gravity + joint damping + contact (the
  // reference's CalcInverseDynamicsSingleTimeStep, TO.cc:228-245); !full = the
  // mass-matrix column mode (no gravity, no damping, no contact; caller passes v = 0).
  void InverseDynamics(const double* q, const double* v, const double* a, bool full, double* tau) const {
    const Model& m = model;
    std::vector<BodyKin> kin;
    Kinematics(q, v, a, &kin);
    std::vector<Wrench> ext(m.nb);
    if (full && m.npairs > 0) ContactForces(kin, &ext);
    const V3 g = full ? m.gravity : V3{0, 0, 0};
    std::vector<Wrench> tot(m.nb);
    for (int i = 0; i < m.nb; ++i) {
      const BodyKin& b = kin[i];
      const V3 cW = b.R * m.com[i];
      const V3 t1 = cross(b.al, cW);
      const V3 t2 = cross(b.w, cross(b.w, cW));
      const V3 acom = (b.a + t1) + t2;
      const V3 f_in = (acom - g) * m.mass[i];
      const V3 wB = TMul(b.R, b.w), alB = TMul(b.R, b.al);
      const double* I = &m.inertia[6 * i];  // xx yy zz xy xz yz
      auto Imul = [&](V3 x) {
        return V3{fma3(I[0], x.x, I[3], x.y, I[4], x.z), fma3(I[3], x.x, I[1], x.y, I[5], x.z),
                  fma3(I[4], x.x, I[5], x.y, I[2], x.z)};
      };
      const V3 nB = Imul(alB) + cross(wB, Imul(wB));
      const V3 n_in = b.R * nB + cross(cW, f_in);
      tot[i].f = f_in - ext[i].f;
      tot[i].n = n_in - ext[i].n;
    }
    // Backward pass.  Chain bodies have at most one child inside their path
    // (the next body of the path); the common body sums its children with the
    // butterfly order.
    V3 cf[IDTO_MAX_PATHS], cn[IDTO_MAX_PATHS];
    for (int k = 0; k < IDTO_MAX_PATHS; ++k) cf[k] = cn[k] = {0, 0, 0};
    for (int i = m.nb - 1; i >= 0; --i) {
      if (i == m.common_body) {
        tot[i].f = tot[i].f + TreeSum(cf, m.npaths);
        tot[i].n = tot[i].n + TreeSum(cn, m.npaths);
      }
      const BodyKin& b = kin[i];
      const int vs = m.vstart[i];
      const V3 f = tot[i].f, n = tot[i].n;
      switch (m.jtype[i]) {
        case IDTO_JOINT_REVOLUTE: tau[vs] = dot(b.hW, n); break;
        case IDTO_JOINT_PRISMATIC: tau[vs] = dot(b.hW, f); break;
        case IDTO_JOINT_PLANAR:
          tau[vs] = dot(Col(b.R_WF, 0), f);
          tau[vs + 1] = dot(Col(b.R_WF, 1), f);
          tau[vs + 2] = dot(Col(b.R_WF, 2), n);
          break;
        case IDTO_JOINT_FLOATING: {
          const V3 nF = TMul(b.R_WF, n), fF = TMul(b.R_WF, f);
          tau[vs] = nF.x; tau[vs + 1] = nF.y; tau[vs + 2] = nF.z;
          tau[vs + 3] = fF.x; tau[vs + 4] = fF.y; tau[vs + 5] = fF.z;
        } break;
      }
      const int lam = m.parent[i];
      if (lam >= 0) {
        const V3 cfi = f, cni = n + cross(b.r, f);
        if (lam == m.common_body) {
          const int path = m.body_path[i];
          cf[path] = cfi;  // exactly one chain root per path hangs off the common body
          cn[path] = cni;
        } else {
          tot[lam].f = tot[lam].f + cfi;
          tot[lam].n = tot[lam].n + cni;
        }
      }
    }
    if (full)
      for (int j = 0; j < m.nv; ++j) tau[j] = tau[j] + m.damping[j] * v[j];
  }

  // Column-major nv x nv mass matrix, column j = ID(q, 0, e_j) without gravity.
  void MassMatrix(const double* q, double* M) const {
    const int nv = model.nv;
    std::vector<double> zero(nv, 0.0), e(nv, 0.0);
    for (int j = 0; j < nv; ++j) {
      e[j] = 1.0;
      InverseDynamics(q, zero.data(), e.data(), false, M + (size_t)j * nv);
      e[j] = 0.0;
    }
  }

  // N+(q): nv x nq column-major, v = N+(q) qdot.  Identity blocks except the
  // 3x4 quaternion block 2 L(q~)^T (I - q~ q~^T)/|q|.
  void Nplus(const double* q, double* N) const {
    const Model& m = model;
    std::memset(N, 0, sizeof(double) * m.nv * m.nq);
    for (int i = 0; i < m.nb; ++i) {
      const int qs = m.qstart[i], vs = m.vstart[i];
      switch (m.jtype[i]) {
        case IDTO_JOINT_REVOLUTE:
        case IDTO_JOINT_PRISMATIC: N[(size_t)qs * m.nv + vs] = 1.0; break;
        case IDTO_JOINT_PLANAR:
          for (int k = 0; k < 3; ++k) N[(size_t)(qs + k) * m.nv + vs + k] = 1.0;
          break;
        case IDTO_JOINT_FLOATING: {
          const double* qq = q + qs;
          const double nrm = std::sqrt(((qq[0] * qq[0] + qq[1] * qq[1]) + qq[2] * qq[2]) + qq[3] * qq[3]);
          const double t[4] = {qq[0] / nrm, qq[1] / nrm, qq[2] / nrm, qq[3] / nrm};
          // LT = L(2 q~)^T, 3x4
          const double w2 = 2.0 * t[0], x2 = 2.0 * t[1], y2 = 2.0 * t[2], z2 = 2.0 * t[3];
          const double LT[3][4] = {{-x2, w2, -z2, y2}, {-y2, z2, w2, -x2}, {-z2, -y2, x2, w2}};
          double D[4][4];
          for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) D[r][c] = ((r == c ? 1.0 : 0.0) - t[r] * t[c]) / nrm;
          for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
              double acc = LT[r][0] * D[0][c];
              for (int k = 1; k < 4; ++k) acc += LT[r][k] * D[k][c];
              N[(size_t)(qs + c) * m.nv + vs + r] = acc;
            }
          for (int k = 0; k < 3; ++k) N[(size_t)(qs + 4 + k) * m.nv + vs + 3 + k] = 1.0;
        } break;
      }
    }
  }
};

}  // namespace oracle
