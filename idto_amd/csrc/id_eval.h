// id_eval.h — one inverse-dynamics evaluation with compliant contact, spread over
// the `npaths` adjacent lanes of a wavefront that cooperate on it.
//
// Replaces, for the hot path, what the reference asks Drake for in
// CalcInverseDynamicsSingleTimeStep (reference optimizer/trajectory_optimizer.cc:228-245):
// force elements (:232), CalcContactForceContribution (:247-386, restated in
// contact_pair() below line by line) and plant.CalcInverseDynamics (:244).
//
// Formulation: world-frame recursive Newton-Euler about the body origins.  Each
// lane owns one *path* of the model's star decomposition (include/idto_model.h):
// it evaluates the optional common root body redundantly, then its own chain of
// up to MAXC bodies entirely in registers (all loops over chain slots are
// unrolled so the per-slot state never leaves VGPRs), evaluates the contact
// pairs assigned to its path, and meets the other lanes of the evaluation only
// in two butterfly sums (contact wrench on the common body, chain-root wrenches).
// The association order of every floating-point sum is the one documented in
// DESIGN.md §3.2, which is what makes the result bit-identical to the serial
// CPU oracle.
#pragma once

#include "dev_math.h"
#include "idto_model.h"

namespace idto_dev {

struct DevModel {
  int nb, nq, nv, npaths, common_body, ngeoms, npairs, maxpp;
  const int* parent;
  const int* jtype;
  const int* qstart;
  const int* vstart;
  const double* X_PF;     // 12 per body
  const double* axis;     // 3
  const double* mass;     // 1
  const double* com;      // 3
  const double* inertia;  // 6
  const double* damping;  // nv
  double gravity[3];
  const int* geom_type;
  const double* geom_X;     // 12
  const double* geom_size;  // 3
  // star decomposition tables
  const int* chain;        // [npaths*MAXCHAIN] body index (-1 = unused slot)
  const int* nchain;       // [npaths]
  const int* pkind;        // [npaths*MAXCHAIN] 0 world, 1 common, 2 previous slot
  const int* path_npairs;  // [npaths]
  const int* path_pairs;   // [npaths*maxpp] pair index, ascending
  const int* pair_ga;      // geometry A of the pair
  const int* pair_gb;
  const int* pair_sa;      // body slot of geometry A: -2 world, -1 common, >= 0 chain slot
  const int* pair_sb;
  // all tables above live in ONE device buffer (doubles first, then the int tables) so that a
  // kernel can stage the whole model into LDS with one coalesced copy and rebase the pointers
  const double* blob;
  int blob_n;              // size in doubles
  // fd_kernel's prologue: the floating joints (nfloat < 0: more than four, nplus_pair looks them up), the constant part
  // of N+ (nv x nq column-major: ones and zeros, NaN where a quaternion block goes; in the blob, read from global memory)
  // and where the non-zeros of each column / row of N+ are (first | count << 16)
  int nfloat;
  int float_qs[4], float_vs[4];
  const double* nplus_const;
  const int* colinfo;
  const int* rowinfo;
  // id_fast.h: the instantiated tree shape this model has (0: none, id_eval<MAXC> serves it) and
  // the gathered records of its bodies and contact pairs (inside the blob)
  int fast_shape;
  int fast_lo, fast_n;     // the part of the blob (doubles) a kernel of that shape needs: the records, jtype, qstart, vstart, f_seg
  int f_maxpp;
  const double* f_body;
  const double* f_cbody;
  const double* f_pairs;
  const int* f_seg;
};

// The model with every table pointer rebased from the global blob to a copy at `dst`
// (LDS: the per-body tables sit on the dependent chain of id_eval; from L2 each access costs
// ~1-2 us of exposed latency, from LDS ~100 cycles).
IDTO_DEV DevModel rebase_model(const DevModel& M, const double* dst) {
  DevModel L = M;
  auto d = [&](const double* p) { return dst + (p - M.blob); };
  auto i = [&](const int* p) {
    return reinterpret_cast<const int*>(dst) + (p - reinterpret_cast<const int*>(M.blob));
  };
  L.parent = i(M.parent); L.jtype = i(M.jtype); L.qstart = i(M.qstart); L.vstart = i(M.vstart);
  L.X_PF = d(M.X_PF); L.axis = d(M.axis); L.mass = d(M.mass); L.com = d(M.com); L.inertia = d(M.inertia);
  L.damping = d(M.damping);
  L.geom_type = i(M.geom_type); L.geom_X = d(M.geom_X); L.geom_size = d(M.geom_size);
  L.chain = i(M.chain); L.nchain = i(M.nchain); L.pkind = i(M.pkind);
  L.path_npairs = i(M.path_npairs); L.path_pairs = i(M.path_pairs);
  L.pair_ga = i(M.pair_ga); L.pair_gb = i(M.pair_gb); L.pair_sa = i(M.pair_sa); L.pair_sb = i(M.pair_sb);
  L.f_body = d(M.f_body); L.f_cbody = d(M.f_cbody); L.f_pairs = d(M.f_pairs); L.f_seg = i(M.f_seg);
  L.blob = dst;
  return L;
}

struct DevContact {
  double k, vd, vs, mu, sigma, threshold;
};

enum { PK_WORLD = 0, PK_COMMON = 1, PK_PREV = 2 };

struct BodyState {  // what later stages need of a body
  M3 R;
  V3 p, w, v;
};

struct JointOut {
  M3 R_FM;
  V3 d2, w_rel, v_rel, al_rel, a_rel, hW;
};

// Joint kinematics in the world frame given the joint frame R_WF (restates the
// switch of oracle/rigid_body.h Kinematics()).
IDTO_DEV JointOut joint_kin(int jtype, const M3& R_WF, V3 axis, const double* qi, const double* vi, const double* ai) {
  JointOut o;
  const V3 zero = mk(0, 0, 0);
  o.R_FM = ident3();
  o.d2 = zero; o.w_rel = zero; o.v_rel = zero; o.al_rel = zero; o.a_rel = zero; o.hW = zero;
  if (jtype == IDTO_JOINT_REVOLUTE) {
    double s, c;
    idto::detmath::sincos(qi[0], &s, &c);
    o.R_FM = axis_angle(axis, s, c);
    o.hW = R_WF * axis;
    o.w_rel = o.hW * vi[0];
    o.al_rel = o.hW * ai[0];
  } else if (jtype == IDTO_JOINT_PRISMATIC) {
    o.hW = R_WF * axis;
    o.d2 = o.hW * qi[0];
    o.v_rel = o.hW * vi[0];
    o.a_rel = o.hW * ai[0];
  } else if (jtype == IDTO_JOINT_PLANAR) {
    double s, c;
    idto::detmath::sincos(qi[2], &s, &c);
    o.R_FM.m[0] = c; o.R_FM.m[1] = -s; o.R_FM.m[3] = s; o.R_FM.m[4] = c;
    const V3 ex = col(R_WF, 0), ey = col(R_WF, 1), ez = col(R_WF, 2);
    o.d2 = ex * qi[0] + ey * qi[1];
    o.v_rel = ex * vi[0] + ey * vi[1];
    o.a_rel = ex * ai[0] + ey * ai[1];
    o.w_rel = ez * vi[2];
    o.al_rel = ez * ai[2];
  } else {  // floating
    o.R_FM = quat_to_rot(qi);
    o.d2 = R_WF * mk(qi[4], qi[5], qi[6]);
    o.w_rel = R_WF * mk(vi[0], vi[1], vi[2]);
    o.v_rel = R_WF * mk(vi[3], vi[4], vi[5]);
    o.al_rel = R_WF * mk(ai[0], ai[1], ai[2]);
    o.a_rel = R_WF * mk(ai[3], ai[4], ai[5]);
  }
  return o;
}

struct SdResult {
  bool valid;
  double phi;
  V3 n, Ca, Cb;
};

// Signed distance between two primitives (restates oracle SignedDistance()).
IDTO_DEV SdResult signed_distance(int typeA, const M3& RA, V3 pA, V3 sA, int typeB, const M3& RB, V3 pB, V3 sB) {
  SdResult out;
  out.valid = false; out.phi = 0; out.n = mk(0, 0, 1); out.Ca = mk(0, 0, 0); out.Cb = mk(0, 0, 0);
  if (typeA == IDTO_GEOM_SPHERE && typeB == IDTO_GEOM_SPHERE) {
    const V3 d = pB - pA;
    const double dist = __builtin_sqrt(dot(d, d));
    out.phi = (dist - sA.x) - sB.x;
    out.n = d / dist;
    out.Ca = pA + out.n * sA.x;
    out.Cb = pB - out.n * sB.x;
    out.valid = true;
  } else if (typeA != typeB) {  // sphere-box in either order
    const bool sphere_is_A = (typeA == IDTO_GEOM_SPHERE);
    const V3 pS = sphere_is_A ? pA : pB;
    const double rad = sphere_is_A ? sA.x : sB.x;
    const M3& RX = sphere_is_A ? RB : RA;
    const V3 pX = sphere_is_A ? pB : pA;
    const V3 h = sphere_is_A ? sB : sA;
    const V3 c = tmul(RX, pS - pX);
    V3 pc = c;
    bool outside = false;
    if (pc.x > h.x) { pc.x = h.x; outside = true; } else if (pc.x < -h.x) { pc.x = -h.x; outside = true; }
    if (pc.y > h.y) { pc.y = h.y; outside = true; } else if (pc.y < -h.y) { pc.y = -h.y; outside = true; }
    if (pc.z > h.z) { pc.z = h.z; outside = true; } else if (pc.z < -h.z) { pc.z = -h.z; outside = true; }
    V3 g;
    double phi;
    if (outside) {
      const V3 dv = c - pc;
      const double dist = __builtin_sqrt(dot(dv, dv));
      g = dv / dist;
      phi = dist - rad;
    } else {
      const double dx = h.x - __builtin_fabs(c.x), dy = h.y - __builtin_fabs(c.y), dz = h.z - __builtin_fabs(c.z);
      g = mk(0, 0, 0);
      double depth;
      if (dx <= dy && dx <= dz) { depth = dx; g.x = (c.x >= 0) ? 1.0 : -1.0; pc.x = g.x * h.x; }
      else if (dy <= dz) { depth = dy; g.y = (c.y >= 0) ? 1.0 : -1.0; pc.y = g.y * h.y; }
      else { depth = dz; g.z = (c.z >= 0) ? 1.0 : -1.0; pc.z = g.z * h.z; }
      phi = -depth - rad;
    }
    const V3 gW = RX * g;
    const V3 boxW = pX + RX * pc;
    const V3 sphW = pS - gW * rad;
    out.phi = phi;
    if (sphere_is_A) { out.n = -gW; out.Ca = sphW; out.Cb = boxW; }
    else { out.n = gW; out.Ca = boxW; out.Cb = sphW; }
    out.valid = true;
  } else {  // box (moving) vs world-fixed axis-aligned box: lowest vertex against the top face
    const double ztop = pB.z + sB.z;
    double zmin = 0;
    V3 best = mk(0, 0, 0);
    bool first = true;
    for (int ix = 0; ix < 2; ++ix)
      for (int iy = 0; iy < 2; ++iy)
        for (int iz = 0; iz < 2; ++iz) {
          const V3 cb = mk(ix ? sA.x : -sA.x, iy ? sA.y : -sA.y, iz ? sA.z : -sA.z);
          const V3 cw = pA + RA * cb;
          if (first || cw.z < zmin) { zmin = cw.z; best = cw; first = false; }
        }
    out.phi = zmin - ztop;
    out.n = mk(0, 0, -1);
    out.Ca = best;
    out.Cb = mk(best.x, best.y, ztop);
    out.valid = true;
  }
  return out;
}

struct PairForce {
  bool active;
  V3 fA, nA, fB, nB;  // wrenches on bodies A and B about their origins (world frame)
};

// reference TO.cc:281-385 for one signed-distance pair
IDTO_DEV PairForce contact_pair(const DevModel& M, const DevContact& cp, int ga, int gb, const BodyState& A,
                                const BodyState& B) {
  PairForce out;
  out.active = false;
  const M3 RgA = A.R * ldm3(M.geom_X + 12 * ga);
  const V3 pgA = A.p + A.R * ldv3(M.geom_X + 12 * ga + 9);
  const M3 RgB = B.R * ldm3(M.geom_X + 12 * gb);
  const V3 pgB = B.p + B.R * ldv3(M.geom_X + 12 * gb + 9);
  const SdResult sd = signed_distance(M.geom_type[ga], RgA, pgA, ldv3(M.geom_size + 3 * ga), M.geom_type[gb], RgB,
                                      pgB, ldv3(M.geom_size + 3 * gb));
  if (!sd.valid) return out;
  if (sd.phi > cp.threshold) return out;
  const V3 nhat = sd.n;
  const V3 pC = (sd.Ca + sd.Cb) * 0.5;
  const V3 pAC = pC - A.p, pBC = pC - B.p;
  const V3 vAc = A.v + cross(A.w, pAC);
  const V3 vBc = B.v + cross(B.w, pBC);
  const V3 vrel = vBc - vAc;
  const double vn = dot(nhat, vrel);
  const V3 vt = vrel - nhat * vn;
  double dissipation = 0.0;
  const double s = vn / cp.vd;
  if (s < 0) dissipation = 1 - s;
  else if (s < 2) dissipation = (s - 2) * (s - 2) / 4;
  double compliant_fn;
  const double exponent = -sd.phi / cp.sigma;
  if (exponent >= 37) compliant_fn = -cp.k * sd.phi;
  else compliant_fn = cp.sigma * cp.k * idto::detmath::log(1 + idto::detmath::exp(exponent));
  const double fn = compliant_fn * dissipation;
  const V3 that = (-vt) / __builtin_sqrt(cp.vs * cp.vs + dot(vt, vt));
  const V3 ft = (that * cp.mu) * fn;
  out.fB = nhat * fn + ft;
  out.fA = -out.fB;
  out.nB = cross(pBC, out.fB);
  out.nA = cross(pAC, out.fA);
  out.active = true;
  return out;
}

// Inertial wrench of body b about its origin, world frame.
IDTO_DEV void inertial_wrench(const DevModel& M, int b, const M3& R, V3 w, V3 al, V3 a, V3 g, V3* f_in, V3* n_in) {
  const V3 cW = R * ldv3(M.com + 3 * b);
  const V3 t1 = cross(al, cW);
  const V3 t2 = cross(w, cross(w, cW));
  const V3 acom = (a + t1) + t2;
  *f_in = (acom - g) * M.mass[b];
  const V3 wB = tmul(R, w), alB = tmul(R, al);
  const double* I = M.inertia + 6 * b;
  const V3 Iw = mk(fma3(I[0], wB.x, I[3], wB.y, I[4], wB.z), fma3(I[3], wB.x, I[1], wB.y, I[5], wB.z),
                   fma3(I[4], wB.x, I[5], wB.y, I[2], wB.z));
  const V3 Ial = mk(fma3(I[0], alB.x, I[3], alB.y, I[4], alB.z), fma3(I[3], alB.x, I[1], alB.y, I[5], alB.z),
                    fma3(I[4], alB.x, I[5], alB.y, I[2], alB.z));
  const V3 nB = Ial + cross(wB, Iw);
  *n_in = R * nB + cross(cW, *f_in);
}

// Generalised forces of body b's joint from the total wrench (f, n) on the body.
IDTO_DEV void put_tau(int j, double t, bool full, const double* damping, const double* v, double* tau) {
  tau[j] = full ? t + damping[j] * v[j] : t;
}
// Planar and floating joints are attached to the world (checked when the model is built), so
// their joint frame R_WF = I * R_PF is re-formed here from the model table (the same expression
// the forward pass evaluates) instead of being kept in 9 registers per body across the contact
// phase: that was what pushed the kernel over 512 registers per lane.
IDTO_DEV void project_tau(int jtype, int vs, const double* xpf, V3 hW, V3 f, V3 n, bool full, const double* damping,
                          const double* v, double* tau) {
  if (jtype == IDTO_JOINT_REVOLUTE) {
    put_tau(vs, dot(hW, n), full, damping, v, tau);
  } else if (jtype == IDTO_JOINT_PRISMATIC) {
    put_tau(vs, dot(hW, f), full, damping, v, tau);
  } else if (jtype == IDTO_JOINT_PLANAR) {
    const M3 R_WF = ident3() * ldm3(xpf);
    put_tau(vs, dot(col(R_WF, 0), f), full, damping, v, tau);
    put_tau(vs + 1, dot(col(R_WF, 1), f), full, damping, v, tau);
    put_tau(vs + 2, dot(col(R_WF, 2), n), full, damping, v, tau);
  } else {
    const M3 R_WF = ident3() * ldm3(xpf);
    const V3 nF = tmul(R_WF, n), fF = tmul(R_WF, f);
    put_tau(vs, nF.x, full, damping, v, tau);
    put_tau(vs + 1, nF.y, full, damping, v, tau);
    put_tau(vs + 2, nF.z, full, damping, v, tau);
    put_tau(vs + 3, fF.x, full, damping, v, tau);
    put_tau(vs + 4, fF.y, full, damping, v, tau);
    put_tau(vs + 5, fF.z, full, damping, v, tau);
  }
}

// tau = ID(q, v, a) for the lane's path.  q, v, a and tau point to this
// evaluation's arrays in LDS (tau: written for the lane's own DoFs; path 0 also
// writes the common body's).  `full` selects gravity + damping + contact, otherwise
// the mass-matrix column mode.  Must be called by all `npaths` lanes of the
// evaluation together (it contains cross-lane butterfly sums).
template <int MAXC>
IDTO_DEV void id_eval(const DevModel& M, const DevContact& cp, int path, bool full, const double* q, const double* v,
                      const double* a, double* tau) {
  const V3 zero = mk(0, 0, 0);
  const V3 g = full ? mk(M.gravity[0], M.gravity[1], M.gravity[2]) : zero;

  // ---- common root body (identical in every lane of the evaluation)
  BodyState cb;
  cb.R = ident3(); cb.p = zero; cb.w = zero; cb.v = zero;
  V3 cb_al = zero, cb_a = zero, cb_fin = zero, cb_nin = zero;
  V3 cb_hW = zero;
  const int cbody = M.common_body;
  if (cbody >= 0) {
    const M3 cb_RWF = ldm3(M.X_PF + 12 * cbody);  // parent is the world: I * R_PF
    const V3 d1 = ldv3(M.X_PF + 12 * cbody + 9);
    const JointOut j = joint_kin(M.jtype[cbody], cb_RWF, ldv3(M.axis + 3 * cbody), q + M.qstart[cbody],
                                 v + M.vstart[cbody], a + M.vstart[cbody]);
    cb.R = cb_RWF * j.R_FM;
    cb.p = d1 + j.d2;
    cb.w = j.w_rel;
    cb.v = j.v_rel;
    cb_al = j.al_rel;
    cb_a = j.a_rel;
    cb_hW = j.hW;
    inertial_wrench(M, cbody, cb.R, cb.w, cb_al, cb_a, g, &cb_fin, &cb_nin);
  }

  // ---- own chain: forward pass
  BodyState bs[MAXC];
  V3 r[MAXC], hW[MAXC], fin[MAXC], nin[MAXC], fext[MAXC], next[MAXC];
  const int nch = M.nchain[path];
  V3 al_prev = zero, a_prev = zero;
#pragma unroll
  for (int s = 0; s < MAXC; ++s) {
    bs[s].R = ident3(); bs[s].p = zero; bs[s].w = zero; bs[s].v = zero;
    r[s] = zero; hW[s] = zero; fin[s] = zero; nin[s] = zero; fext[s] = zero; next[s] = zero;
    if (s < nch) {
      const int b = M.chain[path * IDTO_MAX_CHAIN + s];
      const int kind = M.pkind[path * IDTO_MAX_CHAIN + s];
      M3 Rp = ident3();
      V3 pp = zero, wp = zero, vp = zero, alp = zero, ap = zero;
      if (kind == PK_COMMON) { Rp = cb.R; pp = cb.p; wp = cb.w; vp = cb.v; alp = cb_al; ap = cb_a; }
      if (s > 0 && kind == PK_PREV) {
        Rp = bs[s > 0 ? s - 1 : 0].R; pp = bs[s > 0 ? s - 1 : 0].p; wp = bs[s > 0 ? s - 1 : 0].w;
        vp = bs[s > 0 ? s - 1 : 0].v; alp = al_prev; ap = a_prev;
      }
      const M3 R_WF = Rp * ldm3(M.X_PF + 12 * b);
      const V3 d1 = Rp * ldv3(M.X_PF + 12 * b + 9);
      const JointOut j = joint_kin(M.jtype[b], R_WF, ldv3(M.axis + 3 * b), q + M.qstart[b], v + M.vstart[b],
                                   a + M.vstart[b]);
      hW[s] = j.hW;
      bs[s].R = R_WF * j.R_FM;
      r[s] = d1 + j.d2;
      bs[s].p = pp + r[s];
      bs[s].w = wp + j.w_rel;
      bs[s].v = (vp + cross(wp, r[s])) + j.v_rel;
      const V3 al = (alp + j.al_rel) + cross(wp, j.w_rel);
      const V3 acc = (((ap + cross(alp, r[s])) + cross(wp, cross(wp, r[s]))) + cross(wp, j.v_rel) * 2.0) + j.a_rel;
      inertial_wrench(M, b, bs[s].R, bs[s].w, al, acc, g, &fin[s], &nin[s]);
      al_prev = al;
      a_prev = acc;
    }
  }

  // ---- contact pairs of this path (TO.cc:247-386)
  V3 cfe = zero, cne = zero;  // this path's partial contact wrench on the common body
  if (full) {
    const int np = M.path_npairs[path];
    for (int k = 0; k < np; ++k) {
      const int pi = M.path_pairs[path * M.maxpp + k];
      const int sa = M.pair_sa[pi], sb = M.pair_sb[pi];
      BodyState A, B;
      A.R = ident3(); A.p = zero; A.w = zero; A.v = zero;
      B = A;
      if (sa == -1) A = cb;
      if (sb == -1) B = cb;
#pragma unroll
      for (int s = 0; s < MAXC; ++s) {
        if (sa == s) A = bs[s];
        if (sb == s) B = bs[s];
      }
      const PairForce pf = contact_pair(M, cp, M.pair_ga[pi], M.pair_gb[pi], A, B);
      if (pf.active) {
        if (sa == -1) { cfe = cfe + pf.fA; cne = cne + pf.nA; }
        if (sb == -1) { cfe = cfe + pf.fB; cne = cne + pf.nB; }
#pragma unroll
        for (int s = 0; s < MAXC; ++s) {
          if (sa == s) { fext[s] = fext[s] + pf.fA; next[s] = next[s] + pf.nA; }
          if (sb == s) { fext[s] = fext[s] + pf.fB; next[s] = next[s] + pf.nB; }
        }
      }
    }
  }

  // ---- backward pass along the chain, joint torques
  V3 child_f = zero, child_n = zero;   // contribution of slot s+1 to slot s
  V3 root_f = zero, root_n = zero;     // contribution of the chain root to the common body
#pragma unroll
  for (int s = MAXC - 1; s >= 0; --s) {
    if (s < nch) {
      const int b = M.chain[path * IDTO_MAX_CHAIN + s];
      V3 f = fin[s] - fext[s];
      V3 n = nin[s] - next[s];
      // slot s+1 hangs off slot s ?
      if (s + 1 < MAXC) {
        const bool has_child = (s + 1 < nch) && (M.pkind[path * IDTO_MAX_CHAIN + (s + 1 < MAXC ? s + 1 : s)] == PK_PREV);
        if (has_child) { f = f + child_f; n = n + child_n; }
      }
      project_tau(M.jtype[b], M.vstart[b], M.X_PF + 12 * b, hW[s], f, n, full, M.damping, v, tau);
      const int kind = M.pkind[path * IDTO_MAX_CHAIN + s];
      const V3 cf = f, cn = n + cross(r[s], f);
      if (kind == PK_PREV) { child_f = cf; child_n = cn; }
      else if (kind == PK_COMMON) { root_f = cf; root_n = cn; }
    }
  }

  // ---- common body: butterfly sums over the lanes of this evaluation
  if (cbody >= 0) {   // uniform over the launch
    const V3 ext_f = tree_sum(cfe, M.npaths);
    const V3 ext_n = tree_sum(cne, M.npaths);
    const V3 ch_f = tree_sum(root_f, M.npaths);
    const V3 ch_n = tree_sum(root_n, M.npaths);
    const V3 f = (cb_fin - ext_f) + ch_f;
    const V3 n = (cb_nin - ext_n) + ch_n;
    if (path == 0) project_tau(M.jtype[cbody], M.vstart[cbody], M.X_PF + 12 * cbody, cb_hW, f, n, full, M.damping, v, tau);
  }
}

}  // namespace idto_dev
