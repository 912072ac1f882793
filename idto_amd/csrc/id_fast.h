// id_fast.h — the inverse-dynamics evaluation of id_eval.h as straight-line code for the tree
// shapes of the reference's examples (reference optimizer/trajectory_optimizer.cc:228-386, the
// same routines id_eval.h replaces).
//
// id_eval<MAXC> reads the joint type, the parent kind and the chain length of every slot from the
// model tables per lane, which turns the recursion into ~240 exec-masked branches with a table
// read from LDS in front of each, and selects the two bodies of a contact pair out of the
// register-resident chain with compare cascades (profiles/r03_pmc_sq_summary.txt: 29 % of the
// wavefront's cycles issue VALU work, 52 % wait).  Here the shape is a template parameter:
//   CJ   joint of the common root body: -1 (none) or IDTO_JOINT_FLOATING, attached to the world
//   J0   joint of chain slot 0: revolute or planar;  K0: its parent, PK_WORLD or PK_COMMON
//   slots 1 .. MAXC-1: revolute, parent = the previous slot; every path has exactly MAXC bodies
// and every contact pair touches at most one chain body.  BuildModel (idto_hip.hip) checks a
// model against the instantiated shapes and gathers, on the host, one record of constants per
// (path, slot) and per contact pair in the order this file walks them, so that every table read
// is `base(path) + immediate` and nothing on the recursion's chain waits for an index.
// The contact pairs of a slot are evaluated right after the slot's kinematics: a finished slot
// keeps 12 doubles (r, hW, f, n) instead of 36, which is what made <4> spill.
//
// Floating point: every value is produced by the same operations in the same order as in
// id_eval.h / oracle/rigid_body.h.  Where a term of the generic expression is a
// product with the world's zero velocity it is replaced by the `+ 0.0` that it amounts to for
// finite operands (x + (+-0 * finite) + 0 == x + 0 bit for bit, including the sign of a zero
// result); the world-frame pose I * X_PF of a body attached to the world is formed on the host
// with the same fused expressions.
#pragma once

#include "id_eval.h"

// profiling build (tools/fd_variants.sh, -DIDTO_FD_STAMPS): shader-clock stamps of the phases of fd_kernel and of the
// evaluation, taken by the lanes tid 0 and tid 192 and returned in v_out (tools/fd_stamps.py)
#ifdef IDTO_FD_STAMPS
#ifndef IDTO_FD_STAMP_TID
#define IDTO_FD_STAMP_TID 192
#endif
#define FD_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); if (idto_fd_st) idto_fd_st[i] = (long long)__builtin_readcyclecounter(); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FD_STAMP(i) do { } while (0)
#endif

namespace idto_dev {

// record of one body (doubles); FB_IDX holds {qstart, vstart} as two ints
enum { FB_XPF = 0, FB_AXIS = 12, FB_MASS = 15, FB_COM = 16, FB_INERTIA = 19, FB_DAMP = 25, FB_IDX = 31, FB_STRIDE = 34 };
// record of one contact pair; FP_INFO holds {type on C, type on the other body, C is the pair's A, the other body is
// the common one} as four ints.
// XC, SC: geometry frame and size on C (the chain slot of the pair's group; the common body for a pair without a
// chain body); XO, SO: on the other body - for the world [I R | 0 + I p], formed on the host
enum { FP_INFO = 0, FP_XC = 2, FP_SC = 14, FP_XO = 17, FP_SO = 29, FP_STRIDE = 34 };

// x + (x of the lane whose index differs in bit 0 / bit 1): the butterfly of dev_math.h tree_sum
// as DPP quad permutations (no LDS round trip); lanes of one evaluation are adjacent and aligned
template <int CTRL>
IDTO_DEV double quad_perm(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int NP>
IDTO_DEV V3 tree_sum_fast(V3 v) {
  if (NP >= 2) {
    v.x = v.x + quad_perm<0xB1>(v.x); v.y = v.y + quad_perm<0xB1>(v.y); v.z = v.z + quad_perm<0xB1>(v.z);
  }
  if (NP >= 4) {
    v.x = v.x + quad_perm<0x4E>(v.x); v.y = v.y + quad_perm<0x4E>(v.y); v.z = v.z + quad_perm<0x4E>(v.z);
  }
  if (NP >= 8) {
    v.x = xor_add(v.x, 4); v.y = xor_add(v.y, 4); v.z = xor_add(v.z, 4);
  }
  return v;
}

// The same LDS address behind an empty asm: the compiler can no longer merge the reads through it with earlier reads
// of the same words.  It otherwise hoists the constants of the LAST phase (damping, the joint frames of the projection,
// the velocities of the damping term) to the top of the evaluation - nothing in between writes to LDS provably - and
// then carries ~40 registers across the whole recursion, or spills them to scratch: six dependent scratch reloads at
// the end of the evaluation cost 4k cycles.
IDTO_DEV const double* lds_launder(const double* p) {
  typedef __attribute__((address_space(3))) const double lds_cdouble;
  unsigned a = (unsigned)(size_t)(lds_cdouble*)p;
  asm volatile("" : "+v"(a));
  return (const double*)(lds_cdouble*)(size_t)a;
}

// Values that are computed in the forward pass and used in the backward pass: pinned where they are computed.  The
// compiler otherwise SINKS their computation to the use (the inertial wrench of every slot, ~110 operations each, moved
// into the backward pass) and keeps the operands alive instead - 18 doubles per slot for 6 - which is what drove the
// kernel over its 512 registers.
IDTO_DEV void pin(V3& a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z)); }

// The 25 constants of a body's record (FB_XPF .. FB_INERTIA) read ONE SLOT AHEAD: volatile LDS reads stay where they are
// written - before the contact pairs of the slot before - and the wait for them is placed at their first use, behind
// those pairs; one wavefront per SIMD has nothing else to cover the ~130 cycles of an LDS round trip with
// (profiles/r04_fd_pmc.txt: 43 % of the evaluation's cycles were waits).
enum { FB_PREFETCH = FB_INERTIA + 6 };
IDTO_DEV void prefetch_record(const double* rec, double (&rc)[FB_PREFETCH]) {
#ifdef IDTO_FAST_NO_PREFETCH
#pragma unroll
  for (int i = 0; i < FB_PREFETCH; ++i) rc[i] = rec[i];
#else
  typedef __attribute__((address_space(3))) const volatile double lds_cvdouble;
  lds_cvdouble* p = (lds_cvdouble*)rec;
#pragma unroll
  for (int i = 0; i < FB_PREFETCH; ++i) rc[i] = p[i];
#endif
}

struct ParentKin {  // what a child needs of its parent
  M3 R;
  V3 p, w, v, al, a;
};

IDTO_DEV V3 sel3(bool c, V3 a, V3 b) { return mk(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }

// One signed-distance pair (reference TO.cc:281-385): id_eval.h contact_pair() + signed_distance()
// with the distance test moved in front of the witness points (three divisions and two 3x3
// products that an inactive pair never needs).  The pair's bodies are C - the chain slot of the
// group the pair is listed in (the common body for the groups without a chain body) - and "the other
// one", O: the common body (`oc`) or the world, whose geometry pose the host has already formed.
//
// Everything is evaluated with C in the role of the pair's body A.  When C is the pair's B
// (`cia` false) that is the reference's computation with the roles exchanged, and it produces the same
// bits: the direction-like intermediates (d, n, v_rel, v_t, t, f) come out exactly negated - negation
// commutes with every rounding, (-a) - (-b) == -(a - b), (-a)(-b) == a b - the even ones (dist, v_n,
// |v_t|^2, the force magnitudes) and the witness points are unchanged, sums of two points commute, and
// "the wrench on the second body" is then the reference's wrench on A, i.e. again the one on O.  The single
// expression that is not symmetric is phi = (dist - r_A) - r_B of two spheres: it takes the radii in the
// pair's own order.  Returns whether the pair is active; (fc, nc) is the wrench on C about its origin,
// (fo, no) the one on O.
template <bool HAS_COMMON>
IDTO_DEV bool pair_eval(const double* pr, const DevContact& cp, const BodyState& C, const BodyState& cb, V3* fc, V3* nc,
                        V3* fo, V3* no) {
  const V3 zero = mk(0, 0, 0);
  const int2 types = *reinterpret_cast<const int2*>(pr + FP_INFO);
  const int2 flags = *reinterpret_cast<const int2*>(pr + FP_INFO + 1);
  const int typeC = types.x, typeO = types.y;
  const bool cia = flags.x != 0, oc = HAS_COMMON && flags.y != 0;
  const V3 pgC = C.p + C.R * ldv3(pr + FP_XC + 9);
  V3 pgO = ldv3(pr + FP_XO + 9);   // (the world's: 0 + I p, formed on the host)
  if (HAS_COMMON) pgO = sel3(oc, cb.p + cb.R * pgO, pgO);
  const V3 sC = ldv3(pr + FP_SC), sO = ldv3(pr + FP_SO);
  double phi = 0;
  V3 n1 = mk(0, 0, 1), Cc = zero, Co = zero;   // unit vector from C's geometry to O's, witness points on C and on O
  bool hit = false;
  if (typeC == IDTO_GEOM_SPHERE && typeO == IDTO_GEOM_SPHERE) {
    const V3 d = pgO - pgC;
    const double dist = __builtin_sqrt(dot(d, d));
    const double rA = cia ? sC.x : sO.x, rB = cia ? sO.x : sC.x;
    phi = (dist - rA) - rB;
    if (!(phi > cp.threshold)) {
      n1 = d / dist;
      Cc = pgC + n1 * sC.x;
      Co = pgO - n1 * sO.x;
      hit = true;
    }
  } else if (typeC != typeO) {  // sphere-box in either order
    const bool sphere_is_C = (typeC == IDTO_GEOM_SPHERE);
    M3 RX;                      // the box's rotation in the world
    if (sphere_is_C) {          // the box is on the other body
      RX = ldm3(pr + FP_XO);    // (the world's: I R, formed on the host)
      if (HAS_COMMON) {
        const M3 RXc = cb.R * RX;
#pragma unroll
        for (int i = 0; i < 9; ++i) RX.m[i] = oc ? RXc.m[i] : RX.m[i];
      }
    } else {
      RX = C.R * ldm3(pr + FP_XC);
    }
    const V3 pS = sphere_is_C ? pgC : pgO;
    const double rad = sphere_is_C ? sC.x : sO.x;
    const V3 pX = sphere_is_C ? pgO : pgC;
    const V3 h = sphere_is_C ? sO : sC;
    const V3 c = tmul(RX, pS - pX);
    V3 pc = c;
    bool outside = false;
    if (pc.x > h.x) { pc.x = h.x; outside = true; } else if (pc.x < -h.x) { pc.x = -h.x; outside = true; }
    if (pc.y > h.y) { pc.y = h.y; outside = true; } else if (pc.y < -h.y) { pc.y = -h.y; outside = true; }
    if (pc.z > h.z) { pc.z = h.z; outside = true; } else if (pc.z < -h.z) { pc.z = -h.z; outside = true; }
    V3 g = zero, dv = zero;
    double dist = 1.0;
    if (outside) {
      dv = c - pc;
      dist = __builtin_sqrt(dot(dv, dv));
      phi = dist - rad;
    } else {
      const double dx = h.x - __builtin_fabs(c.x), dy = h.y - __builtin_fabs(c.y), dz = h.z - __builtin_fabs(c.z);
      double depth;
      if (dx <= dy && dx <= dz) { depth = dx; g.x = (c.x >= 0) ? 1.0 : -1.0; pc.x = g.x * h.x; }
      else if (dy <= dz) { depth = dy; g.y = (c.y >= 0) ? 1.0 : -1.0; pc.y = g.y * h.y; }
      else { depth = dz; g.z = (c.z >= 0) ? 1.0 : -1.0; pc.z = g.z * h.z; }
      phi = -depth - rad;
    }
    if (!(phi > cp.threshold)) {
      if (outside) g = dv / dist;
      const V3 gW = RX * g;             // from the box towards the sphere
      const V3 boxW = pX + RX * pc;
      const V3 sphW = pS - gW * rad;
      if (sphere_is_C) { n1 = -gW; Cc = sphW; Co = boxW; }
      else { n1 = gW; Cc = boxW; Co = sphW; }
      hit = true;
    }
  } else {  // box A on a moving body (always C: BuildModel) vs world-fixed axis-aligned box B: A's lowest vertex against B's top face
    const M3 RgA = C.R * ldm3(pr + FP_XC);
    const double ztop = pgO.z + sO.z;
    double zmin = 0;
    V3 best = zero;
#pragma unroll
    for (int iv = 0; iv < 8; ++iv) {
      const V3 cbx = mk((iv & 4) ? sC.x : -sC.x, (iv & 2) ? sC.y : -sC.y, (iv & 1) ? sC.z : -sC.z);
      const V3 cw = pgC + RgA * cbx;
      if (iv == 0 || cw.z < zmin) { zmin = cw.z; best = cw; }
    }
    phi = zmin - ztop;
    if (!(phi > cp.threshold)) {
      n1 = mk(0, 0, -1);
      Cc = best;
      Co = mk(best.x, best.y, ztop);
      hit = true;
    }
  }
  if (!hit) return false;
  const V3 pC = (Cc + Co) * 0.5;
  const V3 pCC = pC - C.p;
  // (O the world: pC - 0 == pC and 0 + (0 x p) == +0 for a finite p)
  V3 pOC = pC, vOc = zero;
  if (HAS_COMMON) {
    const V3 pOCc = pC - cb.p;
    pOC = sel3(oc, pOCc, pC);
    vOc = sel3(oc, cb.v + cross(cb.w, pOCc), zero);
  }
  const V3 vCc = C.v + cross(C.w, pCC);
  const V3 vrel = vOc - vCc;
  const double vn = dot(n1, vrel);
  const V3 vt = vrel - n1 * vn;
  double dissipation = 0.0;
  const double s = vn / cp.vd;
  if (s < 0) dissipation = 1 - s;
  else if (s < 2) dissipation = (s - 2) * (s - 2) / 4;
  double compliant_fn;
  const double exponent = -phi / cp.sigma;
  if (exponent >= 37) compliant_fn = -cp.k * phi;
  else compliant_fn = cp.sigma * cp.k * idto::detmath::log(1 + idto::detmath::exp(exponent));
  const double fn = compliant_fn * dissipation;
  const V3 that = (-vt) / __builtin_sqrt(cp.vs * cp.vs + dot(vt, vt));
  const V3 ft = (that * cp.mu) * fn;
  *fo = n1 * fn + ft;
  *fc = -*fo;
  *no = cross(pOC, *fo);
  *nc = cross(pCC, *fc);
  return true;
}

// inertial_wrench() of id_eval.h with the body's constants read from its record
IDTO_DEV void inertial_wrench_rec(const double* rec, const M3& R, V3 w, V3 al, V3 a, V3 g, V3* f_in, V3* n_in) {
  const V3 cW = R * ldv3(rec + FB_COM);
  const V3 t1 = cross(al, cW);
  const V3 t2 = cross(w, cross(w, cW));
  const V3 acom = (a + t1) + t2;
  *f_in = (acom - g) * rec[FB_MASS];
  const V3 wB = tmul(R, w), alB = tmul(R, al);
  const double* I = rec + FB_INERTIA;
  const V3 Iw = mk(fma3(I[0], wB.x, I[3], wB.y, I[4], wB.z), fma3(I[3], wB.x, I[1], wB.y, I[5], wB.z),
                   fma3(I[4], wB.x, I[5], wB.y, I[2], wB.z));
  const V3 Ial = mk(fma3(I[0], alB.x, I[3], alB.y, I[4], alB.z), fma3(I[3], alB.x, I[1], alB.y, I[5], alB.z),
                    fma3(I[4], alB.x, I[5], alB.y, I[2], alB.z));
  const V3 nB = Ial + cross(wB, Iw);
  *n_in = R * nB + cross(cW, *f_in);
}

// The contact pairs [start, start + count) of this path's list: C is the chain slot the group
// belongs to (the common body for the two groups without a chain body), cb the common body.
// Wrenches on C are added to (*fext, *next), those on the other body - if it is the common one -
// to (*cfe, *cne), in list order.  (fext, next) may be (cfe, cne): then C is the common body.
// (reading a pair's record a pair ahead, as prefetch_record does for the bodies, was measured and dropped: the 32 more
// registers per record in flight and the copies cost more than the round trips they hide - cheetah 22.5 -> 23.8 us)
template <bool HAS_COMMON>
IDTO_DEV void pair_group(const double* plist, int segword, const DevContact& cp, const BodyState& C, const BodyState& cb,
                         V3* fext, V3* next, V3* cfe, V3* cne) {
  const int start = segword & 0xffff, count = segword >> 16;
  for (int j = start; j < start + count; ++j) {
    const double* pr = plist + j * FP_STRIDE;
    V3 fc, nc, fo, no;
    if (pair_eval<HAS_COMMON>(pr, cp, C, cb, &fc, &nc, &fo, &no)) {
      *fext = *fext + fc; *next = *next + nc;
      if (HAS_COMMON) {
        const bool oc = reinterpret_cast<const int*>(pr + FP_INFO)[3] != 0;
        if (oc) { *cfe = *cfe + fo; *cne = *cne + no; }
      }
    }
  }
}

// The gathered tables (all inside the model blob, so that rebase_model() moves them to LDS)
struct FastTab {
  const double* body;    // [npaths][MAXC] records of FB_STRIDE doubles
  const double* cbody;   // the common body's record
  const double* pairs;   // [npaths][maxpp] records of FP_STRIDE doubles, in processing order
  const int* seg;        // [npaths][MAXC + 2] start | count << 16: pairs without a chain body that come first, slots 0 .. MAXC-1, the rest
  int maxpp;
};

// Where an evaluation's q, v, a come from.  InLds: arrays some other phase has built.
struct InLds {
  const double *q, *v, *a;
  IDTO_DEV double Q(int i) const { return q[i]; }
  IDTO_DEV double V(int j) const { return v[j]; }
  IDTO_DEV double A(int j) const { return a[j]; }
  IDTO_DEV InLds laundered() const { InLds o = *this; o.v = lds_launder(v); return o; }
};
// InFwd: evaluation e of the forward-difference set of one record (reference TO.cc:501-540, kernels.h fd_body),
// formed by the lane that consumes it from q_{k+1}, v_{k+1}, a_k and column `col` of N+_{k+1}, N+_k:
//   kind 0  the record's own tau:      q1, v1, a0
//   kind 1  q_{k+1}[col] + dq:         q1 + dq e_col,  v1 + dv n1,  a0 + da n1            (:514-521)
//   kind 2  q_k[col] + dq:             q1,             v1 - dv n1,  a0 - da (n1 + n0)     (:534-540)
//   kind 3  mass-matrix column col:    q1,             0,           e_col                  (:556-561)
// Written without a branch or a select on a loaded value (the compiler turns those into exec-masked regions
// with the LDS read inside: 28 dependent round trips, 6.4k cycles): every lane reads the same table entries,
// and the kinds differ in the multipliers sdv, sda (dv, da with the sign of the kind - (-x) y == -(x y) and
// a - b == a + (-b) bit for bit - and 0 for the kinds 0 and 3) and in bit masks on v1, a0, n0 (all ones / zero).
// x + 0 * n and n1 + (+0) are x and n1 except that a zero -0 may come out as +0: inputs, and with them the
// outputs, can differ from the expressions of TO.cc in the SIGN OF A ZERO only.
struct InFwd {
  const double *q1, *v1, *a0, *N1, *N0;
  int nv, kind, col;
  double dq, sdv, sda;
  unsigned long long keep, keep0;   // kind 3 drops v1, a0; kinds other than 2 drop n0
  IDTO_DEV static double and_bits(double x, unsigned long long m) {
    return __longlong_as_double((long long)((unsigned long long)__double_as_longlong(x) & m));
  }
  IDTO_DEV double Q(int i) const {
    const double x = q1[i], xp = x + dq;
    return (kind == 1 && i == col) ? xp : x;
  }
  IDTO_DEV double V(int j) const { return and_bits(v1[j], keep) + sdv * N1[col * nv + j]; }
  IDTO_DEV double A(int j) const {
    const double unit = (kind == 3 && j == col) ? 1.0 : 0.0;   // (kind 3: a0 is masked to +0, +0 + 1 == 1)
    return (and_bits(a0[j], keep) + unit) + sda * (N1[col * nv + j] + and_bits(N0[col * nv + j], keep0));
  }
  IDTO_DEV InFwd laundered() const { InFwd o = *this; o.v1 = lds_launder(v1); o.N1 = lds_launder(N1); return o; }
};

// tau = ID(q, v, a) for the lane's path: id_eval<MAXC> for a model of shape (CJ, J0, K0).
// W2 >= 1 (the spinner: finger of two links + the spinner itself in ONE path): slot W2 hangs off the world again, and the
// pairs of its group touch slot W2 - 1 as their other body (pair_eval's "common" argument is that slot's state; the
// force on it is taken out of its wrench after the group - the slot has no pairs of its own, BuildModel checks, so the
// order of the generic sum, fin - (0 + f), is kept).
template <int MAXC, int NP, int CJ, int J0, int K0, int W2, class In>
IDTO_DEV void id_eval_fast(const FastTab& T, const double* gravity, const DevContact& cp, int path, bool full,
                           const In& in_fwd, double* tau, long long* idto_fd_st = nullptr) {
  const In& in = in_fwd;
  constexpr bool HAS_COMMON = (CJ == IDTO_JOINT_FLOATING);
  const V3 zero = mk(0, 0, 0);
  const V3 g = full ? mk(gravity[0], gravity[1], gravity[2]) : zero;
  const double* bt = T.body + (size_t)path * (MAXC * FB_STRIDE);
  const double* ct = T.cbody;
  const double* plist = T.pairs + (size_t)path * (T.maxpp * FP_STRIDE);
  const int* seg = T.seg + path * (MAXC + 2);

  // ---- every input of the lane's joints, then every sine / cosine: none of them depends on the recursion
  int qs[MAXC], vs[MAXC];
#pragma unroll
  for (int s = 0; s < MAXC; ++s) {
    const int2 ix = *reinterpret_cast<const int2*>(bt + s * FB_STRIDE + FB_IDX);
    qs[s] = ix.x; vs[s] = ix.y;
  }
  int cqs = 0, cvs = 0;
  if (HAS_COMMON) {
    const int2 ix = *reinterpret_cast<const int2*>(ct + FB_IDX);
    cqs = ix.x; cvs = ix.y;
  }
  double qj[MAXC], vj[MAXC], aj[MAXC];       // the revolute coordinate of each slot (slot 0 planar: theta)
  double q0x = 0, q0y = 0, v0x = 0, v0y = 0, a0x = 0, a0y = 0;   // slot 0 planar: x, y
#pragma unroll
  for (int s = 0; s < MAXC; ++s) {
    if (s == 0 && J0 == IDTO_JOINT_PLANAR) {
      q0x = in.Q(qs[0]); q0y = in.Q(qs[0] + 1); qj[0] = in.Q(qs[0] + 2);
      v0x = in.V(vs[0]); v0y = in.V(vs[0] + 1); vj[0] = in.V(vs[0] + 2);
      a0x = in.A(vs[0]); a0y = in.A(vs[0] + 1); aj[0] = in.A(vs[0] + 2);
    } else {
      qj[s] = in.Q(qs[s]); vj[s] = in.V(vs[s]); aj[s] = in.A(vs[s]);
    }
  }
  double cq[7], cv[6], ca[6];
  if (HAS_COMMON) {
#pragma unroll
    for (int i = 0; i < 7; ++i) cq[i] = in.Q(cqs + i);
#pragma unroll
    for (int i = 0; i < 6; ++i) { cv[i] = in.V(cvs + i); ca[i] = in.A(cvs + i); }
  }
  FD_STAMP(4);
  double sn[MAXC], cs[MAXC];
#pragma unroll
  for (int s = 0; s < MAXC; ++s) idto::detmath::sincos(qj[s], &sn[s], &cs[s]);
  FD_STAMP(5);

  double rc_next[FB_PREFETCH];
  prefetch_record(bt, rc_next);   // slot 0's constants: in flight while the common body is evaluated
  // ---- common root body (identical in every lane of the evaluation)
  BodyState cb;
  cb.R = ident3(); cb.p = zero; cb.w = zero; cb.v = zero;
  V3 cb_al = zero, cb_a = zero, cb_fin = zero, cb_nin = zero;
  V3 cfe = zero, cne = zero;   // this path's partial contact wrench on the common body
  if (HAS_COMMON) {
    const M3 R_WF = ldm3(ct + FB_XPF);           // I * R_PF, formed on the host
    const V3 d1 = ldv3(ct + FB_XPF + 9);         // I * p_PF
    const M3 R_FM = quat_to_rot(cq);
    const V3 d2 = R_WF * mk(cq[4], cq[5], cq[6]);
    const V3 w_rel = R_WF * mk(cv[0], cv[1], cv[2]);
    const V3 v_rel = R_WF * mk(cv[3], cv[4], cv[5]);
    const V3 al_rel = R_WF * mk(ca[0], ca[1], ca[2]);
    const V3 a_rel = R_WF * mk(ca[3], ca[4], ca[5]);
    cb.R = R_WF * R_FM;
    cb.p = zero + (d1 + d2);
    cb.w = zero + w_rel;
    cb.v = zero + v_rel;
    cb_al = zero + al_rel;
    cb_a = zero + a_rel;
    inertial_wrench_rec(ct, cb.R, cb.w, cb_al, cb_a, g, &cb_fin, &cb_nin);
    pin(cb_fin); pin(cb_nin);
    if (full) pair_group<false>(plist, seg[0], cp, cb, cb, &cfe, &cne, &cfe, &cne);
  }

  FD_STAMP(6);
  // ---- own chain: kinematics, inertial wrench and contact pairs slot by slot
  V3 r[MAXC], hW[MAXC], ft[MAXC], nt[MAXC];   // ft, nt: inertial minus contact wrench of the slot
  BodyState bs_prev;                           // (W2: the state of the slot before)
  bs_prev.R = ident3(); bs_prev.p = zero; bs_prev.w = zero; bs_prev.v = zero;
  ParentKin P;
  P.R = ident3(); P.p = zero; P.w = zero; P.v = zero; P.al = zero; P.a = zero;
#pragma unroll
  for (int s = 0; s < MAXC; ++s) {
    double rec[FB_PREFETCH];
#pragma unroll
    for (int i = 0; i < FB_PREFETCH; ++i) rec[i] = rc_next[i];
    const bool world = (s == 0 && K0 == PK_WORLD) || (W2 >= 1 && s == W2);
    if (s == 0 && K0 == PK_COMMON) { P.R = cb.R; P.p = cb.p; P.w = cb.w; P.v = cb.v; P.al = cb_al; P.a = cb_a; }
    // (a body attached to the world: the record holds I * X_PF)
    const M3 R_WF = world ? ldm3(rec + FB_XPF) : P.R * ldm3(rec + FB_XPF);
    const V3 d1 = world ? ldv3(rec + FB_XPF + 9) : P.R * ldv3(rec + FB_XPF + 9);
    BodyState bs;
    V3 al, acc;
    if (s == 0 && J0 == IDTO_JOINT_PLANAR) {
      M3 R_FM = ident3();
      R_FM.m[0] = cs[0]; R_FM.m[1] = -sn[0]; R_FM.m[3] = sn[0]; R_FM.m[4] = cs[0];
      const V3 ex = col(R_WF, 0), ey = col(R_WF, 1), ez = col(R_WF, 2);
      const V3 d2 = ex * q0x + ey * q0y;
      const V3 v_rel = ex * v0x + ey * v0y;
      const V3 a_rel = ex * a0x + ey * a0y;
      const V3 w_rel = ez * vj[0];
      const V3 al_rel = ez * aj[0];
      hW[0] = zero;
      bs.R = R_WF * R_FM;
      r[0] = d1 + d2;
      // (planar joints hang off the world: every product with the parent's velocity is a zero)
      bs.p = zero + r[0];
      bs.w = zero + w_rel;
      bs.v = zero + v_rel;
      al = zero + al_rel;
      acc = zero + a_rel;
    } else {
      const V3 axis = ldv3(rec + FB_AXIS);
      const M3 R_FM = axis_angle(axis, sn[s], cs[s]);
      hW[s] = R_WF * axis;
      const V3 w_rel = hW[s] * vj[s];
      const V3 al_rel = hW[s] * aj[s];
      bs.R = R_WF * R_FM;
      r[s] = d1 + zero;
      if (world) {
        bs.p = zero + r[s];
        bs.w = zero + w_rel;
        bs.v = zero;
        al = zero + al_rel;
        acc = zero;
      } else {
        bs.p = P.p + r[s];
        bs.w = P.w + w_rel;
        bs.v = (P.v + cross(P.w, r[s])) + zero;
        al = (P.al + al_rel) + cross(P.w, w_rel);
        acc = ((P.a + cross(P.al, r[s])) + cross(P.w, cross(P.w, r[s]))) + zero;
      }
    }
    V3 fin, nin;
    inertial_wrench_rec(rec, bs.R, bs.w, al, acc, g, &fin, &nin);
    pin(fin); pin(nin); pin(r[s]); pin(hW[s]);
    if (s == MAXC - 1) FD_STAMP(8);
    if (s + 1 < MAXC) prefetch_record(bt + (s + 1) * FB_STRIDE, rc_next);   // ... the next slot's, across this slot's pairs
    V3 fext = zero, next = zero;
    if (W2 >= 1 && s == W2) {
      V3 ofe = zero, one = zero;   // ... on the slot before
      if (full) pair_group<true>(plist, seg[1 + s], cp, bs, bs_prev, &fext, &next, &ofe, &one);
      ft[W2 - 1] = ft[W2 - 1] - ofe;
      nt[W2 - 1] = nt[W2 - 1] - one;
      pin(ft[W2 - 1]); pin(nt[W2 - 1]);
    } else if (full) {
      pair_group<HAS_COMMON>(plist, seg[1 + s], cp, bs, cb, &fext, &next, &cfe, &cne);
    }
    ft[s] = fin - fext;
    nt[s] = nin - next;
    pin(ft[s]); pin(nt[s]);
    if (W2 >= 1 && s == W2 - 1) bs_prev = bs;
    P.R = bs.R; P.p = bs.p; P.w = bs.w; P.v = bs.v; P.al = al; P.a = acc;
    if (s == 0) FD_STAMP(7);
    if (s == MAXC - 1) FD_STAMP(9);
  }
  if (HAS_COMMON && full) pair_group<false>(plist, seg[MAXC + 1], cp, cb, cb, &cfe, &cne, &cfe, &cne);

  FD_STAMP(10);
  // ---- backward pass along the chain, joint torques
  const In inb = in_fwd.laundered();
  const double* btb = lds_launder(bt);
  V3 child_f = zero, child_n = zero;
  V3 root_f = zero, root_n = zero;
#pragma unroll
  for (int s = MAXC - 1; s >= 0; --s) {
    const double* rec = btb + s * FB_STRIDE;
    V3 f = ft[s], n = nt[s];
    if (s + 1 < MAXC && s + 1 != W2) { f = f + child_f; n = n + child_n; }   // (slot W2 is no child of the slot before)
    if (s == 0 && J0 == IDTO_JOINT_PLANAR) {
      const M3 R_WF = ldm3(rec + FB_XPF);
      const double t0 = dot(col(R_WF, 0), f), t1 = dot(col(R_WF, 1), f), t2 = dot(col(R_WF, 2), n);
      // (the joint velocities of the damping term are read again rather than kept in registers since the forward pass:
      // the same expression of the same LDS words, hence the same bits)
      tau[vs[0]] = full ? t0 + rec[FB_DAMP] * inb.V(vs[0]) : t0;
      tau[vs[0] + 1] = full ? t1 + rec[FB_DAMP + 1] * inb.V(vs[0] + 1) : t1;
      tau[vs[0] + 2] = full ? t2 + rec[FB_DAMP + 2] * inb.V(vs[0] + 2) : t2;
    } else {
      const double t = dot(hW[s], n);
      tau[vs[s]] = full ? t + rec[FB_DAMP] * inb.V(vs[s]) : t;
    }
    const V3 cf = f, cn = n + cross(r[s], f);
    if (s > 0) { child_f = cf; child_n = cn; }
    else if (K0 == PK_COMMON) { root_f = cf; root_n = cn; }
  }

  FD_STAMP(11);
  // ---- common body: butterfly sums over the lanes of this evaluation
  if (HAS_COMMON) {
    const V3 ext_f = tree_sum_fast<NP>(cfe);
    const V3 ext_n = tree_sum_fast<NP>(cne);
    const V3 ch_f = tree_sum_fast<NP>(root_f);
    const V3 ch_n = tree_sum_fast<NP>(root_n);
    const V3 f = (cb_fin - ext_f) + ch_f;
    const V3 n = (cb_nin - ext_n) + ch_n;
    if (path == 0) {
      const double* ctb = lds_launder(ct);
      const M3 R_WF = ldm3(ctb + FB_XPF);
      const V3 nF = tmul(R_WF, n), fF = tmul(R_WF, f);
      const double t6[6] = {nF.x, nF.y, nF.z, fF.x, fF.y, fF.z};
#pragma unroll
      for (int i = 0; i < 6; ++i) tau[cvs + i] = full ? t6[i] + ctb[FB_DAMP + i] * inb.V(cvs + i) : t6[i];
    }
  }
}

}  // namespace idto_dev
