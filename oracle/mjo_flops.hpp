// mjo_flops.hpp — INSTRUMENTED build of the CPU oracle (test / measurement infrastructure, like everything under oracle/).
//
// SURVEY.md §8(d) asks for the algorithmic flops per env-step "from the CPU restatement with an instrumented build (flop
// counter per phase) rather than a guess".  This header turns oracle/mjo.c itself into that build without touching its
// arithmetic: the file is compiled as C++ with
//     g++ -x c++ -include mjo_flops.hpp -DMJO_FLOPS ... mjo.c        (oracle/Makefile, target libmjo_flops.so)
// and `double` is re-defined below as a one-member class whose operators do the same IEEE operation on the member and count it
// in the bin of the current phase (MJO_PHASE(k) markers at the entry of every stage of mjo.c; no-ops in the normal build).
// The class is trivially copyable and layout-compatible with double, so the C ABI of the library (double* arrays, double
// arguments in SSE registers) is unchanged and oracle/mjo.py loads it like the plain library.
//
// What is counted: add / sub, mul, div, sqrt, transcendental calls (sin cos atan2 pow), comparisons and fabs / fmax / fmin / negation
// separately (not flops).  "flops" = add + mul + div + sqrt + transcendental, every operation as ONE flop (an a*b+c pair = 2).
// What is NOT counted: integer work, copies, memset - the floating-point arithmetic of the algorithm only.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

enum { MJO_PH_KIN = 0, MJO_PH_COM, MJO_PH_TENDON, MJO_PH_CRB, MJO_PH_COLL, MJO_PH_CONS, MJO_PH_VEL, MJO_PH_PASSIVE, MJO_PH_RNE, MJO_PH_ACT,
       MJO_PH_ACC, MJO_PH_SOL_SETUP, MJO_PH_SOL_HESS, MJO_PH_SOL_FACTOR, MJO_PH_SOL_MATVEC, MJO_PH_SOL_LS, MJO_PH_SOL_UPDATE, MJO_PH_SENSORS,
       MJO_PH_INTEG, MJO_PH_OTHER, MJO_NPHASE };
enum { MJO_K_ADD = 0, MJO_K_MUL, MJO_K_DIV, MJO_K_SQRT, MJO_K_TRANS, MJO_K_CMP, MJO_K_MISC, MJO_NKIND };

extern thread_local unsigned long long mjo_flop_bins[MJO_NPHASE][MJO_NKIND];
extern thread_local int mjo_flop_phase;
#define MJO_PHASE(k) (mjo_flop_phase = (k))
#define MJO_CNT(kind) (++mjo_flop_bins[mjo_flop_phase][kind])

struct fdouble {
  double v;
  fdouble() = default;
  fdouble(double x) : v(x) {}
  explicit operator int() const { return (int)v; }
  explicit operator long() const { return (long)v; }
  explicit operator unsigned() const { return (unsigned)v; }
  explicit operator bool() const { return v != 0; }
  fdouble& operator+=(fdouble o) { MJO_CNT(MJO_K_ADD); v += o.v; return *this; }
  fdouble& operator-=(fdouble o) { MJO_CNT(MJO_K_ADD); v -= o.v; return *this; }
  fdouble& operator*=(fdouble o) { MJO_CNT(MJO_K_MUL); v *= o.v; return *this; }
  fdouble& operator/=(fdouble o) { MJO_CNT(MJO_K_DIV); v /= o.v; return *this; }
  fdouble operator-() const { MJO_CNT(MJO_K_MISC); return fdouble(-v); }
  fdouble operator+() const { return *this; }
};
static_assert(sizeof(fdouble) == sizeof(double), "layout-compatible with double");
inline fdouble operator+(fdouble a, fdouble b) { MJO_CNT(MJO_K_ADD); return fdouble(a.v + b.v); }
inline fdouble operator-(fdouble a, fdouble b) { MJO_CNT(MJO_K_ADD); return fdouble(a.v - b.v); }
inline fdouble operator*(fdouble a, fdouble b) { MJO_CNT(MJO_K_MUL); return fdouble(a.v * b.v); }
inline fdouble operator/(fdouble a, fdouble b) { MJO_CNT(MJO_K_DIV); return fdouble(a.v / b.v); }
inline bool operator<(fdouble a, fdouble b) { MJO_CNT(MJO_K_CMP); return a.v < b.v; }
inline bool operator>(fdouble a, fdouble b) { MJO_CNT(MJO_K_CMP); return a.v > b.v; }
inline bool operator<=(fdouble a, fdouble b) { MJO_CNT(MJO_K_CMP); return a.v <= b.v; }
inline bool operator>=(fdouble a, fdouble b) { MJO_CNT(MJO_K_CMP); return a.v >= b.v; }
inline bool operator==(fdouble a, fdouble b) { MJO_CNT(MJO_K_CMP); return a.v == b.v; }
inline bool operator!=(fdouble a, fdouble b) { MJO_CNT(MJO_K_CMP); return a.v != b.v; }
// mixed forms with built-in arithmetic types (literals, ints) resolve through the converting constructor; these keep
// `2 * x`, `x < 0`, `1 / x` unambiguous without a conversion operator back to double
#define MJO_MIXED(op, ret)                                                                  \
  inline ret operator op(fdouble a, double b) { return a op fdouble(b); }                   \
  inline ret operator op(double a, fdouble b) { return fdouble(a) op b; }                   \
  inline ret operator op(fdouble a, int b) { return a op fdouble((double)b); }              \
  inline ret operator op(int a, fdouble b) { return fdouble((double)a) op b; }
MJO_MIXED(+, fdouble) MJO_MIXED(-, fdouble) MJO_MIXED(*, fdouble) MJO_MIXED(/, fdouble)
MJO_MIXED(<, bool) MJO_MIXED(>, bool) MJO_MIXED(<=, bool) MJO_MIXED(>=, bool) MJO_MIXED(==, bool) MJO_MIXED(!=, bool)
#undef MJO_MIXED
inline fdouble sqrt(fdouble a) { MJO_CNT(MJO_K_SQRT); return fdouble(std::sqrt(a.v)); }
inline fdouble fabs(fdouble a) { MJO_CNT(MJO_K_MISC); return fdouble(std::fabs(a.v)); }
inline fdouble sin(fdouble a) { MJO_CNT(MJO_K_TRANS); return fdouble(std::sin(a.v)); }
inline fdouble cos(fdouble a) { MJO_CNT(MJO_K_TRANS); return fdouble(std::cos(a.v)); }
inline fdouble atan2(fdouble a, fdouble b) { MJO_CNT(MJO_K_TRANS); return fdouble(std::atan2(a.v, b.v)); }
inline fdouble pow(fdouble a, fdouble b) { MJO_CNT(MJO_K_TRANS); return fdouble(std::pow(a.v, b.v)); }
inline fdouble pow(fdouble a, int b) { MJO_CNT(MJO_K_TRANS); return fdouble(std::pow(a.v, b)); }
inline fdouble pow(fdouble a, double b) { MJO_CNT(MJO_K_TRANS); return fdouble(std::pow(a.v, b)); }
inline fdouble fmax(fdouble a, fdouble b) { MJO_CNT(MJO_K_MISC); return fdouble(std::fmax(a.v, b.v)); }
inline fdouble fmin(fdouble a, fdouble b) { MJO_CNT(MJO_K_MISC); return fdouble(std::fmin(a.v, b.v)); }
inline fdouble fmax(double a, fdouble b) { return fmax(fdouble(a), b); }
inline fdouble fmax(fdouble a, double b) { return fmax(a, fdouble(b)); }
inline fdouble fmin(double a, fdouble b) { return fmin(fdouble(a), b); }
inline fdouble fmin(fdouble a, double b) { return fmin(a, fdouble(b)); }

// counters of THIS thread: out[MJO_NPHASE * MJO_NKIND]; reset to zero
extern "C" {
void mjo_flops_get(unsigned long long* out);
void mjo_flops_reset(void);
int mjo_flops_nphase(void);
int mjo_flops_nkind(void);
const char* mjo_flops_phase_name(int k);
}

#ifdef MJO_FLOPS_IMPL
thread_local unsigned long long mjo_flop_bins[MJO_NPHASE][MJO_NKIND];
thread_local int mjo_flop_phase = MJO_PH_OTHER;
extern "C" {
void mjo_flops_get(unsigned long long* out) { std::memcpy(out, mjo_flop_bins, sizeof(mjo_flop_bins)); }
void mjo_flops_reset(void) { std::memset(mjo_flop_bins, 0, sizeof(mjo_flop_bins)); mjo_flop_phase = MJO_PH_OTHER; }
int mjo_flops_nphase(void) { return MJO_NPHASE; }
int mjo_flops_nkind(void) { return MJO_NKIND; }
const char* mjo_flops_phase_name(int k) {
  static const char* const names[MJO_NPHASE] = {"kinematics (A1)", "com_pos (A2)", "tendon/transmission (A3)", "crb + factor M (A4)", "collision (A5)", "constraint rows (A6)",
                                                "com_vel (A7)", "passive (A7)", "rne bias (A7)", "actuation (A8)", "M^-1 f (A9)", "solver: costs / warm start (A10)",
                                                "solver: gradient + Hessian assembly (A10)", "solver: Cholesky + solve (A10)", "solver: M v, J v (A10)", "solver: line search (A10)",
                                                "solver: update + J^T f (A10)", "sensors (A12)", "integrator (A11)", "other (checks, ctrl)"};
  return k >= 0 && k < MJO_NPHASE ? names[k] : "?";
}
}
#endif

#define double fdouble
