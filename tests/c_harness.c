/*
 * tests/c_harness.c -- the reference's ccall sequence, performed from plain C.
 *
 * What osqp/OSQP.jl does at the boundary [REF src/interface.jl:113-162 (setup!), 164-217 (solve!), 235-250
 * (update_q!), 219-233 (clean!, version), 740-747 (dimensions)] restated without the Python mirror: the library is
 * opened with dlopen / dlsym (as `ccall((:osqp_setup, OSQP.osqp), ...)` does), `csc` / `OSQPData` /
 * `OSQPSettings` are built by hand, and after `osqp_solve` the results are read the way `unsafe_load` reads the
 * Julia mirrors: through RAW BYTE OFFSETS [REF src/types.jl:173-217 -> data 0, delta_y 120, delta_x 136,
 * solution 200, info 208; src/types.jl:81-99 -> status_val 40, obj_val 56, ...], cross-checked against the field
 * names of include/osqp_amd.h (whose offsets are also asserted at compile time in that header).
 *
 * Problem: the literal data of [REF test/basic.jl:4-21]; expected values G1 [REF test/basic.jl:43-49], G2
 * [REF test/basic.jl:52-76], the infeasible-bounds status of [REF test/primal_infeasibility.jl:41-59] is covered by the
 * Python cases; here a primal-infeasible variant of the same problem exercises the `delta_y` hop.
 *
 *   gcc -std=c11 -O1 -Wall -I<repo> tests/c_harness.c -o c_harness -ldl -lm
 *   ./c_harness <path of libosqp_amd.so | libosqp_oracle.so>
 * Prints one JSON line per step; exit status 0 iff every check passed.
 */
#define _POSIX_C_SOURCE 200809L
#include <dlfcn.h>
#include <signal.h>
#include <sys/time.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "include/osqp_amd.h"

typedef void (*fn_defaults)(OSQPSettings *);
typedef c_int (*fn_setup)(OSQPWorkspace **, const OSQPData *, const OSQPSettings *);
typedef c_int (*fn_w)(OSQPWorkspace *);
typedef c_int (*fn_wv)(OSQPWorkspace *, const c_float *);
typedef c_int (*fn_wf)(OSQPWorkspace *, c_float);
typedef const char *(*fn_ver)(void);
typedef c_int (*fn_wi)(OSQPWorkspace *, c_int);

/* Ctrl-C during a solve [REF src/constants.jl:17 :Interrupted]: the harness's own SIGINT handler (what a host runtime
 * has installed), and a timer that raises SIGINT at this process while osqp_solve runs */
static volatile sig_atomic_t harness_sigint = 0;
static void harness_on_sigint(int sig) { (void)sig; harness_sigint = 1; }
static void harness_on_alarm(int sig) { (void)sig; raise(SIGINT); }

static void *must(void *h, const char *name) {
  void *p = dlsym(h, name);
  if (!p) { fprintf(stderr, "missing symbol %s\n", name); exit(2); }
  return p;
}

/* the raw reads of solve! [REF src/interface.jl:176-205]: nothing but byte offsets */
static void *hop(const void *base, size_t off) { void *p; memcpy(&p, (const char *)base + off, sizeof p); return p; }
static c_int rd_int(const void *base, size_t off) { c_int v; memcpy(&v, (const char *)base + off, sizeof v); return v; }
static c_float rd_f(const void *base, size_t off) { c_float v; memcpy(&v, (const char *)base + off, sizeof v); return v; }

static int failures = 0;
static void check(int ok, const char *what) {
  if (!ok) { failures++; fprintf(stderr, "CHECK FAILED: %s\n", what); }
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s <library>\n", argv[0]); return 2; }
  void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  fn_defaults set_default = (fn_defaults)must(h, "osqp_set_default_settings");
  fn_setup setup = (fn_setup)must(h, "osqp_setup");
  fn_w solve = (fn_w)must(h, "osqp_solve"), cleanup = (fn_w)must(h, "osqp_cleanup");
  fn_wv update_q = (fn_wv)must(h, "osqp_update_lin_cost"), update_u = (fn_wv)must(h, "osqp_update_upper_bound");
  fn_wf update_epi = (fn_wf)must(h, "osqp_update_eps_prim_inf"), update_alpha = (fn_wf)must(h, "osqp_update_alpha");
  fn_ver version = (fn_ver)must(h, "osqp_version");

  check(strcmp(version(), "0.6.2") == 0, "osqp_version [REF src/interface.jl:219-221]");
  check(cleanup(NULL) == 0, "osqp_cleanup(NULL): the finalizer of a never-set-up Model [REF src/interface.jl:24-25]");

  /* --- data of [REF test/basic.jl:4-21], 0-based CSC, P upper triangular, +-Inf clamped to +-1e30 --- */
  c_int Pp[3] = {0, 1, 1}, Pi[1] = {0};
  c_float Px[1] = {11.0};
  c_int Ap[3] = {0, 4, 8}, Ai[8] = {0, 2, 3, 4, 1, 2, 3, 4};
  c_float Ax[8] = {-1.0, -1.0, 2.0, 3.0, -1.0, -3.0, 5.0, 4.0};
  c_float q[2] = {3.0, 4.0};
  c_float l[5] = {-OSQP_INFTY, -OSQP_INFTY, -OSQP_INFTY, -OSQP_INFTY, -OSQP_INFTY};
  c_float u[5] = {0.0, 0.0, -15.0, 100.0, 80.0};
  csc P = {1, 2, 2, Pp, Pi, Px, -1}, A = {8, 5, 2, Ap, Ai, Ax, -1};
  OSQPData data = {2, 5, &P, &A, q, l, u};
  OSQPSettings st;
  memset(&st, 0xAB, sizeof st); /* whatever the defaults do not write would show */
  set_default(&st);
  check(st.rho == 0.1 && st.sigma == 1e-6 && st.scaling == 10 && st.max_iter == 4000 && st.alpha == 1.6 && st.linsys_solver == 0,
        "osqp_set_default_settings [REF src/types.jl:136-142]");
  st.verbose = 0; st.eps_abs = 1e-9; st.eps_rel = 1e-9; st.check_termination = 1; st.polish = 0; st.max_iter = 4000;
  st.rho = 0.1; st.adaptive_rho = 0; st.warm_start = 1;

  OSQPWorkspace *work = NULL;
  c_int flag = setup(&work, &data, &st);
  check(flag == 0 && work != NULL, "osqp_setup [REF src/interface.jl:147-159]");
  if (flag != 0 || !work) { printf("{\"step\": \"setup\", \"exitflag\": %lld}\n", (long long)flag); return 1; }
  /* the caller's arrays are borrowed only for the call [REF src/interface.jl:132-155]: wipe them */
  memset(Px, 0, sizeof Px); memset(Ax, 0, sizeof Ax); memset(q, 0, sizeof q); memset(u, 0, sizeof u); memset(l, 0, sizeof l);

  /* dimensions(model) [REF src/interface.jl:740-747]: two pointer hops */
  const void *dptr = hop(work, 0);
  check(rd_int(dptr, 0) == 2 && rd_int(dptr, 8) == 5, "dimensions through workspace.data");
  check(work->data->n == 2 && work->data->m == 5, "dimensions through the header's field names");

  /* --- G1 [REF test/basic.jl:28-50] --- */
  solve(work); /* return value ignored, as in the reference */
  const void *info = hop(work, 208), *sol = hop(work, 200);
  check(info == (void *)work->info && sol == (void *)work->solution, "raw offsets 208 / 200 agree with the header");
  const c_float *x = (const c_float *)hop(sol, 0), *y = (const c_float *)hop(sol, 8);
  char status[33];
  memcpy(status, (const char *)info + 8, 32); status[32] = 0;
  printf("{\"step\": \"G1\", \"status\": \"%s\", \"status_val\": %lld, \"iter\": %lld, \"x\": [%.12g, %.12g], \"y\": [%.12g, %.12g, %.12g, %.12g, %.12g], "
         "\"obj\": %.12g, \"pri_res\": %.3e, \"dua_res\": %.3e}\n",
         status, (long long)rd_int(info, 40), (long long)rd_int(info, 0), x[0], x[1], y[0], y[1], y[2], y[3], y[4], rd_f(info, 56),
         rd_f(info, 64), rd_f(info, 72));
  check(rd_int(info, 40) == OSQP_SOLVED && strcmp(status, "solved") == 0, "G1 status");
  check(fabs(x[0] - 0.0) < 1e-5 && fabs(x[1] - 5.0) < 1e-5, "G1 x [REF test/basic.jl:43]");
  check(fabs(y[0] - 1.666666666666) < 1e-5 && fabs(y[1]) < 1e-5 && fabs(y[2] - 1.3333333) < 1e-5 && fabs(y[3]) < 1e-5 && fabs(y[4]) < 1e-5,
        "G1 y [REF test/basic.jl:44-48]");
  check(fabs(rd_f(info, 56) - 20.0) < 1e-5, "G1 obj_val [REF test/basic.jl:49]");
  check(rd_int(info, 0) > 0 && rd_int(info, 0) <= 4000 && rd_f(info, 88) >= 0.0 && rd_f(info, 112) >= rd_f(info, 88), "iter / solve_time / run_time");

  /* --- G2: update_q! [REF src/interface.jl:235-250, test/basic.jl:52-76] --- */
  c_float q2[2] = {10.0, 20.0};
  check(update_q(work, q2) == 0, "osqp_update_lin_cost");
  memset(q2, 0, sizeof q2);
  solve(work);
  x = (const c_float *)hop(hop(work, 200), 0); y = (const c_float *)hop(hop(work, 200), 8);
  printf("{\"step\": \"G2\", \"status_val\": %lld, \"x\": [%.12g, %.12g], \"obj\": %.12g}\n", (long long)rd_int(hop(work, 208), 40), x[0], x[1],
         rd_f(hop(work, 208), 56));
  check(fabs(x[0] - 0.0) < 1e-5 && fabs(x[1] - 5.0) < 1e-5, "G2 x [REF test/basic.jl:69]");
  check(fabs(y[0] - 3.33333333) < 1e-5 && fabs(y[2] - 6.66666667) < 1e-5, "G2 y [REF test/basic.jl:70-74]");
  check(fabs(rd_f(hop(work, 208), 56) - 100.0) < 1e-5, "G2 obj_val [REF test/basic.jl:75]");

  /* --- settings entry points: the validation the reference relies on [REF src/interface.jl:515-567] --- */
  check(update_epi(work, -1.0) != 0 && update_epi(work, 0.0) == 0 && update_epi(work, 1e-4) == 0, "eps_prim_inf: negative values refused on update (zero only at setup)");
  check(update_alpha(work, 2.0) != 0 && update_alpha(work, 1.6) == 0, "alpha in (0, 2)");

  /* --- primal infeasible variant: rows 0 and 1 say x >= 0, so 2 x1 + 5 x2 <= -100 (row 3) cannot hold; the
   *     certificate is read through workspace.delta_y [REF src/interface.jl:199-201] --- */
  c_float u3[5] = {0.0, 0.0, -15.0, -100.0, 80.0};
  check(update_u(work, u3) == 0, "osqp_update_upper_bound");
  solve(work);
  info = hop(work, 208);
  const c_float *cert = (const c_float *)hop(work, 120);
  check(cert == work->delta_y, "raw offset 120 = delta_y");
  printf("{\"step\": \"infeasible\", \"status_val\": %lld, \"delta_y\": [%.6g, %.6g, %.6g, %.6g, %.6g]}\n", (long long)rd_int(info, 40), cert[0],
         cert[1], cert[2], cert[3], cert[4]);
  check(rd_int(info, 40) == OSQP_PRIMAL_INFEASIBLE, "status of the infeasible variant");
  {
    /* a certificate: A' dy = 0, u' dy_+ + l' dy_- < 0, normalised to unit infinity norm [REF src/interface.jl:199-201] */
    c_float mx = 0.0, ub = 0.0, aty0, aty1;
    const c_float Ad[5][2] = {{-1, 0}, {0, -1}, {-1, -3}, {2, 5}, {3, 4}};
    aty0 = aty1 = 0.0;
    for (int i = 0; i < 5; i++) { mx = fmax(mx, fabs(cert[i])); ub += u3[i] * fmax(cert[i], 0.0); aty0 += Ad[i][0] * cert[i]; aty1 += Ad[i][1] * cert[i]; }
    check(fabs(mx - 1.0) < 1e-9 && ub < 0.0 && fabs(aty0) < 1e-3 && fabs(aty1) < 1e-3, "delta_y is a primal-infeasibility certificate");
  }
  /* --- interrupted solve: a solve that cannot end on its own (no termination test, 2e9 iterations) gets a SIGINT 30 ms
   *     in; status_val -5 "interrupted", osqp_solve comes back, and the handler installed before the call is in place
   *     again afterwards (the library must neither keep its own nor reset to the default) --- */
  {
    fn_wi update_max_iter = (fn_wi)must(h, "osqp_update_max_iter"), update_check = (fn_wi)must(h, "osqp_update_check_termination");
    c_float u4[5] = {0.0, 0.0, -15.0, 100.0, 80.0};
    check(update_u(work, u4) == 0 && update_max_iter(work, 2000000000) == 0 && update_check(work, 0) == 0, "settings of the endless solve");
    struct sigaction sa, sa_alarm;
    memset(&sa, 0, sizeof sa); sa.sa_handler = harness_on_sigint; sigemptyset(&sa.sa_mask);
    memset(&sa_alarm, 0, sizeof sa_alarm); sa_alarm.sa_handler = harness_on_alarm; sigemptyset(&sa_alarm.sa_mask);
    sigaction(SIGINT, &sa, NULL);
    sigaction(SIGALRM, &sa_alarm, NULL);
    struct itimerval tv = {{0, 0}, {0, 30000}};
    setitimer(ITIMER_REAL, &tv, NULL);
    c_int rc = solve(work);
    info = hop(work, 208);
    memcpy(status, (const char *)info + 8, 32); status[32] = 0;
    x = (const c_float *)hop(hop(work, 200), 0);
    printf("{\"step\": \"interrupt\", \"status\": \"%s\", \"status_val\": %lld, \"exitflag\": %lld, \"harness_handler_ran_during_solve\": %d}\n", status,
           (long long)rd_int(info, 40), (long long)rc, (int)harness_sigint);
    check(rd_int(info, 40) == OSQP_SIGINT && strcmp(status, "interrupted") == 0, "status of the interrupted solve [REF src/constants.jl:17]");
    check(harness_sigint == 0, "the library's handler, not the caller's, took the SIGINT raised during the solve");
    check(x[0] != x[0], "no solution is stored by an interrupted solve");
    raise(SIGINT);
    check(harness_sigint == 1, "the caller's SIGINT handler is back after osqp_solve");
    signal(SIGINT, SIG_DFL); signal(SIGALRM, SIG_DFL);
    /* and the workspace is still usable: back to a terminating configuration, solved from the interrupted iterate */
    check(update_max_iter(work, 4000) == 0 && update_check(work, 1) == 0, "settings back");
    solve(work);
    check(rd_int(hop(work, 208), 40) == OSQP_SOLVED, "a solve after the interrupted one");
  }
  check(cleanup(work) == 0, "osqp_cleanup [REF src/interface.jl:223-233]");
  printf("{\"step\": \"done\", \"failures\": %d}\n", failures);
  return failures ? 1 : 0;
}
