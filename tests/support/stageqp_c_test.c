/* A plain C client of the structured stage form (include/mpcqp.h): the header compiles as C99, the pattern call needs no GPU, and the whole
 * call sequence -- dims -> create -> update (host blocks) -> solve -> get -- runs a batch of tiny LQ problems whose answer is known in closed
 * form.  Exit codes: 0 ok, 3 refused for lack of a GPU (after the host-only checks passed), anything else a failure. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mpcqp.h"

#define NF 4 /* frames */
#define NX 2
#define NU 1
#define F (NX + NU)
#define B 5

int main(void) {
  mpcqp_stageqp_dims d;
  memset(&d, 0, sizeof d);
  d.N = NF; d.nx = NX; d.nu = NU; d.np = 0; d.cost_mask = NULL; d.dyn_mask = NULL;
  int sz[4];
  if (mpcqp_stageqp_pattern(&d, sz, NULL, NULL, NULL, NULL) != MPCQP_OK) { fprintf(stderr, "pattern: %s\n", mpcqp_strerror(MPCQP_ERR_ARG)); return 1; }
  const int n = sz[0], m = sz[1];
  if (n != NF * F || m != n + (NF - 1) * NX || sz[2] != NF * F * F || sz[3] != n + (NF - 1) * NX + (NF - 1) * NX * F) { fprintf(stderr, "sizes %d %d %d %d\n", sz[0], sz[1], sz[2], sz[3]); return 1; }
  int *Pp = malloc(sizeof(int) * (n + 1)), *Pi = malloc(sizeof(int) * sz[2]), *Ap = malloc(sizeof(int) * (n + 1)), *Ai = malloc(sizeof(int) * sz[3]);
  if (mpcqp_stageqp_pattern(&d, NULL, Pp, Pi, Ap, Ai) != MPCQP_OK || Pp[n] != sz[2] || Ap[n] != sz[3]) return 1;
  d.N = 1;
  if (mpcqp_stageqp_pattern(&d, sz, NULL, NULL, NULL, NULL) != MPCQP_ERR_ARG) { fprintf(stderr, "NF = 1 was not refused\n"); return 1; }
  d.N = NF;
  printf("pattern ok: n %d m %d nnz(P) %d nnz(A) %d\n", n, m, Pp[n], Ap[n]);

  /* instance b: double integrator with step h_b, weights 1 on the state and r on the input, s_0 pinned, input free:
   * with every input forced to zero by tight bounds the trajectory is s_{k+1} = A s_k -- known without solving anything */
  static double H[B][NF][F][F], AB[B][NF - 1][NX][F], q[B][NF * F], l[B][NF * F + (NF - 1) * NX], u[B][NF * F + (NF - 1) * NX], want[B][NF][NX];
  memset(H, 0, sizeof H); memset(AB, 0, sizeof AB); memset(q, 0, sizeof q);
  for (int b = 0; b < B; b++) {
    const double h = 0.1 + 0.05 * b;
    for (int k = 0; k < NF; k++) { H[b][k][0][0] = 2.0; H[b][k][1][1] = 2.0; H[b][k][2][2] = 0.2; }
    for (int k = 0; k < NF - 1; k++) { AB[b][k][0][0] = 1.0; AB[b][k][0][1] = h; AB[b][k][1][1] = 1.0; AB[b][k][0][2] = 0.5 * h * h; AB[b][k][1][2] = h; }
    for (int i = 0; i < m; i++) { l[b][i] = -1e30; u[b][i] = 1e30; }
    l[b][0] = u[b][0] = 1.0 + b; l[b][1] = u[b][1] = -0.5;                       /* s_0 */
    for (int k = 0; k < NF; k++) { l[b][k * F + 2] = 0.0; u[b][k * F + 2] = 0.0; }  /* u_k = 0 */
    for (int i = n; i < m; i++) { l[b][i] = 0.0; u[b][i] = 0.0; }               /* s_{k+1} - A s_k - B u_k = 0 */
    want[b][0][0] = 1.0 + b; want[b][0][1] = -0.5;
    for (int k = 0; k + 1 < NF; k++) { want[b][k + 1][0] = want[b][k][0] + h * want[b][k][1]; want[b][k + 1][1] = want[b][k][1]; }
  }
  mpcqp_settings st;
  mpcqp_default_settings(&st);
  st.eps_abs = 1e-6; st.eps_rel = 1e-6;
  mpcqp_stageqp *sq = NULL;
  int rc = mpcqp_stageqp_create(&d, B, &st, &sq);
  if (rc == MPCQP_ERR_NO_GPU) { fprintf(stderr, "refused: %s\n", mpcqp_strerror(rc)); return 3; }
  if (rc != MPCQP_OK) { fprintf(stderr, "create: %s\n", mpcqp_strerror(rc)); return 1; }
  mpcqp_handle *hd = mpcqp_stageqp_handle(sq);
  if ((rc = mpcqp_stageqp_update(sq, &H[0][0][0][0], NULL, NULL, &AB[0][0][0][0], &q[0][0], &l[0][0], &u[0][0], MPCQP_MEM_HOST, NULL)) != MPCQP_OK ||
      (rc = mpcqp_solve(hd, NULL)) != MPCQP_OK) { fprintf(stderr, "update / solve: %s\n", mpcqp_strerror(rc)); return 1; }
  static double x[B][NF * F]; int status[B], iters[B];
  if ((rc = mpcqp_get(hd, &x[0][0], NULL, NULL, status, iters, NULL, MPCQP_MEM_HOST)) != MPCQP_OK) { fprintf(stderr, "get: %s\n", mpcqp_strerror(rc)); return 1; }
  double worst = 0.0;
  for (int b = 0; b < B; b++) {
    if (status[b] != MPCQP_SOLVED) { fprintf(stderr, "instance %d: status %d\n", b, status[b]); return 1; }
    for (int k = 0; k < NF; k++) for (int i = 0; i < NX; i++) worst = fmax(worst, fabs(x[b][k * F + i] - want[b][k][i]));
  }
  printf("stage form from C: %d instances solved, iterations %d..%d, max |s - closed form| %.2e\n", B, iters[0], iters[B - 1], worst);
  mpcqp_stageqp_destroy(sq);
  free(Pp); free(Pi); free(Ap); free(Ai);
  return worst < 1e-4 ? 0 : 1;
}
